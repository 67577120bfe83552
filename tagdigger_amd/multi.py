"""Several libraries across several GPUs: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for CPU rehearsal), libraries dealt round-robin,
ONE integer all-reduce of the samples x tags matrix at the end.

The result equals `combineReadCounts` of the reference (tagdigger_fun.py:1061-1098) applied to
the per-file matrices -- files in sorted order, samples in order of first appearance, equal
sample names summed -- whatever the number of ranks, because integer addition is associative.

On GPUs nothing passes through Python lists: every library is counted into the engine's device
matrix, folded into the run's [samples x tags] uint32 DEVICE tensor by K3 (td_fold_rows), that
tensor is all-reduced in place, and rank 0 copies it to the host once.
"""
import numpy as np
import torch
import torch.distributed as dist


from .tagdigger_fun import sample_rows  # noqa: F401  (the sample order combineReadCounts builds; re-exported)


def _rank_world():
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def count_libraries(bckeys, tags, cutsite, counter=None, device=None, maxreads=5e9, as_array=False, progress=False):
    """Every rank calls this with the same arguments.  Returns [sample names, samples x tags counts]
    on every rank (counts: list of lists like the reference's, or an int64 numpy array with `as_array`).

    counter=None (the product path): this rank's GPU (`device`: a torch.device or an index) counts its
    libraries; barcode rows are folded into sample rows on the device (K3) and the [samples x tags] uint32
    device tensor is all-reduced in place -- no host lists anywhere.
    counter(file, barcodes, tags, cutsite) -> matrix: a stand-in for the per-file counter (the CPU rehearsal
    of the sharding and the reduction in tests/test_multi_gloo.py passes the oracle); the matrices are folded
    with numpy and reduced as an int64 host tensor.
    progress=True: rank 0 prints, file by file in the reference's order, the progress lines its find_tags_fastq
    prints (:268-271) -- every rank keeps the per-window counters of its libraries on its GPU, the lines are gathered."""
    rank, world = _rank_world()
    order, rows = sample_rows(bckeys)
    files = sorted(bckeys.keys())
    mine = [f for k, f in enumerate(files) if k % world == rank]
    if counter is None:
        from . import tagdigger_fun
        if isinstance(device, torch.device):
            dev = device if device.index is not None else torch.device("cuda", 0)
        else:
            dev = torch.device("cuda", int(device or 0))
        eng = tagdigger_fun.default_engine(dev.index)
        total = torch.zeros((len(order), len(tags)), dtype=torch.int32, device=dev)     # uint32 counts: bit pattern == int32's
        torch.cuda.synchronize(dev)
        printed = {}
        eng.set_option("progress", 1 if progress else 0)
        for f in mine:
            eng.set_index(bckeys[f][0], tags, cutsite)          # (kept when the barcode set repeats: only the counts are zeroed)
            eng.count_file(f, maxreads)
            if progress:
                printed[f] = eng.progress_lines(f)
            eng.fold_rows(rows[f], total.data_ptr(), len(order))
        eng.set_option("progress", 0)
        if progress:
            every = [printed]
            if world > 1:
                every = [None] * world
                dist.all_gather_object(every, printed)
            if rank == 0:
                for f in files:
                    for part in every:
                        for line in part.get(f, ()):
                            print(line)
        if world > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM)        # the path's one exchange: RCCL over xGMI
        out = total.cpu().numpy().view(np.uint32).astype(np.int64)
    else:
        total = np.zeros((len(order), len(tags)), dtype=np.int64)
        for f in mine:
            m = np.asarray(counter(f, bckeys[f][0], tags, cutsite), dtype=np.int64).reshape(len(bckeys[f][0]), len(tags))
            np.add.at(total, rows[f], m)
        if world > 1:
            t = torch.from_numpy(total)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = total
    return [order, out if as_array else out.tolist()]


# ---------------------------------------------------------------------------------------------
# One (uncompressed) file across several GPUs: contiguous byte ranges.  A line belongs to the
# shard in which it STARTS; what couples the shards is only the line index their first line has
# (is it a sequence line?), i.e. the number of line terminators before them -- one integer per
# rank, all-gathered -- and the additive count matrix, all-reduced at the end.
# ---------------------------------------------------------------------------------------------
def _first_line_start(fh, size, pos):
    """Smallest line start >= pos (a line starts at 0 and after every \\n, \\r\\n or bare \\r)."""
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    j = pos - 1                          # the byte before pos may already end a line
    while j < size:
        fh.seek(j)
        block = fh.read((1 << 16) + 1)   # one byte of look-ahead for a \\r at the block's end
        body = min(len(block), 1 << 16)
        for k in range(body):
            c = block[k]
            if c == 0x0A:
                return j + k + 1
            if c == 0x0D:
                followed_by_lf = k + 1 < len(block) and block[k + 1] == 0x0A
                return j + k + (2 if followed_by_lf else 1)
        j += body
    return size


def shard_bounds(path, world):
    """[(start, end)] per rank: nominal equal byte ranges moved forward to line starts."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        starts = [_first_line_start(fh, size, size * r // world) for r in range(world)] + [size]
    return [(starts[r], max(starts[r], starts[r + 1])) for r in range(world)]


def count_terminators(data):
    """Line terminators (\\n, \\r\\n, bare \\r) in a bytes-like object of whole lines."""
    a = np.frombuffer(data, dtype=np.uint8)
    n = int((a == 0x0A).sum())
    cr = np.flatnonzero(a == 0x0D)
    if cr.size:
        nxt = np.minimum(cr + 1, a.size - 1)
        n += int(((a[nxt] != 0x0A) | (cr + 1 >= a.size)).sum())
    return n


def count_file_sharded(path, barcodes, tags, cutsite="TGCAG", maxreads=5e9, counter=None, device=None, as_array=False,
                       progress=False):
    """find_tags_fastq on one plain FASTQ file, byte-sharded over the ranks of the default process
    group (every rank calls this with the same arguments; backend "nccl" = RCCL for GPUs).
    Returns the whole file's matrix on every rank, bit-identical to the single-GPU result for any
    number of ranks.

    counter=None (the product path): the shard is read into pinned-able host memory once, copied to this
    rank's GPU, its line terminators are counted THERE (td_count_lines_device), the counts are all-gathered
    (-> this shard's first line index), the shard is counted in place with the global maxreads bound, and the
    device matrix is all-reduced.
    counter(data, barcodes, tags, cutsite, first_line, maxreads) -> matrix stands in for the GPU in the CPU
    rehearsal.
    progress=True (device path): rank 0 prints the reference's progress lines (:268-271) -- every shard keeps its
    per-window counters by GLOBAL read ordinal, so the windows of all ranks simply add up (one more small all-reduce)."""
    import math
    rank, world = _rank_world()
    if path[-2:].lower() == 'gz':
        raise ValueError("byte sharding needs an uncompressed file; split gzip input per library instead")
    start, end = shard_bounds(path, world)[rank]
    data = np.fromfile(path, dtype=np.uint8, count=end - start, offset=start)
    # The maxreads bound is global and so is the line index each shard is counted with: device and oracle
    # both compare the bound with the GLOBAL read ordinal (first_line + lines seen), so it is passed on
    # unchanged; a shard that starts at or past the bound is skipped.
    bound = max(1, int(math.ceil(min(maxreads, 2 ** 62))))

    def first_line_of(my_terminators, where):
        mine = torch.tensor([my_terminators], dtype=torch.int64, device=where)
        if world == 1:
            return 0
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return int(sum(int(t[0]) for t in every[:rank]))

    if counter is None:
        from . import tagdigger_fun
        if isinstance(device, torch.device):
            dev = device if device.index is not None else torch.device("cuda", 0)
        else:
            dev = torch.device("cuda", int(device or 0))
        eng = tagdigger_fun.default_engine(dev.index)
        eng.set_index(barcodes, tags, cutsite)
        eng.set_option("progress", 1 if progress else 0)
        total = torch.zeros(len(barcodes) * len(tags), dtype=torch.int32, device=dev)
        shard = torch.empty(max(16, data.size), dtype=torch.uint8, device=dev)         # (torch allocations are 256-byte aligned)
        if data.size:
            shard[:data.size].copy_(torch.from_numpy(data))
        torch.cuda.synchronize(dev)
        terms = eng.count_lines_device(shard.data_ptr(), int(data.size)) if data.size else 0
        first_line = first_line_of(terms, dev)
        eng.bind_counts(total.data_ptr())
        try:
            if data.size and (first_line + 2) // 4 < bound:
                eng.count_device(shard.data_ptr(), int(data.size), first_line=first_line, maxreads=bound)
            st = eng.stats()                                   # (synchronises; raises what a kernel flagged)
            if progress:
                # reads of the whole file, then every rank's windows laid over the same axis and summed
                # (before the matrix is unbound: unbinding resets the handle's results, the windows with them)
                reads = torch.tensor([st["reads"]], dtype=torch.int64, device=dev)
                if world > 1:
                    dist.all_reduce(reads)
                nwin = int(reads[0]) // 50000
                win = torch.zeros((max(1, nwin), 2), dtype=torch.int64, device=dev)
                mine = eng.progress_windows(nwin)
                if mine:
                    win[:len(mine)] = torch.tensor(mine, dtype=torch.int64, device=dev)
                if world > 1:
                    dist.all_reduce(win)
        finally:
            eng.bind_counts(0)
            eng.set_option("progress", 0)
        if world > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        if progress and rank == 0:
            bar = tag = 0
            for k in range(nwin):
                bar, tag = bar + int(win[k, 0]), tag + int(win[k, 1])
                if 50000 * (k + 1) % 1000000 == 0:
                    print(path)
                print("Reads: {0} With barcode and cut site: {1} With tag: {2}".format(50000 * (k + 1), bar, tag))
        out = total.cpu().numpy().view(np.uint32).astype(np.int64).reshape(len(barcodes), len(tags))
    else:
        first_line = first_line_of(count_terminators(data), "cpu")
        out = np.zeros((len(barcodes), len(tags)), dtype=np.int64)
        if data.size and (first_line + 2) // 4 < bound:     # sequence lines (index 1 mod 4) below first_line: (first_line + 2) // 4
            out += np.asarray(counter(data.tobytes(), barcodes, tags, cutsite, first_line, bound), dtype=np.int64).reshape(out.shape)
        if world > 1:
            dist.all_reduce(torch.from_numpy(out), op=dist.ReduceOp.SUM)
    return out if as_array else out.tolist()
