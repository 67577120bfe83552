"""Several libraries across several GPUs: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for CPU rehearsal), libraries dealt round-robin,
one integer all-reduce of the samples x tags matrix at the end.

The result equals `combineReadCounts` of the reference (tagdigger_fun.py:1061-1098) applied to
the per-file matrices -- files in sorted order, samples in order of first appearance, equal
sample names summed -- whatever the number of ranks, because integer addition is associative.
"""
import torch
import torch.distributed as dist


def sample_rows(bckeys):
    """Global sample order exactly as combineReadCounts builds it, and per file the row of each barcode."""
    order, slot, rows = [], {}, {}
    for f in sorted(bckeys.keys()):
        rows[f] = []
        for sample in bckeys[f][1]:
            if sample not in slot:
                slot[sample] = len(order)
                order.append(sample)
            rows[f].append(slot[sample])
    return order, rows


def count_libraries(bckeys, tags, cutsite, counter=None, device=None):
    """Every rank calls this with the same arguments.  `counter(file, barcodes, tags, cutsite)`
    returns the per-barcode matrix of one file (default: the GPU find_tags_fastq on this rank's
    device).  Returns [sample names, samples x tags counts] on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if counter is None:
        from . import tagdigger_fun
        dev_index = device.index if isinstance(device, torch.device) and device.index is not None else 0

        def counter(f, barcodes, tgs, cs):
            return tagdigger_fun.find_tags_fastq(f, barcodes, tgs, cutsite=cs, device=dev_index)
    order, rows = sample_rows(bckeys)
    total = torch.zeros((len(order), len(tags)), dtype=torch.int64, device=device if device is not None else "cpu")
    for k, f in enumerate(sorted(bckeys.keys())):
        if k % world != rank:
            continue
        m = torch.tensor(counter(f, bckeys[f][0], tags, cutsite), dtype=torch.int64).reshape(len(bckeys[f][0]), len(tags))
        total.index_add_(0, torch.tensor(rows[f], dtype=torch.int64, device=total.device), m.to(total.device))
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return [order, total.cpu().tolist()]


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
    import numpy as np
    a = np.frombuffer(data, dtype=np.uint8)
    n = int((a == 0x0A).sum())
    cr = np.flatnonzero(a == 0x0D)
    if cr.size:
        nxt = np.minimum(cr + 1, a.size - 1)
        n += int(((a[nxt] != 0x0A) | (cr + 1 >= a.size)).sum())
    return n


def count_file_sharded(path, barcodes, tags, cutsite="TGCAG", maxreads=5e9, counter=None, device=None):
    """find_tags_fastq on one plain FASTQ file, byte-sharded over the ranks of the default process
    group (every rank calls this with the same arguments; backend "nccl" = RCCL for GPUs).
    `counter(data, barcodes, tags, cutsite, first_line, maxreads)` returns one shard's matrix
    (default: this rank's GPU).  Returns the whole file's matrix on every rank, bit-identical to the
    single-GPU result for any number of ranks."""
    import math
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if path[-2:].lower() == 'gz':
        raise ValueError("byte sharding needs an uncompressed file; split gzip input per library instead")
    if counter is None:
        from . import tagdigger_fun
        dev_index = device.index if isinstance(device, torch.device) and device.index is not None else 0

        def counter(data, bcs, tgs, cs, first_line, mreads):
            eng = tagdigger_fun.default_engine(dev_index)
            eng.set_index(bcs, tgs, cs)
            eng.count_bytes(data, first_line=first_line, maxreads=mreads)
            return eng.counts()
    start, end = shard_bounds(path, world)[rank]
    with open(path, "rb") as fh:
        fh.seek(start)
        data = fh.read(end - start)
    mine = torch.tensor([count_terminators(data)], dtype=torch.int64, device=device if device is not None else "cpu")
    if world > 1:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        first_line = int(sum(int(t[0]) for t in every[:rank]))
    else:
        first_line = 0
    # The maxreads bound is global and so is the line index each shard is counted with: device and oracle
    # both compare the bound with the GLOBAL read ordinal (first_line + lines seen), so it is passed on
    # unchanged; a shard that starts at or past the bound is skipped.
    bound = max(1, int(math.ceil(min(maxreads, 2 ** 62))))
    reads_before = (first_line + 2) // 4               # sequence lines (index 1 mod 4) below first_line
    total = torch.zeros((len(barcodes), len(tags)), dtype=torch.int64, device=device if device is not None else "cpu")
    if len(data) and reads_before < bound:
        m = counter(data, barcodes, tags, cutsite, first_line, bound)
        total += torch.tensor(m, dtype=torch.int64).reshape(len(barcodes), len(tags)).to(total.device)
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return total.cpu().tolist()
