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


# ---------------------------------------------------------------------------------------------
# The same for a BGZF file (bgzip: independent gzip members of at most 64 KiB): a rank takes a contiguous range of
# MEMBERS (nominal equal shares of the compressed bytes); members do not end at line ends, so a rank's lines run
# from the first line start in its members to the first line start in the members behind them.
# ---------------------------------------------------------------------------------------------
def bgzf_index(path):
    """(member offsets, inflated sizes) of a BGZF file as numpy arrays -- td_bgzf_index (host only)."""
    import ctypes as C
    from . import _binding as B
    L = B.load()
    n = C.c_uint64(0)
    B.check(L.td_bgzf_index(path.encode(), None, None, 0, C.byref(n)))
    off = np.zeros(max(1, n.value), dtype=np.uint64)
    isz = np.zeros(max(1, n.value), dtype=np.uint32)
    B.check(L.td_bgzf_index(path.encode(), C.c_void_p(off.ctypes.data), C.c_void_p(isz.ctypes.data), n.value, C.byref(n)))
    return off[:n.value], isz[:n.value]


def bgzf_member_ranges(member_off, file_size, world):
    """[(first member, end member)] per rank: the members that START in the rank's nominal share of the file."""
    first = [int(np.searchsorted(member_off, np.uint64(file_size * r // world), side="left")) for r in range(world)] + [len(member_off)]
    return [(first[r], max(first[r], first[r + 1])) for r in range(world)]


def first_line_start(data, prev_byte):
    """Offset of the first line START inside `data` (bytes-like), given the byte before it (None: the file begins
    here, so 0); len(data) when no line starts in it.  A line starts after \n, \r\n or a bare \r."""
    n = len(data)
    if prev_byte is None or prev_byte == 0x0A:
        return 0
    if prev_byte == 0x0D:
        return 1 if n and data[0] == 0x0A else 0
    a = np.frombuffer(data, dtype=np.uint8)
    hits = np.flatnonzero((a == 0x0A) | (a == 0x0D))
    if hits.size == 0:
        return n
    i = int(hits[0])
    if a[i] == 0x0D and i + 1 < n and a[i + 1] == 0x0A:
        return i + 2
    return i + 1          # (a \r in the last byte: if a \n opens the next shard, that shard's own rule skips it)


def _bgzf_member_bytes(path, off, end):
    """members [off, end) of the file inflated on the host (zlib): the CPU stand-in, and single members (a shard's neighbour)"""
    import zlib
    out = []
    with open(path, "rb") as fh:
        fh.seek(int(off))
        raw = fh.read(int(end - off))
    pos = 0
    while pos < len(raw):
        d = zlib.decompressobj(31)
        out.append(d.decompress(raw[pos:]))
        pos = len(raw) - len(d.unused_data)
    return b"".join(out)


def _agree(exc, where):
    """Called by every rank where a rank-local phase ends and a collective follows: a rank that failed raises its own
    exception, every other rank raises too -- nobody is left waiting in the collective for a rank that is gone."""
    rank, world = _rank_world()
    failed = exc is not None
    if world > 1:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=where)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = int(flag[0]) != 0
    if exc is not None:
        raise exc
    if failed:
        raise RuntimeError("tagdigger_amd: another rank failed on its share of the file; see its message")


class _DevRange:
    """a stretch of device memory with the one method of a tensor that count_file_sharded's counting block uses"""
    def __init__(self, ptr):
        self._ptr = int(ptr)

    def data_ptr(self):
        return self._ptr


def _gzip_shard_text(eng, path, dev):
    """ONE ordinary gzip file (a single member) over the ranks, each decoding a byte range of the compressed file on its
    device (include/tagdig.h td_gz_shard_*, csrc/gz_gpu.hpp): every rank finds the first block start in its range,
    decodes from there to the next rank's start into symbols -- its first window is unknown -- and leaves the MAP of its
    stretch (what the 32 KiB behind it hold in terms of the 32 KiB in front of it); the maps, all-gathered, give every
    rank its window; the stretches' CRC-32s joined in rank order are checked against the member's trailer.  A rank's
    lines run from behind the first terminator of its text to the first terminator of the next rank's (all-gathered heads);
    the bytes between the 16-byte boundary below that start and the start are overwritten with blanks (a line is stripped,
    reference :256).  Returns (device range, its length) -- or None on EVERY rank where the file is not one this scheme
    takes (several members, a seam that does not close, a line longer than 64 KiB at a seam): one rank counts it then."""
    import os
    import struct
    rank, world = _rank_world()
    fsize = os.path.getsize(path)
    NONE = -1
    HEAD = 1 << 16

    def gather(values):
        mine = torch.tensor(values, dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [[int(x) for x in t.tolist()] for t in every]

    def all_ok(ok):
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag[0]) == 0

    # 1. every rank's first block start
    lo, hi = fsize * rank // world, fsize * (rank + 1) // world
    start, ok = NONE, True
    try:
        s, _ = eng.gz_shard_open(path, lo, hi, rank == 0)
        start = NONE if s is None else s
    except Exception:                          # noqa: BLE001 -- every rank learns of it in all_ok and declines alike
        ok = False
    if not all_ok(ok):
        return None
    starts = [v[0] for v in gather([start])]
    live = [r for r in range(world) if starts[r] != NONE]
    if not live or live[0] != 0:
        return None
    nxt = [starts[r] for r in live[live.index(rank) + 1:]][:1] if rank in live else []
    # 2. the stretches, decoded; the seams
    end, nout, final, ok = 0, 0, 0, True
    mp = np.zeros(32768, dtype=np.uint16)
    if rank in live:
        try:
            end, nout, fin, mp = eng.gz_shard_decode(nxt[0] if nxt else None)
            final = 1 if fin else 0
        except Exception:                      # noqa: BLE001 -- every rank learns of it in all_ok and declines alike
            ok = False
    if not all_ok(ok):
        return None
    info = gather([end, nout, final])
    for a, b in zip(live, live[1:]):
        if info[a][0] != starts[b] or info[a][2]:
            return None                                        # (a false start at a seam, or a member that ends inside the file)
    last = live[-1]
    if not info[last][2]:
        return None
    # the member's trailer, and nothing but zeros behind it (more members: the one-rank path reads them)
    trailer = (info[last][0] + 7) // 8
    total_len = sum(info[r][1] for r in live)
    tail = None
    try:
        with open(path, "rb") as fh:
            fh.seek(trailer)
            tail = fh.read(8 + (1 << 20))                      # (zero padding behind a member is short; more than that: the one-rank path)
    except OSError:
        pass
    if not all_ok(tail is not None):
        return None
    if len(tail) < 8 or any(tail[8:]) or trailer + len(tail) < fsize:
        return None
    want_crc, want_len = struct.unpack("<II", tail[:8])
    if want_len != total_len & 0xFFFFFFFF:
        return None
    # 3. the windows through the maps, the text, the CRC-32
    maps_t = torch.from_numpy(mp.astype(np.int32)).to(dev)
    every = [torch.zeros_like(maps_t) for _ in range(world)]
    dist.all_gather(every, maps_t)
    window = np.zeros(32768, dtype=np.uint8)
    before = 0
    for r in live:
        if r >= rank:
            break
        m = every[r].cpu().numpy().astype(np.uint16)
        window = np.where(m & 0x8000, window[m & 0x7FFF], (m & 0xFF).astype(np.uint8)).astype(np.uint8)
        before += info[r][1]
    ptr, crc, ok = 0, 0, True
    if rank in live:
        try:
            ptr, crc = eng.gz_shard_resolve(window, before)
        except Exception:                      # noqa: BLE001 -- every rank learns of it in all_ok and declines alike
            ok = False
    if not all_ok(ok):
        return None
    crcs = [v[0] for v in gather([crc])]
    joined = 0
    for r in live:
        joined = eng.crc32_join(joined, crcs[r], info[r][1])
    if joined != want_crc:
        return None                                            # (the one-rank path raises what gzip.open raises)
    # 4. lines: a rank gives its text up to and including its first terminator to the rank before it
    head, ok = b"", True
    try:
        if rank in live and nout:
            head = eng.d2h(ptr, min(nout, HEAD))
    except Exception:                          # noqa: BLE001
        ok = False
    if not all_ok(ok):
        return None
    cut = None                                                 # (bytes of my text that are the previous rank's line)
    for k, c in enumerate(head):
        if c == 10:
            cut = k + 1
            break
        if c == 13:
            if k + 1 < len(head):
                cut = k + 2 if head[k + 1] == 10 else k + 1
                break
            if nout == len(head):                              # (a '\r' that ends the text: the next rank's '\n', if any, goes with it)
                cut = k + 1
                break
    if rank == 0:
        cut = 0
    whole = cut is None                                        # (no terminator in sight: all of a short text is the previous rank's)
    ok = not (whole and nout > len(head))
    if not all_ok(ok):
        return None
    give = nout if whole else cut
    heads_t = torch.zeros(HEAD + 2, dtype=torch.int32, device=dev)
    heads_t[0] = give
    heads_t[1] = 1 if whole else 0
    if give:
        heads_t[2:2 + give] = torch.tensor(list(head[:give]), dtype=torch.int32, device=dev)
    every = [torch.zeros_like(heads_t) for _ in range(world)]
    dist.all_gather(every, heads_t)
    gives = [int(every[r][0]) for r in range(world)]
    wholes = [int(every[r][1]) for r in range(world)]

    def tail_of(q):
        """the ranks whose heads end rank q's last line"""
        took = []
        for r in range(q + 1, world):
            took.append(r)
            if r in live and not wholes[r]:                    # (its text has a terminator: the line ends in what it gave)
                break
        return took
    # (every rank sees every rank's numbers: room behind a text for the heads it takes -- td_gz_shard_resolve leaves 1 MiB and an eighth)
    if any(sum(gives[r] for r in tail_of(q)) > (1 << 20) + (info[q][1] >> 3) for q in live):
        return None
    tail_bytes = b"".join(bytes(every[r][2:2 + gives[r]].cpu().numpy().astype(np.uint8).tobytes()) for r in tail_of(rank))
    # a '\r' that ended a rank's text and the '\n' the next rank's text begins with are one terminator: that '\n' is in
    # the head it gave away (its first terminator is that '\n'), so nothing more to do
    if rank not in live or whole:
        return _DevRange(0), 0
    own0, own1 = give, nout
    al = own0 - own0 % 16
    if tail_bytes:
        eng.h2d(ptr + nout, tail_bytes)
        own1 = nout + len(tail_bytes)
    if own0 > al:
        eng.h2d(ptr + al, b" " * (own0 - al))
    return _DevRange(ptr + al), own1 - al


def _count_one_stream(path, barcodes, tags, cutsite, bound, counter, dev, as_array, progress):
    """A gzip file that _gzip_shard_text does not take (several members, a seam that does not close; the CPU stand-in) under
    count_file_sharded: the reference reads any .gz by name (:240-241), so it is counted -- by rank 0 alone, through
    td_count_file (decoded on its device, or by its host's threads), the other ranks adding zeros to the same all-reduce."""
    import gzip
    import sys
    rank, world = _rank_world()
    if rank == 0 and world > 1:
        print("tagdigger_amd: %s is a gzip file the ranks cannot share (several members, a seam between two ranks' ranges that does "
              "not close, or the CPU stand-in); rank 0 reads it alone "
              "(one ordinary gzip member is decoded by all ranks, bgzip-compressed files are sharded by members)" % path, file=sys.stderr)
    if counter is not None:
        out = np.zeros((len(barcodes), len(tags)), dtype=np.int64)
        err = None
        try:
            if rank == 0:
                with gzip.open(path, "rb") as fh:
                    data = fh.read()
                out += np.asarray(counter(data, barcodes, tags, cutsite, 0, bound), dtype=np.int64).reshape(out.shape)
        except Exception as e:                 # noqa: BLE001 -- raised on every rank by _agree
            err = e
        _agree(err, "cpu")
        if world > 1:
            dist.all_reduce(torch.from_numpy(out), op=dist.ReduceOp.SUM)
        return out if as_array else out.tolist()
    from . import tagdigger_fun
    eng = tagdigger_fun.default_engine(dev.index)
    eng.set_index(barcodes, tags, cutsite)
    eng.set_option("progress", 1 if progress and rank == 0 else 0)
    total = torch.zeros(len(barcodes) * len(tags), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    eng.bind_counts(total.data_ptr())
    err = None
    try:
        if rank == 0:
            eng.count_file(path, maxreads=bound)               # (a damaged .gz raises what gzip.open raises: EOFError, gzip.BadGzipFile, zlib.error)
        eng.stats()                                            # (synchronises; raises what a kernel flagged)
        if progress and rank == 0:
            for line in eng.progress_lines(path):
                print(line)
    except Exception as e:                     # noqa: BLE001 -- raised on every rank by _agree
        err = e
    finally:
        eng.bind_counts(0)
        eng.set_option("progress", 0)
    _agree(err, dev)
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    out = total.cpu().numpy().view(np.uint32).astype(np.int64).reshape(len(barcodes), len(tags))
    return out if as_array else out.tolist()


def count_file_sharded(path, barcodes, tags, cutsite="TGCAG", maxreads=5e9, counter=None, device=None, as_array=False,
                       progress=False):
    """find_tags_fastq on one FASTQ file -- plain, BGZF-compressed (bgzip), or ONE ordinary gzip member (device path:
    _gzip_shard_text) -- sharded over the ranks of the default process group; any other gzip file (several members, or the
    CPU stand-in) is counted by rank 0 alone (_count_one_stream) (every rank calls this with the same arguments; backend
    "nccl" = RCCL for GPUs).
    Returns the whole file's matrix on every rank, bit-identical to the single-GPU result for any
    number of ranks.

    counter=None (the product path): the rank's byte range (plain) or member range (BGZF: inflated on the GPU) is
    brought into device memory through the library's pinned staging pieces -- the host never holds the shard --, its
    line terminators are counted THERE (td_count_lines_device), the counts are all-gathered (-> this shard's first line
    index), the shard is counted in place with the global maxreads bound, and the device matrix is all-reduced.
    counter(data, barcodes, tags, cutsite, first_line, maxreads) -> matrix stands in for the GPU in the CPU
    rehearsal.
    progress=True (device path): rank 0 prints the reference's progress lines (:268-271) -- every shard keeps its
    per-window counters by GLOBAL read ordinal, so the windows of all ranks simply add up (one more small all-reduce)."""
    import math
    import os
    rank, world = _rank_world()
    is_gz = path[-2:].lower() == 'gz'
    # The maxreads bound is global and so is the line index each shard is counted with: device and oracle
    # both compare the bound with the GLOBAL read ordinal (first_line + lines seen), so it is passed on
    # unchanged; a shard that starts at or past the bound is skipped.
    bound = max(1, int(math.ceil(min(maxreads, 2 ** 62))))

    def gather_ints(value, where):
        mine = torch.tensor([value], dtype=torch.int64, device=where)
        if world == 1:
            return [int(value)]
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [int(t[0]) for t in every]

    def first_line_of(my_terminators, where):
        return int(sum(gather_ints(my_terminators, where)[:rank]))

    use_gpu = counter is None
    if use_gpu:
        from . import tagdigger_fun
        if isinstance(device, torch.device):
            dev = device if device.index is not None else torch.device("cuda", 0)
        else:
            dev = torch.device("cuda", int(device or 0))
        eng = tagdigger_fun.default_engine(dev.index)
    where = dev if use_gpu else "cpu"

    # ---- this rank's bytes: `shard` (device tensor, 16-byte aligned) or `data` (host bytes, the stand-in), n bytes
    data, shard = None, None
    if not is_gz:
        err = None
        try:
            start, end = shard_bounds(path, world)[rank]
            n = end - start
            if use_gpu:
                shard = torch.empty(max(16, n), dtype=torch.uint8, device=dev)     # (torch allocations are 256-byte aligned)
                # (the library fills it on its own streams: nothing of torch's may still be pending on a block the
                # caching allocator hands out again)
                torch.cuda.current_stream(dev).synchronize()
                if n:
                    eng.load_file_range(path, start, n, shard.data_ptr())
            else:
                data = np.fromfile(path, dtype=np.uint8, count=n, offset=start).tobytes()
        except Exception as e:                 # noqa: BLE001 -- raised on every rank by _agree
            err = e
        _agree(err, where)
    else:
        from ._binding import TagdigError
        try:
            moff, misz = bgzf_index(path)
        except TagdigError as e:
            # only "this is gzip, but not BGZF" sends the file to the one-stream path; an unreadable file is an error
            if "BGZF" not in e.detail:
                raise
            moff = None
        gz_text = None
        if moff is None and use_gpu and world > 1:
            got = _gzip_shard_text(eng, path, dev)
            if got is not None:
                gz_text = got
        if moff is None and gz_text is None:
            return _count_one_stream(path, barcodes, tags, cutsite, bound, counter, dev if use_gpu else None, as_array, progress)
        if gz_text is not None:
            shard, n = gz_text
        else:
            fsize = os.path.getsize(path)
            ends = np.append(moff[1:], np.uint64(fsize))
            gpos = np.concatenate(([0], np.cumsum(misz.astype(np.int64))))            # inflated offset of every member (and the total)
            m0, m1 = bgzf_member_ranges(moff, fsize, world)[rank]
            own_len = int(gpos[m1] - gpos[m0])
            # the byte before my members (the last byte of the nearest non-empty member before them)
            prev_byte = None
            k = m0 - 1
            while k >= 0 and misz[k] == 0:
                k -= 1
            if k >= 0:
                prev_byte = _bgzf_member_bytes(path, moff[k], ends[k])[-1]
            err, a = None, 0
            try:
                if use_gpu:
                    buf = torch.empty(max(16, own_len + (1 << 20)), dtype=torch.uint8, device=dev)
                    torch.cuda.current_stream(dev).synchronize()      # (the library inflates into it on its own streams)
                    got = eng.bgzf_inflate_range(path, int(moff[m0]) if m0 < len(moff) else fsize, int(moff[m1]) if m1 < len(moff) else fsize,
                                                 buf.data_ptr(), own_len) if own_len else 0
                    if got != own_len:
                        raise RuntimeError("tagdigger_amd: members %d..%d of %s inflate to %d bytes, their headers say %d" % (m0, m1, path, got, own_len))
                    # my first line start: the first terminator is looked for in pieces copied back (the first one holds it)
                    a, seen, pb = own_len, 0, prev_byte
                    while seen < own_len:
                        piece = buf[seen:min(own_len, seen + (1 << 20))].cpu().numpy().tobytes()
                        f = first_line_start(piece, pb)
                        if f < len(piece):
                            a = seen + f
                            break
                        seen += len(piece)
                        pb = piece[-1]           # (a terminator in the piece's last byte: the next piece's first byte decides)
                else:
                    own = _bgzf_member_bytes(path, moff[m0], ends[m1 - 1]) if m1 > m0 else b""
                    a = first_line_start(own, prev_byte)
            except Exception as e:                 # noqa: BLE001 -- raised on every rank by _agree
                err = e
            _agree(err, where)
            # every rank's first line start as an offset into the inflated file; mine end where the next one's begin
            starts = gather_ints(int(gpos[m0]) + a if a < own_len else -1, where) + [int(gpos[-1])]
            for r in range(world - 1, -1, -1):                                         # (a shard without a line start owns nothing)
                if starts[r] < 0:
                    starts[r] = starts[r + 1]
            g0, g1 = starts[rank], starts[rank + 1]
            n = g1 - g0
            # the part of my lines that lies in the members behind mine (a rank that owns no line has no such part)
            tail_len = max(0, g1 - int(gpos[m1])) if n else 0
            err = None
            try:
                if use_gpu:
                    if tail_len:
                        mt = int(np.searchsorted(gpos, g1, side="left"))               # members m1 .. mt - 1 hold it
                        cap = int(gpos[mt] - gpos[m1])
                        if own_len + cap > buf.numel():
                            bigger = torch.empty(own_len + cap, dtype=torch.uint8, device=dev)
                            bigger[:own_len].copy_(buf[:own_len])
                            buf = bigger
                        torch.cuda.current_stream(dev).synchronize()  # (the copy above, before the library writes behind it)
                        eng.bgzf_inflate_range(path, int(moff[m1]), int(moff[mt]) if mt < len(moff) else fsize, buf.data_ptr() + own_len, cap)
                    lo = g0 - int(gpos[m0])
                    if n == 0:
                        shard = torch.empty(16, dtype=torch.uint8, device=dev)
                    elif lo % 16 == 0:
                        shard = buf[lo:lo + n]                        # (where it lies: the kernels want 16-byte alignment, no more)
                    else:
                        # moved down inside the buffer to the next lower multiple of 16, front to back through a 64 MiB piece
                        # (a copy of the whole shard would hold it twice; a piece's destination ends before the next piece's source)
                        al = lo - lo % 16
                        step = 64 << 20
                        for o in range(0, n, step):
                            m = min(step, n - o)
                            buf[al + o:al + o + m].copy_(buf[lo + o:lo + o + m].clone())
                        shard = buf[al:al + n]
                else:
                    if tail_len:
                        mt = int(np.searchsorted(gpos, g1, side="left"))
                        own = own + _bgzf_member_bytes(path, moff[m1], ends[mt - 1])
                    lo = g0 - int(gpos[m0])
                    data = own[lo:lo + n]
            except Exception as e:                 # noqa: BLE001 -- raised on every rank by _agree
                err = e
            _agree(err, where)

    if use_gpu:
        eng.set_index(barcodes, tags, cutsite)
        eng.set_option("progress", 1 if progress else 0)
        total = torch.zeros(len(barcodes) * len(tags), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        terms = eng.count_lines_device(shard.data_ptr(), int(n)) if n else 0
        first_line = first_line_of(terms, dev)
        eng.bind_counts(total.data_ptr())
        err, st = None, None
        try:
            try:
                if n and (first_line + 2) // 4 < bound:
                    eng.count_device(shard.data_ptr(), int(n), first_line=first_line, maxreads=bound)
                st = eng.stats()                               # (synchronises; raises what a kernel flagged)
            except Exception as e:             # noqa: BLE001 -- raised on every rank by _agree
                err = e
            _agree(err, dev)
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
        err = None
        try:
            if n and (first_line + 2) // 4 < bound:     # sequence lines (index 1 mod 4) below first_line: (first_line + 2) // 4
                out += np.asarray(counter(bytes(data), barcodes, tags, cutsite, first_line, bound), dtype=np.int64).reshape(out.shape)
        except Exception as e:                 # noqa: BLE001 -- raised on every rank by _agree
            err = e
        _agree(err, "cpu")
        if world > 1:
            dist.all_reduce(torch.from_numpy(out), op=dist.ReduceOp.SUM)
    return out if as_array else out.tolist()
