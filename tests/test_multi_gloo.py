"""The N > 1 path rehearsed on CPU: two processes over gloo, libraries dealt across ranks, one
all-reduce -- must equal combineReadCounts over the per-file matrices.  The CPU oracle stands in
for the per-rank GPU counter (there is no GPU here); the sharding and the reduction are the code
under test."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs(tmpdir):
    from helpers import dirty_fastq, small_index
    rnd = random.Random(42)
    _, tags, cutsites = small_index(rnd, "TGCAG", nbar=1, ntag=24)
    bckeys = {}
    for k, name in enumerate(["libC.fq", "libA.fq", "libB.fq"]):
        barcodes, _, _ = small_index(rnd, "TGCAG", nbar=4 + k, ntag=1)
        data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=300)
        path = os.path.join(tmpdir, name)
        open(path, "wb").write(data)
        samples = ["s%d" % ((i * 2 + k) % 5) for i in range(len(barcodes))]   # names shared across files
        bckeys[path] = [barcodes, samples]
    return bckeys, tags


def _oracle_counter(f, barcodes, tags, cutsite):
    from oracle import c_oracle
    return c_oracle.find_tags_fastq(f, barcodes, tags, cutsite=cutsite)


def _worker(rank, world, port, bckeys, tags, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_libraries(bckeys, tags, "TGCAG", counter=_oracle_counter)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_combine_read_counts(tmp_path):
    from tagdigger_amd import tagdigger_fun as tf
    bckeys, tags = _make_inputs(str(tmp_path))
    countsdict = {f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}
    want = tf.combineReadCounts(countsdict, bckeys)
    assert sum(map(sum, want[1])) > 100
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), bckeys, tags, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got == want


def test_single_process_path(tmp_path):
    from tagdigger_amd import multi, tagdigger_fun as tf
    bckeys, tags = _make_inputs(str(tmp_path))
    countsdict = {f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}
    assert multi.count_libraries(bckeys, tags, "TGCAG", counter=_oracle_counter) == tf.combineReadCounts(countsdict, bckeys)


# ---------------------------------------------------------------- one file, byte-sharded
def _oracle_shard_counter(data, barcodes, tags, cutsite, first_line, maxreads):
    from oracle import c_oracle
    return c_oracle.COracle(barcodes, tags, cutsite).count_bytes(data, maxreads=maxreads, first_line=first_line).tolist()


def _dirty_file(tmp_path, style, seed=7):
    from helpers import dirty_fastq, small_index
    rnd = random.Random(seed)
    barcodes, tags, cutsites = small_index(rnd, "TGCAG", nbar=5, ntag=30)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=400)
    if style == "crlf":
        data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n").replace(b"\n", b"\r\n")
    elif style == "cr":
        data = data.replace(b"\r\n", b"\n").replace(b"\n", b"\r")
    elif style == "nofinal":
        data = data.rstrip(b"\r\n")
    path = str(tmp_path / ("one_%s.fq" % style))
    open(path, "wb").write(data)
    return path, data, barcodes, tags


@pytest.mark.parametrize("style", ["mixed", "crlf", "cr", "nofinal"])
def test_shard_bounds_and_line_index(tmp_path, style):
    """Shards tile the file, start at line starts, and counted with their own first line index
    add up to the whole file's matrix -- for every number of ranks."""
    from oracle import c_oracle
    from tagdigger_amd import multi
    path, data, barcodes, tags = _dirty_file(tmp_path, style)
    ora = c_oracle.COracle(barcodes, tags, "TGCAG")
    want = ora.count_bytes(data)
    assert want.sum() > 50
    for world in (1, 2, 3, 5, 16):
        bounds = multi.shard_bounds(path, world)
        assert bounds[0][0] == 0 and bounds[-1][1] == len(data)
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        total = 0 * want
        first_line = 0
        for a, b in bounds:
            if a > 0 and a < len(data):
                assert data[a - 1:a] in (b"\n", b"\r") and not (data[a - 1:a] == b"\r" and data[a:a + 1] == b"\n")
            piece = data[a:b]
            if piece:
                total = total + c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(piece, first_line=first_line)
            first_line += multi.count_terminators(piece) if piece else 0
        assert (total == want).all(), world


def _shard_worker(rank, world, port, path, barcodes, tags, maxreads, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_file_sharded(path, barcodes, tags, "TGCAG", maxreads=maxreads, counter=_oracle_shard_counter)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("maxreads", [5e9, 150, 3, 250, 300, 399])   # 250-399: the bound falls inside the second shard
def test_two_ranks_shard_one_file(tmp_path, maxreads):
    from oracle import c_oracle
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=11)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_shard_worker, args=(2, _free_port(), path, barcodes, tags, maxreads, out), nprocs=2, join=True)
    assert torch.load(out) == want


# ---------------------------------------------------------------- the device-resident path (GPU box)
def _device_worker(rank, world, port, bckeys, tags, out):
    """count_libraries with the product's own counter: this rank's GPU counts, K3 folds on the device, the
    [samples x tags] device tensor is all-reduced (gloo here: both ranks rehearse on GPU 0)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_libraries(bckeys, tags, "TGCAG", device=torch.device("cuda", 0))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_device_path_equals_combine_read_counts(tmp_path):
    from tagdigger_amd import multi, tagdigger_fun as tf
    bckeys, tags = _make_inputs(str(tmp_path))
    countsdict = {f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}
    want = tf.combineReadCounts(countsdict, bckeys)
    # one process: every library on GPU 0, folded on the device, no collective
    assert multi.count_libraries(bckeys, tags, "TGCAG", device=0) == want
    got = multi.count_libraries(bckeys, tags, "TGCAG", device=torch.device("cuda", 0), as_array=True)
    assert got[0] == want[0] and got[1].tolist() == want[1]
    # two ranks
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_worker, args=(2, _free_port(), bckeys, tags, out), nprocs=2, join=True)
    assert torch.load(out) == want


def _device_shard_worker(rank, world, port, path, barcodes, tags, maxreads, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_file_sharded(path, barcodes, tags, "TGCAG", maxreads=maxreads, device=torch.device("cuda", 0))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("maxreads", [5e9, 150, 300])
def test_device_path_byte_sharded_file(tmp_path, maxreads):
    """One file over two ranks, the shard's terminators counted on the device (td_count_lines_device), the global
    maxreads bound applied in the shard it falls into."""
    from oracle import c_oracle
    from tagdigger_amd import multi
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=11)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    assert multi.count_file_sharded(path, barcodes, tags, "TGCAG", maxreads=maxreads, device=0) == want
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_shard_worker, args=(2, _free_port(), path, barcodes, tags, maxreads, out), nprocs=2, join=True)
    assert torch.load(out) == want


@pytest.mark.gpu
def test_fold_rows_after_a_flush_to_the_host_accumulator():
    """A library of more than ~17 GB moves part of its uint32 counts into the host accumulator (launch_count's flush);
    K3 must fold that part too (round 2's td_fold_rows refused it).  The threshold is lowered through the test option
    flush_limit so that the second of three launches flushes the first one's counts."""
    import numpy as np
    import tagdigger_amd
    from helpers import dirty_fastq, small_index
    from oracle import c_oracle
    rnd = random.Random(7)
    barcodes, tags, cutsites = small_index(rnd, "TGCAG", nbar=6, ntag=20)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=400)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
    rows = [0, 2, 1, 0, 2, 1]
    folded = np.zeros((3, len(tags)), dtype=np.int64)
    np.add.at(folded, rows, want.astype(np.int64))
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_index(barcodes, tags, "TGCAG")
        eng.set_option("flush_limit", 1000)
        for _ in range(3):
            eng.count_bytes(data)
        assert (eng.counts_numpy() == 3 * want).all()              # (td_get_counts merges the accumulator)
        total = torch.zeros((3, len(tags)), dtype=torch.int32, device="cuda:0")
        eng.fold_rows(rows, total.data_ptr(), 3)
        assert (total.cpu().numpy().astype(np.int64) == 3 * folded).all()
        eng.set_option("flush_limit", 0)
    finally:
        eng.close()


# ---------------------------------------------------------------- one BGZF file over the ranks (members, not bytes)
def _bgzf_file(tmp_path, style, block, seed=7):
    from helpers import bgzf_bytes
    path, data, barcodes, tags = _dirty_file(tmp_path, style, seed=seed)
    gz = path + ".gz"
    open(gz, "wb").write(bgzf_bytes(data, block=block, level=6))
    return gz, data, barcodes, tags


@pytest.mark.parametrize("style,block", [("mixed", 4096), ("crlf", 997), ("cr", 1500), ("nofinal", 0xFF00), ("mixed", 61)])
def test_bgzf_member_shards_tile_the_file_at_line_starts(tmp_path, style, block):
    """The pieces of the ownership rule, for every number of ranks: a rank's lines run from the first line start in its
    members to the first line start behind them (members do not end at line ends; with 61-byte members most ranks'
    members hold no line start at all); the ranges tile the inflated file, begin at line starts, and counted with
    their own first line index add up to the whole file's matrix."""
    from oracle import c_oracle
    from tagdigger_amd import multi
    gz, data, barcodes, tags = _bgzf_file(tmp_path, style, block)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
    moff, misz = multi.bgzf_index(gz)
    assert int(misz.astype("int64").sum()) == len(data) and misz[-1] == 0            # (the end-of-file member)
    import numpy as np
    gpos = np.concatenate(([0], np.cumsum(misz.astype(np.int64))))
    fsize = os.path.getsize(gz)
    for world in (1, 2, 3, 7, 16):
        ranges = multi.bgzf_member_ranges(moff, fsize, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(moff) and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        starts = []
        for m0, m1 in ranges:
            own = data[int(gpos[m0]):int(gpos[m1])]
            prev = data[int(gpos[m0]) - 1] if gpos[m0] > 0 else None
            a = multi.first_line_start(own, prev)
            starts.append(int(gpos[m0]) + a if a < len(own) else -1)
        starts.append(len(data))
        for r in range(world - 1, -1, -1):
            if starts[r] < 0:
                starts[r] = starts[r + 1]
        assert starts[0] == 0
        total, first_line = 0 * want, 0
        for r in range(world):
            a, b = starts[r], starts[r + 1]
            assert a <= b
            if 0 < a < len(data) and a < b:
                assert data[a - 1:a] in (b"\n", b"\r") and not (data[a - 1:a] == b"\r" and data[a:a + 1] == b"\n")
            piece = data[a:b]
            if piece:
                total = total + c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(piece, first_line=first_line)
                first_line += multi.count_terminators(piece)
        assert (total == want).all(), world


def _bgzf_worker(rank, world, port, path, barcodes, tags, maxreads, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_file_sharded(path, barcodes, tags, "TGCAG", maxreads=maxreads, counter=_oracle_shard_counter)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,block,maxreads", [(2, 4096, 5e9), (2, 997, 250), (3, 61, 5e9), (3, 1500, 150)])
def test_ranks_shard_one_bgzf_file(tmp_path, world, block, maxreads):
    """count_file_sharded on a BGZF file (CPU stand-in for the counter, members inflated with zlib): the whole
    file's matrix for any number of ranks, the maxreads bound applied in the shard it falls into."""
    from oracle import c_oracle
    gz, data, barcodes, tags = _bgzf_file(tmp_path, "mixed", block, seed=11)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_bgzf_worker, args=(world, _free_port(), gz, barcodes, tags, maxreads, out), nprocs=world, join=True)
    assert torch.load(out) == want


@pytest.mark.parametrize("world,maxreads", [(1, 5e9), (2, 5e9), (3, 200)])
def test_an_ordinary_gzip_file_is_counted_by_rank_0(tmp_path, world, maxreads):
    """A gzip file that is not BGZF has no place to cut it: count_file_sharded counts it all the same (the reference reads
    any .gz by name, :240-241) -- rank 0 alone, the other ranks adding zeros to the same all-reduce."""
    import gzip
    from oracle import c_oracle
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=5)
    gz = path + ".gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(data)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    if world == 1:
        from tagdigger_amd import multi
        assert multi.count_file_sharded(gz, barcodes, tags, "TGCAG", maxreads=maxreads, counter=_oracle_shard_counter) == want
        return
    out = str(tmp_path / "res.pt")
    mp.spawn(_bgzf_worker, args=(world, _free_port(), gz, barcodes, tags, maxreads, out), nprocs=world, join=True)
    assert torch.load(out) == want


def _device_bgzf_worker(rank, world, port, path, barcodes, tags, maxreads, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_file_sharded(path, barcodes, tags, "TGCAG", maxreads=maxreads, device=torch.device("cuda", 0))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("block,maxreads", [(4096, 5e9), (997, 300), (61, 5e9)])
def test_device_path_member_sharded_bgzf_file(tmp_path, block, maxreads):
    """The product path on a BGZF file: the rank's members inflated on the GPU into device memory
    (td_bgzf_inflate_range), its first line start found there, the lines' tail fetched from the members behind its
    own, counted in place; one process, then two ranks rehearsing on GPU 0."""
    from oracle import c_oracle
    from tagdigger_amd import multi
    gz, data, barcodes, tags = _bgzf_file(tmp_path, "mixed", block, seed=11)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    assert multi.count_file_sharded(gz, barcodes, tags, "TGCAG", maxreads=maxreads, device=0) == want
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_bgzf_worker, args=(2, _free_port(), gz, barcodes, tags, maxreads, out), nprocs=2, join=True)
    assert torch.load(out) == want


@pytest.mark.gpu
@pytest.mark.parametrize("maxreads", [5e9, 300])
def test_device_path_counts_an_ordinary_gzip_file_on_rank_0(tmp_path, maxreads, capfd):
    """The product path on a gzip file that is not BGZF: rank 0 counts it through td_count_file, the other rank adds zeros;
    one process, then two ranks rehearsing on GPU 0 (rank 0 says on stderr that it reads the file alone)."""
    import gzip
    from oracle import c_oracle
    from tagdigger_amd import multi
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=5)
    gz = path + ".gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(data)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    assert multi.count_file_sharded(gz, barcodes, tags, "TGCAG", maxreads=maxreads, device=0) == want
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_bgzf_worker, args=(2, _free_port(), gz, barcodes, tags, maxreads, out), nprocs=2, join=True)
    assert torch.load(out) == want


@pytest.mark.gpu
def test_load_file_range_streams_through_two_pinned_pieces(tmp_path):
    """td_load_file_range: a byte range of a file into device memory (pieces of 32 MiB: this range takes three), byte
    for byte; the host side holds the staging pieces only."""
    import numpy as np
    import tagdigger_amd
    rng = np.random.default_rng(5)
    blob = rng.integers(0, 256, 80 * (1 << 20) + 12345, dtype=np.uint8)
    path = str(tmp_path / "blob.bin")
    blob.tofile(path)
    eng = tagdigger_amd.Engine(0)
    try:
        off, n = 4097, 70 * (1 << 20) + 777
        dst = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        eng.load_file_range(path, off, n, dst.data_ptr())
        assert bool((dst.cpu().numpy() == blob[off:off + n]).all())
        with pytest.raises(tagdigger_amd._binding.TagdigError):
            eng.load_file_range(path, blob.size - 10, 100, dst.data_ptr())      # (beyond the file's end)
    finally:
        eng.close()


@pytest.mark.gpu
def test_bgzf_inflate_range_rejects_what_is_not_a_member_start(tmp_path):
    """td_bgzf_inflate_range: an offset in the middle of a member, and a destination too small, fail loudly."""
    import tagdigger_amd
    from tagdigger_amd._binding import TagdigError
    gz, data, barcodes, tags = _bgzf_file(tmp_path, "mixed", 4096)
    eng = tagdigger_amd.Engine(0)
    try:
        dst = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
        assert eng.bgzf_inflate_range(gz, 0, os.path.getsize(gz), dst.data_ptr(), len(data)) == len(data)
        assert bytes(dst[:len(data)].cpu().numpy()) == data
        with pytest.raises(TagdigError):
            eng.bgzf_inflate_range(gz, 5, os.path.getsize(gz), dst.data_ptr(), len(data))
        with pytest.raises(TagdigError):
            eng.bgzf_inflate_range(gz, 0, os.path.getsize(gz), dst.data_ptr(), len(data) - 1)
    finally:
        eng.close()


# ---------------------------------------------------------------- the eight-rank shape (BASELINE configs 4 and 5)
# Eight ranks of this code before the first real 8-GPU run: the rank arithmetic (libraries dealt out, shard bounds, member
# ranges, the bound that falls into a late shard, ranks that own nothing) at the world size the driver launches.  Here with
# the CPU stand-in for the counter (world size 8); on the GPU box with the product's counter at world size 5 -- the pool
# admits six processes of one job on a card, the test runner being one -- and 8 libraries over those 5 ranks.
def _make_eight_libraries(tmpdir, nrec=120):
    """Config 4's shape in small: 8 libraries, each with its own barcodes, whose samples map onto shared sample names."""
    from helpers import dirty_fastq, small_index
    rnd = random.Random(4)
    _, tags, cutsites = small_index(rnd, "TGCAG", nbar=1, ntag=40)
    bckeys = {}
    for k in range(8):
        barcodes, _, _ = small_index(rnd, "TGCAG", nbar=6 + (k % 3), ntag=1)
        data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=nrec + 17 * k)
        path = os.path.join(tmpdir, "lib%d.fq" % k)
        open(path, "wb").write(data)
        bckeys[path] = [barcodes, ["sample%02d" % ((3 * i + k) % 11) for i in range(len(barcodes))]]
    return bckeys, tags


def test_eight_ranks_eight_libraries_equal_combine_read_counts(tmp_path):
    from tagdigger_amd import tagdigger_fun as tf
    bckeys, tags = _make_eight_libraries(str(tmp_path))
    want = tf.combineReadCounts({f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}, bckeys)
    assert len(want[0]) == 11 and sum(map(sum, want[1])) > 300
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(8, _free_port(), bckeys, tags, out), nprocs=8, join=True)
    assert torch.load(out) == want


@pytest.mark.parametrize("maxreads", [5e9, 700])            # 700: the bound falls into the sixth of eight shards
def test_eight_ranks_shard_one_file(tmp_path, maxreads):
    from oracle import c_oracle
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=23)
    data = data * 3
    open(path, "wb").write(data)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_shard_worker, args=(8, _free_port(), path, barcodes, tags, maxreads, out), nprocs=8, join=True)
    assert torch.load(out) == want


@pytest.mark.parametrize("block,maxreads", [(4096, 5e9), (997, 700), (61, 5e9)])
def test_eight_ranks_shard_one_bgzf_file(tmp_path, block, maxreads):
    from oracle import c_oracle
    gz, data, barcodes, tags = _bgzf_file(tmp_path, "mixed", block, seed=23)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_bgzf_worker, args=(8, _free_port(), gz, barcodes, tags, maxreads, out), nprocs=8, join=True)
    assert torch.load(out) == want


def _config5_shard_counter(data, barcodes, tags, cutsite, first_line, maxreads):
    from oracle import c_oracle
    return c_oracle.COracle(barcodes, tags, cutsite).count_bytes(data, first_line=first_line, maxreads=maxreads)


def _config5_worker(rank, world, port, path, barcodes, tags, cutsite, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_file_sharded(path, barcodes, tags, cutsite, counter=_config5_shard_counter)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_the_config5_stream(tmp_path):
    """BASELINE config 5's stream -- CWGC, barcodes of 4-10 bp, tri-allelic markers, adapter read-through -- byte-sharded
    over eight ranks."""
    from helpers import synth_host_bytes
    from oracle import c_oracle
    from tagdigger_amd.synth import CONFIGS, SynthConfig
    cfg = SynthConfig(**dict(CONFIGS[5], nreads=20_000, nbar=24, nmarkers=300))
    data = synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
    path = str(tmp_path / "c5.fq")
    open(path, "wb").write(data)
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(data).tolist()
    assert sum(map(sum, want)) > 5000
    out = str(tmp_path / "res.pt")
    mp.spawn(_config5_worker, args=(8, _free_port(), path, list(cfg.barcodes), list(cfg.tags), cfg.cutsite, out), nprocs=8, join=True)
    assert torch.load(out) == want


GPU_REHEARSAL_RANKS = 5            # (the pool admits six processes of one job on a card: the test runner itself is one of them)


@pytest.mark.gpu
def test_five_ranks_on_one_gpu_eight_libraries(tmp_path):
    """Config 4's shape on the device path: 8 libraries over 5 ranks rehearsing on GPU 0 (three ranks take two libraries),
    K3 fold on the device, ONE all-reduce of the [samples x tags] device tensor == combineReadCounts."""
    from tagdigger_amd import tagdigger_fun as tf
    bckeys, tags = _make_eight_libraries(str(tmp_path), nrec=400)
    want = tf.combineReadCounts({f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}, bckeys)
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_worker, args=(GPU_REHEARSAL_RANKS, _free_port(), bckeys, tags, out), nprocs=GPU_REHEARSAL_RANKS, join=True)
    assert torch.load(out) == want


@pytest.mark.gpu
@pytest.mark.parametrize("maxreads", [5e9, 700])
def test_five_ranks_on_one_gpu_byte_sharded_file(tmp_path, maxreads):
    from oracle import c_oracle
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=23)
    data = data * 3
    open(path, "wb").write(data)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data, maxreads=maxreads).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_shard_worker, args=(GPU_REHEARSAL_RANKS, _free_port(), path, barcodes, tags, maxreads, out), nprocs=GPU_REHEARSAL_RANKS, join=True)
    assert torch.load(out) == want


@pytest.mark.gpu
def test_five_ranks_on_one_gpu_member_sharded_bgzf_file(tmp_path):
    from oracle import c_oracle
    gz, data, barcodes, tags = _bgzf_file(tmp_path, "mixed", 997, seed=23)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_device_bgzf_worker, args=(GPU_REHEARSAL_RANKS, _free_port(), gz, barcodes, tags, 5e9, out), nprocs=GPU_REHEARSAL_RANKS, join=True)
    assert torch.load(out) == want


# ---------------------------------------------------------------- a rank that fails takes the job down, not into a hang
def _failing_worker(rank, world, port, path, barcodes, tags, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi

    def counter(data, b, t, cutsite, first_line, maxreads):
        if rank == 1:
            raise OSError("rank 1 cannot read its share")
        return _oracle_shard_counter(data, b, t, cutsite, first_line, maxreads)
    try:
        multi.count_file_sharded(path, barcodes, tags, "TGCAG", counter=counter)
        verdict = "returned"
    except OSError as e:
        verdict = "OSError: %s" % e
    except RuntimeError as e:
        verdict = "RuntimeError: %s" % e
    with open("%s.%d" % (out, rank), "w") as fh:
        fh.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_raises_on_every_rank(tmp_path):
    """One rank's share fails (I/O, a damaged member): that rank raises its exception, the others raise too -- before, they
    waited forever in the all-reduce."""
    path, data, barcodes, tags = _dirty_file(tmp_path, "mixed", seed=11)
    out = str(tmp_path / "verdict")
    mp.spawn(_failing_worker, args=(3, _free_port(), path, barcodes, tags, out), nprocs=3, join=True)
    got = [open("%s.%d" % (out, r)).read() for r in range(3)]
    assert got[1].startswith("OSError: rank 1 cannot read")
    assert got[0].startswith("RuntimeError") and got[2].startswith("RuntimeError")


def _gzip_shard_worker(rank, world, port, path, barcodes, tags, cutsite, maxreads, out, false_seams=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import io
    import contextlib
    from tagdigger_amd import multi, tagdigger_fun
    if false_seams:
        tagdigger_fun.default_engine(0).set_option("gz_gpu_false_every", 1)      # (every rank's start but the first is moved by 4099 bits)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        res = multi.count_file_sharded(path, barcodes, tags, cutsite, maxreads=maxreads, device=torch.device("cuda", 0))
    if rank == 0:
        torch.save((res, err.getvalue()), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,maxreads,style", [(2, 5e9, "lf"), (3, 123_457, "lf"), (GPU_REHEARSAL_RANKS, 5e9, "crlf"), (4, 5e9, "cr")])
def test_one_ordinary_gzip_file_over_the_ranks(tmp_path, world, maxreads, style):
    """ONE gzip member decoded by several ranks (multi._gzip_shard_text over td_gz_shard_*: every rank a byte range of the
    compressed file, symbols first, the windows through the ranks' maps, the CRC-32s joined, lines handed across the seams):
    the matrix of the whole file, with the bound inside a later rank's share, with CRLF and CR-only lines; no rank says that
    it reads the file alone."""
    import gzip
    from oracle import c_oracle
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes
    cfg = SynthConfig.from_id(2, nreads=200_000)
    raw = synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
    if style == "crlf":
        raw = raw.replace(b"\n", b"\r\n")
    elif style == "cr":
        raw = raw.replace(b"\n", b"\r")
    gz = str(tmp_path / "lib.fq.gz")
    with open(gz, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=6))
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw, maxreads=int(min(maxreads, 10 ** 12))).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_gzip_shard_worker, args=(world, _free_port(), gz, list(cfg.barcodes), list(cfg.tags), cfg.cutsite, maxreads, out), nprocs=world, join=True)
    res, err = torch.load(out)
    assert res == want
    assert "reads it alone" not in err, err


@pytest.mark.gpu
def test_gzip_files_the_ranks_cannot_share_go_to_rank_0(tmp_path):
    """Two members, and a damaged member: the sharded scheme declines on every rank alike and rank 0 counts the file (or
    raises what gzip.open raises) as before."""
    import gzip
    from oracle import c_oracle
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes
    cfg = SynthConfig.from_id(2, nreads=120_000)
    raw = synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
    half = raw.index(b"\n@", len(raw) // 2) + 1
    gz = str(tmp_path / "two.fq.gz")
    with open(gz, "wb") as fh:
        fh.write(gzip.compress(raw[:half], compresslevel=6) + gzip.compress(raw[half:], compresslevel=1))
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_gzip_shard_worker, args=(3, _free_port(), gz, list(cfg.barcodes), list(cfg.tags), cfg.cutsite, 5e9, out), nprocs=3, join=True)
    res, err = torch.load(out)
    assert res == want and "reads it alone" in err


@pytest.mark.gpu
def test_a_false_block_start_at_a_seam_sends_the_file_to_rank_0(tmp_path):
    """A rank whose first block start is none: the rank before it ends elsewhere, the seam does not close, every rank sees
    that in the gathered positions and rank 0 counts the file alone -- the matrix is the same."""
    import gzip
    from oracle import c_oracle
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes
    cfg = SynthConfig.from_id(2, nreads=150_000)
    raw = synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
    gz = str(tmp_path / "lib.fq.gz")
    with open(gz, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=6))
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw).tolist()
    out = str(tmp_path / "res.pt")
    mp.spawn(_gzip_shard_worker, args=(3, _free_port(), gz, list(cfg.barcodes), list(cfg.tags), cfg.cutsite, 5e9, out, True), nprocs=3, join=True)
    res, err = torch.load(out)
    assert res == want and "reads it alone" in err
