"""Progress lines of find_tags_fastq (reference tagdigger_fun.py:268-271): the file name after every 1 000 000
reads, `Reads: .. With barcode and cut site: .. With tag: ..` after every 50 000.  tests/golden/progress.json
holds what the REAL reference printed (tests/golden/make_progress_golden.py) on four inputs that seeded
generators rebuild here; the device keeps the two counters per window of 50 000 reads (td_get_progress)."""
import contextlib
import gzip
import hashlib
import io
import json
import os
import random

import numpy as np
import pytest

import helpers
from helpers import DEFAULT_MODE, KERNEL_MODES, apply_mode, mode_id
from oracle import c_oracle

_ALL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "progress.json")))
GOLD = [c for c in _ALL if "splitter" not in c]
SPLIT = [c for c in _ALL if "splitter" in c]
_bytes = {}


def case_bytes(case):
    """The input the reference read, rebuilt from its recipe (checked against the recorded SHA-256)."""
    if case["name"] not in _bytes:
        r = case["recipe"]
        if r["kind"] == "synth":
            from tagdigger_amd.synth import SynthConfig
            cfg = SynthConfig(**{k: tuple(v) if k == "bclen" else v for k, v in r["config"].items()})
            assert cfg.barcodes == case["barcodes"] and cfg.tags == case["tags"]
            raw = helpers.synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
        else:
            rnd = random.Random(r["seed"])
            barcodes, tags, cutsites = helpers.small_index(rnd, r.get("cutsite", "TGCAG"), nbar=r["nbar"], ntag=r["ntag"])
            assert barcodes == case["barcodes"] and tags == case["tags"]
            raw = helpers.dirty_fastq(rnd, barcodes, tags, cutsites, r["nrec"], nl_choices=tuple(r["nl_choices"]),
                                      long_lines=r.get("long_lines", False), permanent_shifts=r.get("permanent_shifts", False))
        assert hashlib.sha256(raw).hexdigest() == case["sha256"], "the generators drifted from the recorded input"
        _bytes[case["name"]] = raw
    return _bytes[case["name"]]


def printed_numbers(case):
    return [tuple(int(x) for x in ln.replace("Reads: ", "").replace(" With barcode and cut site: ", " ").replace(" With tag: ", " ").split())
            for ln in case["stdout"] if ln.startswith("Reads: ")]


@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
def test_oracle_counters_at_the_boundaries_equal_what_the_reference_printed(case):
    """Pins the checker the other tests use: the C oracle stopped at read 50 000 k has the reference's printed counters."""
    raw = case_bytes(case)
    o = c_oracle.COracle(case["barcodes"], case["tags"], case["kwargs"]["cutsite"])
    nums = printed_numbers(case)
    for reads, bar, tag in nums[:: max(1, len(nums) // 3)] + nums[-1:]:
        st = {}
        o.count_bytes(raw, maxreads=reads, stats=st)
        assert (st["reads"], st["barcut"], st["tag"]) == (reads, bar, tag)
    assert [ln for ln in case["stdout"] if not ln.startswith("Reads: ")] == [case["file"]] * (nums[-1][0] // 1000000)


MODES = [DEFAULT_MODE, KERNEL_MODES[2], KERNEL_MODES[1], KERNEL_MODES[3], KERNEL_MODES[5]]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES, ids=mode_id)
@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
def test_find_tags_fastq_prints_what_the_reference_prints(case, mode, tmp_path):
    from tagdigger_amd import tagdigger_fun as tf
    raw = case_bytes(case)
    with open(tmp_path / case["file"], "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=1) if case["file"].endswith("gz") else raw)
    eng = tf.default_engine(0)
    old = os.getcwd()
    os.chdir(tmp_path)
    out = io.StringIO()
    try:
        apply_mode(eng, mode)
        with contextlib.redirect_stdout(out):
            counts = tf.find_tags_fastq(case["file"], case["barcodes"], case["tags"], **case["kwargs"])
    finally:
        apply_mode(eng, DEFAULT_MODE)
        os.chdir(old)
    assert counts == case["counts"]
    assert out.getvalue().splitlines() == case["stdout"]


def boundary_stats(o, host, nreads):
    want = []
    for k in range(1, nreads // 50000 + 1):
        st = {}
        o.count_bytes(host, maxreads=50000 * k, stats=st)
        want.append((st["barcut"], st["tag"]))
    return want


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [DEFAULT_MODE, KERNEL_MODES[1], KERNEL_MODES[5]], ids=mode_id)
def test_windows_at_config3_shape(mode):
    """384 barcodes x 100 000 tags, 420 000 reads resident in HBM, then the same bytes as a host buffer in staged
    pieces: the running sums of the windows are the oracle's counters at every 50 000th read."""
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig.from_id(3, nreads=420_000)
    host = helpers.synth_host_bytes(cfg, 0, cfg.nreads)
    o = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    want = boundary_stats(o, host, cfg.nreads)
    eng = tagdigger_amd.Engine(0)
    try:
        apply_mode(eng, mode)
        eng.set_option("progress", 1)
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        d = eng.dev_alloc(host.nbytes)
        eng.h2d(d, host.tobytes())
        for attempt in ("device", "host"):
            eng.reset()
            if attempt == "device":
                eng.count_device(d, host.nbytes)
            else:
                eng.count_bytes(host)
            w = eng.progress_windows()
            assert len(w) == -(-cfg.nreads // 50000)
            got = list(zip(np.cumsum([a for a, _ in w]).tolist(), np.cumsum([b for _, b in w]).tolist()))
            assert got[:len(want)] == want, (attempt, mode)
            st = eng.stats()
            assert got[-1] == (st["barcut"], st["tag"])
        eng.dev_free(d)
    finally:
        eng.close()


@pytest.mark.gpu
def test_windows_with_short_lines_and_a_bound(tmp_path):
    """Reads of a dozen bases (more wanted lines in a tile than its record holds: the fix-up pass counts them), and a
    maxreads bound inside the third staged piece of a larger file."""
    import tagdigger_amd
    rnd = random.Random(5)
    barcodes, tags, cutsites = helpers.small_index(rnd, "TGCAG", nbar=6, ntag=30, taglens=(4, 9))
    recs = []
    for i in range(260_000):
        b, t = rnd.choice(barcodes), rnd.choice(tags)
        seq = (b + t) if rnd.random() < 0.7 else "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 14)))
        recs.append("@%d\n%s\n+\n%s\n" % (i, seq, "I" * len(seq)))
    raw = "".join(recs).encode()
    o = c_oracle.COracle(barcodes, tags, "TGCAG")
    want = boundary_stats(o, raw, 260_000)
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_option("progress", 1)
        eng.set_index(barcodes, tags, "TGCAG")
        eng.count_bytes(raw)
        w = eng.progress_windows()
        got = list(zip(np.cumsum([a for a, _ in w]).tolist(), np.cumsum([b for _, b in w]).tolist()))
        assert got[:len(want)] == want
        # a bound in a late piece of a 3-piece file
        from tagdigger_amd.synth import SynthConfig
        cfg = SynthConfig.from_id(2, nreads=400_000)
        host = helpers.synth_host_bytes(cfg, 0, cfg.nreads)
        path = str(tmp_path / "three_pieces.fq")
        host.tofile(path)
        o2 = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
        want2 = boundary_stats(o2, host, 330_000)
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        eng.count_file(path, maxreads=330_000)
        lines = eng.progress_lines("three_pieces.fq")
        assert lines == ["Reads: {0} With barcode and cut site: {1} With tag: {2}".format(50000 * (k + 1), a, b)
                         for k, (a, b) in enumerate(want2)]
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPLIT, ids=[c["name"] for c in SPLIT])
def test_barcode_splitter_prints_what_the_reference_prints(case, tmp_path):
    """barcodeSplitter's own progress lines (:1357-1360: counted by the writer threads per window of 50 000 reads) and,
    while there, its output files at 120 000 reads (SHA-256 of what the reference wrote)."""
    from tagdigger_amd import tagdigger_fun as tf
    raw = case_bytes(case)
    with open(tmp_path / case["file"], "wb") as fh:
        fh.write(raw)
    outs = ["split_%d.fq" % k for k in range(len(case["barcodes"]))]
    old = os.getcwd()
    os.chdir(tmp_path)
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            tf.barcodeSplitter(case["file"], case["barcodes"], outs, cutsite=case["kwargs"]["cutsite"],
                               adapter=tf.adapters[case["splitter"]["adapter"]], maxreads=case["splitter"]["maxreads"])
        sums = [hashlib.sha256(open(o, "rb").read()).hexdigest() for o in outs]
    finally:
        os.chdir(old)
    assert sums == case["outputs_sha256"]
    assert out.getvalue().splitlines() == case["stdout"]


@pytest.mark.gpu
def test_cli_on_several_devices_prints_the_progress_lines_in_file_order(tmp_path, capfd):
    """--td-devices (one process per GPU; the same GPU twice here: the rehearsal form): two libraries counted on two
    ranks, their progress lines gathered and printed by rank 0 in the reference's order -- sorted file names, each
    file's lines together -- with the counters the C oracle has at every 50 000th read of that file."""
    import csv
    from tagdigger_amd import tagdigger_script
    from tagdigger_amd.synth import SynthConfig, merged_rows
    cfg = SynthConfig(nreads=290_000, nbar=8, nmarkers=50, seed=4321, cutsite="TGCAG", bclen=(4, 8))
    host = helpers.synth_host_bytes(cfg, 0, cfg.nreads)
    parts = {"lib_1.fq": host[:160_000 * cfg.record_bytes], "lib_2.fq": host[160_000 * cfg.record_bytes:]}
    o = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    want = []
    for name in sorted(parts):
        parts[name].tofile(str(tmp_path / name))
        n = len(parts[name]) // cfg.record_bytes
        for k, (bar, tag) in enumerate(boundary_stats(o, parts[name], n)):
            want.append("Reads: {0} With barcode and cut site: {1} With tag: {2}".format(50000 * (k + 1), bar, tag))
    with open(tmp_path / "key.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["File", "Barcode", "Sample"])
        for name in sorted(parts):
            for k, b in enumerate(cfg.barcodes):
                w.writerow([name, b, "S%d" % k])
    with open(tmp_path / "tags.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Marker name", "Tag sequence"])
        w.writerows(merged_rows(cfg.tags, len(cfg.tags) // 2, np.random.default_rng(1), 0))
    old = os.getcwd()
    try:
        tagdigger_script.main(["-c", cfg.cutsite, "--MergedTags", "tags.csv", "-b", "key.csv", "-o", "counts.csv", "-w", str(tmp_path),
                               "--td-devices", "0,0"])
    finally:
        os.chdir(old)
    got = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("Reads: ")]
    assert got == want and len(want) == 5


@pytest.mark.gpu
def test_progress_lines_with_tassel_tagcount(tmp_path, capsys):
    """tassel_tagcount=True (the exact kernel with the header's count=N as the weight, :251-253,:264-265): tagcount in
    the progress line still counts READS (:263), the matrix counts weights."""
    from tagdigger_amd import tagdigger_fun as tf
    rnd = random.Random(99)
    barcodes, tags, cutsites = helpers.small_index(rnd, "TGCAG", nbar=6, ntag=30)
    recs = []
    for i in range(120_000):
        b, t = rnd.choice(barcodes), rnd.choice(tags)
        u = rnd.random()
        seq = b + t + "ACGT" if u < 0.6 else b + "TGCAG" + "".join(rnd.choice("ACGT") for _ in range(30)) if u < 0.8 else \
            "".join(rnd.choice("ACGTN") for _ in range(50))
        recs.append("@t%d count=%d\n%s\n+\n%s\n" % (i, rnd.randint(1, 40), seq, "I" * len(seq)))
    raw = "".join(recs).encode()
    (tmp_path / "tassel.fq").write_bytes(raw)
    o = c_oracle.COracle(barcodes, tags, "TGCAG")
    want_lines = []
    for k in (1, 2):
        st = {}
        o.count_bytes(raw, maxreads=50000 * k, tassel_tagcount=True, stats=st)
        want_lines.append("Reads: {0} With barcode and cut site: {1} With tag: {2}".format(50000 * k, st["barcut"], st["tag"]))
    want = o.count_bytes(raw, tassel_tagcount=True)
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        got = tf.find_tags_fastq("tassel.fq", barcodes, tags, tassel_tagcount=True)
    finally:
        os.chdir(old)
    assert capsys.readouterr().out.splitlines() == want_lines
    assert got == [[int(v) for v in row] for row in want.astype("int64")]


def _sharded_worker(rank, world, port, path, barcodes, tags, cutsite, maxreads):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    multi.count_file_sharded(path, barcodes, tags, cutsite, maxreads=maxreads, device=torch.device("cuda", 0), progress=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("maxreads", [5e9, 170_000])
def test_byte_sharded_file_prints_the_progress_lines(tmp_path, capfd, maxreads):
    """One file over two ranks (gloo rehearsal on one GPU): each shard keeps its windows by global read ordinal, the
    windows are summed, rank 0 prints what one process would have printed."""
    import socket
    import torch.multiprocessing as mp
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=260_000, nbar=8, nmarkers=50, seed=2468, cutsite="TGCAG", bclen=(4, 8))
    host = helpers.synth_host_bytes(cfg, 0, cfg.nreads)
    path = str(tmp_path / "one_file.fq")
    host.tofile(path)
    o = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    n = int(min(maxreads, cfg.nreads))
    want = ["Reads: {0} With barcode and cut site: {1} With tag: {2}".format(50000 * (k + 1), a, b)
            for k, (a, b) in enumerate(boundary_stats(o, host, n))]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, path, cfg.barcodes, cfg.tags, cfg.cutsite, maxreads), nprocs=2, join=True)
    got = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("Reads: ")]
    assert got == want and len(want) == n // 50000


@pytest.mark.gpu
def test_windows_with_a_barcode_index_too_large_for_k_fast2():
    """900 barcodes x 4 concrete cut sites (RCATGY): the barcode blob (~45 KB) fits the exact kernel's LDS layout but
    not k_fast2's, even with 16 KiB tiles.  Round 2 then ran k_fast -- which keeps no progress records -- with the
    windows on, and k_resolve added uninitialised words to them; now the exact kernel counts when progress is on.
    Windows == the C oracle's counters at every 50 000th read; counts == the oracle's."""
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=160_000, nbar=900, nmarkers=400, seed=77, cutsite="RCATGY", bclen=(7, 10))
    host = helpers.synth_host_bytes(cfg, 0, cfg.nreads)
    o = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    want = boundary_stats(o, host, cfg.nreads)
    full = o.count_bytes(host)
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        for progress in (1, 0):
            eng.set_option("progress", progress)
            eng.reset()
            eng.count_bytes(host)
            assert (eng.counts_numpy() == full).all()
            if progress:
                w = eng.progress_windows()
                got = list(zip(np.cumsum([a for a, _ in w]).tolist(), np.cumsum([b for _, b in w]).tolist()))
                assert got[:len(want)] == want
    finally:
        eng.close()
