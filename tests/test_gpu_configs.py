"""Every BASELINE.json config's INDEX SHAPE under `pytest -m gpu`, against the C oracle on identical bytes.

The goldens stop at 12 barcodes x 30 tags; the regimes that define configs 2-5 -- 96 x 10 k,
384 x 100 k (8 MB tag table > one XCD's L2, overflow buckets, 154 MB matrix), 384 x 500 k (34 MB
table, 768 MB matrix), `CWGC` x 384 x 100 k with barcodes of 4-10 bp -- are exercised here with
1-2 M reads of the canonical stream each: the whole matrix and the three counters
of reference tagdigger_fun.py:246-248 must equal the oracle's, for the free-running kernel at both
tile sizes and for the exact look-back kernel.  Config 3's shape also goes through the file path
(plain, gzip, BGZF: several staged pieces), and one case forces the matrix-size fallback.
"""
import gzip
import os

import numpy as np
import pytest

from helpers import DEFAULT_MODE, KERNEL_MODES, apply_mode, bgzf_bytes, synth_host_bytes
from oracle import c_oracle

pytestmark = pytest.mark.gpu

# name -> (config id of tagdigger_amd.synth.CONFIGS, reads in the sample, kernel modes (helpers.KERNEL_MODES))
SHAPES = {
    "C2_96x10k": (2, 1_000_000, KERNEL_MODES),
    "C3_384x100k": (3, 2_000_000, KERNEL_MODES),
    "C4_384x500k": (4, 1_000_000, [KERNEL_MODES[0], KERNEL_MODES[3], KERNEL_MODES[5], KERNEL_MODES[6]]),
    "C5_CWGC_384x100k": (5, 1_000_000, KERNEL_MODES),
}


@pytest.fixture(scope="module")
def eng():
    import tagdigger_amd
    e = tagdigger_amd.Engine(0)
    yield e
    e.close()


_cache = {}


def shape(name):
    """(config, host bytes, oracle matrix, oracle stats) of a shape, built once per session (3.3 GB of host
    memory in all, config 4's uint64 matrix being half of it)."""
    if name not in _cache:
        from tagdigger_amd.synth import SynthConfig
        cid, nreads, _ = SHAPES[name]
        cfg = SynthConfig.from_id(cid, nreads=nreads)
        host = synth_host_bytes(cfg, 0, nreads)
        ost = {}
        want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(host, stats=ost)
        _cache[name] = (cfg, host, want, ost)
    return _cache[name]


def check(eng, want, ost, what):
    got = eng.counts_numpy()
    st = eng.stats()
    assert (got == want).all(), what
    assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), what


@pytest.mark.parametrize("name", list(SHAPES))
def test_config_shape_device_resident(eng, name):
    """The generator's bytes written straight into HBM (== the host reference bytes), counted in place."""
    cfg, host, want, ost = shape(name)
    assert want.sum() > 0.6 * cfg.nreads                    # ~70 % of the stream are hits
    nb = cfg.nbytes()
    d = eng.dev_alloc(nb)
    try:
        cfg.fill_device(eng, d, 0, cfg.nreads)
        assert eng.d2h(d, 1 << 20) == bytes(host[:1 << 20]) and eng.d2h(d + nb - 4096, 4096) == bytes(host[-4096:])
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        try:
            for mode in SHAPES[name][2]:
                apply_mode(eng, mode)
                eng.reset()
                eng.count_device(d, nb)
                check(eng, want, ost, (name, mode))
                if mode["fastpath"]:
                    # (k_fast2 leaves the buffer's first tile and the one or two whose window crosses its end to the fix-up pass)
                    assert eng.debug_counters()[11] <= (3 if mode.get("kernel", 1) >= 2 else 0), "well-formed FASTQ must leave the fix-up queue empty"
        finally:
            apply_mode(eng, DEFAULT_MODE)
    finally:
        eng.dev_free(d)


@pytest.mark.parametrize("s", [1.0, 1.5])
def test_skewed_hits(eng, s):
    """The Zipf variant of the stream (hot cells: many atomics on few addresses): device bytes == host
    reference bytes, counts == oracle == the generator's own expected matrix."""
    from helpers import synth_expected
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=600_000, nbar=96, nmarkers=5_000, seed=21, skew=s)
    nb = cfg.nbytes()
    host = synth_host_bytes(cfg, 0, cfg.nreads)
    ost = {}
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(host, stats=ost)
    exp, hits = synth_expected(cfg, 0, cfg.nreads)
    assert (want == exp).all() and want.max() > 50 * want.mean()
    d = eng.dev_alloc(nb)
    try:
        cfg.fill_device(eng, d, 0, cfg.nreads)
        assert eng.d2h(d, nb) == host.tobytes()
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        eng.count_device(d, nb)
        check(eng, want, ost, ("skew", s))
        # the same with the hot-cell cache off, and with a cache that never rests (a measurement setting)
        try:
            for hc in (0, 2):
                eng.set_option("hot_cache", hc)
                eng.reset()
                eng.count_device(d, nb)
                check(eng, want, ost, ("skew", s, "hot_cache", hc))
        finally:
            eng.set_option("hot_cache", 1)
        # and the device-side expected matrix the bench checks against
        dw = eng.dev_alloc(want.size * 4)
        try:
            eng.h2d(dw, bytes(want.size * 4))
            assert cfg.expected_device(eng, dw, 0, cfg.nreads) == hits
            assert (np.frombuffer(eng.d2h(dw, want.size * 4), dtype=np.uint32).reshape(want.shape) == want).all()
        finally:
            eng.dev_free(dw)
    finally:
        eng.dev_free(d)


def test_matrix_size_fallback(eng):
    """The free-running kernel addresses count cells as base + 32-bit offset, so matrices of 4 GiB and
    more go to the exact kernel (tagdig.hip launch_count).  The threshold is lowered to force that switch
    on config 2's shape; the fix-up queue counter shows which kernel ran."""
    cfg, host, want, ost = shape("C2_96x10k")
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    try:
        eng.set_option("fast_max_matrix_bytes", 1 << 20)      # 96 x 10 k x 4 B = 3.84 MB is above it
        eng.count_bytes(host)
        check(eng, want, ost, "fallback")
        eng.set_option("fast_max_matrix_bytes", 0)            # back to the built-in 4 GiB
        eng.reset()
        eng.count_bytes(host)
        check(eng, want, ost, "fast again")
    finally:
        eng.set_option("fast_max_matrix_bytes", 0)


@pytest.mark.parametrize("kind", ["plain", "gz", "bgzf"])
def test_config3_shape_through_the_file_path(eng, tmp_path, kind):
    """td_count_file at config 3's index shape: 438 MB of FASTQ = 14 staged pieces of 32 MiB, read
    plain, inflated from ordinary gzip (chunk-parallel decoder) and from BGZF (member-parallel)."""
    cfg, host, want, ost = shape("C3_384x100k")
    raw = host.tobytes()
    if kind == "plain":
        path, blob = str(tmp_path / "c3.fq"), raw
    elif kind == "gz":
        path, blob = str(tmp_path / "c3.fq.gz"), gzip.compress(raw, compresslevel=1)
    else:
        path, blob = str(tmp_path / "c3.bgzf.fq.gz"), bgzf_bytes(raw, level=1)
    with open(path, "wb") as fh:
        fh.write(blob)
    del blob, raw
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_file(path)
    check(eng, want, ost, kind)
    os.unlink(path)


@pytest.mark.parametrize("kind,maxreads", [("plain", 150_001), ("gz", 150_001), ("plain", 1_999_999), ("bgzf", 150_001), ("bgzf", 5e9)])
def test_file_path_stops_reading_after_maxreads(eng, tmp_path, kind, maxreads):
    """maxreads inside the first of 14 staged pieces, and one read before the end: the host stops reading the file
    once a drained piece's line index has passed the bound (reference :272 breaks its loop); same matrix and the
    same three counters as the reference's loop either way."""
    cfg, host, _, _ = shape("C3_384x100k")
    ost = {}
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(host, maxreads=maxreads, stats=ost)
    raw = host.tobytes()
    path = str(tmp_path / ("c3.fq" if kind == "plain" else "c3.fq.gz"))
    with open(path, "wb") as fh:
        fh.write(raw if kind == "plain" else gzip.compress(raw, compresslevel=1) if kind == "gz" else bgzf_bytes(raw, level=1, threads=8))
    del raw
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    if kind == "bgzf":
        eng.set_option("zb_members", 1024)           # (seven GPU batches instead of one: the bound stops the loop after the first)
    try:
        eng.count_file(path, maxreads=maxreads)
    finally:
        eng.set_option("zb_members", 1 << 30)
    check(eng, want, ost, (kind, maxreads))
    assert ost["reads"] == min(maxreads, 2_000_000)
    os.unlink(path)


def test_config5_count_and_trim_over_one_buffer(eng):
    """BASELINE config 5 as SURVEY App. B writes it: degenerate cut site, barcodes of 4-10 bp, 5 % tri-allelic markers
    expanded by readTags_Merged, the common cutter's adapter read through in 20 % of the tag-bearing reads -- the
    counting path and the splitter's per-read branch over ONE resident buffer, against the two oracles (the C
    oracle for the matrix at 1 M reads, the Python restatement of barcodeSplitter for the first 60 k decisions).
    The reference's splitter takes a single ACGT cut site (tagdigger_fun.py:1292): the first concrete one."""
    import contextlib
    import io
    from oracle import tagdigger_oracle as po
    from tagdigger_amd import tagdigger_fun as tf
    cfg, host, want, ost = shape("C5_CWGC_384x100k")
    assert cfg.tag_names is not None and len(cfg.tags) > 2 * 50_000          # the third alleles are there
    adapter = [tuple(x) for x in tf.adapters["PstI-MspI-Hall"]]
    site = cfg.cutsites[0]
    through = sum(1 for i in range(0, 20_000) if b"CCGCTCAGGC" in host[i * 219 + 15:i * 219 + 116].tobytes())
    assert 0.10 * 20_000 < through < 0.18 * 20_000                           # 20 % of the 75 % tag-bearing reads
    with contextlib.redirect_stdout(io.StringIO()):
        ends = tf._adapter_ends(adapter, cfg.barcodes)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_splitter(cfg.barcodes, site, adapter[0][0].replace("^", ""), adapter[1][0].replace("^", ""), ends)
    nb = cfg.nbytes()
    d = eng.dev_alloc(nb)
    try:
        cfg.fill_device(eng, d, 0, cfg.nreads)
        res, terms = eng.count_and_split_device(d, nb)
    finally:
        eng.dev_free(d)
    check(eng, want, ost, "config 5: counts")
    assert terms == 4 * cfg.nreads
    n = 60_000
    decisions = []
    po.barcode_splitter_bytes(host[:n * cfg.record_bytes].tobytes(), cfg.barcodes, site, adapter, decisions=decisions)
    got = [(int(a), int(b)) for a, b in res[:n]]
    assert got == [(b, 999 if b < 0 else c) for b, c in decisions]
    clipped = sum(1 for b, c in decisions if b >= 0 and c != 999)
    assert clipped > 0.03 * n                                                # the read-through reads of this site's barcodes are trimmed
