"""CPU-only checks of the host side: the C-ABI library loads and exports every
declared symbol, fails loudly without a GPU, and the synthetic-stream spec is
self-consistent against the oracle.  No compute calls on a GPU here."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

from conftest import ROOT
from helpers import synth_expected, synth_host_bytes
from oracle import c_oracle
from oracle import tagdigger_oracle as orc


def test_header_symbols_exported():
    from tagdigger_amd import _binding as B
    L = B.load()
    hdr = open(os.path.join(ROOT, "include", "tagdig.h")).read()
    declared = set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(B.EXPORTS)


def test_no_gpu_fails_loudly():
    from tagdigger_amd import _binding as B
    L = B.load()
    h = C.c_void_p()
    rc = L.td_create(C.byref(h), 0)
    if rc == 0:          # running on a GPU box: nothing to check here
        L.td_destroy(h)
        pytest.skip("GPU present")
    assert rc == -1 and b"no CPU fallback" in L.td_last_error()
    import tagdigger_amd
    with pytest.raises(tagdigger_amd.TagdigError):
        tagdigger_amd.Engine(0)


def test_product_does_not_import_oracle():
    """The oracle is a checker only: nothing under tagdigger_amd/ may import, link or open it."""
    pkg = os.path.join(ROOT, "tagdigger_amd")
    pat = re.compile(r"^\s*(from|import)\s+.*\boracle\b|oracle/|liboracle|c_oracle|tagdigger_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                hits = [m.group(0) for m in pat.finditer(src)]
                assert not hits, (f, hits)


def test_engine_setup_matches_oracle_lists():
    """The product's own cut-site enumeration equals the oracle's (hence the reference's)."""
    from tagdigger_amd.engine import enumerate_cut_sites, effective_maxreads
    for cs in ["TGCAG", "", "CWGC", "BN", "RY", "NN", "RCATGY", "VH", "NWS"]:
        assert enumerate_cut_sites(cs) == orc.enumerate_cut_sites(cs)
    for m in [5e9, 3, 2.5, 0, -4, 1, 1.0001]:
        assert effective_maxreads(m) == c_oracle.effective_maxreads(m)


def test_synth_records_and_expected_matrix():
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=20000, nbar=8, nmarkers=50, seed=1234)
    data = synth_host_bytes(cfg, 0, cfg.nreads)
    assert data.nbytes == cfg.nreads * 219
    first = bytes(data[:219]).split(b"\n")
    assert first[0] == b"@r000000000000" and len(first[1]) == 100 and first[2] == b"+" and first[3] == b"I" * 100
    # shards are position independent
    mid = synth_host_bytes(cfg, 777, 10)
    assert bytes(mid) == bytes(data[777 * 219:787 * 219])
    want, hits = synth_expected(cfg, 0, cfg.nreads)
    st = {}
    got = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(data, stats=st)
    assert (got == want).all() and hits == st["tag"] == int(want.sum())
    assert st["reads"] == cfg.nreads and 0.83 * cfg.nreads < st["barcut"] < 0.9 * cfg.nreads
    # and the slow Python oracle agrees on a prefix
    sub = bytes(data[:219 * 1500])
    assert orc.count_bytes(sub, cfg.barcodes, cfg.tags, cfg.cutsite) == \
        c_oracle.count_bytes(sub, cfg.barcodes, cfg.tags, cfg.cutsite)


def test_synth_multicut_config():
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=5000, nbar=12, nmarkers=40, seed=5, cutsite="CWGC", bclen=(4, 10))
    data = synth_host_bytes(cfg, 0, cfg.nreads)
    want, _ = synth_expected(cfg, 0, cfg.nreads)
    got = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(data)
    assert (got == want).all()


def test_write_counts_numpy_path_is_byte_identical(tmp_path):
    """writeCounts (reference tagdigger_fun.py:1100-1111) given a numpy matrix writes what csv.writer writes for the
    list form -- names that need quoting, and an EMPTY sample name (a lone empty field is quoted by csv.writer, the
    first field of a longer row is not)."""
    from tagdigger_amd import tagdigger_fun as tf
    names = ["", "s,1", 's"2', " plain", "x\ny"]
    tags = ["M_A_0", "M,1", ""]
    counts = [[0, 1, 2], [3, 4, 5], [2 ** 40, 7, 8], [9, 10, 11], [12, 13, 14]]
    a, b = str(tmp_path / "list.csv"), str(tmp_path / "array.csv")
    tf.writeCounts(a, counts, names, tags)
    tf.writeCounts(b, np.array(counts, dtype=np.int64), names, tags)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read().split(b"\r\n")[1] == b",0,1,2"


def test_crc32_join_against_zlib():
    """td_crc32_join (host arithmetic, no GPU): CRC-32 of a || b from the two halves' CRC-32s and len(b), against zlib over
    lengths with every bit pattern the decoders meet -- the device decoder's 64 KiB blocks, odd remainders, empty halves, and
    lengths past 2^32 (by the law join(join(a, b), c) == join(a, join(b, c)))."""
    import zlib
    from tagdigger_amd import _binding as B
    L = B.load()
    rng = random.Random(12)
    data = rng.randbytes(1 << 20)
    for _ in range(300):
        la = rng.choice([0, 1, 7, 65536, rng.randrange(1 << 18)])
        lb = rng.choice([0, 1, 3, 255, 256, 65535, 65536, 65537, rng.randrange(1 << 19)])
        a, b = data[:la], data[la:la + lb]
        assert L.td_crc32_join(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b), (la, lb)
    # a run of joins in the decoder's order equals the CRC-32 of the whole
    run, at = zlib.crc32(b""), 0
    while at < len(data):
        n = min(len(data) - at, rng.choice([65536, 65536, 65536, rng.randrange(1, 65536)]))
        run = L.td_crc32_join(run, zlib.crc32(data[at:at + n]), n)
        at += n
    assert run == zlib.crc32(data)
    # lengths no buffer here holds: associativity
    ca, cb, cc = (rng.getrandbits(32) for _ in range(3))
    for lb, lc in [((1 << 33) + 12345, (1 << 35) + 1), (1 << 40, (1 << 32) - 1), (5, 1 << 50)]:
        left = L.td_crc32_join(L.td_crc32_join(ca, cb, lb), cc, lc)
        right = L.td_crc32_join(ca, L.td_crc32_join(cb, cc, lc), lb + lc)
        assert left == right, (lb, lc)
