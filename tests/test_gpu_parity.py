"""GPU parity tests proper: the HIP path through the C-ABI (libtagdig.so) against
the committed golden fixtures and against the CPU oracle on identical bytes.
Bit-exact: the path is integer/byte work, no tolerance anywhere.
Run on an MI355X with `pytest -m gpu`."""
import os
import random

import numpy as np
import pytest

from conftest import load_golden, write_case_file, case_payload
from helpers import (DEFAULT_MODE, KERNEL_MODES, apply_mode, dirty_fastq, fuzz_case, mode_id, small_index, synth_expected,
                     synth_host_bytes)
from oracle import c_oracle
from oracle import tagdigger_oracle as orc

pytestmark = pytest.mark.gpu

CASES = load_golden("hotpath_cases.json") + load_golden("hotpath_random.json")
EXC = {"AssertionError": AssertionError, "IndexError": IndexError, "TypeError": TypeError,
       "ValueError": ValueError, "FileNotFoundError": FileNotFoundError}
# the reference raises lazily at its first lookup into a leaf root; the product at index build
ROOTLEAF = {"tag equals cutsite first"}


@pytest.fixture(scope="module")
def eng():
    import tagdigger_amd
    e = tagdigger_amd.Engine(0)
    yield e
    e.close()


def gpu_counts(eng, data, barcodes, tags, cutsite="TGCAG", **kw):
    eng.set_index(barcodes, tags, cutsite)
    eng.count_bytes(data, **kw)
    return eng.counts_numpy()


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_golden_find_tags_fastq(case, tmp_path):
    """The drop-in function itself, file in -> matrix out, against reference outputs."""
    from tagdigger_amd import tagdigger_fun as tf
    path = write_case_file(case, tmp_path)
    if case.get("filename_override"):
        path = str(tmp_path / "nope" / case["filename_override"])
    if "raises" in case:
        with pytest.raises(EXC[case["raises"]]) as ei:
            tf.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        if case["message"] and case["raises"] == "AssertionError":
            assert str(ei.value) == case["message"]
    else:
        got = tf.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        assert got == case["counts"]


@pytest.mark.parametrize("mode", KERNEL_MODES + [dict(prescan=1, tile_kb=16), dict(prescan=1, tile_kb=32)], ids=mode_id)
def test_golden_all_kernel_modes(eng, mode):
    """Every non-raising, non-gz fixture through the in-memory path in each kernel configuration."""
    import base64, gzip
    try:
        for case in CASES:
            if "raises" in case or case["kwargs"].get("tassel_tagcount"):
                continue
            data = case_payload(case)
            if case["filename"][-2:].lower() == "gz":
                data = gzip.decompress(data)
            eng.set_index(case["barcodes"], case["tags"], case["kwargs"].get("cutsite", "TGCAG"))
            apply_mode(eng, mode)
            eng.count_bytes(data, maxreads=case["kwargs"].get("maxreads", 5e9))
            assert eng.counts() == case["counts"], case["name"]
    finally:
        eng.set_option("prescan", 0)
        apply_mode(eng, DEFAULT_MODE)


def test_synth_device_generator_and_counts(eng):
    """K0 bytes == host reference bytes; counts == oracle == generator's own expected matrix."""
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=100_000, nbar=8, nmarkers=50, seed=1234)
    nb = cfg.nbytes()
    d = eng.dev_alloc(nb)
    try:
        cfg.fill_device(eng, d, 0, cfg.nreads)
        dev = eng.d2h(d, nb)
        host = synth_host_bytes(cfg, 0, cfg.nreads)
        assert dev == bytes(host)
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        eng.count_device(d, nb)
        got = eng.counts_numpy()
        st = eng.stats()
    finally:
        eng.dev_free(d)
    want, hits = synth_expected(cfg, 0, cfg.nreads)
    ost = {}
    ora = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(host, stats=ost)
    assert (got == ora).all() and (got == want).all()
    assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"])
    assert st["lines"] == 4 * cfg.nreads and st["tag"] == hits


@pytest.mark.parametrize("cutsite,nl", [("TGCAG", ("\n",)), ("CWGC", ("\n",)), ("TGCAG", ("\r\n",)),
                                        ("TGCAT", ("\r",)), ("", ("\n", "\r\n", "\r")), ("RCATGY", ("\n", "\r\n"))])
@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_vs_oracle(eng, cutsite, nl, seed):
    """Dirty FASTQ a few tiles long: every terminator style, blanks, N, case, phase shifts."""
    rnd = random.Random(1000 * seed + len(cutsite) + len(nl))
    barcodes, tags, cutsites = small_index(rnd, cutsite)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=1500, nl_choices=nl, long_lines=(seed == 2),
                       permanent_shifts=(seed == 2))
    ora = c_oracle.COracle(barcodes, tags, cutsite)
    ost = {}
    want = ora.count_bytes(data, stats=ost)
    try:
        for mode in KERNEL_MODES:
            eng.set_index(barcodes, tags, cutsite)
            apply_mode(eng, mode)
            lines = eng.count_bytes(data)
            got = eng.counts_numpy()
            st = eng.stats()
            assert (got == want).all(), (cutsite, nl, mode)
            assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), mode
            assert lines == st["lines"] == data.count(b"\n") + data.count(b"\r") - data.count(b"\r\n")
    finally:
        apply_mode(eng, DEFAULT_MODE)


@pytest.mark.parametrize("seed", [12519])
def test_fuzz_cases_that_once_failed(eng, seed):
    """Campaign cases kept by their seed.  12519: a few workgroups take sixteen tiles through k_fast4's ring of five slots, and
    a line that opens with a blank was re-read from global memory by a tile number taken from its slot AFTER the slot had
    been given back (the next tile's number by then) -- every producer count, several runs (the failure was a race)."""
    barcodes, tags, cutsite, _, data = fuzz_case(seed)
    ost = {}
    want = c_oracle.COracle(barcodes, tags, cutsite).count_bytes(data, stats=ost)
    eng.set_index(barcodes, tags, cutsite)
    try:
        for nprod in (0, 7, 8, 9, 10, 11, 12, 13):
            apply_mode(eng, dict(fastpath=1, kernel=4, f4_nprod=nprod))
            for _ in range(3):
                eng.reset()
                eng.count_bytes(data)
                st = eng.stats()
                assert (eng.counts_numpy() == want).all(), nprod
                assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), nprod
    finally:
        apply_mode(eng, dict(f4_nprod=0))
        apply_mode(eng, DEFAULT_MODE)


def test_fuzz_campaign(eng):
    """Randomised cases against the C oracle for TD_FUZZ_SECONDS seconds (default 20; a soak run on the
    GPU box uses minutes): random index shapes (barcode and tag counts and lengths, cut sites with IUPAC
    codes, tags longer than 32 and 64 bases), every terminator style, long lines and phase shifts, both
    tile sizes, both kernels."""
    import time
    budget = float(os.environ.get("TD_FUZZ_SECONDS", "20"))
    seed0 = int(os.environ.get("TD_FUZZ_SEED", "12345"))
    t_end = time.time() + budget
    next_note = time.time() + 30
    ncase = 0
    try:
        while time.time() < t_end:
            barcodes, tags, cutsite, nl, data = fuzz_case(seed0 + ncase)
            ost = {}
            want = c_oracle.COracle(barcodes, tags, cutsite).count_bytes(data, stats=ost)
            eng.set_index(barcodes, tags, cutsite)
            for mode in KERNEL_MODES:
                apply_mode(eng, mode)
                eng.reset()
                eng.count_bytes(data)
                st = eng.stats()
                assert (eng.counts_numpy() == want).all(), ("seed", seed0 + ncase, cutsite, nl, mode)
                assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), ("seed", seed0 + ncase, mode)
            ncase += 1
            if time.time() >= next_note:                      # (a long soak must not look hung)
                print(" [%d cases so far] " % ncase, end="", flush=True)
                next_note = time.time() + 30
    finally:
        apply_mode(eng, DEFAULT_MODE)
    print(" [fuzz campaign: %d cases] " % ncase, end="")
    assert ncase > 0


@pytest.mark.parametrize("mode", KERNEL_MODES, ids=mode_id)
def test_tile_boundary_sweep(eng, mode):
    """Slide a record across a tile boundary byte by byte, with \\r\\n split across it -- the boundary
    between the buffer's second and third tiles, so that both are ordinary tiles of the main pass (the
    first and the last go through other code)."""
    barcodes, tags = ["AACG", "TTGACC"], ["TGCAGAAAC", "TGCAGGGGT"]
    rec = b"@h\r\nAACGTGCAGAAACTT\r\n+\r\nIIII\r\n"
    eng.set_index(barcodes, tags, "TGCAG")
    ora = c_oracle.COracle(barcodes, tags, "TGCAG")
    tile = 1024 * (24 if mode.get("kernel") == 4 else mode["tile_kb2"] if mode.get("kernel") == 2 else mode["tile_kb"])

    def padded(n):          # one record of exactly n bytes
        return b"@p\nGGGG\n+\n" + b"I" * (n - 10 - 1) + b"\n"
    try:
        apply_mode(eng, mode)
        for pad in list(range(tile - 40, tile + 8)):
            # tile 0 is one padded record; a second one ends `pad` bytes into tile 1 ... so that rec straddles 2 * tile
            data = padded(tile) + padded(pad) + rec + rec + padded(tile) + padded(tile // 2) + rec
            eng.reset()
            eng.count_bytes(data)
            assert (eng.counts_numpy() == ora.count_bytes(data)).all(), pad
    finally:
        apply_mode(eng, DEFAULT_MODE)


def test_long_lines_and_phase_shifts(eng):
    """Lines longer than a tile and permanent line-phase shifts (every later tile mispredicts its
    neighbours' phase at the shift): the fix-up pass puts the matrix right."""
    rnd = random.Random(77)
    barcodes, tags, cutsites = small_index(rnd, "TGCAG", nbar=12, ntag=60)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=4000, long_lines=True, permanent_shifts=True)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
    try:
        for mode in KERNEL_MODES:
            eng.set_index(barcodes, tags, "TGCAG")
            apply_mode(eng, mode)
            eng.count_bytes(data)
            assert (eng.counts_numpy() == want).all(), mode
    finally:
        apply_mode(eng, DEFAULT_MODE)


def test_many_short_lines_overflow_rounds(eng):
    """More sequence-line starts in a tile than the per-tile list holds (1-byte lines)."""
    barcodes, tags = ["A"], ["CC", "GT"]
    eng.set_index(barcodes, tags, "")
    body = b"\n".join([b"x", b"ACC", b"y", b"z"] * 3000) + b"\n" + b"\n" * 9000 + b"@\nAGT\n+\n!\n"
    ora = c_oracle.COracle(barcodes, tags, "")
    try:
        for mode in KERNEL_MODES:
            eng.reset()
            apply_mode(eng, mode)
            eng.count_bytes(body)
            assert (eng.counts_numpy() == ora.count_bytes(body)).all(), mode
    finally:
        apply_mode(eng, DEFAULT_MODE)


def test_streamed_pieces_and_maxreads(eng):
    """> 32 MiB host buffer: several staged pieces with the line index carried on the device."""
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=400_000, nbar=24, nmarkers=500, seed=77)
    host = synth_host_bytes(cfg, 0, cfg.nreads)          # 87.6 MB
    shifted = b"\n" + bytes(host)                        # every line index moves by one
    ora = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_bytes(host)
    want, _ = synth_expected(cfg, 0, cfg.nreads)
    assert (eng.counts_numpy() == want).all()
    for maxreads in (1, 123_457, 399_999):
        eng.reset()
        eng.count_bytes(host, maxreads=maxreads)
        w, _ = synth_expected(cfg, 0, maxreads)
        assert (eng.counts_numpy() == w).all() and eng.stats()["reads"] == maxreads
    eng.reset()
    eng.count_bytes(shifted)
    assert int(eng.counts_numpy().sum()) == int(ora.count_bytes(shifted).sum()) == 0
    # first_line argument: the same bytes declared to start on line 3 read as shifted by one more
    eng.reset()
    eng.count_bytes(shifted, first_line=3)
    assert (eng.counts_numpy() == want).all()


def test_count_lines_device(eng):
    rnd = random.Random(9)
    data = bytes(rnd.choice(b"ACGT\n\r\r\nxyz") for _ in range(200_003))
    d = eng.dev_alloc(len(data))
    try:
        eng.h2d(d, data)
        got = eng.count_lines_device(d, len(data))
    finally:
        eng.dev_free(d)
    want = data.count(b"\n") + data.count(b"\r") - data.count(b"\r\n")
    assert got == want


def test_nonascii_policy(eng):
    import tagdigger_amd
    barcodes, tags = ["AACG"], ["TGCAGAAAC"]
    eng.set_index(barcodes, tags, "TGCAG")
    ok = "@h\xe9\nAACGTGCAGAAAC\n+\n\xff\xfe\n".encode("latin-1")      # high bytes outside sequence lines
    eng.count_bytes(ok)
    assert eng.counts() == [[1]]
    bad = "@h\nAACGTGCAGAAAC\xe9\n+\nII\n".encode("latin-1")
    eng.reset()
    eng.count_bytes(bad)
    with pytest.raises(tagdigger_amd.NonAsciiSequence):
        eng.counts()
    with pytest.raises(orc.NonAsciiSequence):
        orc.count_bytes(bad, barcodes, tags)
    # beyond maxreads it is never looked at
    eng.reset()
    eng.count_bytes(ok + bad, maxreads=1)
    assert eng.counts() == [[1]]


def test_long_tags_all_widths(eng):
    """Tag lengths across every packed-width instantiation (W = 1,2,3,4,6,10 words)."""
    rnd = random.Random(3)
    for maxlen in (20, 60, 90, 125, 190, 300):
        barcodes = ["ACGTAC", "TTGA", "GGGTCCAATC"]
        tags = []
        while len(tags) < 30:
            t = "TGCAG" + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(max(1, maxlen - 40), maxlen)))
            if not any(t.startswith(o) or o.startswith(t) for o in tags):
                tags.append(t)
        data = dirty_fastq(rnd, barcodes, tags, ["TGCAG"], nrec=400)
        want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
        got = gpu_counts(eng, data, barcodes, tags, "TGCAG")
        assert (got == want).all() and want.sum() > 50, maxlen


def test_short_and_mixed_length_tags(eng):
    """Tags shorter than the hashed prefix go through the short list; mixed lengths share buckets."""
    rnd = random.Random(4)
    barcodes = ["ACGT", "TTG"]
    tags = ["A", "CA", "CCGT", "GGGTTTAAACCC", "GGGTTTAAACCG" + "T" * 30, "T" * 40 + "A", "T" * 40 + "C" + "G" * 25]
    for k in range(40):
        t = "CG" + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(5, 70)))
        if not any(t.startswith(o) or o.startswith(t) for o in tags):
            tags.append(t)
    data = dirty_fastq(rnd, barcodes, tags, [""], nrec=1200)
    want = c_oracle.COracle(barcodes, tags, "").count_bytes(data)
    got = gpu_counts(eng, data, barcodes, tags, "")
    assert (got == want).all() and want.sum() > 300


def test_idempotent_relaunch_and_accumulate(eng):
    """Counting the same buffer twice doubles every cell; reset returns to zero."""
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=30_000, nbar=8, nmarkers=50, seed=9)
    host = bytes(synth_host_bytes(cfg, 0, cfg.nreads))
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_bytes(host)
    once = eng.counts_numpy()
    eng.count_bytes(host)
    assert (eng.counts_numpy() == 2 * once).all()
    eng.reset()
    assert eng.counts_numpy().sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["gpu", "host"])
@pytest.mark.parametrize("level", [1, 6, 0])
def test_bgzf_file_counts_like_plain(tmp_path, where, level):
    """A bgzip-style file: its members inflated on the GPU (one per lane, csrc/gpu_inflate.hpp; the default) or
    member-parallel on the host -- same matrix as the plain bytes.  Level 0 = stored blocks."""
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    from helpers import bgzf_bytes, synth_host_bytes, synth_expected
    cfg = SynthConfig(nreads=300_000, nbar=8, nmarkers=50, seed=1234)      # 66 MB: several staging pieces
    data = bytes(synth_host_bytes(cfg, 0, cfg.nreads))
    p = tmp_path / "lib.fq.gz"
    p.write_bytes(bgzf_bytes(data, level=level, threads=8))
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        eng.set_option("gpu_inflate", 1 if where == "gpu" else 0)
        eng.count_file(str(p))
        want, _ = synth_expected(cfg, 0, cfg.nreads)
        assert (eng.counts_numpy() == want).all()
        # a read limit inside the file
        eng.reset()
        eng.count_file(str(p), maxreads=123_456)
        w2, _ = synth_expected(cfg, 0, 123_456)
        assert (eng.counts_numpy() == w2).all()
    finally:
        eng.close()


@pytest.mark.gpu
def test_bgzf_gpu_inflate_dirty_and_damaged(tmp_path):
    """Members that end inside lines (odd block sizes), CRLF and unterminated last lines through the GPU inflater;
    a flipped bit in a member's payload or a wrong CRC-32 in its trailer is an error, not a different matrix."""
    import tagdigger_amd
    from helpers import bgzf_bytes
    rnd = random.Random(5)
    barcodes, tags, cutsites = small_index(rnd, "TGCAG", nbar=8, ntag=40)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=40_000, nl_choices=("\n", "\r\n")).rstrip(b"\r\n")
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_index(barcodes, tags, "TGCAG")
        for block in (0xFF00, 4093, 65280 // 7):
            p = tmp_path / ("d%d.fq.gz" % block)
            p.write_bytes(bgzf_bytes(data, block=block, level=6, threads=8))
            eng.reset()
            eng.count_file(str(p))
            assert (eng.counts_numpy() == want).all(), block
        blob = bytearray(bgzf_bytes(data, level=6, threads=8))
        bad = tmp_path / "bad.fq.gz"
        flipped = bytearray(blob)
        flipped[len(blob) // 2] ^= 0x10                                   # somewhere in a member's deflate stream
        bad.write_bytes(bytes(flipped))
        def as_gzip_open_ends(path):
            import gzip
            import zlib
            try:
                with gzip.open(path, "rb") as fh:
                    while fh.read1(8192):
                        pass
            except (EOFError, OSError, zlib.error) as exc:
                return exc
            raise AssertionError("gzip.open reads this file")
        eng.reset()
        expected = as_gzip_open_ends(str(bad))                           # (reference :240-243: its exception propagates)
        with pytest.raises(type(expected)) as ei:
            eng.count_file(str(bad))
        assert str(ei.value) == str(expected)
        import struct
        bsize = struct.unpack("<H", blob[16:18])[0] + 1                  # first member: its CRC-32 sits 8 bytes before its end
        wrong = bytearray(blob)
        wrong[bsize - 8] ^= 0xFF
        bad.write_bytes(bytes(wrong))
        eng.reset()
        expected = as_gzip_open_ends(str(bad))
        with pytest.raises(type(expected)) as ei:
            eng.count_file(str(bad))
        assert str(ei.value) == str(expected)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("decoder", ["default", "host-threads", "sequential", "zlib"])
def test_gzip_file_counts_like_plain(tmp_path, monkeypatch, capfd, decoder):
    """An ordinary one-member gzip file above 8 MiB is decoded on the device by default (csrc/gz_gpu.hpp); option
    gpu_huffman 0: the chunk-parallel decoder on the host's threads (csrc/par_inflate.hpp; TAGDIG_PAR_INFLATE=0: one
    thread, TAGDIG_ZLIB=1: zlib): same matrix as the plain bytes, also with a read limit that ends the stream early."""
    import gzip
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes, synth_expected
    for k in ("TAGDIG_PAR_INFLATE", "TAGDIG_ZLIB", "TAGDIG_INFLATE_CHUNK", "TAGDIG_INFLATE_THREADS"):
        monkeypatch.delenv(k, raising=False)
    if decoder == "sequential":
        monkeypatch.setenv("TAGDIG_PAR_INFLATE", "0")
    if decoder == "zlib":
        monkeypatch.setenv("TAGDIG_ZLIB", "1")
    monkeypatch.setenv("TAGDIG_INFLATE_STATS", "1")
    cfg = SynthConfig(nreads=300_000, nbar=8, nmarkers=50, seed=4321)      # 66 MB, about 13 MB compressed
    data = bytes(synth_host_bytes(cfg, 0, cfg.nreads))
    p = tmp_path / "lib.fq.gz"
    p.write_bytes(gzip.compress(data, 1))
    assert os.path.getsize(p) > 8 << 20
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        if decoder != "default":
            eng.set_option("gpu_huffman", 0)
        eng.count_file(str(p))
        want, _ = synth_expected(cfg, 0, cfg.nreads)
        assert (eng.counts_numpy() == want).all()
        err = capfd.readouterr().err
        assert ("count_gzip_gpu:" in err) == (decoder == "default") and eng.last_gz_route() == (1 if decoder == "default" else 0)
        assert ("par_inflate:" in err) == (decoder == "host-threads")
        eng.reset()
        eng.count_file(str(p), maxreads=100_000)                           # (the reader stops asking early)
        want, _ = synth_expected(cfg, 0, 100_000)
        assert (eng.counts_numpy() == want).all()
    finally:
        eng.close()


@pytest.mark.gpu
def test_index_is_kept_between_identical_calls(tmp_path):
    """The same barcodes / tags / cut site again: the device index is reused and the counts start from zero;
    anything different rebuilds it."""
    from tagdigger_amd import tagdigger_fun as tf
    case = next(c for c in load_golden("hotpath_random.json") if c["name"] == "random000")
    path = write_case_file(case, tmp_path)
    bars, tags, cut = case["barcodes"], case["tags"], case["kwargs"]["cutsite"]
    first = tf.find_tags_fastq(path, bars, tags, cutsite=cut)
    again = tf.find_tags_fastq(path, list(bars), list(tags), cutsite=cut)          # equal lists, other objects
    assert first == again == case["counts"]
    fewer = tf.find_tags_fastq(path, bars, tags[:-1], cutsite=cut)
    assert fewer == [row[:-1] for row in case["counts"]]
    assert tf.find_tags_fastq(path, bars, tags, cutsite=cut) == case["counts"]
    with pytest.raises(AssertionError, match="Non-ACGT tag"):
        tf.find_tags_fastq(path, bars, tags[:-1] + ["ACGN"], cutsite=cut)
    assert tf.find_tags_fastq(path, bars, tags, cutsite=cut) == case["counts"]      # (after a failed set-up too)


@pytest.mark.gpu
def test_find_tags_fastq_many(tmp_path):
    """The batched call returns, in file order, what the single-file call returns (the first
    file's is a reference output); one barcode list for all files or one per file; an error in one file
    surfaces after the others are done."""
    import gzip
    from tagdigger_amd import tagdigger_fun as tf
    case = next(c for c in load_golden("hotpath_random.json") if c["name"] == "random000")
    first = write_case_file(case, tmp_path)
    data = case_payload(case)
    lines = data.split(b"\n")
    second = str(tmp_path / "half.fq")
    open(second, "wb").write(b"\n".join(lines[:len(lines) // 8 * 4]) + b"\n")
    third = str(tmp_path / "again.fq.gz")
    open(third, "wb").write(gzip.compress(data))
    files = [first, second, third]
    bars, tags, cut = case["barcodes"], case["tags"], case["kwargs"]["cutsite"]
    got = tf.find_tags_fastq_many(files, bars, tags, cutsite=cut, devices=[0])
    assert got[0] == case["counts"] and got[2] == case["counts"]
    assert got == [tf.find_tags_fastq(f, bars, tags, cutsite=cut) for f in files]
    per_file = [bars, bars[:3], bars[1:]]
    got = tf.find_tags_fastq_many(files, per_file, tags, cutsite=cut, devices=[0, 0])
    assert got == [tf.find_tags_fastq(f, b, tags, cutsite=cut) for f, b in zip(files, per_file)]
    with pytest.raises(FileNotFoundError):
        tf.find_tags_fastq_many([first, str(tmp_path / "missing.fq")], bars, tags, cutsite=cut)
    with pytest.raises(ValueError):
        tf.find_tags_fastq_many(files, [bars] * 4, tags, cutsite=cut)


@pytest.mark.gpu
def test_byte_sharded_file_on_the_gpu(tmp_path):
    """One file cut into byte shards at line starts (tagdigger_amd.multi): counted shard by shard
    with each shard's own first line index, the GPU gives the whole file's matrix; and the
    single-rank form of count_file_sharded equals find_tags_fastq."""
    import tagdigger_amd
    from tagdigger_amd import multi, tagdigger_fun as tf
    rnd = random.Random(31)
    barcodes, tags, cutsites = small_index(rnd, "TGCAG", nbar=6, ntag=40)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=3000)
    path = str(tmp_path / "one.fq")
    open(path, "wb").write(data)
    want = c_oracle.COracle(barcodes, tags, "TGCAG").count_bytes(data)
    eng = tagdigger_amd.Engine(0)
    try:
        for world in (2, 3, 7):
            eng.set_index(barcodes, tags, "TGCAG")
            first_line = 0
            for a, b in multi.shard_bounds(path, world):
                if b > a:
                    eng.count_bytes(data[a:b], first_line=first_line)
                    first_line += multi.count_terminators(data[a:b])
            assert (eng.counts_numpy() == want).all(), world
    finally:
        eng.close()
    assert multi.count_file_sharded(path, barcodes, tags, "TGCAG") == tf.find_tags_fastq(path, barcodes, tags, cutsite="TGCAG")
    assert multi.count_file_sharded(path, barcodes, tags, "TGCAG") == want.tolist()


@pytest.mark.gpu
def test_device_expected_matrix_equals_host_reference():
    """td_synth_expected_device (what bench.py checks the kernels against) against the oracle side's
    synth_expected for the same shard of the canonical stream."""
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_expected
    cfg = SynthConfig(nreads=500_000, nbar=24, nmarkers=300, seed=9, cutsite="CWGC", bclen=(4, 10))
    eng = tagdigger_amd.Engine(0)
    try:
        n = len(cfg.barcodes) * len(cfg.tags)
        d = eng.dev_alloc(n * 4)
        eng.h2d(d, bytes(n * 4))
        hits = cfg.expected_device(eng, d, 12345, cfg.nreads)
        got = np.frombuffer(eng.d2h(d, n * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
        eng.dev_free(d)
        want, whits = synth_expected(cfg, 12345, cfg.nreads)
        assert hits == whits and (got == want).all()
    finally:
        eng.close()
