"""Ordinary gzip decoded ON THE DEVICE (csrc/gz_gpu.hpp through td_gunzip_file_gpu / td_count_file: block search, Huffman
decoding into tokens, LZ77 with markers, the windows between chunks, CRC-32 -- what `gzip.open(fqfile, 'rt')` of reference
tagdigger_fun.py:240-241 reads).  The text must equal zlib's byte for byte and the counts the oracle's on the plain bytes,
for every kind of stream the host decoders are tested with; what the device decoder does not take (several members, tokens
that overflow, tiny inputs) must fall to the host decoders unnoticed; damaged streams must end as gzip.open ends."""
import gzip
import os
import random
import struct
import zlib

import pytest

from helpers import gzip_one_member, synth_host_bytes
from oracle import c_oracle

pytestmark = pytest.mark.gpu

NREADS = 300_000


@pytest.fixture(scope="module")
def eng():
    import tagdigger_amd
    e = tagdigger_amd.Engine(0)
    e.set_option("gz_gpu_min", 0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def sample():
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig.from_id(2, nreads=NREADS)
    raw = synth_host_bytes(cfg, 0, NREADS).tobytes()
    ost = {}
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw, stats=ost)
    return cfg, raw, want, ost


def _raw_deflate(data, level, flush_every=0, strategy=zlib.Z_DEFAULT_STRATEGY, flush=zlib.Z_SYNC_FLUSH):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for i in range(0, len(data), flush_every):
        out.append(co.compress(data[i:i + flush_every]))
        out.append(co.flush(flush))
    return b"".join(out) + co.flush()


def _mixed(d):
    """one DEFLATE stream whose middle 700 000 bytes are stored blocks (level 0), the rest level 6 (deflateParams)"""
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    h = len(d) // 2
    a = co.compress(d[:h]) + co.flush(zlib.Z_FULL_FLUSH)
    st = zlib.compressobj(0, zlib.DEFLATED, -15)
    b = st.compress(d[h:h + 700_000]) + st.flush(zlib.Z_FULL_FLUSH)
    co2 = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = co2.compress(d[h + 700_000:]) + co2.flush()
    return a + b + c


def _member(data, body):
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


# name -> (writer, decoded on the device?)
STREAMS = {
    "level6": (lambda d: gzip.compress(d, compresslevel=6), True),
    "level1": (lambda d: gzip.compress(d, compresslevel=1), True),
    "level9": (lambda d: gzip.compress(d, compresslevel=9), True),
    "sync-flushes": (lambda d: _member(d, _raw_deflate(d, 6, flush_every=100_000)), True),
    "full-flushes": (lambda d: gzip_one_member(d, level=1, threads=4, piece=1 << 18), True),
    "fixed-huffman": (lambda d: _member(d, _raw_deflate(d, 6, strategy=zlib.Z_FIXED)), True),
    "huffman-only": (lambda d: _member(d, _raw_deflate(d, 6, strategy=zlib.Z_HUFFMAN_ONLY)), True),
    "rle": (lambda d: _member(d, _raw_deflate(d, 6, strategy=zlib.Z_RLE)), True),
    "stored": (lambda d: gzip.compress(d[:20_000_000], compresslevel=0), True),
    "stored-in-the-middle": (lambda d: _member(d, _mixed(d)), True),
    "header-fields": (lambda d: b"\x1f\x8b\x08\x1c" + b"\0" * 6 + struct.pack("<H", 5) + b"extra" + b"name.fq\0" + b"a comment\0"
                                + _raw_deflate(d, 6) + struct.pack("<II", zlib.crc32(d), len(d) & 0xFFFFFFFF), True),
    "zero-padding": (lambda d: gzip.compress(d, compresslevel=6) + b"\0" * 1000, True),
    "two-members": (lambda d: gzip.compress(d[:len(d) // 3], compresslevel=6) + gzip.compress(d[len(d) // 3:], compresslevel=1), True),
    "three-members-padded": (lambda d: gzip.compress(d[:len(d) // 3], compresslevel=6) + b"\0" * 700 + gzip.compress(d[len(d) // 3:2 * len(d) // 3], compresslevel=1)
                                       + gzip.compress(d[2 * len(d) // 3:], compresslevel=9) + b"\0" * 64, True),
    "small-members": (lambda d: b"".join(gzip.compress(d[i:i + 2_000_000], compresslevel=6) for i in range(0, len(d), 2_000_000)), False),
}


def _check(eng, want, ost, what):
    got = eng.counts_numpy()
    st = eng.stats()
    assert (got == want).all(), what
    assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), what


@pytest.mark.parametrize("terr_kb", [16, 128])
@pytest.mark.parametrize("kind", sorted(STREAMS))
def test_text_and_counts(eng, sample, tmp_path, kind, terr_kb):
    cfg, raw, want, ost = sample
    writer, on_device = STREAMS[kind]
    blob = writer(raw)
    text = gzip.decompress(blob)
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(blob)
    eng.set_option("gz_gpu_terr_kb", terr_kb)
    got = eng.gunzip_file_gpu(path, len(text) + 64)
    if on_device:
        assert got is not None, "the device decoder left the file to the host"
        assert len(got) == len(text) and got == text
    else:
        assert got is None
    if text != raw:
        ost = {}
        want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(text, stats=ost)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.reset()
    eng.count_file(path)
    _check(eng, want, ost, (kind, terr_kb))
    assert eng.last_gz_route() == (1 if on_device else 0)


@pytest.mark.parametrize("seg_kb,margin_kb", [(1024, 256), (3000, 64), (700, 1024)])
@pytest.mark.parametrize("kind", ["level6", "full-flushes", "two-members", "sync-flushes"])
def test_segments_of_the_compressed_file(eng, sample, tmp_path, kind, seg_kb, margin_kb):
    """The file goes through the device in segments of compressed bytes (1 GiB by default; here a megabyte or so): the window,
    the unfinished line and the member's CRC-32 are carried from one to the next."""
    cfg, raw, want, ost = sample
    blob = STREAMS[kind][0](raw)
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(blob)
    eng.set_option("gz_gpu_terr_kb", 64)
    eng.set_option("gz_gpu_seg_kb", seg_kb)
    eng.set_option("gz_gpu_margin_kb", margin_kb)
    try:
        got = eng.gunzip_file_gpu(path, len(raw) + 64)
        assert got is not None and got == raw
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        for maxreads in (5e9, NREADS // 2 + 17):
            ost2 = {}
            want2 = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw, maxreads=int(min(maxreads, 10 ** 12)), stats=ost2)
            eng.reset()
            eng.count_file(path, maxreads=maxreads)
            _check(eng, want2, ost2, (kind, seg_kb, maxreads))
            assert eng.last_gz_route() == 1
    finally:
        eng.set_option("gz_gpu_seg_kb", 1 << 20)
        eng.set_option("gz_gpu_margin_kb", 16384)


@pytest.mark.parametrize("every", [1, 40, 97])
def test_false_block_starts_cost_a_second_decoding_only(eng, sample, tmp_path, every):
    """The chunks are chained by the host: a start that is none (here every n-th found start, moved by 4099 bits) is dropped and
    the stretch behind its predecessor's end decoded again -- the text is the same."""
    cfg, raw, want, ost = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=6))
    eng.set_option("gz_gpu_terr_kb", 64)
    eng.set_option("gz_gpu_false_every", every)
    try:
        got = eng.gunzip_file_gpu(path, len(raw) + 64)
        assert got is None or got == raw          # (None: more false starts than the decoder repairs, eight -- the host decoder's file then)
        if every > 1:
            assert got == raw
    finally:
        eng.set_option("gz_gpu_false_every", 0)


def test_block_search_that_decodes_before_it_believes(eng, sample, tmp_path):
    cfg, raw, want, ost = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=6))
    eng.set_option("gz_gpu_verify", 1)
    try:
        assert eng.gunzip_file_gpu(path, len(raw) + 64) == raw
    finally:
        eng.set_option("gz_gpu_verify", 0)


def test_what_the_device_decoder_leaves_to_the_host(eng, sample, tmp_path):
    """Tiny and empty inputs, text that deflates a thousandfold (more tokens than a chunk's buffer holds), a file below the
    size bound: td_count_file counts them as before."""
    cfg, raw, want, ost = sample
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_option("gz_gpu_terr_kb", 128)
    orc = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    cases = {"empty text": b"", "one read": raw[:raw.index(b"\n@", 10) + 1], "a thousandfold": raw[:2000] * 20000}
    for name, text in cases.items():
        path = str(tmp_path / "x.fq.gz")
        with open(path, "wb") as fh:
            fh.write(gzip.compress(text, compresslevel=6))
        got = eng.gunzip_file_gpu(path, len(text) + 64)
        assert got is None or got == text, name
        st = {}
        w = orc.count_bytes(text, stats=st)
        eng.reset()
        eng.count_file(path)
        _check(eng, w, st, name)
    eng.set_option("gz_gpu_min", 1 << 30)
    try:
        path = str(tmp_path / "y.fq.gz")
        with open(path, "wb") as fh:
            fh.write(gzip.compress(raw, compresslevel=1))
        assert eng.gunzip_file_gpu(path, len(raw) + 64) is None
        eng.reset()
        eng.count_file(path)
        _check(eng, want, ost, "below the bound")
        assert eng.last_gz_route() == 0
    finally:
        eng.set_option("gz_gpu_min", 0)


def test_maxreads_inside_a_gzip_file(eng, sample, tmp_path):
    cfg, raw, _, _ = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=1))
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    for maxreads in (1, 4321, NREADS - 1, NREADS, NREADS + 5):
        ost = {}
        want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw, maxreads=maxreads, stats=ost)
        eng.reset()
        eng.count_file(path, maxreads=maxreads)
        _check(eng, want, ost, maxreads)
        assert eng.last_gz_route() == 1


def test_damaged_streams_end_as_gzip_open_ends(eng, sample, tmp_path):
    """Flipped bits, truncation, a wrong CRC-32, a wrong length, bytes behind the member: the exception of gzip.open (class
    and message; tests/test_gzip_damage.py pins the rules on the real reference), never counts -- and the next good file
    counts on the device again."""
    cfg, raw, want, ost = sample
    good = gzip.compress(raw, compresslevel=6)
    rng = random.Random(5)
    bad = {"truncated": good[:len(good) // 2], "crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:],
           "length": good[:-4] + struct.pack("<I", len(raw) + 1), "junk": good + b"junk"}
    for k in range(6):
        at = rng.randrange(100, len(good) - 100)
        bad["flip %d" % k] = good[:at] + bytes([good[at] ^ (1 << rng.randrange(8))]) + good[at + 1:]
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    for name, blob in bad.items():
        path = str(tmp_path / "bad.fq.gz")
        with open(path, "wb") as fh:
            fh.write(blob)
        try:
            with gzip.open(path, "rt", newline=None) as fh:
                for _ in fh:
                    pass
            expected = None
        except (EOFError, OSError, zlib.error) as exc:
            expected = exc
        eng.reset()
        if expected is None:
            # (a flipped bit that zlib does not notice cannot exist -- the CRC-32 would fail -- but a flip in the header's
            # MTIME or OS byte changes nothing)
            eng.count_file(path)
            continue
        with pytest.raises(type(expected)) as ei:
            eng.count_file(path)
        assert type(ei.value) is type(expected) and str(ei.value) == str(expected), name
    path = str(tmp_path / "good.fq.gz")
    with open(path, "wb") as fh:
        fh.write(good)
    eng.reset()
    eng.count_file(path)
    _check(eng, want, ost, "the good file afterwards")
    assert eng.last_gz_route() == 1


def test_larger_file_with_default_options(sample, tmp_path):
    """A file above the default size bound through a fresh engine with default options: the device decoder is what runs."""
    import tagdigger_amd
    cfg, raw, want, ost = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip_one_member(raw * 3, level=1, threads=6))
    assert os.path.getsize(path) > 8 << 20
    e = tagdigger_amd.Engine(0)
    try:
        e.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        e.count_file(path)
        _check(e, want * 3, {k: 3 * v for k, v in ost.items()}, "three times the sample")
        assert e.last_gz_route() == 1
    finally:
        e.close()


def _structured_bytes(rng, n):
    """Bytes that are no FASTQ: stretches of noise (stored blocks), of one byte (runs of maximal copies), of text with repeats
    at every distance up to the window's 32 KiB, and of short periodic patterns."""
    out = bytearray()
    words = [bytes(rng.choice(b"ACGTNacgt\n@+IF#") for _ in range(rng.randint(1, 40))) for _ in range(200)]
    while len(out) < n:
        kind = rng.randrange(6)
        m = rng.randint(1, 200_000)
        if kind == 0:
            out += rng.randbytes(m)
        elif kind == 1:
            out += bytes([rng.randrange(256)]) * m
        elif kind == 2:
            out += b"".join(rng.choice(words) for _ in range(m // 20 + 1))
        elif kind == 3:
            pat = rng.randbytes(rng.randint(1, 9))
            out += pat * (m // len(pat) + 1)
        elif kind == 4 and len(out) > 40_000:
            d = rng.choice([1, 2, 3, 4, 255, 256, 257, 258, 259, 4095, 4096, 32767, 32768])     # (copy from exactly that far back)
            for _ in range(rng.randint(1, 50)):
                k = rng.randint(3, 600)
                out += out[len(out) - d:len(out) - d + k] if d >= k else (out[len(out) - d:] * (k // d + 1))[:k]
        else:
            out += bytes(rng.randrange(32, 127) for _ in range(min(m, 5000)))
    return bytes(out[:n])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_streams_that_are_no_fastq(eng, tmp_path, seed):
    """The decoder against zlib on structured bytes, compressed in several ways, cut into small blocks by flushes, at small
    territories and segments: the text is zlib's, or the decoder says that the file is the host decoder's (None)."""
    rng = random.Random(seed)
    data = _structured_bytes(rng, 6_000_000 + rng.randrange(2_000_000))
    writers = {
        "level1": lambda d: gzip.compress(d, compresslevel=1),
        "level9": lambda d: gzip.compress(d, compresslevel=9),
        "flush-1k-9k": lambda d: _member(d, _raw_deflate(d, 6, flush_every=rng.randint(1000, 9000))),
        "full-flush-50k": lambda d: _member(d, _raw_deflate(d, rng.choice([1, 4, 9]), flush_every=50_000, flush=zlib.Z_FULL_FLUSH)),
        "filtered": lambda d: _member(d, _raw_deflate(d, 6, strategy=zlib.Z_FILTERED)),
        "memlevel1": lambda d: _member(d, (lambda co: co.compress(d) + co.flush())(zlib.compressobj(6, zlib.DEFLATED, -15, 1))),     # (blocks of 127 symbols)
        "window-512": lambda d: _member(d, (lambda co: co.compress(d) + co.flush())(zlib.compressobj(9, zlib.DEFLATED, -9))),
    }
    took = 0
    try:
        for name, w in writers.items():
            blob = w(data)
            assert gzip.decompress(blob) == data
            path = str(tmp_path / "x.gz")
            with open(path, "wb") as fh:
                fh.write(blob)
            for terr_kb, seg_kb in ((16, 1 << 20), (64, 900)):
                eng.set_option("gz_gpu_terr_kb", terr_kb)
                eng.set_option("gz_gpu_seg_kb", seg_kb)
                eng.set_option("gz_gpu_margin_kb", 512)
                got = eng.gunzip_file_gpu(path, len(data) + 64)
                assert got is None or got == data, (name, terr_kb, seg_kb)
                took += got is not None
    finally:
        eng.set_option("gz_gpu_seg_kb", 1 << 20)
        eng.set_option("gz_gpu_margin_kb", 16384)
        eng.set_option("gz_gpu_terr_kb", 128)
    assert took >= 8          # (most of them are the device decoder's: the thousandfold stretches overflow a chunk's tokens in some)


def test_progress_windows_through_the_device_decoder(sample, tmp_path):
    """find_tags_fastq's progress lines (reference :268-271) for a .gz file decoded on the device in small segments: the
    windows of 50 000 reads are those of the plain bytes counted in one piece."""
    import numpy as np
    import tagdigger_amd
    cfg, raw, want, ost = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=6))
    e = tagdigger_amd.Engine(0)
    try:
        e.set_option("progress", 1)
        e.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        e.count_bytes(raw)
        plain = [tuple(int(x) for x in w) for w in e.progress_windows()]
        lines_plain = e.progress_lines("lib.fq.gz")
        e.reset()
        e.set_option("gz_gpu_min", 0)
        e.set_option("gz_gpu_terr_kb", 64)
        e.set_option("gz_gpu_seg_kb", 2000)
        e.set_option("gz_gpu_margin_kb", 256)
        e.count_file(path)
        assert e.last_gz_route() == 1
        assert [tuple(int(x) for x in w) for w in e.progress_windows()] == plain
        assert e.progress_lines("lib.fq.gz") == lines_plain and len(lines_plain) == NREADS // 50000
        _check(e, want, ost, "with the progress windows")
    finally:
        e.close()


def test_campaign_against_zlib(eng, tmp_path):
    """Random structured bytes, random compressor settings (level, strategy, memLevel, window size, flushes), random territory
    and segment sizes, for TD_GZ_FUZZ_SECONDS seconds (default 15; a soak run uses minutes): the device decoder's text is
    zlib's, or it leaves the file to the host decoders."""
    import time
    budget = float(os.environ.get("TD_GZ_FUZZ_SECONDS", "15"))
    seed0 = int(os.environ.get("TD_GZ_FUZZ_SEED", "4242"))
    t_end = time.time() + budget
    ncase = took = 0
    try:
        while time.time() < t_end:
            rng = random.Random(seed0 + ncase)
            data = _structured_bytes(rng, rng.randint(200_000, 3_000_000))
            level = rng.choice([1, 1, 4, 6, 6, 9])
            strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
            co = zlib.compressobj(level, zlib.DEFLATED, -rng.choice([15, 15, 15, 12, 9]), rng.choice([8, 8, 9, 4, 1]), strategy)
            every = rng.choice([0, 0, 3000, 40_000, 400_000])
            parts = []
            if every:
                for i in range(0, len(data), every):
                    parts.append(co.compress(data[i:i + every]))
                    parts.append(co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
            else:
                parts.append(co.compress(data))
            parts.append(co.flush())
            blob = _member(data, b"".join(parts)) + b"\0" * rng.choice([0, 0, 7, 513])
            path = str(tmp_path / "c.gz")
            with open(path, "wb") as fh:
                fh.write(blob)
            eng.set_option("gz_gpu_terr_kb", rng.choice([16, 32, 64, 128]))
            eng.set_option("gz_gpu_seg_kb", rng.choice([1 << 20, 1 << 20, 300, 1100]))
            eng.set_option("gz_gpu_margin_kb", rng.choice([16384, 128, 512]))
            eng.set_option("gz_gpu_verify", rng.choice([0, 0, 1]))
            got = eng.gunzip_file_gpu(path, len(data) + 64)
            assert got is None or got == data, ("seed", seed0 + ncase)
            took += got is not None
            ncase += 1
    finally:
        for k, v in (("gz_gpu_seg_kb", 1 << 20), ("gz_gpu_margin_kb", 16384), ("gz_gpu_terr_kb", 128), ("gz_gpu_verify", 0)):
            eng.set_option(k, v)
    print(" [gzip campaign: %d cases, %d decoded on the device] " % (ncase, took), end="")
    assert ncase > 0 and took * 2 >= ncase
