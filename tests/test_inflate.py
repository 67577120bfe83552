"""The library's gzip reader (td_gunzip_file: what td_count_file / td_split_file read .gz files
through): ordinary and multi-member gzip via zlib, BGZF member-parallel.  Host only."""
import ctypes as C
import gzip
import os
import random

import pytest

from helpers import bgzf_bytes, gzip_one_member


def gunzip(path, capacity, chunk=0):
    from tagdigger_amd import _binding as B
    L = B.load()
    buf = (C.c_uint8 * max(1, capacity))()
    n = C.c_uint64(0)
    rc = L.td_gunzip_file(str(path).encode(), buf, capacity, chunk, C.byref(n))
    return rc, bytes(buf[:n.value])


@pytest.fixture(autouse=True, params=["sequential", "parallel-3000", "parallel-65536", "pipeline-3000", "pipeline-65536"])
def decoder(request, monkeypatch):
    """Every test runs with the one-thread decoder (fast_inflate.hpp) and with the chunk-parallel one
    (par_inflate.hpp) forced on with chunks so small that these files span many batches, chunks without a
    block start, members that end inside a batch and dropped chunks -- the latter both in its batch form (the host reader) and
    as the pipeline td_count_file drives for the GPU (TAGDIG_GUNZIP_PIPELINE: the same dev_next / dev_release / dev_check
    calls, the markers resolved by the host)."""
    if request.param.startswith("pipeline"):
        monkeypatch.setenv("TAGDIG_GUNZIP_PIPELINE", "1")
    if request.param == "sequential":
        monkeypatch.setenv("TAGDIG_PAR_INFLATE", "0")
    else:
        monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")
        monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", request.param.split("-")[1])
        monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "4")
    return request.param


@pytest.fixture(scope="module")
def fastq():
    rng = random.Random(7)
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGTN") for _ in range(rng.randrange(30, 150))), b"I" * 40)
                    for i in range(20000))


@pytest.mark.parametrize("kind", ["bgzf", "bgzf-small-blocks", "gzip", "two-members", "bgzf-empty", "pigz-like", "pigz-like-small-pieces"])
@pytest.mark.parametrize("chunk", [0, 97, 70000, 1 << 22])
def test_gunzip_file(tmp_path, fastq, kind, chunk):
    data = b"" if kind == "bgzf-empty" else fastq
    blob = {"bgzf": lambda: bgzf_bytes(data), "bgzf-small-blocks": lambda: bgzf_bytes(data, block=777, level=1),
            "gzip": lambda: gzip.compress(data), "two-members": lambda: gzip.compress(data[:999]) + gzip.compress(data[999:]),
            "bgzf-empty": lambda: bgzf_bytes(b""),
            # one member whose DEFLATE stream was compressed in independent pieces closed by full flushes (pigz)
            "pigz-like": lambda: gzip_one_member(data, level=1, threads=4, piece=1 << 17),
            "pigz-like-small-pieces": lambda: gzip_one_member(data, level=6, threads=2, piece=5000)}[kind]()
    p = tmp_path / "x.fq.gz"
    p.write_bytes(blob)
    rc, got = gunzip(p, len(data), chunk)
    assert rc == 0 and got == data
    assert gzip.decompress(blob) == data                 # the fixture itself is valid gzip


def test_gunzip_threads_env(tmp_path, fastq, monkeypatch):
    p = tmp_path / "x.fq.gz"
    p.write_bytes(bgzf_bytes(fastq))
    for n in ("1", "2", "5"):
        monkeypatch.setenv("TAGDIG_INFLATE_THREADS", n)
        rc, got = gunzip(p, len(fastq))
        assert rc == 0 and got == fastq


def test_gunzip_bgzf_corruption_is_an_error(tmp_path, fastq):
    blob = bytearray(bgzf_bytes(fastq))
    blob[len(blob) // 2] ^= 0x55                       # somewhere inside a member's deflate data
    p = tmp_path / "x.fq.gz"
    p.write_bytes(bytes(blob))
    rc, _ = gunzip(p, len(fastq))
    assert rc != 0
    p.write_bytes(bgzf_bytes(fastq)[:-40])             # truncated inside the last members
    rc, _ = gunzip(p, len(fastq))
    assert rc != 0


def test_gunzip_destination_too_small(tmp_path, fastq):
    p = tmp_path / "x.fq.gz"
    p.write_bytes(bgzf_bytes(fastq))
    rc, _ = gunzip(p, len(fastq) - 1)
    assert rc != 0


# ------------------------------------------------------------------ the library's own DEFLATE decoder
def _gz(data, level=6, strategy=0, mem=8):
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return co.compress(data) + co.flush()


_PAYLOADS = {}


def _payloads():
    if not _PAYLOADS:
        _PAYLOADS.update(_make_payloads())
    return _PAYLOADS


def _make_payloads():
    rng = random.Random(99)
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(100)),
                                          bytes(rng.choice(b"FFFFFF:,#") for _ in range(100))) for i in range(12000))
    return {
        "empty": b"",
        "one": b"A",
        "fastq": fq,                                                     # 2.6 MB: several staging refills
        "random": rng.randbytes(300000),                                 # incompressible: stored blocks at level 0, long codes
        "zeros": bytes(3_000_000),                                       # distance-1 runs, length 258
        "period3": b"abc" * 400000,                                      # distances 2..7
        "period7": b"abcdefg" * 200000,
        "far": rng.randbytes(32768) * 40,                                # matches at distance 32768
        "text": (b"the quick brown fox jumps over the lazy dog; " * 3000) + fq[:50000],
    }


@pytest.mark.parametrize("name", ["empty", "one", "fastq", "random", "zeros", "period3", "period7", "far", "text"])
@pytest.mark.parametrize("level,strategy", [(0, 0), (1, 0), (6, 0), (9, 0), (6, 4), (6, 2), (6, 3), (1, 1)])
def test_fast_inflate_equals_zlib(tmp_path, monkeypatch, name, level, strategy):
    """Z_FIXED (4) gives fixed-Huffman blocks, Z_HUFFMAN_ONLY (2) literal-only dynamic blocks, Z_RLE (3)
    distance-1 matches, level 0 stored blocks."""
    monkeypatch.delenv("TAGDIG_ZLIB", raising=False)
    data = _payloads()[name]
    p = tmp_path / "x.gz"
    p.write_bytes(_gz(data, level, strategy))
    for chunk in (0, 1 << 22, 4097):
        rc, got = gunzip(p, len(data), chunk)
        assert rc == 0 and got == data, (name, level, strategy, chunk)


def test_fast_inflate_header_fields_and_members(tmp_path, monkeypatch):
    import struct
    import zlib
    monkeypatch.delenv("TAGDIG_ZLIB", raising=False)
    rng = random.Random(3)
    parts = [bytes(rng.choice(b"ACGTN\n") for _ in range(n)) for n in (70000, 1, 0, 1500000)]

    def member(data, flg):
        raw = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = raw.compress(data) + raw.flush()
        head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\x03"
        if flg & 4:
            head += struct.pack("<H", 5) + b"hello"
        if flg & 8:
            head += b"reads.fq\0"
        if flg & 16:
            head += b"a comment\0"
        if flg & 2:
            head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
        return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)

    blob = b"".join(member(d, f) for d, f in zip(parts, (4 | 8 | 16 | 2, 8, 0, 4)))
    p = tmp_path / "m.gz"
    p.write_bytes(blob + b"\0" * 100)                      # trailing zero padding is ignored, as by gzread
    rc, got = gunzip(p, sum(map(len, parts)))
    assert rc == 0 and got == b"".join(parts)
    assert gzip.decompress(blob) == b"".join(parts)


def test_fast_inflate_rejects_damage(tmp_path, monkeypatch):
    """Flipped bytes and truncation: an error (never a crash, a hang or silently wrong data)."""
    monkeypatch.delenv("TAGDIG_ZLIB", raising=False)
    data = _payloads()["fastq"][:400000]
    good = _gz(data)
    rng = random.Random(17)
    p = tmp_path / "d.gz"
    wrong = 0
    for trial in range(150):
        blob = bytearray(good)
        if trial % 3 == 0:
            blob = blob[:rng.randrange(1, len(blob))]
        else:
            for _ in range(rng.randrange(1, 4)):
                blob[rng.randrange(10, len(blob))] ^= 1 << rng.randrange(8)
        p.write_bytes(bytes(blob))
        rc, got = gunzip(p, len(data) + 1000)
        if rc == 0:
            assert got == data               # (a flip that only touched the header's mtime, say)
        else:
            wrong += 1
    assert wrong > 120


def test_zlib_fallback_env(tmp_path, monkeypatch):
    data = _payloads()["text"]
    p = tmp_path / "x.gz"
    p.write_bytes(_gz(data))
    monkeypatch.setenv("TAGDIG_ZLIB", "1")
    rc, got = gunzip(p, len(data))
    assert rc == 0 and got == data


def _stats_line(capfd):
    err = capfd.readouterr().err
    lines = [ln for ln in err.splitlines() if ln.startswith("par_inflate:")]
    return lines[-1] if lines else None


def test_parallel_decoder_chains_chunks(tmp_path, monkeypatch, capfd, decoder):
    """With 64 KiB chunks the chains are longer than one chunk (block starts are found and confirmed by the
    predecessor); with 3000-byte chunks a batch is shorter than a block, holds no block start, and the first
    chunk decodes alone.  The output is byte-identical either way."""
    import re
    if decoder == "sequential":
        pytest.skip("parallel decoder only")
    monkeypatch.setenv("TAGDIG_INFLATE_STATS", "1")
    data = _payloads()["fastq"]
    p = tmp_path / "x.gz"
    p.write_bytes(_gz(data, 6))
    rc, got = gunzip(p, len(data))
    assert rc == 0 and got == data
    line = _stats_line(capfd)
    assert line, "the chunk-parallel decoder did not run"
    batches, chunks = map(int, re.search(r"(\d+) batches, (\d+) chunks", line).groups())
    if decoder.startswith("pipeline"):       # (there a chunk without a block start is decoded by the chaining thread, and counts)
        assert chunks >= batches >= 1, line
    else:
        assert (chunks > batches if decoder == "parallel-65536" else chunks == batches) and batches >= 1, line


def test_parallel_decoder_is_the_default_for_large_files(tmp_path, monkeypatch, capfd, decoder):
    """No TAGDIG_PAR_INFLATE: files from 8 MiB of compressed data go through the chunk-parallel decoder,
    smaller ones through the one-thread decoder."""
    if decoder != "sequential":
        pytest.skip("once is enough")
    monkeypatch.delenv("TAGDIG_PAR_INFLATE", raising=False)
    monkeypatch.setenv("TAGDIG_INFLATE_STATS", "1")
    monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "4")
    rng = random.Random(5)
    big = rng.randbytes(6 << 20) + _payloads()["fastq"] * 2 + rng.randbytes(3 << 20)     # > 8 MiB even compressed
    p = tmp_path / "big.gz"
    p.write_bytes(_gz(big, 1))
    assert os.path.getsize(p) >= 8 << 20
    rc, got = gunzip(p, len(big))
    assert rc == 0 and got == big
    assert _stats_line(capfd)
    small = _payloads()["fastq"]
    p.write_bytes(_gz(small, 6))
    rc, got = gunzip(p, len(small))
    assert rc == 0 and got == small
    assert _stats_line(capfd) is None


# ------------------------------------------------------------------ the decoder the GPU runs per lane, on the host
def test_lane_decoder_against_zlib():
    """csrc/gpu_inflate.hpp through td_inflate_raw_host: stored, fixed and dynamic blocks, several blocks per
    stream, codes longer than the primary tables' index, every zlib level -- and damaged streams fail cleanly."""
    import ctypes as C
    import random
    import zlib
    from tagdigger_amd import _binding as B
    L = B.load()

    def inflate(comp, n):
        out = (C.c_char * max(1, n))()
        return L.td_inflate_raw_host(comp, len(comp), out, n), bytes(out[:n])
    rnd = random.Random(7)

    def fastq(n):
        return "".join("@r%012d\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGT") for _ in range(100)),
                                                     "".join(rnd.choice("IIIIIHGF#5<") for _ in range(100))) for i in range(n)).encode()
    inputs = [b"", b"a", b"abc" * 5, bytes(rnd.randrange(256) for _ in range(9000)), fastq(290), b"A" * 65000,
              bytes(rnd.choice(b"ACGT") for _ in range(60000)), bytes(range(256)) * 200, fastq(300)[:0xFF00]]
    for data in inputs:
        for level in (0, 1, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                comp = co.compress(data) + co.flush()
                rc, out = inflate(comp, len(data))
                assert rc == 0 and out == data, (len(data), level, strategy, rc)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    d1, d2 = fastq(100), bytes(rnd.randrange(256) for _ in range(3000))
    comp = co.compress(d1) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d2) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d1) + co.flush()
    assert inflate(comp, len(d1 + d2 + d1)) == (0, d1 + d2 + d1)
    assert inflate(comp, len(d1 + d2 + d1) - 3)[0] != 0 and inflate(comp, len(d1 + d2 + d1) + 3)[0] != 0     # the size is part of the contract
    data = fastq(200)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    for _ in range(400):                                                   # no crash, no hang; mostly rejected
        b = bytearray(comp)
        for _ in range(rnd.randint(1, 5)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        inflate(bytes(b), len(data))
    assert inflate(comp[:len(comp) // 2], len(data))[0] != 0


def test_random_streams_through_both_forms_of_the_parallel_decoder(decoder):
    """tests/repro/gzip_pipeline_fuzz.py for TD_FUZZ_SECONDS (default 15) in a child process with a deadline: members of
    independently compressed pieces (every strategy, stored blocks, sync and full flushes), one to three members, random
    chunk sizes / thread counts, damaged copies -- equal to zlib's bytes or an error, and never a hang."""
    import subprocess
    import sys
    if decoder != "sequential":
        pytest.skip("once is enough (the script sets the decoder's environment itself)")
    seconds = float(os.environ.get("TD_FUZZ_SECONDS", "15"))
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "repro", "gzip_pipeline_fuzz.py")
    try:
        r = subprocess.run([sys.executable, script, str(seconds), "20261004"], capture_output=True, text=True, timeout=seconds + 120)
    except subprocess.TimeoutExpired as e:
        raise AssertionError("the decoder hung: " + str(e.stdout)[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout
