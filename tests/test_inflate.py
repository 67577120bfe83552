"""The library's gzip reader (td_gunzip_file: what td_count_file / td_split_file read .gz files
through): ordinary and multi-member gzip via zlib, BGZF member-parallel.  Host only."""
import ctypes as C
import gzip
import os
import random

import pytest

from helpers import bgzf_bytes


def gunzip(path, capacity, chunk=0):
    from tagdigger_amd import _binding as B
    L = B.load()
    buf = (C.c_uint8 * max(1, capacity))()
    n = C.c_uint64(0)
    rc = L.td_gunzip_file(str(path).encode(), buf, capacity, chunk, C.byref(n))
    return rc, bytes(buf[:n.value])


@pytest.fixture(scope="module")
def fastq():
    rng = random.Random(7)
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGTN") for _ in range(rng.randrange(30, 150))), b"I" * 40)
                    for i in range(20000))


@pytest.mark.parametrize("kind", ["bgzf", "bgzf-small-blocks", "gzip", "two-members", "bgzf-empty"])
@pytest.mark.parametrize("chunk", [0, 97, 70000, 1 << 22])
def test_gunzip_file(tmp_path, fastq, kind, chunk):
    data = b"" if kind == "bgzf-empty" else fastq
    blob = {"bgzf": lambda: bgzf_bytes(data), "bgzf-small-blocks": lambda: bgzf_bytes(data, block=777, level=1),
            "gzip": lambda: gzip.compress(data), "two-members": lambda: gzip.compress(data[:999]) + gzip.compress(data[999:]),
            "bgzf-empty": lambda: bgzf_bytes(b"")}[kind]()
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
