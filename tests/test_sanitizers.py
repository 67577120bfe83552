"""The host-side gzip readers under ThreadSanitizer and AddressSanitizer + UBSan (CPU builds only: make -C
tagdigger_amd/csrc san; never the GPU build).  ~3 000 lines of multithreaded C++ stand between a .gz file and the
count kernel -- par_inflate.hpp: sixteen workers and a chaining thread over a ring of chunk buffers; gz_source.hpp: the
BGZF member pool; fast_inflate.hpp; gz_pyrules.hpp -- and san/gz_san.cpp drives all of them over one file and compares
them with what gzip.open does.  Good streams of every kind the decoder distinguishes, damaged ones, tiny chunks (most
territories without a block start, members that end inside a territory)."""
import gzip
import os
import random
import shutil
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tagdigger_amd", "csrc")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")


@pytest.fixture(scope="module")
def binaries():
    subprocess.check_call(["make", "-s", "-C", CSRC, "san"])
    return {"tsan": os.path.join(CSRC, "san", "gz_san_tsan"), "asan": os.path.join(CSRC, "san", "gz_san_asan")}


def _fastq(nrec, seed):
    rng = random.Random(seed)
    out = []
    for i in range(nrec):
        seq = bytes(rng.choice(b"ACGTN") for _ in range(rng.randrange(30, 150)))
        out.append(b"@r%d some text\n%s\n+\n%s\n" % (i, seq, b"I" * len(seq)))
    return b"".join(out)


def _raw(data, level, flush_every=0, full=False):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for i in range(0, len(data), flush_every):
        out.append(co.compress(data[i:i + flush_every]))
        out.append(co.flush(zlib.Z_FULL_FLUSH if full else zlib.Z_SYNC_FLUSH))
    return b"".join(out) + co.flush()


def _member(data, body):
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def _streams():
    d = _fastq(12000, 5)                       # ~2.5 MB of text
    rng = random.Random(9)
    good = {
        "level1": gzip.compress(d, 1), "level6": gzip.compress(d, 6), "level9": gzip.compress(d, 9),
        "two-members": gzip.compress(d[:len(d) // 3], 6) + gzip.compress(d[len(d) // 3:], 1),
        "member-of-nothing-inside-a-line": gzip.compress(d[:1001], 6) + gzip.compress(b"") + gzip.compress(d[1001:], 6),
        "sync-flushes": _member(d, _raw(d, 6, 50_000)), "full-flushes": _member(d, _raw(d, 1, 100_000, full=True)),
        "stored": gzip.compress(d[:300_000], 0) + gzip.compress(d[300_000:], 6),
        "zero-padding": gzip.compress(d, 6) + b"\0" * 100,
    }
    g6 = good["level6"]
    bad = {"truncated": g6[:len(g6) // 2], "crc": g6[:-8] + bytes([g6[-8] ^ 1]) + g6[-7:], "junk": g6 + b"junk",
           "second-member-cut": good["two-members"][:-3000]}
    for k in range(3):
        at = rng.randrange(1000, len(g6) - 1000)
        bad["flip-%d" % k] = g6[:at] + bytes([g6[at] ^ (1 << rng.randrange(8))]) + g6[at + 1:]
    return good, bad


GOOD, BAD = _streams()


@pytest.mark.parametrize("chunk", ["3000", "65536"])
@pytest.mark.parametrize("san", ["tsan", "asan"])
@pytest.mark.parametrize("name", sorted(GOOD) + sorted(BAD))
def test_readers_under_sanitizers(binaries, tmp_path, name, san, chunk):
    blob = GOOD.get(name) or BAD[name]
    p = tmp_path / "x.fq.gz"
    p.write_bytes(blob)
    env = dict(os.environ, TAGDIG_INFLATE_THREADS="6", TAGDIG_INFLATE_CHUNK=chunk,
               TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("TAGDIG_ZLIB", None)
    r = subprocess.run([binaries[san], str(p)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, (name, san, chunk, r.stdout[-500:], r.stderr[-3000:])
    assert ("gzip.open reads it" in r.stdout) == (name in GOOD), r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
