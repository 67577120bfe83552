#!/usr/bin/env python3
"""Random gzip streams through the chunk-parallel decoder, in both of its forms (the host reader's batches and the pipeline
td_count_file drives for the GPU, here with the markers resolved on the host: TAGDIG_GUNZIP_PIPELINE), against zlib.

  tests/repro/gzip_pipeline_fuzz.py SECONDS [SEED]

Streams: members made of independently compressed pieces (random level and strategy -- fixed-Huffman, Huffman-only, RLE,
stored -- closed by full flushes, sync flushes inside), over text that looks like FASTQ, random bytes, zeros and short
periods; one to three members; damaged copies now and then (must fail, not hang or return other bytes).  Chunk size,
thread count and over-subscription change from case to case (the library reads them from the environment at every open).
Prints one line per failure and a summary; exit code 1 on any failure.  Run by tests/test_inflate.py in a child process
with a deadline, so that a hang is a failure too."""
import ctypes as C
import os
import random
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tagdigger_amd import _binding as B   # noqa: E402

L = B.load()


def gunzip(path, capacity):
    buf = (C.c_uint8 * max(1, capacity))()
    n = C.c_uint64(0)
    rc = L.td_gunzip_file(path.encode(), buf, capacity, 0, C.byref(n))
    return rc, bytes(buf[:n.value]) if rc == 0 else b""


def content(rng, n):
    kind = rng.random()
    if kind < 0.5:
        out = []
        size = 0
        i = rng.randrange(10 ** 6)
        qual = rng.random() < 0.5
        while size < n:
            ln = rng.randrange(20, 160)
            rec = b"@r%09d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGTN") for _ in range(ln)),
                                            b"I" * ln if qual else bytes(rng.choice(b"FGHIJ#5,") for _ in range(ln)))
            out.append(rec)
            size += len(rec)
            i += 1
        return b"".join(out)[:n]
    if kind < 0.65:
        return rng.randbytes(n)
    if kind < 0.8:
        return bytes(n)
    period = rng.randbytes(rng.randrange(1, 40))
    return (period * (n // len(period) + 1))[:n]


def member(rng, data):
    pieces, pos = [], 0
    while True:
        n = min(len(data) - pos, rng.choice([1, 200, 5000, 70000, 400000, 3000000]))
        last = pos + n >= len(data)
        level = rng.choice([0, 1, 1, 6, 6, 9])
        strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 8, 9]), strategy)
        part = data[pos:pos + n]
        body = []
        if rng.random() < 0.3 and n > 10:
            cut = rng.randrange(1, n)
            body.append(co.compress(part[:cut]))
            body.append(co.flush(zlib.Z_SYNC_FLUSH))
            body.append(co.compress(part[cut:]))
        else:
            body.append(co.compress(part))
        body.append(co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH))
        pieces.append(b"".join(body))
        pos += n
        if last:
            break
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + b"".join(pieces)
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gzfuzz_%d.gz" % os.getpid())
    deadline = time.time() + seconds
    cases = bad = damaged_ok = 0
    os.environ["TAGDIG_PAR_INFLATE"] = "1"
    try:
        while time.time() < deadline:
            datas = [content(rng, rng.choice([0, 1, 3000, 200_000, 2_000_000, 6_000_000])) for _ in range(rng.choice([1, 1, 1, 2, 3]))]
            blob = b"".join(member(rng, d) for d in datas)
            want = b"".join(datas)
            for pipeline in (False, True):
                os.environ["TAGDIG_INFLATE_CHUNK"] = str(rng.choice([1024, 3000, 17000, 65536, 1 << 20]))
                os.environ["TAGDIG_INFLATE_THREADS"] = str(rng.choice([2, 3, 8]))
                os.environ["TAGDIG_INFLATE_OVERSUB"] = str(rng.choice([1, 2, 4]))
                if pipeline:
                    os.environ["TAGDIG_GUNZIP_PIPELINE"] = "1"
                else:
                    os.environ.pop("TAGDIG_GUNZIP_PIPELINE", None)
                what = "seed %d case %d pipeline=%d chunk=%s threads=%s oversub=%s" % (
                    seed, cases, pipeline, os.environ["TAGDIG_INFLATE_CHUNK"], os.environ["TAGDIG_INFLATE_THREADS"], os.environ["TAGDIG_INFLATE_OVERSUB"])
                with open(path, "wb") as fh:
                    fh.write(blob)
                rc, got = gunzip(path, len(want) + 1)
                cases += 1
                if rc != 0 or got != want:
                    bad += 1
                    print("FAIL (good stream): rc %d, %d bytes of %d; %s" % (rc, len(got), len(want), what), flush=True)
                if len(blob) > 40 and rng.random() < 0.3:
                    b = bytearray(blob)
                    if rng.random() < 0.4:
                        b = b[:rng.randrange(11, len(b))]
                    else:
                        b[rng.randrange(10, len(b))] ^= 1 << rng.randrange(8)
                    with open(path, "wb") as fh:
                        fh.write(bytes(b))
                    rc, got = gunzip(path, len(want) + 1)
                    if rc == 0 and got != want:
                        try:
                            ref = zlib.decompressobj(31).decompress(bytes(b))      # (what zlib makes of the first member)
                        except zlib.error:
                            ref = None
                        if ref is None or not want.startswith(got):
                            bad += 1
                            print("FAIL (damaged stream returned other bytes): %s" % what, flush=True)
                    else:
                        damaged_ok += 1
    finally:
        if os.path.exists(path):
            os.remove(path)
    print("gzip pipeline fuzz: seed %d, %d cases, %d damaged handled, %d failures" % (seed, cases, damaged_ok, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
