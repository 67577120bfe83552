#!/usr/bin/env python3
"""Generate tests/golden/gzip_damage.json from the REAL reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_gzip_damage_golden.py [/root/reference]

What find_tags_fastq (reference tagdigger_fun.py:240-243, the exception leaves the loop at :250) does with a
.gz file that is damaged, padded or empty: which exception class comes out, with which message, or which
matrix.  Only DATA is written (the payloads, deterministic: mtime 0, and what the reference returned / raised);
the payloads are built here, none of the reference's text is.
"""
import base64
import contextlib
import gzip
import io
import json
import os
import random
import struct
import sys
import tempfile
import zlib

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import tagdigger_fun as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
B = ["AACG", "TTGACC"]
T = ["TGCAGAAAC", "TGCAGGGGT"]


def rec(seq, hdr):
    return ("%s\n%s\n+\n%s\n" % (hdr, seq, "I" * len(seq))).encode()


def records(n, seed):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        body = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(5, 60)))
        out.append(rec(rnd.choice(B) + rnd.choice(T) + body if rnd.random() < 0.8 else body, "@r%d" % i))
    return b"".join(out)


def gz(data, level=6):
    return gzip.compress(data, compresslevel=level, mtime=0)


def run_ref(payload, **kw):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.fq.gz")
        with open(path, "wb") as fh:
            fh.write(payload)
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out):
                res = ref.find_tags_fastq(path, list(B), list(T), **kw)
            return {"counts": res}
        except Exception as e:
            mod = type(e).__module__
            return {"raises": type(e).__name__, "module": mod, "message": str(e),
                    "bases": [c.__name__ for c in type(e).__mro__[1:-2]]}


def main():
    a = records(40, 1)            # member 1
    b = records(25, 2)            # member 2
    big = records(3000, 3)        # > 64 KiB of text: several deflate blocks at level 1, reads beyond the first buffers
    bases = {"ga": gz(a), "gb": gz(b), "gbig": gz(big, 1)}
    cases = []

    # a payload is a list of parts: a slice of a base member with bits flipped in it, or literal bytes (so that the three
    # members are written down once)
    def P(base, lo=0, hi=None, flips=()):
        n = len(bases[base])
        hi = n if hi is None else hi
        return {"base": base, "lo": lo % (n + 1) if lo >= 0 else n + lo, "hi": hi if hi >= 0 else n + hi,
                "flips": [[at if at >= 0 else n + at, bit] for at, bit in flips]}

    def L(raw):
        return {"lit": base64.b64encode(raw).decode("ascii")}

    def build(parts):
        out = b""
        for part in parts:
            if "lit" in part:
                out += base64.b64decode(part["lit"])
                continue
            q = bytearray(bases[part["base"]])
            for at, bit in part["flips"]:
                q[at] ^= 1 << bit
            out += bytes(q[part["lo"]:part["hi"]])
        return out

    def case(name, parts, **kw):
        payload = build(parts)
        r = {"name": name, "parts": parts, "kwargs": kw}
        r.update(run_ref(payload, **kw))
        cases.append(r)
        print("%-44s %s" % (name, r.get("raises", "counts") + (": " + r["message"] if "raises" in r else "")))

    na, nb, nbig = len(bases["ga"]), len(bases["gb"]), len(bases["gbig"])
    case("good one member", [P("ga")])
    case("good two members", [P("ga"), P("gb")])
    # ---- the stream ends early
    case("truncated in the deflate data", [P("ga", 0, na // 2)])
    case("truncated: last byte missing", [P("ga", 0, -1)])
    case("truncated: ISIZE missing", [P("ga", 0, -4)])
    case("truncated: CRC and ISIZE missing", [P("ga", 0, -8)])
    case("truncated: trailer and final byte missing", [P("ga", 0, -9)])
    case("truncated in the header", [P("ga", 0, 5)])
    case("truncated after the magic", [P("ga", 0, 2)])
    case("one byte of magic", [P("ga", 0, 1)])
    case("second member truncated", [P("ga"), P("gb", 0, nb // 2)])
    case("second member: header only", [P("ga"), P("gb", 0, 10)])
    case("second member: magic only", [P("ga"), P("gb", 0, 2)])
    case("second member: one byte of magic", [P("ga"), P("gb", 0, 1)])
    case("big member truncated late", [P("gbig", 0, -100)])
    case("big member truncated early", [P("gbig", 0, 3000)])
    # ---- a member fails its checks
    case("CRC-32 flipped", [P("ga", flips=[(-8, 0)])])
    case("ISIZE flipped", [P("ga", flips=[(-4, 0)])])
    case("CRC-32 and ISIZE flipped", [P("ga", flips=[(-8, 0), (-4, 0)])])
    case("second member: CRC-32 flipped", [P("ga"), P("gb", flips=[(-6, 3)])])
    case("second member: ISIZE flipped", [P("ga"), P("gb", flips=[(-1, 7)])])
    case("big member: CRC-32 flipped", [P("gbig", flips=[(-7, 5)])])
    # ---- what follows the last member
    case("trailing junk", [P("ga"), L(b"junk!")])
    case("trailing junk, one byte", [P("ga"), L(b"x")])
    case("trailing junk with quotes", [P("ga"), L(b"'\"")])
    case("trailing junk after zero padding", [P("ga"), L(b"\0" * 7 + b"junk")])
    case("zero padding after the last member", [P("ga"), L(b"\0" * 512)])
    case("one zero byte after the last member", [P("ga"), L(b"\0")])
    case("zero padding between members", [P("ga"), L(b"\0" * 33), P("gb")])
    case("a second magic with a bad method", [P("ga"), L(b"\x1f\x8b\x07"), P("gb", 3)])
    # ---- nothing at all
    case("empty file", [])
    case("member of nothing", [L(gz(b""))])
    case("two members of nothing", [L(gz(b"") + gz(b""))])
    case("member of nothing, then data", [L(gz(b"")), P("ga")])
    case("data, a member of nothing inside a line, data", [L(gz(a[:100]) + gz(b"") + gz(a[100:]))])
    case("only zero bytes", [L(b"\0" * 64)])
    # ---- header variants gzip.open accepts or refuses
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(a) + raw.flush()
    trailer = struct.pack("<II", zlib.crc32(a), len(a) & 0xFFFFFFFF)
    case("header with a file name", [L(b"\x1f\x8b\x08\x08" + b"\0" * 4 + b"\x00\xff" + b"name.fq\0" + body + trailer)])
    case("header with extra field, name, comment and header CRC",
         [L(b"\x1f\x8b\x08\x1e" + b"\0" * 4 + b"\x00\xff" + b"\x04\x00abcd" + b"n\0" + b"c\0" + b"\x12\x34" + body + trailer)])
    case("header with reserved flag bits", [L(b"\x1f\x8b\x08\xe0" + b"\0" * 4 + b"\x00\xff" + body + trailer)])
    case("header whose file name never ends", [L(b"\x1f\x8b\x08\x08" + b"\0" * 4 + b"\x00\xff" + b"name")])
    case("header whose extra field is cut", [L(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + b"\x10\x00abc")])
    case("unknown compression method", [L(b"\x1f\x8b\x07\x00" + b"\0" * 4 + b"\x00\xff" + body + trailer)])
    case("not gzip at all", [L(b"@r0\nAACGTGCAGAAAC\n+\nIIIIIIIIIIIII\n")])
    # ---- damage inside the deflate data (what comes out depends on where it lands: the reference's answer is recorded)
    rnd = random.Random(99)
    for k in range(14):
        at = rnd.randrange(10, nbig - 8)
        case("big member: bit flipped at %d" % at, [P("gbig", flips=[(at, rnd.randrange(8))])])
    for k in range(10):
        at = rnd.randrange(10, na - 8)
        case("small member: bit flipped at %d" % at, [P("ga", flips=[(at, rnd.randrange(8))])])
    # ---- damage the loop never reaches (reference :272 leaves at maxreads; what it has read ahead by then decides)
    case("maxreads=3, CRC-32 flipped", [P("ga", flips=[(-8, 0)])], maxreads=3)
    case("maxreads=3, second member truncated", [P("ga"), P("gb", 0, nb // 2)], maxreads=3)
    case("maxreads=3, trailing junk", [P("ga"), L(b"junk!")], maxreads=3)
    case("maxreads=40 (all of member 1), second member CRC-32 flipped", [P("ga"), P("gb", flips=[(-6, 3)])], maxreads=40)
    case("maxreads=41, second member CRC-32 flipped", [P("ga"), P("gb", flips=[(-6, 3)])], maxreads=41)
    case("maxreads=100, big member truncated late", [P("gbig", 0, -100)], maxreads=100)
    case("maxreads=2999, big member: CRC-32 flipped", [P("gbig", flips=[(-7, 5)])], maxreads=2999)
    case("maxreads=3000, big member: CRC-32 flipped", [P("gbig", flips=[(-7, 5)])], maxreads=3000)
    # (how far ahead of the loop the damage must lie: truncations walked towards the bound, and an invalid code)
    for cut in (3000, 2200, 1500, 1100, 800):
        for mr in (20, 60, 100):
            case("maxreads=%d, big member cut at %d" % (mr, cut), [P("gbig", 0, cut)], maxreads=mr)
    bad_at = None
    for at in range(2000, 2400):
        q = bytearray(bases["gbig"]); q[at] ^= 0x10
        try:
            gzip.decompress(bytes(q))
        except zlib.error:
            bad_at = at
            break
        except Exception:
            pass
    if bad_at is not None:
        for mr in (20, 40, 60, 80, 100, 140):
            case("maxreads=%d, big member: invalid data at %d" % (mr, bad_at), [P("gbig", flips=[(bad_at, 4)])], maxreads=mr)
    with open(os.path.join(HERE, "gzip_damage.json"), "w") as fh:
        json.dump({"barcodes": B, "tags": T, "bases": {k: base64.b64encode(v).decode("ascii") for k, v in bases.items()},
                   "cases": cases}, fh, indent=0, separators=(",", ":"))
        fh.write("\n")
    print("wrote gzip_damage.json", os.path.getsize(os.path.join(HERE, "gzip_damage.json")), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
