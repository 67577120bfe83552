"""A .gz input that is damaged, padded or empty ends the way the reference ends it (tagdigger_fun.py:240-243, :250,
:272-273): gzip.open's exception -- EOFError for a stream that stops early, gzip.BadGzipFile for a member that fails
its check or bytes that are no header, zlib.error for invalid DEFLATE data -- with its message, or the matrix where the
loop is through before it meets the damage.  tests/golden/gzip_damage.json holds what the REAL reference did on 96
files (make_gzip_damage_golden.py).

Here, without a GPU: the library's statement of the reference's reading rules (csrc/gz_pyrules.hpp, through
td_gzip_check) on every case; td_gunzip_file -- all three forms of the decoder -- on the cases without a bound; the Python
oracle.  With a GPU (-m gpu): find_tags_fastq itself, on every gzip route."""
import base64
import ctypes as C
import gzip
import zlib

import pytest

from conftest import load_golden

GOLD = load_golden("gzip_damage.json")
BASES = {k: base64.b64decode(v) for k, v in GOLD["bases"].items()}
CASES = GOLD["cases"]
CLASSES = {"EOFError": EOFError, "BadGzipFile": gzip.BadGzipFile, "error": zlib.error}


def payload(case):
    out = b""
    for part in case["parts"]:
        if "lit" in part:
            out += base64.b64decode(part["lit"])
            continue
        q = bytearray(BASES[part["base"]])
        for at, bit in part["flips"]:
            q[at] ^= 1 << bit
        out += bytes(q[part["lo"]:part["hi"]])
    return out


def ids(cases):
    return [c["name"].replace(" ", "_") for c in cases]


def expect(case, call):
    """call() must end as the reference did: same class, same message, or no exception."""
    if "raises" in case:
        with pytest.raises(CLASSES[case["raises"]]) as ei:
            call()
        assert type(ei.value) is CLASSES[case["raises"]]
        assert str(ei.value) == case["message"]
        return None
    return call()


def test_fixture_covers_the_three_classes_and_the_bound():
    kinds = {c.get("raises", "counts") for c in CASES}
    assert kinds == {"counts", "EOFError", "BadGzipFile", "error"}
    bounded = [c for c in CASES if c["kwargs"]]
    assert any("raises" in c for c in bounded) and any("counts" in c for c in bounded)
    assert issubclass(gzip.BadGzipFile, OSError)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_reading_rules_give_the_reference_outcome(case, tmp_path):
    from tagdigger_amd import _binding as B
    from tagdigger_amd.engine import effective_maxreads
    L = B.load()
    p = tmp_path / "x.fq.gz"
    p.write_bytes(payload(case))
    expect(case, lambda: B.check(L.td_gzip_check(str(p).encode(), effective_maxreads(case["kwargs"].get("maxreads", 5e9)))))


UNBOUNDED = [c for c in CASES if not c["kwargs"]]


@pytest.mark.parametrize("decoder", ["sequential", "parallel-3000", "pipeline-3000"])
@pytest.mark.parametrize("case", UNBOUNDED, ids=ids(UNBOUNDED))
def test_gunzip_file_ends_as_gzip_open_does(case, decoder, tmp_path, monkeypatch):
    from tagdigger_amd import _binding as B
    L = B.load()
    if decoder.startswith("pipeline"):
        monkeypatch.setenv("TAGDIG_GUNZIP_PIPELINE", "1")
    if decoder == "sequential":
        monkeypatch.setenv("TAGDIG_PAR_INFLATE", "0")
    else:
        monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")
        monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", decoder.split("-")[1])
        monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "4")
    blob = payload(case)
    p = tmp_path / "x.fq.gz"
    p.write_bytes(blob)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)()
    n = C.c_uint64(0)

    def call():
        B.check(L.td_gunzip_file(str(p).encode(), buf, cap, 0, C.byref(n)))
        return bytes(buf[:n.value])
    got = expect(case, call)
    if got is not None:
        with gzip.open(str(p), "rb") as fh:
            assert got == fh.read()


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_python_oracle_on_the_damaged_files(case, tmp_path):
    """The restatement reads .gz files the way the reference does (gzip.open): it must end the same way."""
    from oracle import tagdigger_oracle as O
    p = tmp_path / "x.fq.gz"
    p.write_bytes(payload(case))
    got = expect(case, lambda: O.find_tags_fastq(str(p), list(GOLD["barcodes"]), list(GOLD["tags"]), **case["kwargs"]))
    if got is not None:
        assert [list(r) for r in got] == case["counts"]


ROUTES = {
    # what td_count_file does with a .gz that is not BGZF, by size and options (csrc/tagdig.hip td_count_file)
    "sequential": {"env": {"TAGDIG_PAR_INFLATE": "0"}, "gpu_resolve": 1},                       # below 8 MiB: fast_inflate.hpp on the calling thread
    "chunks-gpu-resolve": {"env": {"TAGDIG_PAR_INFLATE": "1", "TAGDIG_INFLATE_CHUNK": "3000", "TAGDIG_INFLATE_THREADS": "4"}, "gpu_resolve": 1},   # count_gzip_dev
    "chunks-host-resolve": {"env": {"TAGDIG_PAR_INFLATE": "1", "TAGDIG_INFLATE_CHUNK": "3000", "TAGDIG_INFLATE_THREADS": "4"}, "gpu_resolve": 0},  # par_inflate.hpp, batches
    # single-member files of 8 MiB and more: Huffman decoding and LZ77 on the device (gz_gpu.hpp); what it does not chain falls to the routes above
    "device-decoder": {"env": {"TAGDIG_PAR_INFLATE": "0"}, "gpu_resolve": 1, "opts": {"gz_gpu_min": 0, "gz_gpu_terr_kb": 16}},
}
ROUTE_DEFAULTS = {"gz_gpu_min": 8 << 20, "gz_gpu_terr_kb": 128}


@pytest.mark.gpu
@pytest.mark.parametrize("route", sorted(ROUTES))
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_find_tags_fastq_on_the_damaged_files(case, route, tmp_path, capsys, monkeypatch):
    import tagdigger_amd.tagdigger_fun as tf
    from tagdigger_amd.engine import default_engine
    p = tmp_path / "x.fq.gz"
    p.write_bytes(payload(case))
    for k, v in ROUTES[route]["env"].items():
        monkeypatch.setenv(k, v)
    eng = default_engine(0)
    eng.set_option("gpu_resolve", ROUTES[route]["gpu_resolve"])
    for k, v in ROUTES[route].get("opts", {}).items():
        eng.set_option(k, v)
    try:
        got = expect(case, lambda: tf.find_tags_fastq(str(p), list(GOLD["barcodes"]), list(GOLD["tags"]), **case["kwargs"]))
    finally:
        eng.set_option("gpu_resolve", 1)
        for k, v in ROUTE_DEFAULTS.items():
            eng.set_option(k, v)
    capsys.readouterr()
    if got is not None:
        assert got == case["counts"]
    # the next good file counts as if nothing had happened
    good = tmp_path / "good.fq.gz"
    good.write_bytes(BASES["ga"])
    assert tf.find_tags_fastq(str(good), list(GOLD["barcodes"]), list(GOLD["tags"]), progress=False) == CASES[0]["counts"]


@pytest.mark.gpu
def test_damaged_bgzf_files_end_as_gzip_open_ends_them(tmp_path, capsys):
    """BGZF is inflated member-parallel (on the GPU, or on the host's threads): a damaged file still ends in the
    exception gzip.open raises on it, read member after member."""
    import random
    import tagdigger_amd.tagdigger_fun as tf
    from tagdigger_amd.engine import default_engine
    from helpers import bgzf_bytes
    from oracle import tagdigger_oracle as O
    rnd = random.Random(5)
    recs = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, ("AACGTGCAGAAAC" + "".join(rnd.choice("ACGT") for _ in range(40))).encode(), b"I" * 53)
                    for i in range(6000))
    good = bgzf_bytes(recs, block=20000)
    Bc, Tg = list(GOLD["barcodes"]), list(GOLD["tags"])
    variants = {
        "truncated": good[:len(good) * 2 // 3],
        "crc": good[:5000] + bytes([good[5000] ^ 0x40]) + good[5001:],
        "junk": good + b"junk",
        "padding": good + b"\0" * 100,
    }
    eng = default_engine(0)
    for gpu_inflate in (1, 0):
        eng.set_option("gpu_inflate", gpu_inflate)
        try:
            for name, blob in variants.items():
                p = tmp_path / ("%s.fq.gz" % name)
                p.write_bytes(blob)
                for kw in ({}, {"maxreads": 50}):
                    try:
                        want = O.find_tags_fastq(str(p), Bc, Tg, **kw)
                        exc = None
                    except Exception as e:       # noqa: BLE001 -- what gzip.open raised
                        want, exc = None, e
                    if exc is None:
                        assert tf.find_tags_fastq(str(p), Bc, Tg, progress=False, **kw) == want, (name, kw, gpu_inflate)
                    else:
                        with pytest.raises(type(exc)) as ei:
                            tf.find_tags_fastq(str(p), Bc, Tg, progress=False, **kw)
                        assert type(ei.value) is type(exc) and str(ei.value) == str(exc), (name, kw, gpu_inflate)
        finally:
            eng.set_option("gpu_inflate", 1)
    capsys.readouterr()


def _loop_over_gzip_open(path, lines_needed):
    """What a loop that takes `lines_needed` complete lines out of gzip.open(path, 'rt') and then breaks (the shape of the
    reference's loops, :250 with :272-273) ends in: None, or the exception gzip raised on the way."""
    try:
        with gzip.open(path, "rt", encoding="latin-1") as fh:
            for k, _ in enumerate(fh):
                if lines_needed is not None and k + 1 >= lines_needed:
                    break
    except (EOFError, OSError, zlib.error) as exc:
        return exc
    return None


def test_reading_rules_campaign_against_gzip_open(tmp_path):
    """Random streams -- members at several levels, members of nothing, zero padding, \\r\\n and bare \\r lines -- with random
    damage (cuts, flipped bits, junk) and random bounds: td_gzip_check must end as Python's own gzip.open ends under a loop
    of the reference's shape.  (The 96 goldens pin the reference itself; this pins the restatement on Lib/gzip.py over a
    few hundred more files.)"""
    import os
    import random
    import time
    from tagdigger_amd import _binding as B
    L = B.load()
    rng = random.Random(int(os.environ.get("TD_FUZZ_SEED", "20261005")))
    budget = float(os.environ.get("TD_FUZZ_SECONDS", "20"))
    t0 = time.time()
    ncase = nraise = 0
    p = tmp_path / "x.fq.gz"
    while time.time() - t0 < budget:
        nl = rng.choice([b"\n", b"\n", b"\r\n", b"\r"])
        recs = []
        for i in range(rng.randrange(1, 400)):
            seq = bytes(rng.choice(b"ACGTN") for _ in range(rng.randrange(0, 120)))
            recs.append(b"@r%d" % i + nl + seq + nl + b"+" + nl + b"I" * len(seq) + nl)
        text = b"".join(recs)
        cuts = sorted(rng.randrange(0, len(text) + 1) for _ in range(rng.randrange(0, 4)))
        blob = b""
        for a, b in zip([0] + cuts, cuts + [len(text)]):
            blob += gzip.compress(text[a:b], compresslevel=rng.choice([0, 1, 6, 9]), mtime=0)
            if rng.random() < 0.2:
                blob += b"\0" * rng.randrange(1, 40)
        kind = rng.random()
        if kind < 0.25:
            blob = blob[:rng.randrange(0, len(blob) + 1)]
        elif kind < 0.55:
            q = bytearray(blob)
            for _ in range(rng.randrange(1, 3)):
                q[rng.randrange(len(q))] ^= 1 << rng.randrange(8)
            blob = bytes(q)
        elif kind < 0.65:
            blob += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 12)))
        p.write_bytes(blob)
        nreads = len(recs)
        maxreads = rng.choice([None, None, rng.randrange(1, nreads + 3), rng.randrange(1, nreads + 3)])
        need = None if maxreads is None else 4 * (maxreads - 1) + 2
        expected = _loop_over_gzip_open(str(p), need)
        rc = L.td_gzip_check(str(p).encode(), maxreads if maxreads is not None else 1 << 62)
        if expected is None:
            assert rc == 0, (ncase, maxreads, (L.td_last_error() or b"").decode())
        else:
            with pytest.raises(type(expected)) as ei:
                B.check(rc)
            assert type(ei.value) is type(expected) and str(ei.value) == str(expected), (ncase, maxreads)
            nraise += 1
        ncase += 1
    print(" [gzip reading rules campaign: %d cases, %d raise] " % (ncase, nraise), end="")
    assert ncase > 50 and nraise > 10
