"""Ordinary gzip through td_count_file with the markers resolved on the GPU (count_gzip_dev in csrc/tagdig.hip: the host
threads decode DEFLATE into 16-bit symbols, k_gz_resolve turns markers into bytes, k_gz_crc takes the CRC-32 -- what
`gzip.open(fqfile, 'rt')` of reference tagdigger_fun.py:240-241 reads).  Counts must equal the oracle's on the plain bytes
for every kind of stream, with the resolution on the GPU and on the host; damaged streams must raise, never count."""
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


def _raw_deflate(data, level, flush_every=0):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for i in range(0, len(data), flush_every):
        out.append(co.compress(data[i:i + flush_every]))
        out.append(co.flush(zlib.Z_SYNC_FLUSH))                 # (an empty stored block, the window kept: what pigz writes)
    return b"".join(out) + co.flush()


def _member(data, body):
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


STREAMS = {
    "level6": lambda d: gzip.compress(d, compresslevel=6),
    "level1": lambda d: gzip.compress(d, compresslevel=1),
    "level9": lambda d: gzip.compress(d, compresslevel=9),
    "sync-flushes": lambda d: _member(d, _raw_deflate(d, 6, flush_every=100_000)),
    "full-flushes": lambda d: gzip_one_member(d, level=1, threads=4, piece=1 << 18),
    "two-members": lambda d: gzip.compress(d[:len(d) // 3], compresslevel=6) + gzip.compress(d[len(d) // 3:], compresslevel=1),
    "stored-in-the-middle": lambda d: gzip.compress(d[:len(d) // 2], compresslevel=6) + gzip.compress(d[len(d) // 2:len(d) // 2 + 700_000], compresslevel=0)
                                      + gzip.compress(d[len(d) // 2 + 700_000:], compresslevel=6),
    "header-fields": lambda d: b"\x1f\x8b\x08\x1c" + b"\0" * 6 + struct.pack("<H", 5) + b"extra" + b"name.fq\0" + b"a comment\0"
                               + _raw_deflate(d, 6) + struct.pack("<II", zlib.crc32(d), len(d) & 0xFFFFFFFF),
}


def _check(eng, want, ost, what):
    got = eng.counts_numpy()
    st = eng.stats()
    assert (got == want).all(), what
    assert (st["reads"], st["barcut"], st["tag"]) == (ost["reads"], ost["barcut"], ost["tag"]), what


@pytest.mark.parametrize("chunk", ["65536", "1048576"])
@pytest.mark.parametrize("where", ["gpu", "host"])
@pytest.mark.parametrize("kind", sorted(STREAMS))
def test_gzip_file_counts_like_plain(eng, sample, tmp_path, monkeypatch, kind, where, chunk):
    cfg, raw, want, ost = sample
    blob = STREAMS[kind](raw)
    assert gzip.decompress(blob) == raw
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(blob)
    monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")               # (the chunk-parallel decoder also below 8 MiB)
    monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", chunk)
    monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "6")
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_option("gpu_resolve", 1 if where == "gpu" else 0)
    try:
        eng.reset()
        eng.count_file(path)
        _check(eng, want, ost, (kind, where, chunk))
    finally:
        eng.set_option("gpu_resolve", 1)


def test_gpu_resolve_is_what_runs(eng, sample, tmp_path, monkeypatch, capfd):
    """With the device decoder off, the path of a gzip file of this size is count_gzip_dev (its statistics line says so), not
    the host reader."""
    cfg, raw, want, ost = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw * 3, compresslevel=1))       # (> 8 MiB compressed: no TAGDIG_PAR_INFLATE needed)
    assert os.path.getsize(path) > 8 << 20
    monkeypatch.delenv("TAGDIG_PAR_INFLATE", raising=False)
    monkeypatch.setenv("TAGDIG_INFLATE_STATS", "1")
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_option("gpu_huffman", 0)                            # (the device decoder, tests/test_gzip_gpu_huffman.py, would take the file)
    try:
        eng.reset()
        eng.count_file(path)
    finally:
        eng.set_option("gpu_huffman", 1)
    _check(eng, want * 3, {k: 3 * v for k, v in ost.items()}, "three times the sample")
    err = capfd.readouterr().err
    assert "count_gzip_dev:" in err and "par_inflate:" in err, err


def test_maxreads_inside_a_gzip_file(eng, sample, tmp_path, monkeypatch):
    cfg, raw, _, _ = sample
    path = str(tmp_path / "lib.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=1))
    monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")
    monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", "65536")
    monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "6")
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    for maxreads in (1, 4321, NREADS - 1, NREADS, NREADS + 5):
        ost = {}
        want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(raw, maxreads=maxreads, stats=ost)
        eng.reset()
        eng.count_file(path, maxreads=maxreads)
        _check(eng, want, ost, maxreads)
        assert ost["reads"] == min(maxreads, NREADS)


def test_damaged_gzip_raises_and_the_engine_goes_on(eng, sample, tmp_path, monkeypatch):
    """Flipped bits, truncation, a wrong CRC-32, a wrong length, bytes behind the last member: td_count_file ends in the
    exception gzip.open ends in (reference :240-243; tests/test_gzip_damage.py pins the rules on what the real reference
    did) -- class and message -- and the next good file counts."""
    cfg, raw, want, ost = sample
    good = gzip.compress(raw, compresslevel=6)
    monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")
    monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", "65536")
    monkeypatch.setenv("TAGDIG_INFLATE_THREADS", "6")
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    path = str(tmp_path / "lib.fq.gz")
    rng = random.Random(11)
    cases = []
    for _ in range(6):
        b = bytearray(good)
        b[rng.randrange(len(b) // 10, len(b) - 100)] ^= 1 << rng.randrange(8)
        cases.append(("flip", bytes(b)))
    cases.append(("truncated", good[:len(good) * 2 // 3]))
    cases.append(("no trailer", good[:-8]))
    b = bytearray(good); b[-8] ^= 0x40
    cases.append(("wrong crc", bytes(b)))
    b = bytearray(good); b[-4] ^= 0x01
    cases.append(("wrong length", bytes(b)))
    cases.append(("junk behind the member", good + b"\x00\x01\x02 not gzip"))
    for what, blob in cases:
        with open(path, "wb") as fh:
            fh.write(blob)
        try:
            with gzip.open(path, "rb") as fh:
                while fh.read1(8192):
                    pass
            raise AssertionError("gzip.open reads this file: " + what)
        except (EOFError, OSError, zlib.error) as exc:
            expected = exc
        eng.reset()
        with pytest.raises(type(expected)) as ei:
            eng.count_file(path)
        # (a flip may land in a place that changes nothing that is checked... it cannot: the CRC-32 covers every byte)
        assert type(ei.value) is type(expected) and str(ei.value) == str(expected), what
    # zero padding behind the last member is skipped, as gzip.open skips it
    for blob in (good, good + b"\x00" * 300):
        with open(path, "wb") as fh:
            fh.write(blob)
        eng.reset()
        eng.count_file(path)
        _check(eng, want, ost, "after the damaged ones")


def test_gzip_fuzz_campaign(eng, tmp_path, monkeypatch):
    """TD_FUZZ_SECONDS (default 20) of random cases through the GPU-resolved gzip path against the C oracle on the plain bytes:
    irregular FASTQ (every terminator style, blank and long lines, phase shifts) so that batches end inside lines, inside
    \\r\\n pairs and inside runs of blank lines; members of independently compressed pieces (levels, strategies, stored blocks,
    sync and full flushes), one to three members; chunk sizes from 1 KiB, two to eight decoder threads."""
    import time
    from helpers import dirty_fastq, small_index
    budget = float(os.environ.get("TD_FUZZ_SECONDS", "20"))
    seed0 = int(os.environ.get("TD_FUZZ_SEED", "424242"))
    monkeypatch.setenv("TAGDIG_PAR_INFLATE", "1")
    path = str(tmp_path / "f.fq.gz")
    t_end, next_note, ncase = time.time() + budget, time.time() + 30, 0

    def member(rnd, data):
        pieces, pos = [], 0
        while True:
            n = min(len(data) - pos, rnd.choice([1, 300, 7000, 90000, 600000]))
            last = pos + n >= len(data)
            co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, rnd.choice([1, 8, 9]),
                                  rnd.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
            part = data[pos:pos + n]
            body = co.compress(part[:n // 2]) + (co.flush(zlib.Z_SYNC_FLUSH) if rnd.random() < 0.3 else b"") + co.compress(part[n // 2:])
            pieces.append(body + co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH))
            pos += n
            if last:
                return _member(data, b"".join(pieces))

    while time.time() < t_end:
        rnd = random.Random(seed0 + ncase)
        cutsite = rnd.choice(["TGCAG", "CWGC", ""])
        nl = rnd.choice([("\n",), ("\r\n",), ("\r",), ("\n", "\r\n", "\r")])
        barcodes, tags, cutsites = small_index(rnd, cutsite, nbar=rnd.randint(1, 12), ntag=rnd.randint(1, 60))
        datas = [dirty_fastq(rnd, barcodes, tags, cutsites, nrec=rnd.choice([1, 50, 3000, 20000]), nl_choices=nl,
                             long_lines=rnd.random() < 0.3, permanent_shifts=rnd.random() < 0.3) for _ in range(rnd.choice([1, 1, 2, 3]))]
        raw = b"".join(datas)
        with open(path, "wb") as fh:
            fh.write(b"".join(member(rnd, d) for d in datas))
        ost = {}
        want = c_oracle.COracle(barcodes, tags, cutsite).count_bytes(raw, stats=ost)
        monkeypatch.setenv("TAGDIG_INFLATE_CHUNK", str(rnd.choice([1024, 3000, 17000, 65536, 1 << 20])))
        monkeypatch.setenv("TAGDIG_INFLATE_THREADS", str(rnd.choice([2, 3, 8])))
        monkeypatch.setenv("TAGDIG_INFLATE_OVERSUB", str(rnd.choice([1, 2, 4])))
        eng.set_index(barcodes, tags, cutsite)
        eng.reset()
        eng.count_file(path)
        _check(eng, want, ost, ("seed", seed0 + ncase))
        ncase += 1
        if time.time() >= next_note:
            print(" [%d cases so far] " % ncase, end="", flush=True)
            next_note = time.time() + 30
    print(" [gzip fuzz campaign: %d cases] " % ncase, end="")
    assert ncase > 0
