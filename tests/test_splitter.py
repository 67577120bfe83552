"""Barcode splitter: the oracle's restatement and the product's host logic against
fixtures captured from the real reference (tests/golden/make_splitter_golden.py), then the GPU
path -- through the C-ABI -- against both."""
import base64
import contextlib
import gzip
import io
import os
import random

import numpy as np
import pytest

from conftest import load_golden
from oracle import tagdigger_oracle as po

G = load_golden("splitter.json")


def adapter_of(name):
    return [tuple(x) for x in G["adapters"][name]]


# ------------------------------------------------------------------ CPU: oracle vs reference
@pytest.mark.parametrize("case", G["find"], ids=lambda c: "%s-%d" % (c["adapter"], c["barcode"]))
def test_oracle_find_adapter_seq(case):
    ad = adapter_of(case["adapter"])
    log = []
    trees = po.build_adapter_tree(ad, case["barcodes"], log)
    assert "".join(x + "\n" for x in log) == case["stdout"]
    s0, s1 = ad[0][0].replace("^", ""), ad[1][0].replace("^", "")
    start = len(case["barcodes"][case["barcode"]]) + len(case["cutsite"])
    got = [po.find_adapter_seq(r, trees[case["barcode"]], s0, s1, start) for r in case["reads"]]
    assert got == case["values"]


def good_split_cases():
    return [c for c in G["split"] if "raises" not in c]


@pytest.mark.parametrize("k", range(len(good_split_cases())))
def test_oracle_splitter_outputs(k):
    c = good_split_cases()[k]
    outs, _ = po.barcode_splitter_bytes(base64.b64decode(c["fastq_b64"]), c["barcodes"], c["cutsite"], adapter_of(c["adapter"]),
                                        maxreads=c["maxreads"] if c["maxreads"] is not None else 500000000)
    assert [base64.b64encode(o).decode() for o in outs] == c["outputs_b64"]


# ------------------------------------------------------------------ CPU: product host logic
@pytest.mark.parametrize("case", G["find"], ids=lambda c: "%s-%d" % (c["adapter"], c["barcode"]))
def test_adapter_ends_match_the_trie(case):
    """The flat list the product hands to the GPU decides every read end like the oracle's trie,
    and the messages printed while resolving it are the reference's."""
    from tagdigger_amd import tagdigger_fun as tf
    ad = adapter_of(case["adapter"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ends = tf._adapter_ends(ad, case["barcodes"])
    assert buf.getvalue() == case["stdout"]
    trees = po.build_adapter_tree(ad, case["barcodes"])
    rng = random.Random(5)
    for b, bc in enumerate(case["barcodes"]):
        reads = list(case["reads"])
        for seq, _ in ends[b]:                       # every stored beginning, alone and behind random bases
            reads += [seq, "ACGT" * 3 + seq, seq[1:], seq + "A"]
        reads += ["".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 30))) for _ in range(200)]
        for r in reads:
            want = po.sequence_index_lookup(r[::-1], trees[b][0])
            want = 999 if want == -1 else trees[b][1][want]
            hits = [sl for seq, sl in ends[b] if len(seq) <= len(r) and r.endswith(seq)]
            assert len(hits) <= 1
            assert (hits[0] if hits else 999) == want, (bc, r)


def test_trie_survivors_rules():
    from tagdigger_amd import tagdigger_fun as tf
    assert tf._trie_survivors(["AC", "AC", "ACG", "T"]) == [("AC", 0), ("T", 3)]
    with pytest.raises(AssertionError, match="Problematic sequence: 1"):
        tf._trie_survivors(["ACG", "AC"])
    assert tf._trie_survivors([]) == []


# ------------------------------------------------------------------ GPU
def run_product(tmp_path, c):
    from tagdigger_amd import tagdigger_fun as tf
    old = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        name = "in.fq.gz" if c["gz"] else "in.fq"
        raw = base64.b64decode(c["fastq_b64"])
        with open(name, "wb") as fh:
            fh.write(gzip.compress(raw, mtime=0) if c["gz"] else raw)
        outs = ["out%d.fq" % i for i in range(len(c["barcodes"]))]
        kw = {"cutsite": c["cutsite"], "adapter": adapter_of(c["adapter"])}
        if c["maxreads"] is not None:
            kw["maxreads"] = c["maxreads"]
        buf = io.StringIO()
        rec = {}
        try:
            with contextlib.redirect_stdout(buf):
                tf.barcodeSplitter(name, c["barcodes"], outs, **kw)
        except Exception as e:
            rec["raises"] = type(e).__name__
            rec["message"] = str(e)
        rec["stdout"] = buf.getvalue()
        rec["outputs_b64"] = [base64.b64encode(open(o, "rb").read()).decode() if os.path.exists(o) else None for o in outs]
        return rec
    finally:
        os.chdir(old)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(G["split"])))
def test_gpu_splitter_matches_reference(k, tmp_path):
    c = G["split"][k]
    got = run_product(tmp_path, c)
    assert got.get("raises") == c.get("raises")
    if "raises" in c:
        assert got["message"] == c["message"]
    assert got["stdout"] == c["stdout"]
    assert got["outputs_b64"] == c["outputs_b64"]


def synth_reads(rng, barcodes, cutsite, ad, n, first_line_shift=0):
    """FASTQ bytes with every kind of read end, plus the oracle's decisions."""
    s0, s1 = ad[0][0].replace("^", ""), ad[1][0].replace("^", "")
    recs = []
    for i in range(n):
        bc = rng.choice(barcodes)
        body = "".join(rng.choice("ACGT") for _ in range(rng.randrange(20, 120)))
        kind = rng.randrange(8)
        if kind == 0:
            seq = "".join(rng.choice("ACGTN") for _ in range(rng.randrange(0, 80)))
        elif kind == 1:
            seq = bc + cutsite + body + s0 + body[:7]
        elif kind == 2:
            seq = bc + cutsite + body + s1 + body[:5]
        elif kind == 3:
            full = ad[0][0][:ad[0][0].find("^")] + ad[0][1]
            seq = bc + cutsite + body + full[:rng.randrange(1, len(full) + 1)]
        elif kind == 4:
            full = ad[1][0][:ad[1][0].find("^")] + ad[1][1].replace("[barcode]", po.reverse_complement(bc))
            seq = bc + cutsite + body + full[:rng.randrange(1, len(full) + 1)]
        elif kind == 5:
            seq = (bc + cutsite + body).lower()
        else:
            seq = bc + cutsite + body
        r = rng.random()
        if r < 0.01:                       # a read far longer than the window k_split2 stages behind a tile (its global-memory path)
            seq = seq[:len(seq) // 2] + "".join(rng.choice("AT") for _ in range(rng.randrange(400, 3000))) + seq[len(seq) // 2:]
        elif r < 0.025:                    # a read of a few hundred bases: the site search's second and third window of 128 positions
            k = rng.randrange(0, len(seq) + 1)
            seq = seq[:k] + "".join(rng.choice("AT") for _ in range(rng.randrange(90, 380))) + seq[k:]
        elif r < 0.045:                    # blanks str.strip() takes off, and a read of a few bases
            seq = rng.choice([" ", "\t", "  "]) + seq[:rng.randrange(0, 12)] + rng.choice(["", " ", "\t "])
        recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * len(seq)))
    return ("\n" * first_line_shift + "".join(recs)).encode("ascii")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,adapter", [(1, "PstI-MspI-Hall"), (2, "NsiI-MspI-Clark"), (3, "repeat")])
def test_gpu_split_device_decisions(seed, adapter):
    """Several tiles of reads: every decision equals the oracle's, for any first_line."""
    import tagdigger_amd
    from tagdigger_amd import tagdigger_fun as tf
    rng = random.Random(seed)
    barcodes = ["AACG", "TTGACC", "CGT", "GATTACAG", "CCA"]
    cutsite = "TGCAT" if adapter.startswith("Nsi") else "TGCAG"
    ad = adapter_of(adapter)
    data = synth_reads(rng, barcodes, cutsite, ad, 4000)
    with contextlib.redirect_stdout(io.StringIO()):
        ends = tf._adapter_ends(ad, barcodes)
    eng = tagdigger_amd.Engine(0)
    try:
        eng.set_splitter(barcodes, cutsite, ad[0][0].replace("^", ""), ad[1][0].replace("^", ""), ends)
        d = eng.dev_alloc(len(data))
        try:
            eng.h2d(d, data)
            for first_line in (0, 4, 8):
                res, terms = eng.split_device(d, len(data), first_line=first_line)
                assert terms == data.count(b"\n")
                want = []
                po.barcode_splitter_bytes(data, barcodes, cutsite, ad, decisions=want)
                got = [(int(a), int(b)) for a, b in res[:len(want)]]
                assert got == [(b, 999 if b < 0 else c) for b, c in want]
            # a buffer that starts on a sequence line (first_line 1): same reads, same decisions
            cut = data.index(b"\n") + 1
            d2 = eng.dev_alloc(len(data))
            try:
                eng.h2d(d2, data[cut:])
                res, _ = eng.split_device(d2, len(data) - cut, first_line=1)
                want = []
                po.barcode_splitter_bytes(data, barcodes, cutsite, ad, decisions=want)
                got = [(int(a), int(b)) for a, b in res[:len(want)]]
                assert got == [(b, 999 if b < 0 else c) for b, c in want]
            finally:
                eng.dev_free(d2)
        finally:
            eng.dev_free(d)
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_split_fuzz_campaign():
    """Random barcode sets, adapter sets, read mixes, line ends and first-line offsets for
    TD_FUZZ_SECONDS seconds (default 20): every decision of k_split equals the oracle's."""
    import time
    import tagdigger_amd
    from tagdigger_amd import tagdigger_fun as tf
    budget = float(os.environ.get("TD_FUZZ_SECONDS", "20"))
    seed0 = int(os.environ.get("TD_FUZZ_SEED", "4242"))
    t_end = time.time() + budget
    next_note = time.time() + 30
    names = sorted(G["adapters"].keys())
    eng = tagdigger_amd.Engine(0)
    ncase = 0
    try:
        while time.time() < t_end:
            rng = random.Random(seed0 + ncase)
            name = rng.choice(names)
            ad = adapter_of(name)
            cutsite = "TGCAT" if name.startswith("Nsi") else "TGCAG"
            barcodes = []
            while len(barcodes) < rng.randint(1, 12):
                b = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 9)))
                if not any((b + cutsite).startswith(o + cutsite) or (o + cutsite).startswith(b + cutsite) for o in barcodes):
                    barcodes.append(b)
            data = synth_reads(rng, barcodes, cutsite, ad, rng.randint(1, 4000))
            nl = rng.choice([b"\n", b"\n", b"\r\n", b"\r"])
            data = data.replace(b"\n", nl)
            with contextlib.redirect_stdout(io.StringIO()):
                ends = tf._adapter_ends(ad, barcodes)
            eng.set_splitter(barcodes, cutsite, ad[0][0].replace("^", ""), ad[1][0].replace("^", ""), ends)
            eng.set_option("split_kernel", 2 if ncase % 3 else 1)          # k_split2 (tile in LDS) twice, k_split once
            d = eng.dev_alloc(len(data))
            try:
                eng.h2d(d, data)
                res, _ = eng.split_device(d, len(data), first_line=4 * rng.randint(0, 3))
                want = []
                po.barcode_splitter_bytes(data, barcodes, cutsite, ad, decisions=want)
                got = [(int(a), int(b)) for a, b in res[:len(want)]]
                assert got == [(b, 999 if b < 0 else c) for b, c in want], ("seed", seed0 + ncase, name, nl, "split_kernel", 2 if ncase % 3 else 1)
            finally:
                eng.dev_free(d)
            ncase += 1
            if time.time() >= next_note:                      # (a long soak must not look hung)
                print(" [%d cases so far] " % ncase, end="", flush=True)
                next_note = time.time() + 30
    finally:
        eng.close()
    print(" [splitter fuzz campaign: %d cases] " % ncase, end="")
    assert ncase > 0


@pytest.mark.gpu
def test_gpu_splitter_on_damaged_gzip_ends_as_gzip_open_does(tmp_path):
    """barcodeSplitter reads .gz inputs through gzip.open as find_tags_fastq does (reference :1318-1319) and leaves its
    loop behind the quality line of read number maxreads (:1361-1362): a damaged input ends in gzip.open's exception
    -- unless the loop is through first, and then the files are the ones the good input gives."""
    import gzip
    import zlib
    from tagdigger_amd import tagdigger_fun as tf
    rng = random.Random(3)
    barcodes = ["AACG", "TTGACC", "CGT", "GATTACAG"]
    ad = adapter_of("PstI-MspI-Hall")
    data = synth_reads(rng, barcodes, "TGCAG", ad, 4000)
    good = gzip.compress(data, compresslevel=1, mtime=0)
    outs = [str(tmp_path / ("o%d.fq" % i)) for i in range(len(barcodes))]

    def ends(blob, lines_needed):
        """What a loop that takes `lines_needed` complete lines (None: all) out of gzip.open(..., 'rt') meets."""
        p = tmp_path / "probe.gz"
        p.write_bytes(blob)
        try:
            with gzip.open(str(p), "rt", newline=None) as fh:
                for k, _ in enumerate(fh):
                    if lines_needed is not None and k + 1 >= lines_needed:
                        break
        except (EOFError, OSError, zlib.error) as exc:
            return exc
        return None
    variants = {"truncated": good[:len(good) // 2], "crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], "junk": good + b"junk",
                "padding": good + b"\0" * 64, "flip": good[:3000] + bytes([good[3000] ^ 0x20]) + good[3001:]}
    for name, blob in variants.items():
        for maxreads in (500000000, 100):
            src = tmp_path / ("%s.fq.gz" % name)
            src.write_bytes(blob)
            expected = ends(blob, None if maxreads > 10 ** 6 else 4 * maxreads)
            with contextlib.redirect_stdout(io.StringIO()):
                if expected is None:
                    tf.barcodeSplitter(str(src), barcodes, outs, cutsite="TGCAG", adapter=ad, maxreads=maxreads)
                else:
                    with pytest.raises(type(expected)) as ei:
                        tf.barcodeSplitter(str(src), barcodes, outs, cutsite="TGCAG", adapter=ad, maxreads=maxreads)
                    assert type(ei.value) is type(expected) and str(ei.value) == str(expected), (name, maxreads)
                    continue
            # (what the loop saw: a flipped bit may decode -- to other text -- long before the member's CRC-32 says so)
            seen = b""
            with gzip.open(str(src), "rb") as fh:
                try:
                    while seen.count(b"\n") < 4 * maxreads:
                        piece = fh.read1(8192)
                        if not piece:
                            break
                        seen += piece
                except (EOFError, OSError, zlib.error):
                    pass
            want, _ = po.barcode_splitter_bytes(seen, barcodes, "TGCAG", ad, maxreads=maxreads)
            for o, w in zip(outs, want):
                assert open(o, "rb").read() == w, (name, maxreads)


@pytest.mark.gpu
@pytest.mark.parametrize("newline,threads", [(b"\r\n", None), (b"\n", None), (b"\n", "3"), (b"\n", "1")])
def test_gpu_split_file_large(tmp_path, monkeypatch, newline, threads):
    """More than one 32 MiB piece through td_split_file (records straddle the pieces): files equal the
    oracle's.  CRLF files take the writers' byte-by-byte walk, LF files the shared line index; any number
    of writer threads gives the same files."""
    if threads:
        monkeypatch.setenv("TAGDIG_SPLIT_THREADS", threads)
    import tagdigger_amd
    from tagdigger_amd import tagdigger_fun as tf
    rng = random.Random(11)
    barcodes = ["AACG", "TTGACC", "CGT", "GATTACAG"]
    ad = adapter_of("PstI-MspI-Hall")
    block = synth_reads(rng, barcodes, "TGCAG", ad, 3001).replace(b"\n", newline)
    data = block * (((70 << 20) // len(block)) + 1)
    src = tmp_path / "big.fq"
    src.write_bytes(data)
    outs = [str(tmp_path / ("o%d.fq" % i)) for i in range(len(barcodes))]
    with contextlib.redirect_stdout(io.StringIO()):
        tf.barcodeSplitter(str(src), barcodes, outs, cutsite="TGCAG", adapter=ad)
    want, stats = po.barcode_splitter_bytes(block, barcodes, "TGCAG", ad)
    reps = len(data) // len(block)
    for o, w in zip(outs, want):
        got = open(o, "rb").read()
        assert len(got) == len(w) * reps
        assert got[:len(w)] == w and got[-len(w):] == w
        assert got == w * reps


@pytest.mark.gpu
def test_gpu_splitter_cli_matches_reference(tmp_path):
    """barcode_splitter_script.py of the reference, same key file and FASTQ: same files, same stdout."""
    from tagdigger_amd import barcode_splitter_script
    c = G["cli"]
    (tmp_path / "lane1.fq").write_bytes(base64.b64decode(c["fastq_b64"]))
    (tmp_path / "key.csv").write_text(c["key_csv"])
    old = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            barcode_splitter_script.main(c["argv"])
        assert buf.getvalue() == c["stdout"]
        for name, want in c["outputs_b64"].items():
            assert base64.b64encode(open(name, "rb").read()).decode() == want
    finally:
        os.chdir(old)


def test_splitter_cpu_restatement_rate(capsys):
    """Not a check: prints the pure-Python restatement's rate on 20 k canonical reads (what DESIGN.md
    quotes beside the GPU path's numbers; run with -s to see it)."""
    import time
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes
    cfg = SynthConfig(nreads=20_000, nbar=96, nmarkers=500, seed=3)
    data = bytes(synth_host_bytes(cfg, 0, cfg.nreads))
    t0 = time.perf_counter()
    outs, stats = po.barcode_splitter_bytes(data, cfg.barcodes, cfg.cutsite, adapter_of("PstI-MspI-Hall"))
    dt = time.perf_counter() - t0
    assert stats[0] == cfg.nreads and sum(map(len, outs)) > 0
    with capsys.disabled():
        print(" [python restatement of barcodeSplitter: %.1f k reads/s] " % (cfg.nreads / dt / 1e3), end="")


@pytest.mark.gpu
def test_gpu_split_entries_that_do_not_fit_the_compact_form():
    """An adapter of more than 128 characters (its beginnings do not fit k_split2's compact entries, whose characters
    live in master strings of 128 bytes): the splitter must fall back to k_split whatever split_kernel says, with the
    oracle's decisions; and a long-period repeat that fills a three-character group beyond eight entries."""
    import tagdigger_amd
    from tagdigger_amd import tagdigger_fun as tf
    rng = random.Random(99)
    long_tail = "".join(rng.choice("ACGT") for _ in range(150))
    for ad in ([("CCG^G", long_tail), ("CTGCA^G", "[barcode]AGATCGGAAGAGC")],
               [("CCG^G", "ACG" * 30), ("CTGCA^G", "[barcode]" + "ACG" * 25)]):
        barcodes = ["ACGT", "GGTCA", "TTAGC"]
        data = synth_reads(rng, barcodes, "TGCAG", ad, 1500)
        with contextlib.redirect_stdout(io.StringIO()):
            ends = tf._adapter_ends(ad, barcodes)
        eng = tagdigger_amd.Engine(0)
        try:
            eng.set_splitter(barcodes, "TGCAG", ad[0][0].replace("^", ""), ad[1][0].replace("^", ""), ends)
            for kern in (2, 1):
                eng.set_option("split_kernel", kern)
                d = eng.dev_alloc(len(data))
                try:
                    eng.h2d(d, data)
                    res, _ = eng.split_device(d, len(data))
                    want = []
                    po.barcode_splitter_bytes(data, barcodes, "TGCAG", ad, decisions=want)
                    got = [(int(a), int(b)) for a, b in res[:len(want)]]
                    assert got == [(b, 999 if b < 0 else c) for b, c in want], kern
                finally:
                    eng.dev_free(d)
        finally:
            eng.close()
