#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (build container only).

Usage (from anywhere):
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [/root/reference]

The reference is imported read-only from its checkout, run on small inputs in
a scratch directory, and only DATA (inputs + observed outputs) is written next
to this script.  The reference's source never enters this repository and never
travels to the GPU box; the fixtures do.

Fixture files
  hotpath_primitives.json   enumerate_cut_sites / build_sequence_tree+lookup tables
  hotpath_cases.json        find_tags_fastq: hand-written edge cases (SURVEY A.7 and more)
  hotpath_random.json       find_tags_fastq: seeded random small cases
  hotpath_errors.json       index-build assertion / shadowing behaviour
"""
import base64
import contextlib
import gzip
import io
import json
import os
import random
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import tagdigger_fun as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def b64(b):
    return base64.b64encode(b).decode("ascii")


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=0, separators=(",", ":"))
        fh.write("\n")
    print("wrote", name, os.path.getsize(path), "bytes")


def run_ref(filename, payload, barcodes, tags, **kw):
    """Write payload under `filename` in a scratch dir and call the reference."""
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, filename)
        with open(path, "wb") as fh:
            fh.write(payload)
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out):
                res = ref.find_tags_fastq(path, list(barcodes), list(tags), **kw)
            return {"counts": res}
        except Exception as e:  # recorded, not hidden
            return {"raises": type(e).__name__, "message": str(e)}


def case(name, filename, payload, barcodes, tags, **kw):
    rec = {"name": name, "filename": filename, "fastq_b64": b64(payload),
           "barcodes": list(barcodes), "tags": list(tags), "kwargs": kw}
    rec.update(run_ref(filename, payload, barcodes, tags, **kw))
    return rec


# ---------------------------------------------------------------- primitives
def primitives():
    out = {"enumerate_cut_sites": {}, "lookup": []}
    for cs in ["TGCAG", "", "R", "Y", "K", "M", "S", "W", "B", "D", "H", "V", "N",
               "RY", "YR", "BN", "CWGC", "NN", "GWRC", "RCATGY", "VH", "TGCAT",
               "CCNGG", "RRY", "NWS"]:
        out["enumerate_cut_sites"][cs] = ref.enumerate_cut_sites(cs)

    def table(seqs, numseq, queries):
        tree = ref.build_sequence_tree(list(seqs), numseq)
        res = []
        for q in queries:
            try:
                res.append(ref.sequence_index_lookup(q, tree))
            except Exception as e:
                res.append({"raises": type(e).__name__})
        return {"sequences": list(seqs), "numseq": numseq, "queries": list(queries),
                "result": res}

    q1 = ["ACG", "ACGTTT", "AC", "ACN", "G", "GAAA", "T", "", "N", "acg", "ACT", "ACTA",
          "ACA", "GN", "CG"]
    out["lookup"].append(table(["ACG", "ACT", "G"], 3, q1))
    # multi-cut-site wrap (index mod numseq)
    bc = ref.combine_barcode_and_cutsite(["AA", "CCC"], "CAGC") + \
        ref.combine_barcode_and_cutsite(["AA", "CCC"], "CTGC")
    out["lookup"].append(table(bc, 2, ["AACAGCTTT", "AACTGCTTT", "CCCCTGCA", "CCCCAGC",
                                        "CCCCGGC", "AACAG", "AAC", "CCCCTGN"]))
    # special lone-empty tree
    out["lookup"].append(table([""], 1, ["A", "C", "G", "T", "N", "", "AN", "a", " A"]))
    # duplicates / extension shadowing (first wins, silently)
    out["lookup"].append(table(["AC", "AC"], 2, ["AC", "ACG", "A"]))
    out["lookup"].append(table(["AC", "ACG", "T"], 3, ["AC", "ACG", "ACGT", "T", "TT"]))
    out["lookup"].append(table(["AC", "ACG", "ACGT", "ACT"], 4, ["ACGT", "ACT", "AC"]))
    # root is a leaf: first sequence empty but not the special case
    out["lookup"].append(table(["", "A"], 2, ["A", "C", "G", "T", "", "N"]))
    out["lookup"].append(table(["", ""], 2, ["A", "C", "", "N"]))
    # long
    rnd = random.Random(7)
    seqs = []
    while len(seqs) < 40:
        s = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(3, 12)))
        if not any(s.startswith(t) or t.startswith(s) for t in seqs):
            seqs.append(s)
    qs = [s + "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 4))) for s in seqs]
    qs += [s[:-1] for s in seqs] + ["".join(rnd.choice("ACGT") for _ in range(10)) for _ in range(30)]
    out["lookup"].append(table(seqs, len(seqs), qs))
    return out


# ---------------------------------------------------------------- hand cases
def rec(seq, hdr="@r0", qual=None, nl="\n", final_nl=True):
    qual = "I" * len(seq) if qual is None else qual
    s = hdr + nl + seq + nl + "+" + nl + qual + (nl if final_nl else "")
    return s.encode("latin-1")


def hand_cases():
    B = ["AACG", "TTGACC"]
    T = ["TGCAGAAAC", "TGCAGGGGT"]
    plain = "AACGTGCAGAAACTTTT"
    c = []
    c.append(case("plain", "x.fq", rec(plain), B, T))
    c.append(case("lower", "x.fq", rec(plain.lower()), B, T))
    c.append(case("mixed case", "x.fq", rec("aAcGtGcAgAaAcTtTt"), B, T))
    c.append(case("crlf", "x.fq", rec(plain, nl="\r\n"), B, T))
    c.append(case("bare cr", "x.fq", rec(plain, nl="\r"), B, T))
    c.append(case("mixed terminators", "x.fq",
                  b"@h\r" + plain.encode() + b"\n+\r\nIIII\n@h2\r\nTTGACCTGCAGGGGTAA\r+\nII", B, T))
    c.append(case("cr then lf lines", "x.fq", b"@h\r\r" + plain.encode() + b"\n+\nI\n", B, T))
    c.append(case("cr cr lf", "x.fq", b"@h\n" + plain.encode() + b"\r\r\n+\nI\n" + rec(plain), B, T))
    c.append(case("N after tag", "x.fq", rec("AACGTGCAGAAACNNNN"), B, T))
    c.append(case("N in tag", "x.fq", rec("AACGTGCAGAANCTTTT"), B, T))
    c.append(case("N in barcode", "x.fq", rec("ANCGTGCAGAAACTTTT"), B, T))
    c.append(case("N in cutsite", "x.fq", rec("AACGTGNAGAAACTTTT"), B, T))
    c.append(case("short read", "x.fq", rec("AACGTGCAGAAA"), B, T))
    c.append(case("exact len", "x.fq", rec("AACGTGCAGAAAC"), B, T))
    c.append(case("barcode only", "x.fq", rec("AACGTGCAG"), B, T))
    c.append(case("barcode partial", "x.fq", rec("AACGTG"), B, T))
    c.append(case("empty seq line", "x.fq", rec(""), B, T))
    c.append(case("lead/trail space", "x.fq", rec("  AACGTGCAGAAAC "), B, T))
    c.append(case("lead tab vt ff", "x.fq", rec("\t\x0b\x0c AACGTGCAGAAAC\t"), B, T))
    c.append(case("lead fs gs rs us", "x.fq", rec("\x1c\x1d\x1e\x1fAACGTGCAGAAAC\x1f"), B, T))
    c.append(case("inner space", "x.fq", rec("AACG TGCAGAAAC"), B, T))
    c.append(case("inner space after tag", "x.fq", rec("AACGTGCAGAAAC TTT"), B, T))
    c.append(case("nul byte lead", "x.fq", rec("\x00AACGTGCAGAAAC"), B, T))
    c.append(case("punct in tag", "x.fq", rec("AACGTGCAGA.ACTTTT"), B, T))
    c.append(case("bracket chars", "x.fq", rec("AACGTGCAG`AACTTTT") + rec("[ACGTGCAGAAAC") + rec("AACGTGCAGAA{C"), B, T))
    c.append(case("no final newline", "x.fq", rec(plain, final_nl=False), B, T))
    c.append(case("trunc 2nd rec nl", "x.fq", rec(plain) + b"@h2\nTTGACCTGCAGGGGTAA\n", B, T))
    c.append(case("trunc 2nd rec no nl", "x.fq", rec(plain) + b"@h2\nTTGACCTGCAGGGGTAA", B, T))
    c.append(case("leading blank line", "x.fq", b"\n" + rec(plain), B, T))
    c.append(case("four blank lead", "x.fq", b"\n\n\n\n" + rec(plain), B, T))
    c.append(case("only newlines", "x.fq", b"\n" * 37, B, T))
    c.append(case("empty file", "x.fq", b"", B, T))
    c.append(case("single line no nl", "x.fq", b"@hdr", B, T))
    c.append(case("two lines no nl", "x.fq", b"@hdr\n" + plain.encode(), B, T))
    c.append(case("seq in wrong line", "x.fq", ("@h\n+\n" + plain + "\nIII\n").encode(), B, T))
    c.append(case("2 member gzip", "x.fq.gz",
                  gzip.compress(rec(plain)) + gzip.compress(rec("TTGACCTGCAGGGGTAA", "@r1")), B, T))
    c.append(case("GZ suffix", "x.fq.GZ", gzip.compress(rec(plain)), B, T))
    c.append(case("gz split mid record", "x.fastq.gz",
                  gzip.compress(rec(plain)[:9]) + gzip.compress(rec(plain)[9:] + rec(plain)), B, T))
    c.append(case("tags without site", "x.fq", rec(plain), B, ["AAAC", "GGGT"]))
    c.append(case("mixed tags", "x.fq", rec("AACGTGCAGTGCAGAAAC"), B, ["TGCAGAAAC", "GGGT"]))
    two = rec("AACGCAGCAAAC") + rec("AACGCTGCAAAC", "@r1")
    c.append(case("multi-cut tags carry site", "x.fq", two, B, ["CAGCAAAC", "CTGCAAAC"], cutsite="CWGC"))
    c.append(case("multi-cut tags w/o site", "x.fq", two, B, ["AAAC", "GGGG"], cutsite="CWGC"))
    c.append(case("multi-cut one tag w site", "x.fq", two, B, ["CAGCAAAC", "GGGG"], cutsite="CWGC"))
    c.append(case("lowercase cutsite", "x.fq", two, B, ["cagcaaac", "ctgcaaac"], cutsite="cwgc"))
    c.append(case("tassel_tagcount", "x.fq", rec(plain, "@x count=17"), B, T, tassel_tagcount=True))
    c.append(case("tassel two recs", "x.fq", rec(plain, "@x count=17") + rec("TTGACCTGCAGGGGTAA", "@length=64count=4000000000 "),
                  B, T, tassel_tagcount=True))
    c.append(case("tassel bad header", "x.fq", rec(plain, "@x nocount"), B, T, tassel_tagcount=True))
    c.append(case("maxreads=3", "x.fq", rec(plain) * 5, B, T, maxreads=3))
    c.append(case("maxreads=2.5", "x.fq", rec(plain) * 5, B, T, maxreads=2.5))
    c.append(case("maxreads=0", "x.fq", rec(plain) * 5, B, T, maxreads=0))
    c.append(case("maxreads=-4", "x.fq", rec(plain) * 5, B, T, maxreads=-4))
    c.append(case("maxreads=1", "x.fq", rec("GGGG") + rec(plain) * 3, B, T, maxreads=1))
    c.append(case("empty barcode", "x.fq", rec("TGCAGAAACTT"), [""], T))
    c.append(case("empty barcode + empty site", "x.fq", rec("AAACTT") + rec("NAAAC", "@r1") + rec("", "@r2"),
                  [""], ["AAAC", "GGGT"], cutsite=""))
    c.append(case("empty site", "x.fq", rec("AACGAAACTT") + rec("TTGACCGGGTC", "@r1"), B, ["AAAC", "GGGT"], cutsite=""))
    c.append(case("lower-case keys", "x.fq", rec(plain), ["aacg", "ttgacc"], ["tgcagaaac", "tgcagggGT"]))
    c.append(case("tag equals cutsite first", "x.fq", rec(plain) + rec("AACGTGCAGC"), B, ["TGCAG", "TGCAGGGGT"]))
    c.append(case("tag equals cutsite only", "x.fq", rec(plain) + rec("AACGTGCAG") + rec("AACGTGCAGN"), B, ["TGCAG"]))
    c.append(case("dup tags", "x.fq", rec(plain) * 2, B, ["TGCAGAAAC", "TGCAGAAAC", "TGCAGGGGT"]))
    c.append(case("tag extends earlier tag", "x.fq", rec(plain) + rec("AACGTGCAGAAACG"), B, ["TGCAGAAAC", "TGCAGAAACG"]))
    c.append(case("tag prefix of earlier tag", "x.fq", rec(plain), B, ["TGCAGAAACG", "TGCAGAAAC"]))
    c.append(case("dup barcodes", "x.fq", rec(plain), ["AACG", "AACG"], T))
    c.append(case("barcode extends earlier", "x.fq", rec("AACGTTGCAGAAAC") + rec(plain), ["AACG", "AACGT"], T))
    c.append(case("barcode+site prefix clash", "x.fq", rec("AACGTGCAGTGCAGAAAC") + rec(plain), ["AACGTGCAG", "AACG"], T))
    c.append(case("non-ACGT barcode", "x.fq", rec(plain), ["AANG"], T))
    c.append(case("non-ACGT tag", "x.fq", rec(plain), B, ["TGCAGNAAC"]))
    c.append(case("invalid cutsite", "x.fq", rec(plain), B, T, cutsite="TGXAG"))
    c.append(case("no tags", "x.fq", rec(plain), B, []))
    c.append(case("no barcodes", "x.fq", rec(plain), [], T))
    c.append(case("missing file", "x.fq", b"", B, T))
    c[-1]["filename_override"] = "does_not_exist.fq"
    # long-ish tags / reads crossing 32- and 64-base packing boundaries
    rnd = random.Random(11)
    body = "".join(rnd.choice("ACGT") for _ in range(150))
    longtags = ["TGCAG" + body[:n] for n in (26, 27, 28, 59, 60, 91, 92, 123)]
    longtags = [t[:-1] + ("A" if t[-1] != "A" else "C") if i < 7 else t for i, t in enumerate(longtags)]
    reads = b"".join(rec("AACG" + t + "ACGT", "@l%d" % i) for i, t in enumerate(longtags))
    reads += rec("AACGTGCAG" + body[:122])  # one base short of the longest tag
    c.append(case("long tags", "x.fq", reads, B, longtags))
    c.append(case("long barcodes", "x.fq", rec("ACGTACGTACGTACTGCAGAAACT") + rec("ACGTACGTACGTACGTGCAGGGGT"),
                  ["ACGTACGTACGTAC", "ACGTACGTACGTACG"], T))
    return c


def missing_file_fix(cases):
    for k in cases:
        if k.get("filename_override"):
            try:
                ref.find_tags_fastq("/nonexistent_dir_zz/" + k["filename_override"], k["barcodes"], k["tags"])
            except Exception as e:
                k.pop("counts", None)
                k["raises"] = type(e).__name__
                k["message"] = ""


# ---------------------------------------------------------------- random cases
def random_cases(n=60, seed=20261003):
    rnd = random.Random(seed)
    out = []
    sites = ["TGCAG", "TGCAG", "CWGC", "TGCAT", "", "CATGG", "RCATGY", "GWC"]
    for ci in range(n):
        cutsite = rnd.choice(sites)
        cutsites = ref.enumerate_cut_sites(cutsite)
        nbar = rnd.randint(1, 12)
        barcodes = []
        guard = 0
        while len(barcodes) < nbar and guard < 1000:
            guard += 1
            b = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0 if nbar == 1 else 3, 9)))
            if b == "" and cutsite == "" and nbar > 1:
                continue
            ok = True
            for o in barcodes:
                for c1 in cutsites:
                    for c2 in cutsites:
                        if (b + c1).startswith(o + c2) or (o + c2).startswith(b + c1):
                            ok = False
            if ok:
                barcodes.append(b)
        mode = rnd.choice(["with_site", "without_site", "mixed"])
        ntag = rnd.randint(1, 30)
        tags = []
        guard = 0
        while len(tags) < ntag and guard < 2000:
            guard += 1
            L = rnd.choice([rnd.randint(1, 12), rnd.randint(20, 40), rnd.randint(60, 70)])
            body = "".join(rnd.choice("ACGT") for _ in range(L))
            if mode == "with_site" or (mode == "mixed" and rnd.random() < 0.5):
                t = rnd.choice(cutsites) + body
            else:
                t = body
            if any(t.startswith(o) or o.startswith(t) for o in tags):
                continue
            tags.append(t)
        # stripped tags must stay prefix-free too (single-site strip branch)
        cl = len(cutsite)
        if set(t[:cl] for t in tags) <= set(cutsites) and len(cutsites) == 1:
            st = [t[cl:] for t in tags]
            if any(a != b and (a.startswith(b)) for a in st for b in st) or "" in st:
                tags = [t for t in tags if len(t) > cl]
                st = [t[cl:] for t in tags]
                keep = []
                for t, s in zip(tags, st):
                    if not any(s != s2 and (s.startswith(s2) or s2.startswith(s)) for s2 in [x[cl:] for x in keep]):
                        keep.append(t)
                tags = keep
        if not tags:
            tags = ["ACGTACGTAC"]
        nl = rnd.choice(["\n", "\n", "\r\n", "\r"])
        nrec = rnd.randint(1, 60)
        chunks = []
        for ri in range(nrec):
            u = rnd.random()
            b = rnd.choice(barcodes)
            cs = rnd.choice(cutsites)
            t = rnd.choice(tags)
            if u < 0.55:
                if t[:len(cs)] in cutsites and len(cs) > 0:
                    seq = b + t
                else:
                    seq = b + cs + t
                seq += "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 20)))
            elif u < 0.7:
                seq = b + cs + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 50)))
            elif u < 0.8:
                seq = "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 90)))
            else:
                seq = b + (t if t[:len(cs)] in cutsites and len(cs) > 0 else cs + t)
                if seq:
                    p = rnd.randrange(len(seq))
                    seq = seq[:p] + rnd.choice("Nn.-*RYX") + seq[p + 1:]
            if rnd.random() < 0.2:
                seq = seq.lower()
            if rnd.random() < 0.1:
                seq = rnd.choice([" ", "\t", "  "]) + seq + rnd.choice(["", " ", "\t "])
            if rnd.random() < 0.1:
                seq = seq[:rnd.randint(0, len(seq))]
            hdr = "@r%d" % ri + ("" if rnd.random() < 0.8 else " some text here")
            qual = "I" * rnd.randint(0, len(seq) + 3)
            thisnl = nl if rnd.random() < 0.9 else rnd.choice(["\n", "\r\n", "\r"])
            chunks.append((hdr + thisnl + seq + thisnl + "+" + thisnl + qual + thisnl).encode("latin-1"))
        if rnd.random() < 0.15:
            chunks.insert(rnd.randrange(len(chunks) + 1), b"\n")  # phase shift
        payload = b"".join(chunks)
        if rnd.random() < 0.3 and payload:
            payload = payload.rstrip(b"\r\n")
        kw = {"cutsite": cutsite}
        if rnd.random() < 0.2:
            kw["maxreads"] = rnd.randint(1, nrec + 2)
        fname = "r%d.fq" % ci
        blob = payload
        if rnd.random() < 0.3:
            fname += ".gz"
            blob = gzip.compress(payload)
        r = {"name": "random%03d" % ci, "filename": fname, "fastq_b64": b64(blob),
             "barcodes": barcodes, "tags": tags, "kwargs": kw}
        r.update(run_ref(fname, blob, barcodes, tags, **kw))
        out.append(r)
    return out


# ---------------------------------------------------------------- error cases
def build_errors():
    out = []

    def attempt(seqs, numseq):
        try:
            ref.build_sequence_tree(list(seqs), numseq)
            return {"sequences": seqs, "numseq": numseq, "ok": True}
        except Exception as e:
            return {"sequences": seqs, "numseq": numseq, "raises": type(e).__name__, "message": str(e)}
    for seqs, n in [(["ACG", "AC"], 2), (["AC", "ACG"], 2), (["AC", "AC"], 2), (["ACG", "ACT", "AC"], 3),
                    (["TT", "ACG", "AC", "A"], 4), (["A", "ACG", "AC"], 3), (["ACG", "AC", "ACG", "AC"], 2),
                    ([], 0), (["", "A"], 2), (["A", ""], 2), (["GAT", "GA", "G"], 3)]:
        out.append(attempt(seqs, n))
    return out


def main():
    os.chdir(tempfile.gettempdir())
    dump("hotpath_primitives.json", primitives())
    hc = hand_cases()
    missing_file_fix(hc)
    dump("hotpath_cases.json", hc)
    dump("hotpath_random.json", random_cases())
    dump("hotpath_errors.json", build_errors())


if __name__ == "__main__":
    main()
