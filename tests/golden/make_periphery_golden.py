#!/usr/bin/env python3
"""Generate tests/golden/periphery.json and cli_config1.json from the REAL reference
(build container only; see make_golden.py for the rules -- data only, never reference code).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_periphery_golden.py [/root/reference]
"""
import base64
import contextlib
import gzip
import io
import json
import os
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tagdigger_fun as ref  # noqa: E402


def call(fn, files, *args, **kw):
    """Run fn(*args) in a scratch dir holding `files` {name: text}; record result + stdout + files written."""
    with tempfile.TemporaryDirectory() as d:
        old = os.getcwd()
        os.chdir(d)
        try:
            for name, text in files.items():
                mode = "wb" if isinstance(text, bytes) else "w"
                with open(name, mode, **({} if mode == "wb" else {"newline": ""})) as fh:
                    fh.write(text)
            before = set(os.listdir("."))
            out = io.StringIO()
            rec = {}
            try:
                with contextlib.redirect_stdout(out):
                    rec["result"] = fn(*args, **kw)
            except Exception as e:
                rec["raises"] = type(e).__name__
                rec["message"] = str(e)
            rec["stdout"] = out.getvalue()
            written = {}
            for name in sorted(set(os.listdir(".")) - before):
                written[name] = base64.b64encode(open(name, "rb").read()).decode()
            rec["written_b64"] = written
            return rec
        finally:
            os.chdir(old)


def entry(func, files, args, kw=None):
    kw = kw or {}
    files_json = {k: (base64.b64encode(v).decode() if isinstance(v, bytes) else v) for k, v in files.items()}
    binary = [k for k, v in files.items() if isinstance(v, bytes)]
    import copy
    rec = {"func": func, "files": files_json, "binary_files": binary, "args": copy.deepcopy(args), "kwargs": copy.deepcopy(kw)}
    rec.update(call(getattr(ref, func), files, *args, **kw))
    return rec


def main():
    E = []
    # ---- key file
    key = "File,Barcode,Sample\nlib01.fasta.gz,AACG,PI230189\nlib01.fasta.gz,TTGACC,KD-230-a\nmy-second-lib_fasta.txt,CCGA,foo-2705\n"
    E.append(entry("readBarcodeKeyfile", {"k.csv": key}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "Sample,x,Barcode,File\n s1 ,q, aacg , a.fq \n,,,\ns2,q,,b.fq\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "File,Barcode,Sample\n,,\nf.fq,AAX,s\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "File,Barcode,Sample\nf.fq,AAC,s\nf.fq,AAC,t\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "File,Barcode,Sample\nf.fq,AAC,\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "File,Barcode,Sample\n,AAC,s\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "Files,Barcode,Sample\nf.fq,AAC,s\n"}, ["k.csv"]))
    E.append(entry("readBarcodeKeyfile", {}, ["missing.csv"]))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "Input File,Barcode,Output File\na.fq,AA,o1.fq\na.fq,CC,o1.fq\n"}, ["k.csv"], {"forSplitter": True}))
    E.append(entry("readBarcodeKeyfile", {"k.csv": "Input File,Barcode,Output File\na.fq,AA,o1.fq\na.fq,CC,o2.fq\n"}, ["k.csv"], {"forSplitter": True}))
    # ---- isFastq
    fq = "@h\nACGTNacgtn\n+\nIIIIIIIIII\n"
    E.append(entry("isFastq", {"a.fq": fq}, ["a.fq"]))
    E.append(entry("isFastq", {"a.fq.gz": gzip.compress(fq.encode())}, ["a.fq.gz"]))
    E.append(entry("isFastq", {"a.fq": ">h\nACGT\n+\nIIII\n"}, ["a.fq"]))
    E.append(entry("isFastq", {"a.fq": "@h\nACGU\n+\nIIII\n"}, ["a.fq"]))
    E.append(entry("isFastq", {"a.fq": "@h\nACGT\n-\nIIII\n"}, ["a.fq"]))
    E.append(entry("isFastq", {}, ["nope.fq"]))
    # ---- marker names
    E.append(entry("readMarkerNames", {"m.txt": "TP276,\n  Mrkr2010 \n\n,\nMrkr2011\n"}, ["m.txt"]))
    E.append(entry("readMarkerNames", {}, ["nope.txt"]))
    # ---- Merged
    merged = ("Marker name,Tag sequence\n"
              "TP276,TGCAGAAAAACACGT[A/C]TCTTTGCTTCTACCAGATGCACAAAGAGAGGGGAAATAGGCAAGA\n"
              "Mrkr2010,TGCAGACCTTCTTCTTCTCGT[AAG/GAC]CAACAAACAAGGTAGTAAAGACCACCACAACCACGTGC\n"
              "Mrkr2011,TGCAGCGAACAATG[CAC/TAC/TAT]TGTACATTGAAGAACACTACAGACTATTACAAGCTCACACGTC\n"
              "Mrkr2012,TGCAGTTTTCCC[AG/C-]AGAGAGA\n")
    E.append(entry("readTags_Merged", {"t.csv": merged}, ["t.csv"]))
    E.append(entry("readTags_Merged", {"t.csv": merged}, ["t.csv"], {"toKeep": ["TP276", "Mrkr2012"]}))
    E.append(entry("readTags_Merged", {"t.csv": merged + "Dup1,TGCAGAAAAACACGT[A/G]TCTTTGCTTCTACCAGATGCACAAAGAGAGGGGAAATAGGCAAGA\n"}, ["t.csv"]))
    E.append(entry("readTags_Merged", {"t.csv": merged + "Dup1,TGCAGAAAAACACGT[A/G]TCTTTGCTTCTACCAGATGCACAAAGAGAGGGGAAATAGGCAAGA\n"}, ["t.csv"], {"allowDuplicates": True}))
    E.append(entry("readTags_Merged", {"t.csv": "Marker name,Tag sequence\nM_1,ACGT[A/C]T\n"}, ["t.csv"]))
    E.append(entry("readTags_Merged", {"t.csv": "Marker name,Tag sequence\nM1,ACGTAT\n"}, ["t.csv"]))
    E.append(entry("readTags_Merged", {"t.csv": "Marker name,Tag sequence\nM1,ACGN[A/C]T\n"}, ["t.csv"]))
    E.append(entry("readTags_Merged", {"t.csv": "Marker,Tag\nM1,ACG[A/C]T\n"}, ["t.csv"]))
    E.append(entry("readTags_Merged", {}, ["nope.csv"]))
    # ---- Rows / Columns
    rows = ("Marker name,Allele name,Tag sequence\nTP276,0,TGCAGAAAAACACGTATCT\nTP276,1,TGCAGAAAAACACGTCTCT\n"
            "Mrker2035,dom,TGCAGCCCCC\nMrker4050,0,TGCAGAGAG\nMrker4050,1,TGCAGAGTG\nMrker4050,2,tgcagagcg\n")
    E.append(entry("readTags_Rows", {"r.csv": rows}, ["r.csv"]))
    E.append(entry("readTags_Rows", {"r.csv": rows}, ["r.csv"], {"toKeep": ["Mrker4050"]}))
    E.append(entry("readTags_Rows", {"r.csv": rows + "X,0,TGCAGCCCCC\n"}, ["r.csv"]))
    E.append(entry("readTags_Rows", {"r.csv": "Marker name,Allele name,Tag sequence\nA_B,0,ACGT\n"}, ["r.csv"]))
    E.append(entry("readTags_Rows", {"r.csv": "Marker name,Allele name,Tag sequence\nAB,0,ACGU\n"}, ["r.csv"]))
    cols = "Marker name,Tag sequence 0,Tag sequence 1\nTP276,TGCAGAAAAACACGTATCT,TGCAGAAAAACACGTCTCT\nM2,ACGTACGTAA,ACCTACGTTA\n"
    E.append(entry("readTags_Columns", {"c.csv": cols}, ["c.csv"]))
    E.append(entry("readTags_Columns", {"c.csv": cols + "M3,ACGTACGTAA,ACGTACGTAC\n"}, ["c.csv"]))
    E.append(entry("readTags_Columns", {"c.csv": "Marker name,Tag sequence 0\nM,ACGT\n"}, ["c.csv"]))
    # ---- UNEAK
    uneak = (">TP276_query_64\nTGCAGAAAAACACGTATCTTTGCTTCTACCAGATGCACAAAGAGAGGGGAAATAGGCAAGAGCAA\n"
             ">TP276_hit_64\nTGCAGAAAAACACGTCTCTTTGCTTCTACCAGATGCACAAAGAGAGGGGAAATAGGCAAGAGCAA\n"
             ">TP539_query_30\nTGCAGAAAACACAGAAACAGAACCATGCACAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n"
             ">TP539_hit_64\nTGCAGAAAACACAGAAACAGAACTATGCACGAGTCACCAGCGGCTGAAAAACATGAATGATAGAG\n")
    E.append(entry("readTags_UNEAK_FASTA", {"u.fa": uneak}, ["u.fa"]))
    E.append(entry("readTags_UNEAK_FASTA", {"u.fa": uneak}, ["u.fa"], {"toKeep": ["TP539"]}))
    E.append(entry("readTags_UNEAK_FASTA", {"u.fa": uneak.replace(">TP539_hit", ">TP540_hit")}, ["u.fa"]))
    E.append(entry("readTags_UNEAK_FASTA", {"u.fa": ">XP1_query_4\nACGT\n"}, ["u.fa"]))
    E.append(entry("readTags_UNEAK_FASTA", {"u.fa": ">TP1_query_8\nACGTACGT\n>TP1_hit_4\nACGTTTTT\n"}, ["u.fa"]))
    # ---- reverseComplement
    E.append(entry("reverseComplement", {}, ["ACGTTGCAN"]))
    E.append(entry("reverseComplement", {}, ["acgtA"]))
    # ---- Stacks catalog (v1 columns: locus 2, sequence 9, haplotype 3, SNP column 3; v2: 1, 5, 2, 2)
    def stacks_v1(rows_tags, rows_snps, rows_alleles):
        t = "# comment\n" + "".join("0\t1\t%s\t\t0\t+\tconsensus\t0\t\t%s\t0\t0\t0\n" % r for r in rows_tags)
        sn = "# c\n" + "".join("0\t1\t%s\t%d\tE\t0\tA\tC\n" % r for r in rows_snps)
        al = "# c\n" + "".join("0\t1\t%s\t%s\t50.0\t10\n" % r for r in rows_alleles)
        return {"cat.tags.tsv": t, "cat.snps.tsv": sn, "cat.alleles.tsv": al}
    st = stacks_v1([("1", "TGCAGAAAACCCCGGGGTTTT"), ("2", "TGCAGTTTTGGGGCCCCAAAA"), ("3", "TGCAGACGTACGTACGTACGN"), ("4", "tgcagcccccaaaaa")],
                   [("1", 7), ("1", 12), ("2", 9), ("3", 6)],
                   [("1", "AC"), ("1", "GT"), ("1", "AT"), ("2", "C"), ("2", "A"), ("3", "G"), ("4", "")])
    sargs = ["cat.tags.tsv", "cat.snps.tsv", "cat.alleles.tsv"]
    E.append(entry("readTags_Stacks", st, sargs))
    E.append(entry("readTags_Stacks", st, sargs, {"binaryOnly": True}))
    E.append(entry("readTags_Stacks", st, sargs, {"toKeep": ["2", "4"]}))
    stgz = dict(st)
    stgz["cat.tags.tsv.gz"] = gzip.compress(st["cat.tags.tsv"].encode(), mtime=0)
    del stgz["cat.tags.tsv"]
    E.append(entry("readTags_Stacks", stgz, ["cat.tags.tsv.gz", "cat.snps.tsv", "cat.alleles.tsv"]))
    v2 = {"t.tsv": "# c\n1\t7\t\t\t\tTGCAGAAAACCCC\n1\t8\t\t\t\tTGCAGTTTTGGGG\n",
          "s.tsv": "1\t7\t6\tE\n1\t8\t5\tE\n", "a.tsv": "1\t7\tA\n1\t7\tG\n1\t8\tC\n1\t8\tA\n"}
    E.append(entry("readTags_Stacks", v2, ["t.tsv", "s.tsv", "a.tsv"], {"version": 2}))
    E.append(entry("readTags_Stacks", v2, ["t.tsv", "s.tsv", "a.tsv"], {"version": 2, "binaryOnly": True}))
    E.append(entry("readTags_Stacks", {}, ["no.tsv", "no2.tsv", "no3.tsv"]))
    bad = dict(st); bad["cat.snps.tsv"] = "0\t1\t1\tx\n"
    E.append(entry("readTags_Stacks", bad, sargs))                       # ValueError
    bad = dict(st); bad["cat.alleles.tsv"] = "0\t1\t9\tA\n"
    E.append(entry("readTags_Stacks", bad, sargs))                       # KeyError
    bad = dict(st); bad["cat.alleles.tsv"] = "0\t1\n"
    E.append(entry("readTags_Stacks", bad, sargs))                       # IndexError
    bad = dict(st); bad["cat.alleles.tsv"] = st["cat.alleles.tsv"] + "0\t1\t1\tAC\t1\t1\n"
    E.append(entry("readTags_Stacks", bad, sargs, {"binaryOnly": True}))  # non-unique names
    bad = dict(st); bad["cat.alleles.tsv"] = "0\t1\t2\tCAG\n"
    E.append(entry("readTags_Stacks", bad, sargs))                       # haplotype longer than the SNP list
    # ---- TASSEL-GBSv2 SAM
    sam = ("@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:Chr01\tLN:43270923\n@SQ\tSN:scaffold_12\tLN:9000\n@PG\tID:bowtie2\n"
           "tagSeq=A\t0\tChr01\t1000\t42\t20M\t*\t0\t0\tTGCAGAAAACCCCGGGGTTT\tIIII\n"
           "tagSeq=B\t0\tChr01\t1000\t42\t20M\t*\t0\t0\tTGCAGAAAACCCCGGTGTTT\tIIII\n"
           "tagSeq=C\t16\tChr01\t2000\t42\t10M2D8M\t*\t0\t0\tAAACCCGGGTTTACTGCA\tIIII\n"
           "tagSeq=D\t16\tChr01\t2000\t42\t10M2D8M\t*\t0\t0\tAAACCCGGGTTAACTGCA\tIIII\n"
           "tagSeq=E\t4\t*\t0\t0\t*\t*\t0\t0\tTGCAGGGGGGGGGG\tIIII\n"
           "tagSeq=F\t0\tscaffold_12\t77\t42\t12M\t*\t0\t0\tTGCAGTTTTTTT\tIIII\n"
           "tagSeq=G\t0\tscaffold_12\t77\t42\t16M\t*\t0\t0\tTGCAGTTTTTTTACGT\tIIII\n"
           "tagSeq=H\t0\tscaffold_12\t77\t42\t12M\t*\t0\t0\tTGCAGTTATTTT\tIIII\n"
           "tagSeq=I\t0\tscaffold_12\t77\t42\t8M\t*\t0\t0\tTGCAGTTA\tIIII\n"
           "tagSeq=J\t16\tChr01\t500\t42\t4M1I10M\t*\t0\t0\tAAACCCGGGTTCTGCA\tIIII\n"
           "tagSeq=K\t0\tchromosome03\t31\t42\t9M\t*\t0\t0\tTGCAGACGT\tIIII\n"
           "tagSeq=L\t0\tchromosome03\t31\t42\t9M\t*\t0\t0\tTGCAGACGA\tIIII\n"
           "tagSeq=M\t0\tchromosome03\t31\t42\t9M\t*\t0\t0\tTGCAGTCGA\tIIII\n")
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"]))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"binaryOnly": True}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"noMonomorphic": True}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"toKeep": ["S01_1015", "S03_31", "S01_1989"]}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"toKeep": ["nothing"]}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"writeMarkerKey": True, "keyfilename": "key_out.csv"}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"writeMarkerKey": True, "keyfilename": "key_out.csv", "binaryOnly": True,
                                                                   "toKeep": ["S03_31", "S03_35"]}))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam}, ["t.sam"], {"writeMarkerKey": True}))
    E.append(entry("readTags_TASSELSAM", {}, ["missing.sam"]))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam + "\n"}, ["t.sam"]))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam + "x\t0\tChr01\t9\t1\t5M\t*\t0\t0\tTGCNG\tI\n"}, ["t.sam"]))
    E.append(entry("readTags_TASSELSAM", {"t.sam": sam + "x\tzz\tChr01\t9\t1\t5M\t*\t0\t0\tTGCAG\tI\n"}, ["t.sam"]))
    # ---- pyRAD .alleles
    pyrad = (">s1_0    TGCAGAAAACCCCGGGG--\n>s1_1    TGCAGAAAACCCCGGGG--\n>s2_0    TGCAGAAAATCCCGGGGTT\n>s2_1    TGCAGAAAACCCCGGGG\n"
             "//                   *              |0|\n"
             ">s1_0    TGCAGTTNT\n>s1_1    TGCAGTTAT\n>s2_0    TGCAGTTCT\n>s2_1    TGCAGTT-T\n"
             "//             -     |1|\n"
             ">s1_0    TGCAGGG\n>s1_1    TGCAGGG\n"
             "//           |2|\n"
             ">s1_0    TGCAGCA\n>s1_1    TGCAGCC\n>s2_0    TGCAGCG\n"
             "//      * |13|\n")
    E.append(entry("readTags_pyRAD", {"p.alleles": pyrad}, ["p.alleles"]))
    E.append(entry("readTags_pyRAD", {"p.alleles": pyrad}, ["p.alleles"], {"binaryOnly": True}))
    E.append(entry("readTags_pyRAD", {"p.alleles": pyrad}, ["p.alleles"], {"toKeep": ["1", "13"]}))
    E.append(entry("readTags_pyRAD", {"p.alleles": pyrad + ">s1_0    TGCAGXA\n"}, ["p.alleles"]))
    E.append(entry("readTags_pyRAD", {"p.alleles": pyrad + "junk\n"}, ["p.alleles"]))
    E.append(entry("readTags_pyRAD", {"p.alleles": "//   |5|\n"}, ["p.alleles"]))
    E.append(entry("readTags_pyRAD", {"p.alleles": ">a   ---\n>b   ---\n//   |5|\n"}, ["p.alleles"]))
    E.append(entry("readTags_pyRAD", {}, ["missing.alleles"]))
    # ---- compareTags
    E.append(entry("compareTags", {}, [["ACGTA", "ACCTA", "ACGTT"]]))
    E.append(entry("compareTags", {}, [["ACGTA", "ACC"]]))
    E.append(entry("compareTags", {}, [["ACGTA", "ACC"]], {"trim": False}))
    # ---- sanitizeTags
    E.append(entry("sanitizeTags", {}, [[["A_x_0", "A_y_1", "B_x_0", "B_y_1", "C_x_0"], ["ACGT", "ACGA", "TTT", "TTTG", "GGG"]]]))
    E.append(entry("sanitizeTags", {}, [[["TP27_a_0", "TP27_c_1", "TP276_a_0", "TP276_c_1", "X_a_0", "X_c_1"],
                                         ["AAAA", "AAAC", "CCCC", "CCCG", "AAAAT", "GGGG"]]]))
    E.append(entry("sanitizeTags", {}, [[["A_0", "B_0", "C_0"], ["ACG", "ACG", "ACGT"]]]))
    E.append(entry("sanitizeTags", {}, [[["A_0", "B_0"], ["ACG", "TTT"]]]))
    # ---- combine / writers / extractMarkers
    bck = {"b.fq": [["AA", "CC"], ["s2", "s1"]], "a.fq": [["GG", "TT", "AC"], ["s1", "s3", "s1"]]}
    cd = {"b.fq": [[1, 2], [3, 4]], "a.fq": [[10, 20], [30, 40], [100, 200]]}
    E.append(entry("combineReadCounts", {}, [cd, bck]))
    E.append(entry("writeCounts", {}, ["out.csv", [[1, 2], [3, 4]], ["s,1", 's"2'], ["M_A_0", "M_C_1"]]))
    E.append(entry("extractMarkers", {}, [["M1_A_0", "M2_G_1", "M1_C_1", "M2_T_0", "Z_only_0"]]))
    E.append(entry("writeDiploidGeno", {}, ["g.csv", [[1, 0, 0, 0], [0, 5, 2, 2], [3, 3, 0, 1]], ["x", "y", "z"],
                                            ["M1_A_0", "M1_C_1", "M2_G_1", "M2_T_0"]]))
    E.append(entry("writeDiploidGeno", {}, ["g.csv", [[1, 0]], ["x"], ["M1_A_0", "M1_C_2"]]))
    with open(os.path.join(HERE, "periphery.json"), "w") as fh:
        json.dump(E, fh, indent=0, separators=(",", ":"))
        fh.write("\n")
    print("wrote periphery.json", len(E), "entries")

    # ---- end-to-end: the reference CLI on a config-1 style library (synthetic stream, host generator)
    from tagdigger_amd.synth import SynthConfig
    from helpers import synth_host_bytes
    cfg = SynthConfig(nreads=3000, nbar=8, nmarkers=50, seed=1234)
    fastq = bytes(synth_host_bytes(cfg, 0, cfg.nreads))
    samples = ["S0", "S1", "S2", "S3", "S4", "S5", "S0", "S2"]            # 8 barcodes -> 6 samples
    key = "File,Barcode,Sample\n" + "".join("lib.fq.gz,%s,%s\n" % (b, s) for b, s in zip(cfg.barcodes, samples))
    # Merged rows from the biallelic pairs (differ at exactly one base)
    rows = ["Marker name,Tag sequence"]
    for m in range(len(cfg.tags) // 2):
        a, b = cfg.tags[2 * m], cfg.tags[2 * m + 1]
        pos = [i for i in range(len(a)) if a[i] != b[i]][0]
        rows.append("Mk%d,%s[%s/%s]%s" % (m, a[:pos], a[pos], b[pos], a[pos + 1:]))
    tagscsv = "\n".join(rows) + "\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "key.csv"), "w").write(key)
        open(os.path.join(d, "tags.csv"), "w").write(tagscsv)
        gzbytes = gzip.compress(fastq, mtime=0)
        open(os.path.join(d, "lib.fq.gz"), "wb").write(gzbytes)
        cmd = [sys.executable, os.path.join(REF, "tagdigger_script.py"), "-e", "PstI", "--MergedTags", "tags.csv",
               "-b", "key.csv", "-o", "counts.csv", "-g", "geno.csv"]
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        p = subprocess.run(cmd, cwd=d, capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        cli = {"argv": cmd[2:], "key_csv": key, "tags_csv": tagscsv, "fastq_gz_b64": base64.b64encode(gzbytes).decode(),
               "stdout": p.stdout,
               "counts_csv_b64": base64.b64encode(open(os.path.join(d, "counts.csv"), "rb").read()).decode(),
               "geno_csv_b64": base64.b64encode(open(os.path.join(d, "geno.csv"), "rb").read()).decode()}
    with open(os.path.join(HERE, "cli_config1.json"), "w") as fh:
        json.dump(cli, fh, separators=(",", ":"))
        fh.write("\n")
    print("wrote cli_config1.json", os.path.getsize(os.path.join(HERE, "cli_config1.json")))


if __name__ == "__main__":
    main()
