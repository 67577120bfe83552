import contextlib, io, os, random, sys, re
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_splitter as ts
import tagdigger_amd
from tagdigger_amd import tagdigger_fun as tf
from oracle import tagdigger_oracle as po
rng = random.Random(1)
barcodes = ["AACG", "TTGACC", "CGT", "GATTACAG", "CCA"]
cutsite = "TGCAG"; ad = ts.adapter_of("PstI-MspI-Hall")
data = ts.synth_reads(rng, barcodes, cutsite, ad, 4000)
with contextlib.redirect_stdout(io.StringIO()):
    ends = tf._adapter_ends(ad, barcodes)
eng = tagdigger_amd.Engine(0)
eng.set_splitter(barcodes, cutsite, ad[0][0].replace("^", ""), ad[1][0].replace("^", ""), ends)
d = eng.dev_alloc(len(data)); eng.h2d(d, data)
res, _ = eng.split_device(d, len(data))
want = []
po.barcode_splitter_bytes(data, barcodes, cutsite, ad, decisions=want)
got = [(int(a), int(b)) for a, b in res[:len(want)]]
want = [(b, 999 if b < 0 else c) for b, c in want]
pos, lines = 0, []
for m in re.finditer(rb"\r\n|\n|\r", data):
    lines.append((pos, m.start())); pos = m.end()
bad = [i for i in range(len(want)) if got[i] != want[i]]
print(len(bad), "differences of", len(want))
for i in bad[:8]:
    a, b = lines[4 * i + 1]
    print(i, "got", got[i], "want", want[i], "tile %d offset %d len %d" % (a // 24576, a % 24576, b - a), data[a:b][-30:])
# entries of barcode 3 ending like that
b3 = ends[3] if isinstance(ends, list) else None
print(type(ends), str(ends)[:300])
