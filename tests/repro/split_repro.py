#!/usr/bin/env python3
"""One case of tests/test_splitter.py::test_gpu_split_fuzz_campaign by seed: where the decisions differ, with the record's place in the buffer."""
import contextlib, io, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_splitter as ts
import tagdigger_amd
from tagdigger_amd import tagdigger_fun as tf
from oracle import tagdigger_oracle as po

seed = int(sys.argv[1]); kern = int(sys.argv[2]) if len(sys.argv) > 2 else 2
case0 = int(os.environ.get("TD_FUZZ_SEED", "4242"))
rng = random.Random(seed)
names = sorted(ts.G["adapters"].keys())
name = rng.choice(names); ad = ts.adapter_of(name)
cutsite = "TGCAT" if name.startswith("Nsi") else "TGCAG"
barcodes = []
while len(barcodes) < rng.randint(1, 12):
    b = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 9)))
    if not any((b + cutsite).startswith(o + cutsite) or (o + cutsite).startswith(b + cutsite) for o in barcodes):
        barcodes.append(b)
data = ts.synth_reads(rng, barcodes, cutsite, ad, rng.randint(1, 4000))
nl = rng.choice([b"\n", b"\n", b"\r\n", b"\r"])
data = data.replace(b"\n", nl)
with contextlib.redirect_stdout(io.StringIO()):
    ends = tf._adapter_ends(ad, barcodes)
eng = tagdigger_amd.Engine(0)
eng.set_splitter(barcodes, cutsite, ad[0][0].replace("^", ""), ad[1][0].replace("^", ""), ends)
eng.set_option("split_kernel", kern)
d = eng.dev_alloc(len(data)); eng.h2d(d, data)
res, _ = eng.split_device(d, len(data), first_line=4 * rng.randint(0, 3))
want = []
po.barcode_splitter_bytes(data, barcodes, cutsite, ad, decisions=want)
got = [(int(a), int(b)) for a, b in res[:len(want)]]
want = [(b, 999 if b < 0 else c) for b, c in want]
print(name, repr(nl), len(data), "bytes", len(want), "reads; sites", ad[0][0], ad[1][0])
# offsets of sequence lines
import re
pos, lines = 0, []
for m in re.finditer(rb"\r\n|\n|\r", data):
    lines.append((pos, m.start())); pos = m.end()
if pos < len(data): lines.append((pos, len(data)))
bad = [i for i in range(len(want)) if got[i] != want[i]]
print(len(bad), "differences")
for i in bad[:10]:
    a, b = lines[4 * i + 1]
    print(i, "got", got[i], "want", want[i], "line bytes [%d, %d) tile %d offset %d len %d" % (a, b, a // 24576, a % 24576, b - a), data[a:b][:60], b"...", data[a:b][-45:])
