import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tagdigger_amd
eng = tagdigger_amd.Engine(0)
B = ["AACG", "TTGACC"]; T = ["TGCAGAAAC", "TGCAGGGGT"]
for name, seq in [("plain", "AACGTGCAGAAACTTTT"), ("lead", "  AACGTGCAGAAAC "), ("tab", "\tAACGTGCAGAAAC"), ("N", "NACGTGCAGAAAC")]:
    data = ("@r0\n" + seq + "\n+\n" + "I" * len(seq) + "\n").encode()
    for fp in (1, 0):
        eng.set_index(B, T, "TGCAG")
        eng.set_option("fastpath", fp)
        eng.count_bytes(data)
        print(name, "fastpath", fp, eng.counts(), eng.stats(), eng.debug_counters()[12:20])
