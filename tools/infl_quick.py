import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig
from compress_formats import bgzf_bytes
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb); cfg.fill_device(eng, d, 0, reads); host = eng.d2h(d, nb); eng.dev_free(d)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
eng.count_bytes(host); want = eng.counts_numpy().copy()
path = "/tmp/q.bgzf.fq.gz"
for level in (1, 6):
    open(path, "wb").write(bgzf_bytes(host, level=level, threads=32))
    for crc in (1, 0):
        eng.set_option("gpu_inflate_crc", crc)
        best = 1e9
        for _ in range(3):
            eng.reset(); t0 = time.perf_counter(); eng.count_file(path); eng.sync(); best = min(best, time.perf_counter() - t0)
        assert (eng.counts_numpy() == want).all()
        print("level %d crc %d: %.2f GB compressed, count_file %.2f GB/s = %.1f Mreads/s" % (level, crc, os.path.getsize(path)/1e9, nb / best / 1e9, reads / best / 1e6), flush=True)
os.remove(path)
