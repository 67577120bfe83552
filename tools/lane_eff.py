#!/usr/bin/env python3
"""Phase D lane efficiency against the record length (config 3's index): a tile's wanted lines are matched
in wave passes of 64 lanes, so the cost per byte has a saw-tooth in (wanted lines per tile) mod 64.
  usage: tools/lane_eff.py [first_len last_len]   (TD_OPTS='tile_kb2=24 ...')"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tagdigger_amd
from tagdigger_amd.synth import CONFIGS, SynthConfig

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (38, 52)
eng = tagdigger_amd.Engine(0)
opts = dict(kv.split('=') for kv in os.environ.get('TD_OPTS', '').split())
for k, v in opts.items():
    eng.set_option(k, int(v))
tile = int(opts.get('tile_kb2', 24)) * 1024
base = dict(CONFIGS[3])
for read_len in range(lo, hi + 1):
    c = dict(base, read_len=read_len, nreads=int(4e9 // (2 * read_len + 19)))
    c['body'] = min(c.get('body', 30), read_len - 16)
    cfg = SynthConfig(**c)
    nb = cfg.nbytes()
    d = eng.dev_alloc(nb)
    cfg.fill_device(eng, d, 0, cfg.nreads)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_device(d, nb); eng.sync()
    eng.reset(); eng.set_option("timing", 1)
    for _ in range(4):
        eng.count_device(d, nb)
    eng.sync()
    ms, n = eng.kernel_time_ms()
    eng.set_option("timing", 0)
    per_tile = tile / cfg.record_bytes
    print("read length %3d (%3d B/record, %5.1f wanted lines per tile, %d passes, lanes %4.1f %%): %6.3f ms/GB  %5.2f TB/s  %6.2f Gread/s"
          % (read_len, cfg.record_bytes, per_tile, -(-int(per_tile + 0.999) // 64), per_tile / (64 * -(-int(per_tile + 0.999) // 64)) * 100,
             ms / (nb / 1e9), nb / ms / 1e9, cfg.nreads / ms / 1e6), flush=True)
    eng.dev_free(d)
eng.close()
