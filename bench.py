#!/usr/bin/env python3
"""Headline benchmark: FASTQ reads/s of the tag-counting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (libtagdig's fused count kernel, through
the C-ABI) over one synthetic FASTQ library already resident in HBM, plus --
for N > 1 -- the one RCCL all-reduce of the integer count matrix.  Workload at
N=1: the configuration BASELINE.json's metric is quoted on (200 M reads x 384
barcodes x 100 k tags, 100 bp reads, 219 B/record = 43.8 GB), one such library
per GPU (weak scaling, BASELINE config 4's file-per-GPU sharding).

Rank 0 prints ONE JSON line (see the driver contract in the task statement).
The `roofline` object prices the count kernel against HBM bandwidth using
ALGORITHMIC bytes = 219 B x reads per launch and the kernel's own duration
from HIP events on the launch stream; `cpu_baseline` times the oracle's C
restatement on a bounded prefix of the same stream on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=200_000_000, help="reads per GPU")
    ap.add_argument("--barcodes", type=int, default=384)
    ap.add_argument("--markers", type=int, default=50_000, help="tags = 2 x markers")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cutsite", default="TGCAG", help="cut site, IUPAC codes allowed (BASELINE config 5: CWGC)")
    ap.add_argument("--bclen-max", type=int, default=8, help="barcodes are 4..N bases long (config 5: 10)")
    ap.add_argument("--tile-kb", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="reads in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-python-sample", type=int, default=400_000,
                    help="reads for the pure-Python restatement's timing inside cpu_baseline (0 = skip)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--stagger", type=int, default=-1)
    ap.add_argument("--table-load", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    ap.add_argument("--prio", type=int, default=-1)
    ap.add_argument("--debug-ablate", type=int, default=0, help="timing-only kernel ablation bits (implies --no-check)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    # TD_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- a correctness rehearsal of the N > 1 code
    # path on a one-GPU box (never a measurement)
    rehearsal = os.environ.get("TD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig

    cfg = SynthConfig(nreads=args.reads, nbar=args.barcodes, nmarkers=args.markers, seed=args.seed,
                      cutsite=args.cutsite, bclen=(4, args.bclen_max))
    eng = tagdigger_amd.Engine(local_rank)
    if args.tile_kb:
        eng.set_option("tile_kb", args.tile_kb)
    if args.blocks_per_cu:
        eng.set_option("blocks_per_cu", args.blocks_per_cu)
    if args.stagger >= 0:
        eng.set_option("stagger", args.stagger)
    if args.table_load:
        eng.set_option("table_load_pct", args.table_load)
    if args.nt >= 0:
        eng.set_option("nt_loads", args.nt)
    if args.prio >= 0:
        eng.set_option("prio", args.prio)
    if args.debug_ablate:
        eng.set_option("debug_ablate", args.debug_ablate)
        args.no_check = True
    nbytes = cfg.nbytes()
    fastq = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    first_read = rank * cfg.nreads                      # this GPU's library = its own slice of the stream
    cfg.fill_device(eng, fastq.data_ptr(), first_read, cfg.nreads)
    # The matrix lives in torch tensors so that RCCL can reduce it in place.  With several GPUs there are
    # two: the all-reduce of one pass (the path's one exchange: an integer sum over xGMI) runs on RCCL's
    # stream while the next pass counts into the other matrix.
    nmat = 2 if world > 1 else 1
    mats = [torch.zeros(len(cfg.barcodes) * len(cfg.tags), dtype=torch.int32, device=dev) for _ in range(nmat)]
    reducing = [None] * nmat
    counts = mats[0]
    eng.bind_counts(counts.data_ptr())
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    stream = torch.cuda.current_stream().cuda_stream
    passes = [0]

    def step():
        b = passes[0] % nmat
        passes[0] += 1
        if reducing[b] is not None:       # this matrix's previous all-reduce (two passes ago)
            reducing[b].wait()
            reducing[b] = None
        mats[b].zero_()
        eng.bind_counts(mats[b].data_ptr())
        eng.count_device(fastq.data_ptr(), nbytes, stream=stream)
        if world > 1:
            reducing[b] = dist.all_reduce(mats[b], async_op=True)
        return mats[b]

    def fence():
        for b in range(nmat):
            if reducing[b] is not None:
                reducing[b].wait()
                reducing[b] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    # ---- correctness of what is being timed (rank-local shard, before any all-reduce)
    check = None
    if not args.no_check:
        # the whole matrix against the one the generator's own choices imply (built on the device from
        # the shared spec include/td_synth_spec.h; nothing is parsed, nothing of oracle/ is involved)
        counts.zero_()
        eng.reset()
        eng.bind_counts(counts.data_ptr())
        eng.count_device(fastq.data_ptr(), nbytes, stream=stream)
        torch.cuda.synchronize()
        want = torch.zeros_like(counts)
        hits = cfg.expected_device(eng, want.data_ptr(), first_read, cfg.nreads)
        st = eng.stats()
        ok = bool(torch.equal(counts, want)) and st["tag"] == hits and st["reads"] == cfg.nreads
        check = {"bit_exact_vs_expected": ok, "reads": int(st["reads"]), "barcut": int(st["barcut"]), "tag": int(st["tag"])}
        if not ok:
            print("bench.py: rank %d COUNT MISMATCH against the generator's expected matrix" % rank, file=sys.stderr)
            sys.exit(3)

    eng.set_option("timing", 1)
    fence()
    t0 = time.perf_counter()
    last = counts
    for _ in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    kms, klaunches = eng.kernel_time_ms()
    fixups = eng.debug_counters()[11]
    eng.set_option("timing", 0)

    if world > 1:
        t = torch.tensor([elapsed, kms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kms = float(t[0]), float(t[1])
        # every rank now holds the same summed matrix: its total must equal the sum of all shards' hits
        tot = torch.tensor([int(last.to(torch.int64).sum())], dtype=torch.int64, device=dev)
        mine = torch.tensor([check["tag"] if check else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(mine)
        if check and int(tot[0]) != int(mine[0]):
            print("bench.py: all-reduced matrix total %d != sum of shard hits %d" % (int(tot[0]), int(mine[0])),
                  file=sys.stderr)
            sys.exit(3)

    if rank == 0:
        total_reads = cfg.nreads * world * args.steps
        value = total_reads / elapsed
        algo_bytes = cfg.record_bytes * cfg.nreads
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        metric = "FASTQ reads/sec (whole node), 200M-read \u00d7 100k-tag synthetic, 1/2/4/8 MI355X"
        try:                                              # (verbatim from BASELINE.json when it is there)
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            pass
        out = {
            "metric": metric,
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] shape, device-resident (tier T1): %d reads x %d barcodes x %d tags "
                                   "per GPU, 100 bp reads, %d B/record, one library per GPU"
                                   % (cfg.nreads, len(cfg.barcodes), len(cfg.tags), cfg.record_bytes),
                       "reads_per_gpu": cfg.nreads, "barcodes": len(cfg.barcodes), "tags": len(cfg.tags),
                       "fastq_bytes_per_gpu": nbytes, "sharding": "library-per-GPU + RCCL all-reduce(int32 count matrix) per pass, overlapped with the next pass"
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tdk::k_fast (+k_resolve, fix-up pass)", "kernel_ms": kms, "fixup_queue": fixups, "kernel_launches": klaunches,
                         "algorithmic_bytes_per_launch": algo_bytes},
            "check": check,
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, min(args.cpu_sample, cfg.nreads), min(args.cpu_python_sample, cfg.nreads))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def cpu_baseline(cfg, sample_reads, python_reads=0):
    """The oracle's C restatement (oracle/oracle.c, scalar, one thread) on the first
    `sample_reads` reads of the same stream, produced by the host reference generator.
    `python_restatement`: the pure-Python restatement (oracle/tagdigger_oracle.py: the reference's own
    nested-list trie and per-line loop, the closest thing to the reference that can travel; BASELINE.md
    has its calibration against the real reference: 41.6 k vs 28.2 k reads/s on the build host)."""
    from helpers import synth_host_bytes
    from oracle import c_oracle
    data = synth_host_bytes(cfg, 0, sample_reads)
    t0 = time.perf_counter()
    ora = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ora.count_bytes(data)
    loop_s = time.perf_counter() - t0
    out = {"value": sample_reads / loop_s, "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": "first %d reads of the same synthetic stream (%.2f GB), record loop only; "
                     "trie build %.2f s timed separately; scalar C restatement oracle/oracle.c"
                     % (sample_reads, data.nbytes / 1e9, build_s),
           "index_build_s": build_s, "loop_s": loop_s}
    # the same C restatement on every core this process may use (at most 16: the GPU box's share per GPU),
    # one shard of whole records per thread, matrices summed -- the "own CPU path at all cores" line of SURVEY 8d
    try:
        ncore = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        ncore = max(1, min(16, os.cpu_count() or 1))
    if ncore > 1:
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        per = (sample_reads + ncore - 1) // ncore
        shards = [(k * per, min(sample_reads, (k + 1) * per)) for k in range(ncore) if k * per < sample_reads]
        oracles = [c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite) for _ in shards]
        mats = [np.zeros((len(cfg.barcodes), len(cfg.tags)), dtype=np.uint64) for _ in shards]
        for m in mats:
            m.fill(0)                     # (pages touched before the clock starts)

        def run(k):
            a, b = shards[k]
            return oracles[k].count_bytes(data[a * cfg.record_bytes:b * cfg.record_bytes], first_line=4 * a, counts=mats[k])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(len(shards)) as ex:
            parts = list(ex.map(run, range(len(shards))))
        par_s = time.perf_counter() - t0
        out["all_cores"] = {"value": sample_reads / par_s, "unit": "reads/s", "cores": len(shards), "loop_s": par_s,
                            "matrix_total": int(sum(int(m.sum()) for m in parts))}
    if python_reads > 0:
        from oracle import tagdigger_oracle as po
        pdata = bytes(data[:python_reads * cfg.record_bytes])
        t0 = time.perf_counter()
        index = po.prepare_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        pbuild = time.perf_counter() - t0
        t0 = time.perf_counter()
        po.count_bytes(pdata, cfg.barcodes, cfg.tags, cfg.cutsite, index=index)
        ploop = time.perf_counter() - t0
        out["python_restatement"] = {"value": python_reads / ploop, "unit": "reads/s", "cores": 1,
                                     "sample": "first %d reads, record loop only" % python_reads,
                                     "index_build_s": pbuild, "loop_s": ploop}
    return out


if __name__ == "__main__":
    main()
