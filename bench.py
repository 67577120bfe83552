#!/usr/bin/env python3
"""Headline benchmark: FASTQ reads/s of the tag-counting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (libtagdig's fused count kernel, through
the C-ABI) over one synthetic FASTQ library already resident in HBM, plus --
for N > 1 -- the one RCCL all-reduce of the integer count matrix.  Workload at
N=1: the configuration BASELINE.json's metric is quoted on (200 M reads x 384
barcodes x 100 k tags, 100 bp reads, 219 B/record = 43.8 GB).
  --scaling weak   (default) one such library per GPU: BASELINE config 4's file-per-GPU sharding
  --scaling strong ONE library byte-sharded over the N GPUs (each shard counted with its true
                   first line index), the same single all-reduce
  --config 2|4|5   the other BASELINE index shapes (parity-test cases; not the headline line)

Rank 0 prints ONE JSON line (see the driver contract in the task statement).
The `roofline` object prices the count kernel against HBM bandwidth using
ALGORITHMIC bytes = 219 B x reads per launch and the kernel's own duration
from HIP events on the launch stream; `cpu_baseline` times the oracle (the
reference-equivalent Python restatement, and its C port beside it) on a bounded
prefix of the same stream on this box's host cores; `tiers` reports the
PCIe- and file-inclusive rates (never `value`).
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 4, 5),
                    help="BASELINE config whose index shape and seed to use (3 = the one the metric is quoted on)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (weak) / in the library (strong); 0 = 200 M, config 2: 50 M")
    ap.add_argument("--barcodes", type=int, default=0)
    ap.add_argument("--markers", type=int, default=0, help="tags = 2 x markers")
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--cutsite", default="", help="cut site, IUPAC codes allowed (BASELINE config 5: CWGC)")
    ap.add_argument("--bclen-max", type=int, default=0, help="barcodes are 4..N bases long (config 5: 10)")
    ap.add_argument("--skew", type=float, default=0.0, help="Zipf exponent of the hit distribution over tags and barcodes (0 = uniform)")
    ap.add_argument("--tile-kb", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="reads in the C port's cpu_baseline sample (0 = skip cpu_baseline)")
    ap.add_argument("--cpu-python-sample", type=int, default=400_000,
                    help="reads for the pure-Python restatement's timing (cpu_baseline.value; 0 = skip)")
    ap.add_argument("--oracle-sample", type=int, default=2_000_000,
                    help="reads of the resident stream counted by the C oracle and compared with a GPU pass over the same bytes (0 = skip)")
    ap.add_argument("--tier-reads", type=int, default=16_000_000, help="reads in the T2/T3 tier measurements (0 = skip)")
    ap.add_argument("--other-configs", default="auto",
                    help="BASELINE configs measured after the headline at N=1 (3 steps each, full read count, bit-exact check); "
                         "auto = 2,4,5 with the default workload, none otherwise; '' = skip")
    ap.add_argument("--other-reads", type=int, default=0, help="reads of those passes (0 = each config's own count)")
    ap.add_argument("--traffic", default="auto", choices=("auto", "off"),
                    help="roofline.traffic: 'auto' = two rocprofv3 --pmc child runs of this script (FETCH_SIZE, WRITE_SIZE) when "
                         "rocprofv3 is on PATH, else null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # (the counter passes' run: launches only)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--stagger", type=int, default=-1)
    ap.add_argument("--table-load", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    ap.add_argument("--prio", type=lambda v: int(v, 0), default=-1)
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to td_set_option (repeatable)")
    ap.add_argument("--debug-ablate", type=int, default=0, help="timing-only kernel ablation bits (implies --no-check)")
    return ap.parse_args(argv)


def self_launch(args):
    """`bench.py --gpus N` from a bare shell: N fresh ranks under torch.distributed.run, started BEFORE this
    process has touched the GPU (it never does); rank 0's JSON line and the launcher's exit code are relayed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def is_default_workload(args):
    return (args.config == 3 and not (args.reads or args.barcodes or args.markers or args.cutsite or args.bclen_max
                                      or args.skew or args.seed >= 0 or args.tile_kb or args.blocks_per_cu or args.opt
                                      or args.debug_ablate or args.stagger >= 0 or args.table_load or args.nt >= 0 or args.prio >= 0))


def measure_traffic(args):
    """roofline.traffic, live: this script again as a child of `rocprofv3 --pmc <counter>` (one pass per counter:
    the TCC slots do not hold both), launches only, before this process touches the GPU.  Returns the dict that
    goes into the line, or None (no rocprofv3, a failed pass)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return None
    # (this run is itself being profiled: no profiler inside a profiler)
    if any(("rocprof" in (os.environ.get(k) or "").lower()) for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR")):
        return None
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="tdpmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", "2", "--warmup", "1"]
            try:
                p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            except OSError:
                return None
            try:
                p.wait(timeout=300)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None
            vals = {}
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and ("k_fast4" in row.get("Kernel_Name", "") or "k_fast2" in row.get("Kernel_Name", "")):
                            vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not vals:
                return None
            name, v = max(vals.items(), key=lambda kv: max(kv[1]))
            got[counter] = (name, sum(v) / len(v) * 1024.0, len(v))           # (reported in KiB per dispatch)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]
    return {"fabric_bytes_per_launch": 2 * fetch + write, "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
            "kernel": got["FETCH_SIZE"][0], "dispatches": got["FETCH_SIZE"][2], "seconds": time.perf_counter() - t0,
            "what": "2 x FETCH_SIZE + WRITE_SIZE of the main pass, two rocprofv3 --pmc child runs of this script in this run "
                    "(gfx950 tallies a wide coalesced read stream at half its bytes: MI355X_MICROARCH.md, HBM); requests at the "
                    "L2's memory side, Infinity-Cache hits included: fabric traffic, an upper bound of the HBM bytes"}


def main_kernel_name(args):
    """The main pass the options select (td_set_option "kernel": 4 -- the default -- k_fast4, 2 k_fast2, 1 k_fast)."""
    opts = dict(kv.split("=") for kv in (args.opt or []))
    if int(opts.get("fastpath", "1")) == 0:
        return "k_count"
    return {4: "k_fast4", 2: "k_fast2", 1: "k_fast"}.get(int(opts.get("kernel", "4")), "k_fast4")


def make_config(args, cid, reads=0):
    from tagdigger_amd.synth import CONFIGS, SynthConfig
    base = dict(CONFIGS[cid])
    if cid in (4, 5):
        base["nreads"] = 200_000_000                    # (one library; config 5's 1 B reads are five of them)
    if reads:
        base["nreads"] = reads
    if cid == args.config:
        if args.barcodes:
            base["nbar"] = args.barcodes
        if args.markers:
            base["nmarkers"] = args.markers
        if args.seed >= 0:
            base["seed"] = args.seed
        if args.cutsite:
            base["cutsite"] = args.cutsite
        if args.bclen_max:
            base["bclen"] = (4, args.bclen_max)
        if args.skew:
            base["skew"] = args.skew
    return SynthConfig(**base)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.debug_ablate or args.pmc_child:
        args.no_check = True
    default_workload = is_default_workload(args)
    # the counter passes come first: their children must be started by a process that has not initialised HIP
    traffic = None
    if world == 1 and default_workload and args.traffic == "auto" and not args.pmc_child:
        traffic = measure_traffic(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    # TD_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- a correctness rehearsal of the N > 1 code
    # path on a one-GPU box (never a measurement)
    rehearsal = os.environ.get("TD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    # TD_BENCH_FORCE_DIST=1: ONE rank goes through the collective path all the same -- backend nccl (RCCL) with a
    # communicator of one: what a one-GPU box can show of the N > 1 code (init, the in-place int32 all-reduce of the
    # bound matrix overlapped with the next pass, barrier); never a scaling measurement
    dist_on = world > 1 or os.environ.get("TD_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if rehearsal else "nccl"
        if world == 1:
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import tagdigger_amd

    cfg = make_config(args, args.config, args.reads)
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
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v, 0))
    if args.debug_ablate:
        eng.set_option("debug_ablate", args.debug_ablate)

    # The matrix lives in torch tensors so that RCCL can reduce it in place.  With several GPUs there are
    # two: the all-reduce of one pass (the path's one exchange: an integer sum over xGMI) runs on RCCL's
    # stream while the next pass counts into the other matrix.
    nmat = 2 if dist_on else 1
    mats = [torch.zeros(len(cfg.barcodes) * len(cfg.tags), dtype=torch.int32, device=dev) for _ in range(nmat)]
    reducing = [None] * nmat
    counts = mats[0]
    eng.bind_counts(counts.data_ptr())
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    stream = torch.cuda.current_stream().cuda_stream
    # this rank's share of the stream: weak = its own library (a file of its own: line 0), strong = a byte range of
    # ONE file, cut at line starts and counted with its true line index
    weak_reads = cfg.nreads
    lo, hi = cfg.nreads * rank // world, cfg.nreads * (rank + 1) // world
    fastq = torch.empty(max(weak_reads, hi - lo) * cfg.record_bytes, dtype=torch.uint8, device=dev)

    def fence():
        for b in range(nmat):
            if reducing[b] is not None:
                reducing[b].wait()
                reducing[b] = None
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def run_scaling(scaling):
        """warm-up, the check of what is timed, the timed region; returns this rank's view (rank 0's is printed)"""
        if scaling == "weak":
            my_reads, first_read, first_line, job_reads = weak_reads, rank * weak_reads, 0, weak_reads * world
        else:
            my_reads, first_read, first_line, job_reads = hi - lo, lo, 4 * lo, cfg.nreads
        nbytes = my_reads * cfg.record_bytes
        cfg.fill_device(eng, fastq.data_ptr(), first_read, my_reads)
        passes = [0]

        def step():
            b = passes[0] % nmat
            passes[0] += 1
            if reducing[b] is not None:       # this matrix's previous all-reduce (two passes ago)
                reducing[b].wait()
                reducing[b] = None
            mats[b].zero_()
            eng.bind_counts(mats[b].data_ptr())
            eng.count_device(fastq.data_ptr(), nbytes, first_line=first_line, stream=stream)
            if dist_on:
                reducing[b] = dist.all_reduce(mats[b], async_op=True)
            return mats[b]

        for _ in range(args.warmup):
            step()
        fence()

        # ---- correctness of what is being timed (rank-local shard, before any all-reduce)
        check = None
        if not args.no_check:
            # (1) the whole matrix against the one the generator's own choices imply (built on the device from
            # the shared spec include/td_synth_spec.h; nothing is parsed, nothing of oracle/ is involved)
            counts.zero_()
            eng.reset()
            eng.bind_counts(counts.data_ptr())
            eng.count_device(fastq.data_ptr(), nbytes, first_line=first_line, stream=stream)
            torch.cuda.synchronize()
            want = torch.zeros_like(counts)
            hits = cfg.expected_device(eng, want.data_ptr(), first_read, my_reads)
            st = eng.stats()
            ok = bool(torch.equal(counts, want)) and st["tag"] == hits and st["reads"] == my_reads
            check = {"bit_exact_vs_expected": ok, "reads": int(st["reads"]), "barcut": int(st["barcut"]), "tag": int(st["tag"]),
                     "what": "whole matrix + counters against the generator's expected matrix (product code sharing include/td_synth_spec.h "
                             "with the generator); the independent C-oracle parse is oracle_sample"}
            del want
            if not ok:
                print("bench.py: rank %d COUNT MISMATCH against the generator's expected matrix" % rank, file=sys.stderr)
                sys.exit(3)
            # (2) an independent checker on a sample: the first reads of the RESIDENT bytes, copied back and
            # counted by the C oracle, against a GPU pass over exactly that prefix
            if rank == 0 and args.oracle_sample > 0:
                n = min(args.oracle_sample, my_reads)
                counts.zero_()
                eng.reset()
                eng.count_device(fastq.data_ptr(), n * cfg.record_bytes, first_line=first_line, stream=stream)
                torch.cuda.synchronize()
                gst = eng.stats()
                sample = fastq[:n * cfg.record_bytes].cpu().numpy()
                okc, ost = oracle_check(cfg, sample, first_line, counts.cpu().numpy().view(np.uint32))
                okc = okc and (gst["reads"], gst["barcut"], gst["tag"]) == (ost["reads"], ost["barcut"], ost["tag"])
                check["oracle_sample"] = {"reads": n, "bit_exact_vs_c_oracle": bool(okc), "tag": int(ost["tag"])}
                del sample
                if not okc:
                    print("bench.py: GPU counts differ from the C oracle on the first %d reads" % n, file=sys.stderr)
                    sys.exit(3)
            counts.zero_()
            eng.reset()

        eng.set_option("timing", 1)
        eng.kernel_times_ms()                 # (drop what the check launched)
        fence()
        t0 = time.perf_counter()
        last = counts
        for _ in range(args.steps):
            last = step()
        fence()
        elapsed = time.perf_counter() - t0
        ktimes = eng.kernel_times_ms()
        kms = sum(ktimes) / len(ktimes) if ktimes else 0.0
        fixups = eng.debug_counters()[11]
        eng.set_option("timing", 0)

        if dist_on:
            t = torch.tensor([elapsed, kms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, kms = float(t[0]), float(t[1])
            # every rank now holds the same summed matrix: its total must equal the sum of all shards' hits
            tot = torch.tensor([int(last.to(torch.int64).sum())], dtype=torch.int64, device=dev)
            mine = torch.tensor([check["tag"] if check else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(mine)
            if check:
                check["reduced_total_equals_sum_of_shard_hits"] = int(tot[0]) == int(mine[0])
                if int(tot[0]) != int(mine[0]):
                    print("bench.py: all-reduced matrix total %d != sum of shard hits %d" % (int(tot[0]), int(mine[0])),
                          file=sys.stderr)
                    sys.exit(3)
        return {"scaling": scaling, "my_reads": my_reads, "nbytes": nbytes, "first_line": first_line, "job_reads": job_reads,
                "elapsed": elapsed, "kms": kms, "ktimes": ktimes, "fixups": fixups, "check": check, "step": step}

    def allreduce_alone():
        """the path's one exchange by itself: the int32 [barcodes x tags] matrix, in place, median of 5 (max over ranks)"""
        ts = []
        for _ in range(6):
            fence()
            t0 = time.perf_counter()
            dist.all_reduce(mats[0])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = torch.tensor([sorted(ts[1:])[2]], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mats[0].zero_()
        return float(t[0]) * 1e3

    main_run = run_scaling(args.scaling)
    strong_run = None
    if world > 1 and args.scaling == "weak" and not args.pmc_child:
        strong_run = run_scaling("strong")                 # (the metric's literal reading: ONE library over the N GPUs)
    allreduce_ms = allreduce_alone() if dist_on else None
    R = main_run
    my_reads, nbytes, first_line, kms, ktimes = R["my_reads"], R["nbytes"], R["first_line"], R["kms"], R["ktimes"]
    step = R["step"]

    if rank == 0 and not args.pmc_child:
        value = R["job_reads"] * args.steps / R["elapsed"]
        algo_bytes = cfg.record_bytes * my_reads
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        metric = "FASTQ reads/sec (whole node), 200M-read × 100k-tag synthetic, 1/2/4/8 MI355X"
        try:                                              # (verbatim from BASELINE.json when it is there)
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            pass
        if world == 1:
            sharding = "single GPU"
        elif args.scaling == "weak":
            sharding = "library-per-GPU + RCCL all-reduce(int32 count matrix) per pass, overlapped with the next pass"
        else:
            sharding = "one library byte-sharded over the GPUs (true first line index per shard) + RCCL all-reduce(int32 count matrix) per pass"
        out = {
            "metric": metric,
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": R["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d] shape, device-resident (tier T1): %d reads x %d barcodes x %d tags "
                                   "per GPU, 100 bp reads, %d B/record, cut site %s%s, %s"
                                   % (args.config - 1, my_reads, len(cfg.barcodes), len(cfg.tags), cfg.record_bytes, cfg.cutsite,
                                      ", Zipf %.2f hits" % args.skew if args.skew else "",
                                      "one library per GPU" if args.scaling == "weak" else "one library over all GPUs"),
                       "reads_per_gpu": my_reads, "barcodes": len(cfg.barcodes), "tags": len(cfg.tags),
                       "fastq_bytes_per_gpu": nbytes, "sharding": sharding},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["fabric_bytes_per_launch"] if traffic else None,
                         "traffic_kind": "fabric_bytes_per_launch", "traffic_detail": traffic,
                         "kernel": "tdk::%s main pass + k_resolve + k_fast fix-up pass (HIP events around the three)" % main_kernel_name(args), "kernel_ms": kms,
                         "kernel_ms_min": min(ktimes) if ktimes else None, "kernel_ms_max": max(ktimes) if ktimes else None,
                         "fixup_queue": R["fixups"], "kernel_launches": len(ktimes),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "check": R["check"],
        }
        def note(key):                                     # (stderr: what has been measured so far, should a later stage fail)
            print("bench.py: %s = %s" % (key, json.dumps(out.get(key))[:400]), file=sys.stderr)
            sys.stderr.flush()
        note("value")
        note("roofline")
        if dist_on:
            out["rccl_ranks"] = dist.get_world_size()
            out["collective_backend"] = backend + (" (RCCL over xGMI)" if backend == "nccl" else " (REHEARSAL on one GPU: not a measurement)")
            out["allreduce_ms"] = allreduce_ms
            out["allreduce_bytes"] = int(mats[0].numel()) * 4
        if strong_run is not None:
            S = strong_run
            out["strong"] = {"value": S["job_reads"] * args.steps / S["elapsed"], "unit": "reads/s",
                             "ms_per_step": S["elapsed"] / args.steps * 1e3, "library_reads": S["job_reads"],
                             "reads_per_gpu": S["my_reads"], "kernel_ms": S["kms"], "fixup_queue": S["fixups"], "check": S["check"],
                             "allreduce_ms": allreduce_ms,
                             "what": "ONE library byte-sharded over the GPUs, each shard counted with its true first line index, "
                                     "the same all-reduce per pass (overlapped with the next pass)"}
        if world == 1 and not args.debug_ablate:
            # the same pass with the reference's progress counters kept per window of 50 000 reads (find_tags_fastq's
            # default here, reference :268-271) -- untimed above, reported beside it
            # Passes with and without the counters ALTERNATE in one loop (after one untimed pass of the recording kernel, whose
            # first launch loads its code object): the box's clock drifts by a few per cent over a run, and both see the same drift.
            eng.set_option("timing", 1)
            eng.set_option("progress", 1)
            step()
            fence()
            eng.kernel_times_ms()
            pk, qk = [], []
            for _ in range(4):
                for prog, into in ((0, qk), (1, pk)):                      # (the same preparation before either kind of pass)
                    eng.set_option("progress", prog)
                    eng.reset()
                    step()
                    fence()
                    into += eng.kernel_times_ms()
            win = eng.progress_windows()
            pst = eng.stats()
            med = lambda v: sorted(v)[len(v) // 2] if v else None
            out["progress_windows"] = {"kernel_ms": med(pk), "plain_kernel_ms_same_loop": med(qk),
                                       "ratio": med(pk) / med(qk) if pk and qk else None, "windows": len(win),
                                       "sums_equal_counters": (sum(a for a, _ in win), sum(b for _, b in win)) == (pst["barcut"], pst["tag"]),
                                       "what": "medians of 4 + 4 alternating passes; the headline is measured without the counters"}
            eng.set_option("timing", 0)
            eng.set_option("progress", 0)
        if world == 1 and args.config == 5 and not args.debug_ablate:
            out["count_and_trim"] = count_and_trim(eng, cfg, fastq, nbytes, first_line, my_reads, stream)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, min(args.cpu_sample, my_reads), min(args.cpu_python_sample, my_reads))
            note("cpu_baseline")
        others = ("2,4,5" if default_workload else "") if args.other_configs == "auto" else args.other_configs
        if world == 1 and others and not args.debug_ablate:
            del fastq
            R = main_run = None
            eng.bind_counts(0)
            del mats, counts
            torch.cuda.empty_cache()
            out["other_configs"] = {}
            for cid in [int(c) for c in others.split(",") if c]:
                out["other_configs"]["c%d" % cid] = other_config(eng, args, cid, dev, stream)
                note("other_configs")
            fastq = None
        if world == 1 and args.tier_reads > 0 and not args.debug_ablate:
            fastq = None
            R = main_run = None
            torch.cuda.empty_cache()
            eng.bind_counts(0)
            out["tiers"] = tiers(eng, cfg, min(args.tier_reads, my_reads))
        print(json.dumps(out))
        sys.stdout.flush()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def other_config(eng, args, cid, dev, stream, steps=3):
    """One of the other BASELINE index shapes at its full read count: bit-exact check against the generator's
    expected matrix, then `steps` timed passes (HIP events around the kernels); config 5 also runs the trim branch."""
    import torch
    t_all = time.perf_counter()
    cfg = make_config(args, cid, args.other_reads)
    reads = cfg.nreads
    nbytes = reads * cfg.record_bytes
    fastq = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cfg.fill_device(eng, fastq.data_ptr(), 0, reads)
    counts = torch.zeros(len(cfg.barcodes) * len(cfg.tags), dtype=torch.int32, device=dev)
    eng.bind_counts(counts.data_ptr())
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_device(fastq.data_ptr(), nbytes, stream=stream)              # (warm; also the pass that is checked)
    torch.cuda.synchronize()
    want = torch.zeros_like(counts)
    hits = cfg.expected_device(eng, want.data_ptr(), 0, reads)
    st = eng.stats()
    ok = bool(torch.equal(counts, want)) and st["tag"] == hits and st["reads"] == reads
    del want
    eng.set_option("timing", 1)
    eng.kernel_times_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        counts.zero_()
        eng.count_device(fastq.data_ptr(), nbytes, stream=stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    kt = eng.kernel_times_ms()
    eng.set_option("timing", 0)
    kms = sum(kt) / len(kt) if kt else 0.0
    res = {"reads": reads, "barcodes": len(cfg.barcodes), "tags": len(cfg.tags), "cutsite": cfg.cutsite,
           "ms": kms, "ms_per_step": wall * 1e3, "reads_per_s": reads / wall,
           "frac": cfg.record_bytes * reads / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None,
           "bit_exact": ok, "tag": int(st["tag"]), "steps": steps}
    if cid == 5:
        eng.reset()
        res["count_and_trim"] = count_and_trim(eng, cfg, fastq, nbytes, 0, reads, stream)
    eng.bind_counts(0)
    del fastq, counts
    torch.cuda.empty_cache()
    res["seconds"] = time.perf_counter() - t_all
    return res


def count_and_trim(eng, cfg, fastq, nbytes, first_line, reads, stream):
    """BASELINE config 5's other half: counting AND the splitter's per-read branch (barcode, first full restriction
    site, adapter run-off: reference :1251-1283, :1328-1363) over the same resident buffer -- td_count_and_split_device,
    wall time of the call (count pass, line prefix, decisions; two synchronisations inside), three repetitions."""
    import contextlib
    import ctypes as C
    import io
    import torch
    from tagdigger_amd import tagdigger_fun as tf
    adapter = [tuple(x) for x in tf.adapters["PstI-MspI-Hall"]]
    with contextlib.redirect_stdout(io.StringIO()):
        ends = tf._adapter_ends(adapter, cfg.barcodes)
    eng.set_splitter(cfg.barcodes, cfg.cutsites[0], adapter[0][0].replace("^", ""), adapter[1][0].replace("^", ""), ends)
    cap = reads + 8
    d_out = torch.empty((cap, 2), dtype=torch.int32, device=fastq.device)
    terms = C.c_uint64(0)
    times = []
    for _ in range(4):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = eng._L.td_count_and_split_device(eng._h, C.c_void_p(fastq.data_ptr()), nbytes, first_line, 1 << 62,
                                              C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(stream) if stream else None, C.byref(terms))
        times.append(time.perf_counter() - t0)
        if rc:
            return {"error": rc}
    ms = sorted(times[1:])[1] * 1e3
    with_bar = int((d_out[:reads, 0] >= 0).sum())
    clipped = int(((d_out[:reads, 0] >= 0) & (d_out[:reads, 1] != 999)).sum())
    return {"ms": ms, "reads_per_s": reads / (ms * 1e-3), "terminators": int(terms.value), "with_barcode": with_bar, "clipped": clipped,
            "what": "count pass + line prefix + k_split2 decisions (8 B per read, left in HBM) over one resident buffer; median of 3"}


def oracle_check(cfg, sample, first_line, got_flat):
    """The checker: oracle/oracle.c over `sample` (numpy uint8, whole records) against the GPU's matrix."""
    import numpy as np
    from oracle import c_oracle
    ost = {}
    want = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite).count_bytes(sample, first_line=first_line, stats=ost)
    return bool((want.reshape(-1) == got_flat.astype(np.uint64)).all()), ost


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sample_reads, python_reads=0):
    """The reference CPU path, timed on this box's host cores on a bounded prefix of the same stream
    (host reference generator).  `value` is the pure-Python restatement (oracle/tagdigger_oracle.py: the
    reference's own nested-list trie and per-line loop, one interpreter thread -- the closest thing to
    the reference that can travel; BASELINE.md section 3 has its calibration against the real reference:
    it is 1.47x FASTER than the real thing on the build host).  `c_port` / `c_port_all_cores`: the
    scalar C restatement (oracle/oracle.c) on one core and on every core this process may use -- a fairer
    CPU line, labelled as not-the-reference."""
    from helpers import synth_host_bytes
    from oracle import c_oracle
    data = synth_host_bytes(cfg, 0, sample_reads)
    t0 = time.perf_counter()
    ora = c_oracle.COracle(cfg.barcodes, cfg.tags, cfg.cutsite)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ora.count_bytes(data)
    loop_s = time.perf_counter() - t0
    c_port = {"value": sample_reads / loop_s, "unit": "reads/s", "cores": 1,
              "sample": "first %d reads of the same synthetic stream (%.2f GB), record loop only; "
                        "trie build %.2f s timed separately; scalar C restatement oracle/oracle.c"
                        % (sample_reads, data.nbytes / 1e9, build_s),
              "index_build_s": build_s, "loop_s": loop_s}
    # kind: "restatement" = `value` is the pure-Python restatement of the reference's loop (the closest thing to the
    # reference that can travel); "port" when only the scalar C port was timed
    out = {"value": None, "unit": "reads/s", "cores": 1, "kind": "restatement" if python_reads > 0 else "port",
           "cpu": host_cpu_model(), "c_port": c_port}
    c_port["kind"] = "port"
    # the same C restatement on every core this process may use (at most 16: the GPU box's share per GPU),
    # one shard of whole records per thread, matrices summed
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
        out["c_port_all_cores"] = {"value": sample_reads / par_s, "unit": "reads/s", "cores": len(shards), "loop_s": par_s,
                                   "matrix_total": int(sum(int(m.sum()) for m in parts))}
        del oracles, mats, parts
    if python_reads > 0:
        from oracle import tagdigger_oracle as po
        pdata = bytes(data[:python_reads * cfg.record_bytes])
        t0 = time.perf_counter()
        index = po.prepare_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        pbuild = time.perf_counter() - t0
        t0 = time.perf_counter()
        po.count_bytes(pdata, cfg.barcodes, cfg.tags, cfg.cutsite, index=index)
        ploop = time.perf_counter() - t0
        out["value"] = python_reads / ploop
        out["sample"] = ("first %d reads of the same synthetic stream, record loop only (the nested-list trie build, %.1f s, "
                         "is timed separately); pure-Python restatement oracle/tagdigger_oracle.py, one interpreter thread"
                         % (python_reads, pbuild))
        out["index_build_s"] = pbuild
        out["loop_s"] = ploop
    else:                                 # (no Python sample asked for: the C port is the only line there is)
        out["value"] = c_port["value"]
        out["sample"] = c_port["sample"]
    return out


def tiers(eng, cfg, reads):
    """Tiers T2/T3 of SURVEY 8d on `reads` reads of the same stream and index -- inputs NOT resident in HBM:
    T2 host buffer -> pinned staging -> hipMemcpyAsync overlapped with counting; T3 from a file: plain,
    ordinary gzip (decoded on the device; and as before: chunk-parallel decode on the host, markers and CRC-32 on the GPU), BGZF (member-parallel inflate).  Each result is checked
    against the generator's expected matrix.  Never the bench `value`."""
    import gzip
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from compress_formats import bgzf_bytes, gzip_one_member      # (writers of the test files: no product code, no oracle)
    rb = cfg.record_bytes
    nb = reads * rb
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)               # (the other configs' passes left theirs)
    d = eng.dev_alloc(nb)
    cfg.fill_device(eng, d, 0, reads)
    host = np.frombuffer(eng.d2h(d, nb), dtype=np.uint8)
    eng.dev_free(d)
    cells = len(cfg.barcodes) * len(cfg.tags)

    def expected(n):
        dw = eng.dev_alloc(cells * 4)
        eng.h2d(dw, bytes(cells * 4))
        cfg.expected_device(eng, dw, 0, n)
        w = np.frombuffer(eng.d2h(dw, cells * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
        eng.dev_free(dw)
        return w
    want = expected(reads)
    out = {"reads": reads, "stage_threads": int(os.environ.get("TAGDIG_STAGE_THREADS", "16" if (os.cpu_count() or 1) >= 32 else "8")),
           "inflate_threads": os.environ.get("TAGDIG_INFLATE_THREADS", "default (host cores, at most %s)" % os.environ.get("TAGDIG_INFLATE_MAX", "16"))}

    def timed(fn, n, w):
        eng.reset()
        t0 = time.perf_counter()
        fn()
        eng.sync()
        dt = time.perf_counter() - t0
        ok = bool((eng.counts_numpy() == w).all())
        return {"reads_per_s": n / dt, "GB_per_s": n * rb / dt / 1e9, "bit_exact": ok}
    eng.count_bytes(host)                                           # (warm: pinned buffers, page faults)
    out["T2_host_buffer_pinned_h2d_overlap"] = timed(lambda: eng.count_bytes(host), reads, want)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR")) as tmp:
        plain = os.path.join(tmp, "tiers_lib.fq")
        with open(plain, "wb") as fh:
            fh.write(host)
        out["T3_plain_file_page_cache"] = timed(lambda: eng.count_file(plain), reads, want)
        os.unlink(plain)
        # ordinary gzip: ONE member, one DEFLATE stream (what gzip / pigz write; compressed here on threads the way
        # pigz does it, level 1)
        gzp = os.path.join(tmp, "tiers_lib.fq.gz")
        with open(gzp, "wb") as fh:
            fh.write(gzip_one_member(host, level=1, threads=16))
        # the whole of it on the device (csrc/gz_gpu.hpp: block search, Huffman decoding, LZ77, windows, CRC-32: only the
        # compressed bytes cross PCIe) ...
        eng.count_file(gzp)                                             # (warm: the decoder's buffers)
        r = timed(lambda: eng.count_file(gzp), reads, want)
        r["reads"] = reads
        r["gz_bytes"] = os.path.getsize(gzp)
        r["decoded_on_device"] = bool(eng.last_gz_route() == 1)
        out["T3_gzip_file_gpu_inflate"] = r
        # ... DEFLATE decoded into symbols by the host threads, markers -> bytes and the CRC-32 on the GPU (count_gzip_dev:
        # round 3's path, and the fallback of the one above); "gpu_resolve" 0: all of it on the host, as in rounds 1-2
        eng.set_option("gpu_huffman", 0)
        eng.count_file(gzp)                                             # (warm: the decoder's pinned buffers)
        r = timed(lambda: eng.count_file(gzp), reads, want)
        r["reads"] = reads
        r["gz_bytes"] = os.path.getsize(gzp)
        out["T3_gzip_file_host_decode_gpu_resolve"] = r
        eng.set_option("gpu_resolve", 0)
        r = timed(lambda: eng.count_file(gzp), reads, want)
        eng.set_option("gpu_resolve", 1)
        eng.set_option("gpu_huffman", 1)
        r["reads"] = reads
        out["T3_gzip_file_host_inflate"] = r
        os.unlink(gzp)
        # BGZF (bgzip's level 6), the whole sample twice over: a batch of the GPU inflater is 49 152 members (3 GB of FASTQ)
        bz = os.path.join(tmp, "tiers_lib.bgzf.fq.gz")
        one = bgzf_bytes(host.tobytes(), level=6, threads=16)[:-28]          # (without the end-of-file member)
        with open(bz, "wb") as fh:
            fh.write(one)
            fh.write(one)
            fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        w2 = want * 2
        for where, flag in (("gpu", 1), ("host", 0)):
            eng.set_option("gpu_inflate", flag)
            eng.count_file(bz)                                              # (warm: the inflater's buffers)
            r = timed(lambda: eng.count_file(bz), 2 * reads, w2)
            r["reads"] = 2 * reads
            r["gz_bytes"] = os.path.getsize(bz)
            out["T3_bgzf_file_%s_inflate" % where] = r
        eng.set_option("gpu_inflate", 1)
    return out


if __name__ == "__main__":
    main()
