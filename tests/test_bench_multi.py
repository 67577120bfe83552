"""bench.py's N > 1 path: `python bench.py --gpus N` from a bare shell must start its own ranks (before any
GPU call), relay rank 0's line and report both scalings, the all-reduce's time and the rank count.

The result it stands for is the reference's combineReadCounts (tagdigger_fun.py:1061-1098): matrices of
libraries (or of one library's byte shards) summed cell by cell == ONE integer all-reduce.

CPU: the launcher's command and environment (no GPU is touched).  GPU box (one card): the whole N = 2 bench in
rehearsal mode (TD_BENCH_REHEARSAL=1: both ranks on GPU 0 over gloo), both scalings, with the line's checks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _import_bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_self_launch_command(monkeypatch):
    """--gpus 4 without WORLD_SIZE: torch.distributed.run with 4 ranks on 127.0.0.1, the same arguments, IPC mode set;
    this process imports neither torch nor the library first."""
    bench = _import_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7                                   # the launcher's exit code is the bench's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_exits_2(monkeypatch):
    bench = _import_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 2


def test_default_workload_flag():
    bench = _import_bench()
    assert bench.is_default_workload(bench.parse_args([]))
    assert bench.is_default_workload(bench.parse_args(["--steps", "20", "--warmup", "3"]))
    for extra in (["--reads", "1000"], ["--config", "4"], ["--skew", "1.5"], ["--tile-kb", "16"], ["--opt", "run=4"]):
        assert not bench.is_default_workload(bench.parse_args(extra))


def _run_bench(extra, timeout=900, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra if env_extra is not None else {"TD_BENCH_REHEARSAL": "1"})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, BENCH] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_weak_and_strong_rehearsal():
    out = _run_bench(["--gpus", "2", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--oracle-sample", "200000"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["check"]["bit_exact_vs_expected"] and out["check"]["oracle_sample"]["bit_exact_vs_c_oracle"]
    assert out["check"]["reduced_total_equals_sum_of_shard_hits"]
    assert out["config"]["reads_per_gpu"] == 2000000
    assert out["value"] > 0 and out["allreduce_ms"] > 0
    s = out["strong"]
    assert s["library_reads"] == 2000000 and s["reads_per_gpu"] == 1000000
    assert s["check"]["bit_exact_vs_expected"] and s["check"]["reduced_total_equals_sum_of_shard_hits"]
    assert s["value"] > 0
    assert "REHEARSAL" in out["collective_backend"]


@pytest.mark.gpu
def test_bench_two_ranks_strong_only_rehearsal():
    out = _run_bench(["--gpus", "2", "--scaling", "strong", "--reads", "1000001", "--steps", "2", "--warmup", "1",
                      "--oracle-sample", "100000"])
    assert out["scaling"] == "strong" and "strong" not in out
    assert out["config"]["reads_per_gpu"] == 500000            # rank 0's shard of 1 000 001 reads
    assert out["check"]["bit_exact_vs_expected"] and out["check"]["reduced_total_equals_sum_of_shard_hits"]


@pytest.mark.gpu
def test_bench_single_gpu_line_small():
    """the N = 1 line at a small read count: other configs (with count + trim for config 5), cpu_baseline labels"""
    out = _run_bench(["--reads", "1000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000", "--cpu-python-sample", "20000",
                      "--tier-reads", "0", "--oracle-sample", "100000", "--traffic", "off", "--other-configs", "2,5", "--other-reads", "500000"])
    assert out["n_gpus"] == 1 and out["check"]["bit_exact_vs_expected"]
    assert out["roofline"]["traffic"] is None and out["roofline"]["traffic_kind"] == "fabric_bytes_per_launch"
    assert out["cpu_baseline"]["kind"] == "restatement" and out["cpu_baseline"]["c_port"]["kind"] == "port"
    oc = out["other_configs"]
    assert oc["c2"]["bit_exact"] and oc["c2"]["barcodes"] == 96 and oc["c2"]["tags"] == 10000
    assert oc["c5"]["bit_exact"] and oc["c5"]["cutsite"] == "CWGC" and oc["c5"]["count_and_trim"]["ms"] > 0


@pytest.mark.gpu
def test_bench_one_rank_through_rccl():
    """What a one-GPU box can show of the nccl path: TD_BENCH_FORCE_DIST=1 sends the single rank through
    init_process_group("nccl") -- RCCL with a communicator of one --, the in-place int32 all-reduce of the bound matrix
    overlapped with the next pass, the barriers and the stand-alone all-reduce timing.  (Several ranks need several GPUs:
    RCCL refuses two ranks on one device; the two-rank runs above go over gloo.)"""
    out = _run_bench(["--reads", "2000000", "--steps", "3", "--warmup", "1", "--oracle-sample", "200000", "--cpu-sample", "0",
                      "--tier-reads", "0", "--traffic", "off", "--other-configs", ""], env_extra={"TD_BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    assert out["collective_backend"].startswith("nccl")
    assert out["allreduce_ms"] > 0 and out["allreduce_bytes"] == 384 * 100000 * 4
    assert out["check"]["bit_exact_vs_expected"] and out["check"]["reduced_total_equals_sum_of_shard_hits"]
