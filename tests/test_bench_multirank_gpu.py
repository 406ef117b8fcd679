"""bench.py's N > 1 code path as the driver launches it (python -m torch.distributed.run, one rank per process), rehearsed
with two ranks on the one GPU of the test box: TAHOE_BENCH_BACKEND=gloo makes the ranks share device 0 and sends the
collectives through host memory -- the numbers mean nothing, the code path (rank/row bookkeeping, the self-check of the
process group, the barriers, max over ranks, the K4 legs with their collectives) is the one the 8-GPU run takes with RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(extra, ranks=2, timeout=600, cpu=False):
    env = dict(os.environ, TAHOE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--rows", "40000", "--steps", "2",
           "--warmup", "1", "--no-host"] + ([] if cpu else ["--no-cpu"]) + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-3000:]  # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_two_ranks_row_shards_and_k4_legs():
    line = run_bench(["--k4", "--k4-trees", "400", "--k4-sample", "512"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    c = line["collective"]
    assert c["ranks_seen"] == 2 and c["allreduce_ones_ok"] is True and c["allreduce_ones_sum"] == 2.0 and len(c["device_of_rank"]) == 2
    assert c["ms_per_step_fastest_rank"] <= c["ms_per_step_slowest_rank"] == line["ms_per_step"]
    cfg = line["config"]
    assert cfg["rows_per_gpu"] == 20000 and cfg["rows_per_step"] == 40000 and cfg["sharding"] == "rows"
    assert abs(line["value"] - 40000 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    k4 = line["k4"]
    assert "error" not in k4, k4
    assert k4["row_sharded"]["bit_identical_to_cpu_f32"] is True and "error" not in k4["row_sharded"]
    assert k4["row_sharded"]["rows_per_gpu"] == 20000
    ar = k4["tree_sharded_allreduce64"]
    assert ar["within_stated_bound"] is True and ar["trees_per_gpu"] == 200 and "error" not in ar


def test_two_ranks_without_k4_flag_print_no_k4_legs():
    line = run_bench([])
    assert line["k4"] is None and line["collective"]["ranks_seen"] == 2


def test_two_rank_line_carries_a_cpu_baseline():
    """Every line of a scaling run has its own cpu_baseline (rank 0, a short single-thread leg on its shard's first rows, with
    the bitwise check of the timed buffer) and a roofline object -- a line without both counts as unmeasured."""
    line = run_bench(["--cpu-rows-multi", "3000"], cpu=True)
    c = line["cpu_baseline"]
    assert c is not None and c["cores"] == 1 and c["kind"] == "port" and c["value"] > 0 and c["unit"] == "samples/s"
    assert c["gpu_matches_cpu_bitwise_on_sample"] is True and "3000 rows" in c["sample"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert line["n_gpus"] == 2 and line["configs"] is None  # the per-configuration legs are N = 1 only


def test_one_rank_process_group_on_the_real_backend():
    """TAHOE_BENCH_FORCE_DIST=1: bench.py as a fresh child with a ONE-rank process group on the real backend --
    init_process_group("nccl", world_size=1, rank=0, device_id=...), then the same fence (barrier), max-over-ranks (all-reduce
    MAX / MIN of float64), the all-reduce of ones and the agree() flag as at N > 1 -- so this file's torch.distributed surface
    has met RCCL on hardware before the driver's 8-GPU scaling run."""
    env = dict(os.environ, TAHOE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.pop("TAHOE_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "40000", "--steps", "3", "--warmup", "1", "--no-host",
                        "--no-configs", "--k4", "--k4-trees", "400", "--k4-sample", "512", "--cpu-rows", "3000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    c = line["collective"]
    assert c is not None and c["backend"] == "nccl (RCCL)" and c["ranks_seen"] == 1 and c["allreduce_ones_ok"] is True
    assert c["allreduce_ones_sum"] == 1.0 and c["device_of_rank"] == [0]
    assert c["ms_per_step_fastest_rank"] == c["ms_per_step_slowest_rank"] == line["ms_per_step"]
    assert line["n_gpus"] == 1 and line["cpu_baseline"]["gpu_matches_cpu_bitwise_on_sample"] is True


def test_two_ranks_tree_shards_chained_bit_exact():
    """--shard trees --tree-mode chain: rank 1 continues rank 0's running float32 sums chunk by chunk; the primary line's
    K4 chain leg must be bit-identical to the CPU's sequential sum."""
    line = run_bench(["--shard", "trees", "--tree-mode", "chain", "--chunk-rows", "8192"])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "trees/chain" and line["config"]["rows_per_gpu"] == 40000
    assert line["collective"]["allreduce_ones_ok"] is True
    line = run_bench(["--k4", "--k4-chain", "--k4-trees", "400", "--k4-sample", "512"])
    ch = line["k4"]["tree_sharded_chain"]
    assert ch["bit_identical_to_cpu_f32"] is True and "error" not in ch


def test_one_rank_line_carries_the_same_scaling_label():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "40000", "--steps", "2", "--warmup", "1", "--no-cpu",
                        "--no-host", "--no-k4"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 1 and line["collective"] is None
