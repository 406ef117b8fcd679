#!/usr/bin/env python3
"""bench.py -- samples/sec of the batched tree-ensemble traversal on MI355X.

One "step" = one tahoe_forest_predict over one resident batch (BASELINE.json config 3, "K3": synthetic
forest of 1000 trees of depth 12 over 256 features, 1M rows, float32).  Inputs and the forest are
resident in HBM before the timed region (as in the reference, BaseTahoeTest.h:563-573).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1, --shard rows (default): every rank holds the whole forest and its own 1M rows (rows are independent
    -> no data-path collective); value = N * rows / time; "scaling": "weak".
N > 1, --shard trees: the forest's trees are split across ranks, every rank sees the same 1M rows, partial
    float32 sums are combined by one RCCL all-reduce per step; value = rows / time; "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MISSING = -999.0


def algorithmic_bytes_per_sample(T, depth, cols, bits_bytes):
    """the reference's own traversal byte model (main.cu:38-46) made exact for perfect
    depth-`depth` trees: per (row, tree) `depth` internal visits of one node record (4 B threshold +
    b B packed bits) and one feature (4 B), plus the leaf record; plus the row once and the prediction."""
    return T * (depth * (4 + bits_bytes + 4) + (4 + bits_bytes)) + cols * 4 + 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--shard", choices=["rows", "trees"], default="rows")
    ap.add_argument("--strategy", type=int, default=0, help="0 auto, 1 direct, 2 rowtile, 3 tileblock, 4 tilering")
    ap.add_argument("--cpu-rows", type=int, default=100_000, help="rows of the batch timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-resident (PCIe-inclusive) leg")
    ap.add_argument("--no-tree-leg", action="store_true",
                    help="skip the secondary K4 leg (8000 trees split across the ranks + one RCCL all-reduce per batch)")
    ap.add_argument("--tree-leg-trees", type=int, default=8000)
    args = ap.parse_args()

    import torch

    import tahoe_amd as ta

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # TAHOE_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through host memory); the numbers of such a run mean nothing, the code path is the same.
    backend = os.environ.get("TAHOE_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if backend == "nccl":
            dist.all_reduce(t, **kw)  # RCCL over xGMI
        else:
            host = t.cpu()
            dist.all_reduce(host, **kw)
            t.copy_(host)

    T, D, C, R = args.trees, args.depth, args.cols, args.rows
    nodes = ta.synth_forest(T, D, C, seed=42)
    n_per_tree = ta.capi.tree_num_nodes(D)
    if world > 1 and args.shard == "trees":
        lo, hi = (T * rank) // world, (T * (rank + 1)) // world
        my_nodes, my_T, first_row = nodes[lo * n_per_tree: hi * n_per_tree], hi - lo, 0
    else:
        my_nodes, my_T, first_row = nodes, T, rank * R
    data = ta.synth_data(R, C, seed=43, first_row=first_row)
    x = torch.from_numpy(data).cuda()
    preds = torch.empty(R, dtype=torch.float32, device="cuda")
    forest = ta.Forest(my_nodes, my_T, D, C, missing=MISSING)
    forest.set_strategy(args.strategy)
    info = forest.info()
    stream = torch.cuda.current_stream()

    def step():
        if world > 1 and args.shard == "trees":
            # tahoe_amd/sharding.py, TreeShardedForest.predict: partial sums, one all-reduce, transform
            forest.predict_raw(x, preds, stream=stream)
            all_reduce(preds)  # 4 B/row
            ta.capi.transform_preds(preds, 0, T, 0.0, 0.0, stream=stream)
        else:
            forest.predict(x, preds, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    forest.set_profiling(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = forest.kernel_times_ms()
    prepass_ms = forest.prepass_times_ms()
    forest.set_profiling(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    total_rows = R * world if (world > 1 and args.shard == "rows") else R
    value = total_rows / (dt / args.steps)

    # ---- roofline of the dominant (traversal) kernel, per launch on this rank ----
    b_alg = algorithmic_bytes_per_sample(my_T, D, C, info.bits_bytes) * R
    k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
    achieved = b_alg / (k_ms * 1e-3) / 1e9
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed summary of
    # the separate rocprofv3 --pmc passes (tools/pmc.sh, same command, K3 only) supplies it.
    traffic = None
    strategy_name = ta.STRATEGY_NAMES.get(forest.get_strategy(R), "?")
    if (T, D, C, R) == (1000, 12, 256, 1_000_000) and world == 1:
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic.json")) as fh:
                entry = json.load(fh)["strategies"].get(strategy_name)
            if entry:
                traffic = int(sum((k["FETCH_SIZE_KB"] or 0) + (k["WRITE_SIZE_KB"] or 0) for k in entry.values()) * 1024)
        except (OSError, KeyError, ValueError):
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": strategy_name + "_kernel",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "traffic_unit": "bytes per launch (FETCH_SIZE+WRITE_SIZE, raw; profiles/r01/hbm_traffic.json)",
        "kernel_ms_avg": round(k_ms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4) if len(kernel_ms) else None,
        "prepass_kernel_ms_avg": round(float(np.mean(prepass_ms)), 4) if len(prepass_ms) else 0.0,
        "frac_incl_prepass": round(b_alg / ((k_ms + (float(np.mean(prepass_ms)) if len(prepass_ms) else 0.0)) * 1e-3) / 1e9
                                   / HBM_PEAK_GBPS, 4),
        "algorithmic_bytes_per_launch": b_alg,
        "compulsory_frac": round(((R * C * 4 + my_T * n_per_tree * (4 + info.bits_bytes) + R * 4) / (k_ms * 1e-3) / 1e9)
                                 / HBM_PEAK_GBPS, 5),
    }

    # ---- secondary leg: BASELINE config 4 ("K4"), the forest north_star shards by trees.  Every rank holds
    # trees [T4*rank/N, T4*(rank+1)/N) and the same 1M rows; per batch: partial float32 sums, ONE all-reduce of
    # 4 B/row over RCCL/xGMI, transform on the total.  Strong scaling; reported beside `value`, never as it. ----
    tree_leg = None
    if not args.no_tree_leg and not (world > 1 and args.shard == "trees") and (T, D, C, R) == (1000, 12, 256, 1_000_000):
        try:
            T4 = args.tree_leg_trees
            lo4, hi4 = (T4 * rank) // world, (T4 * (rank + 1)) // world
            # counter-based generator: tree t is the same on every rank; a rank keeps its slice
            nodes4 = ta.synth_forest(T4, D, C, seed=45)[lo4 * n_per_tree: hi4 * n_per_tree].copy()
            x4 = x if rank == 0 and first_row == 0 else torch.from_numpy(ta.synth_data(R, C, seed=43, first_row=0)).cuda()
            f4 = ta.Forest(nodes4, hi4 - lo4, D, C, missing=MISSING)
            f4.reserve(R)
            p4 = torch.empty(R, dtype=torch.float32, device="cuda")

            def step4():
                f4.predict_raw(x4, p4, stream=stream)
                if world > 1:
                    all_reduce(p4)
                ta.capi.transform_preds(p4, 0, T4, 0.0, 0.0, stream=stream)

            k4, w4 = max(3, min(args.steps, 10)), 2
            for _ in range(w4):
                step4()
            fence()
            t4 = time.perf_counter()
            for _ in range(k4):
                step4()
            fence()
            t4 = time.perf_counter() - t4
            if world > 1:
                tt = torch.tensor([t4], dtype=torch.float64, device="cuda")
                all_reduce(tt, op=dist.ReduceOp.MAX)
                t4 = float(tt.item())
            tree_leg = {"workload": f"K4: {T4} trees depth {D}, {C} features, {R} rows; trees split across {world} GPU(s), "
                                    f"one all-reduce of {4 * R} B per batch", "scaling": "strong",
                        "value": round(R / (t4 / k4), 1), "unit": "samples/s", "ms_per_step": round(t4 / k4 * 1e3, 4),
                        "steps": k4, "trees_per_gpu": hi4 - lo4, "tree_groups_per_gpu": f4.info().qring_groups,
                        "bit_exact": world == 1}
            f4.close()
            del x4, p4
        except Exception as err:  # the primary line must survive a failure of the secondary leg
            tree_leg = {"error": f"{type(err).__name__}: {err}"}

    # ---- host-resident batch (rank 0, N = 1 only): the PCIe-inclusive rate, reported beside `value`, never as it ----
    host_leg = None
    if rank == 0 and world == 1 and not args.no_host:
        fence()
        ref = preds.cpu().numpy()
        host_leg = {"unit": "samples/s", "note": "tahoe_forest_predict_host: rows start in host memory, chunked upload "
                    "overlapped with the traversal, predictions back in host memory; PCIe-inclusive, not `value`"}
        pin = ta.PinnedArray(R, C)
        pin.array[:] = data
        out_h = np.empty(R, dtype=np.float32)
        for name, src in (("pinned", pin.array), ("pageable", data)):
            forest.predict_host(src, out_h)  # creates the buffers / warms up
            reps = 3
            th = time.perf_counter()
            for _ in range(reps):
                forest.predict_host(src, out_h)
            th = (time.perf_counter() - th) / reps
            host_leg[name] = {"value": round(R / th, 1), "ms_per_batch": round(th * 1e3, 3),
                              "GBps_over_link": round(R * C * 4 / th / 1e9, 2),
                              "bitwise_equal_to_resident": bool(np.array_equal(out_h.view(np.uint32), ref.view(np.uint32)))}
            if not host_leg[name]["bitwise_equal_to_resident"]:
                raise SystemExit("bench: host-pipeline predictions differ from the resident-batch predictions")
        pin.close()

    # ---- CPU baseline + parity spot check (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle

        n_cpu = min(args.cpu_rows, R)
        tc = time.perf_counter()
        want, _ = oracle.predict(nodes, T, D, data[:n_cpu], MISSING, threads=1)
        cpu_s = time.perf_counter() - tc
        got = preds[:n_cpu].cpu().numpy()
        exact = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        ncores = os.cpu_count() or 1
        n_all = min(R, n_cpu * min(ncores, 16))
        tc = time.perf_counter()
        oracle.predict(nodes, T, D, data[:n_all], MISSING, threads=ncores)
        cpu_all_s = time.perf_counter() - tc
        cpu = {
            "value": round(n_cpu / cpu_s, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"first {n_cpu} rows of the batch, all {T} trees, single thread (the reference's "
                      f"predict_on_cpu is single-threaded)",
            "seconds": round(cpu_s, 2),
            "all_cores": {"value": round(n_all / cpu_all_s, 1), "cores": ncores, "rows": n_all,
                          "seconds": round(cpu_all_s, 2)},
            "gpu_matches_cpu_bitwise_on_sample": exact,
        }
        if not exact:
            raise SystemExit("bench: GPU sums differ from the CPU oracle on the sampled rows")

    if rank == 0:
        out = {
            "metric": "samples/sec, 1000-tree depth-12 forest @1M rows (batched tree-ensemble traversal)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "us_per_batch": round(ms_per_step * 1e3, 1),
            "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.shard == "trees") else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"K3: synthetic forest {T} trees depth {D}, {C} features, {R} rows"
                                   + (" per GPU" if world > 1 and args.shard == "rows" else ""),
                       "trees": T, "depth": D, "cols": C, "rows_per_step": total_rows,
                       "sharding": "none" if world == 1 else args.shard,
                       "strategy": ta.STRATEGY_NAMES.get(forest.get_strategy(R))},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "host_pipeline": host_leg,
            "tree_sharded_k4": tree_leg,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
