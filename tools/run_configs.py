#!/usr/bin/env python3
"""Runs BASELINE.json's five configurations once on cuda:0: parity against the CPU oracle (all rows for the small
ones, a strided sample for the 1M-row ones) and kernel timing.  Writes gpurun_out/configs.json (copy the result to
profiles/).  K3 is what bench.py measures; the others are parity cases with a timing for orientation."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tahoe_amd as ta  # noqa: E402
from oracle import oracle  # noqa: E402

import bench  # noqa: E402  (the byte model, the counter summaries and the workloads are bench.py's)

MISSING = bench.MISSING


def roofline(ms, rows, cols, b, trees, len_sum_per_row, node_bytes, n_nodes, cfg, leaf_bytes=None):
    out = bench.config_roofline(ms, rows, cols, trees, len_sum_per_row, node_bytes, n_nodes, cfg, leaf_bytes)
    out["bits_bytes"] = b
    return out


path_len_sum = bench.dense_path_len_sum


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def time_predict(forest, x, steps=5, warmup=2):
    out = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
    for _ in range(warmup):
        forest.predict_raw(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        forest.predict_raw(x, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def check_dense(forest, nodes, T, D, data, x, sample):
    want, want_leaf = oracle.predict(nodes, T, D, data[sample], MISSING, want_leaf=True, threads=16)
    leaf, sums = forest.predict_leaf_idx(x[torch.from_numpy(sample).cuda()].contiguous())
    forest.check()
    return bool(np.array_equal(bits(leaf.cpu().numpy()), want_leaf) and np.array_equal(bits(sums.cpu().numpy()), bits(want)))


def main():
    res = {}
    # K1: SUSY-like through the text formats (C=18, 500 trees, depth 8, 10K rows)
    T, D, C, R = 500, 8, 18, 10_000
    nodes = ta.synth_forest(T, D, C, seed=11, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=12, missing_prob=0.02, missing=MISSING)
    with tempfile.TemporaryDirectory() as d:
        ta.write_model(os.path.join(d, "m.txt"), nodes, T, D)
        ta.write_data(os.path.join(d, "d.txt"), data, MISSING)
        n2, T2, D2 = ta.load_model(os.path.join(d, "m.txt"))
        x2, miss = ta.load_data(os.path.join(d, "d.txt"))
    assert n2.tobytes() == nodes.tobytes() and x2.tobytes() == data.tobytes()
    f = ta.Forest(n2, T2, D2, C, missing=miss)
    x = torch.from_numpy(x2).cuda()
    ms, _ = time_predict(f, x)
    _, leaf1 = oracle.predict(n2, T2, D2, x2, miss, want_leaf=True, threads=16)
    res["K1"] = {"shape": [T, D, C, R], "strategy": ta.STRATEGY_NAMES[f.get_strategy(R)], "ms": round(ms, 4),
                 "samples_per_s": round(R / ms * 1e3), "parity_all_rows": check_dense(f, n2, T2, D2, x2, x, np.arange(R)),
                 "roofline": roofline(ms, R, C, f.info().bits_bytes, T, path_len_sum(leaf1), 4 + f.info().bits_bytes, T * ta.capi.tree_num_nodes(D), "K1")}
    # K2: SVHN-like width (C=3072, 500 trees, depth 8, 100K rows)
    T, D, C, R = 500, 8, 3072, 100_000
    nodes = ta.synth_forest(T, D, C, seed=21)
    data = ta.synth_data(R, C, seed=22)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    x = torch.from_numpy(data).cuda()
    ms, _ = time_predict(f, x, steps=3, warmup=1)
    res["K2"] = {"shape": [T, D, C, R], "strategy": ta.STRATEGY_NAMES[f.get_strategy(R)], "ms": round(ms, 3),
                 "samples_per_s": round(R / ms * 1e3), "parity_sample_2000": check_dense(f, nodes, T, D, data, x, np.arange(0, R, 50)),
                 "roofline": roofline(ms, R, C, f.info().bits_bytes, T, float(T * D), 4 + f.info().bits_bytes, T * ta.capi.tree_num_nodes(D), "K2")}
    del x, data
    # K3 / K4: 1M rows x 256 features; K4 = 8000 trees, also as 8 tree shards summed in rank order on this one GPU
    C, R, D = 256, 1_000_000, 12
    data = ta.synth_data(R, C, seed=43)
    x = torch.from_numpy(data).cuda()
    sample = np.unique(np.concatenate([np.arange(0, R, 2003), [R - 1]]))
    for name, T in (("K3", 1000), ("K4", 8000)):
        nodes = ta.synth_forest(T, D, C, seed=42)
        tc = time.perf_counter()
        f = ta.Forest(nodes, T, D, C, missing=MISSING)
        create_s = time.perf_counter() - tc
        ms, full = time_predict(f, x, steps=10, warmup=3)
        entry = {"shape": [T, D, C, R], "strategy": ta.STRATEGY_NAMES[f.get_strategy(R)], "ms": round(ms, 3),
                 "create_s": round(create_s, 2), "device_MB": round(f.info().device_bytes / 1e6, 1),
                 "samples_per_s": round(R / ms * 1e3), "parity_sample_%d" % sample.size: check_dense(f, nodes, T, D, data, x, sample),
                 "roofline": roofline(ms, R, C, f.info().bits_bytes, T, float(T * D), 4 + f.info().bits_bytes, T * ta.capi.tree_num_nodes(D), name)}
        if name == "K4":
            per = ta.capi.tree_num_nodes(D)
            from tahoe_amd import sharding
            total = torch.zeros(R, dtype=torch.float64, device="cuda")   # "allreduce64": exact sum of the float32 partials
            chain = torch.zeros(R, dtype=torch.float32, device="cuda")   # "chain": running float32 sums, shard to shard
            shard_ms = []
            for k in range(8):
                lo, hi = T * k // 8, T * (k + 1) // 8
                fs = ta.Forest(nodes[lo * per: hi * per], hi - lo, D, C, missing=MISSING)
                m, part = time_predict(fs, x, steps=2, warmup=1)
                shard_ms.append(m)
                total += part.double()
                fs.predict_accumulate(x, chain)
                fs.check()
                fs.close()
            exact = oracle.predict_f64(nodes, T, D, data[sample], MISSING)
            got = total.float().cpu().numpy()[sample]  # one rounding
            one = full.cpu().numpy()[sample]
            bound = sharding.sum_error_bound(oracle.abs_leaf_sum(nodes, T, D, data[sample], MISSING), exact, T // 8)
            entry["tree_shards_8"] = {"ms_per_shard_avg": round(float(np.mean(shard_ms)), 3),
                                      "allreduce64_max_abs_err_vs_f64": float(np.max(np.abs(got - exact))),
                                      "allreduce64_within_bound": bool(np.all(np.abs(got - exact) <= bound)),
                                      "one_gpu_f32_max_abs_err_vs_f64": float(np.max(np.abs(one - exact))),
                                      "allreduce64_max_rel_diff_vs_1gpu_f32": float(np.max(np.abs(got - one) / np.maximum(np.abs(one), 1e-30))),
                                      "chain_bit_equal_to_1gpu_all_rows": bool(torch.equal(chain.view(torch.int32), full.view(torch.int32)))}
        res[name] = entry
        f.close()
    del x
    # K5: irregular sparse forest (2000 trees, depth 4..24), 200K rows
    T, C, R = 2000, 256, 200_000
    sn, tr = ta.capi.synth_sparse_forest(T, C, 4, 24, 0.32, 65535, 44)
    f = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
    x = torch.from_numpy(data[:R].copy()).cuda()
    per_strategy = {}
    for sid in (ta.STRATEGY_DIRECT, ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK):
        try:
            f.set_strategy(sid)
        except ta.TahoeError:
            continue
        per_strategy[ta.STRATEGY_NAMES[sid]] = round(time_predict(f, x, steps=2, warmup=1)[0], 3)
    f.set_strategy(ta.STRATEGY_AUTO)
    ms, _ = time_predict(f, x, steps=2, warmup=1)
    sample = np.arange(0, R, 400)
    want, want_leaf = oracle.sparse_predict(sn, tr, data[:R][sample], MISSING, want_leaf=True, threads=16)
    leaf, sums = f.predict_leaf_idx(x[torch.from_numpy(sample).cuda()].contiguous())
    f.check()
    sizes = np.diff(np.append(tr, sn.size))
    len5 = bench.sparse_path_len_sum(sn, tr, want_leaf)
    res["K5"] = {"shape": {"trees": T, "cols": C, "rows": R, "nodes": int(sn.size), "nodes_per_tree_mean": float(sizes.mean()),
                           "nodes_per_tree_max": int(sizes.max())},
                 "strategy": "sparse_" + ta.STRATEGY_NAMES[f.get_strategy(R)], "ms": round(ms, 3), "ms_per_strategy": per_strategy, "samples_per_s": round(R / ms * 1e3),
                 "parity_sample_%d" % sample.size: bool(np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
                                                         and np.array_equal(bits(sums.cpu().numpy()), bits(want))),
                 "roofline": dict(roofline(ms, R, C, 4, T, len5, 12, int(sn.size), "K5"),
                                  note="sparse_node_t is 12 bytes AoS (Struct.h:50-54); path lengths from the oracle's leaf nodes on the row sample")}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
