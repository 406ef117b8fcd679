"""Sparse forests of K5's shape (2000 trees, 256 features, 200 k rows) with the depth range cut at 9 / 12 / 24: what the walk
below the 9-level LDS top costs the quantised sparse kernel.  python tools/k5_depths.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
T, C, R = 2000, 256, 200_000
x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
out = torch.empty(R, dtype=torch.float32, device="cuda")
for max_depth in (9, 12, 24):
    sn, tr = ta.capi.synth_sparse_forest(T, C, 4, max_depth, 0.32, 65535, 44)
    f = ta.capi.SparseForest(sn, tr, C, missing=-999.0)
    for _ in range(2): f.predict_raw(x, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f.predict_raw(x, out)
    torch.cuda.synchronize()
    print("depth 4..%d" % max_depth, "nodes/tree %.0f" % (sn.size / T), ta.STRATEGY_NAMES[f.get_strategy(R)], round((time.perf_counter() - t) / 5 * 1e3, 3), "ms")
    f.close()
