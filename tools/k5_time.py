"""K5 timing only (sparse forest, 2000 trees depth 4..24, 200 k rows x 256 cols): python tools/k5_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
T, C, R = 2000, 256, 200_000
sn, tr = ta.capi.synth_sparse_forest(T, C, 4, 24, 0.32, 65535, 44)
x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
out = torch.empty(R, dtype=torch.float32, device="cuda")
f = ta.capi.SparseForest(sn, tr, C, missing=-999.0)
for _ in range(2): f.predict_raw(x, out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): f.predict_raw(x, out)
torch.cuda.synchronize()
print("K5", ta.STRATEGY_NAMES[f.get_strategy(R)], round((time.perf_counter() - t) / 5 * 1e3, 3), "ms")
