"""K4 (8000 trees, depth 12, 256 features) on one GPU: 1 M rows, and the 125 k rows one of 8 row shards walks.  python tools/k4_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
T, D, C = 8000, 12, 256
nodes = ta.synth_forest(T, D, C, seed=42)
f = ta.Forest(nodes, T, D, C, missing=-999.0)
print("groups", f.info().qring_groups if hasattr(f.info(), "qring_groups") else "?")
for R in (1_000_000, 125_000):
    x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
    out = torch.empty(R, dtype=torch.float32, device="cuda")
    for _ in range(2): f.predict_raw(x, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f.predict_raw(x, out)
    torch.cuda.synchronize()
    print("K4", R, "rows", round((time.perf_counter() - t) / 5 * 1e3, 3), "ms")
