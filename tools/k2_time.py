import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import tahoe_amd as ta
T, D, C, R = 500, 8, 3072, 100_000
nodes = ta.synth_forest(T, D, C, seed=21)
x = torch.from_numpy(ta.synth_data(R, C, seed=22)).cuda()
f = ta.Forest(nodes, T, D, C, missing=-999.0)
out = torch.empty(R, dtype=torch.float32, device="cuda")
for _ in range(3): f.predict_raw(x, out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): f.predict_raw(x, out)
torch.cuda.synchronize()
print("K2", ta.STRATEGY_NAMES[f.get_strategy(R)], round((time.perf_counter() - t) / 10 * 1e3, 3), "ms", f.info().qring_tile_rows)
