"""Sparse handles: every strategy timed on a grid of shapes against what TAHOE_STRATEGY_AUTO picks -> gpurun_out/selector_sparse.json"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta

def timeit(f, x, out, steps):
    for _ in range(2): f.predict_raw(x, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): f.predict_raw(x, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

res = []
for (T, C, R, dmin, dmax) in ((2000, 256, 200_000, 4, 24), (2000, 256, 20_000, 4, 24), (2000, 256, 2_000, 4, 24), (200, 64, 50_000, 4, 24),
                              (200, 64, 3_000, 4, 16), (50, 256, 100_000, 4, 12), (150, 256, 100_000, 4, 12), (500, 18, 10_000, 3, 10),
                              (1000, 100, 500_000, 6, 14), (30, 32, 1_000_000, 4, 20)):
    sn, tr = ta.capi.synth_sparse_forest(T, C, dmin, dmax, 0.32, 65535, 44)
    x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
    out = torch.empty(R, dtype=torch.float32, device="cuda")
    f = ta.capi.SparseForest(sn, tr, C, missing=-999.0)
    auto = ta.STRATEGY_NAMES[f.get_strategy(R)]
    per = {}
    for s in (ta.STRATEGY_TILEBLOCK, ta.STRATEGY_QRING, ta.STRATEGY_ROWTILE):
        try:
            f.set_strategy(s)
        except ta.TahoeError:
            continue
        per[ta.STRATEGY_NAMES[f.get_strategy(R)]] = round(timeit(f, x, out, 5 if R * T > 1e8 else 20), 4)
    best = min(per, key=per.get)
    res.append({"trees": T, "cols": C, "rows": R, "depth": [dmin, dmax], "auto": auto, "best": best, "ms": per,
                "auto_over_best": round(per[auto] / per[best], 3)})
    print(res[-1], flush=True)
    f.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/selector_sparse.json", "w"), indent=1)
