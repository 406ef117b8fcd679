"""K2 (500 trees of depth 8, 3072 features, 100 k rows): AUTO, the quantised wide form and the float32 wide form.  python tools/k2_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
T, D, C, R = 500, 8, 3072, 100_000
nodes = ta.synth_forest(T, D, C, seed=21)
x = torch.from_numpy(ta.synth_data(R, C, seed=22)).cuda()
f = ta.Forest(nodes, T, D, C, missing=-999.0)
out = torch.empty(R, dtype=torch.float32, device="cuda")
ref = None
for s in (ta.STRATEGY_AUTO, ta.STRATEGY_QRING, ta.STRATEGY_TILERING):
    try:
        f.set_strategy(s)
    except ta.TahoeError as e:
        print(ta.STRATEGY_NAMES.get(s, s), "unavailable")
        continue
    for _ in range(3): f.predict_raw(x, out)
    f.check()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f.predict_raw(x, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 10 * 1e3
    same = True if ref is None else bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
    ref = out.clone() if ref is None else ref
    print("K2", "auto" if s == ta.STRATEGY_AUTO else "", ta.STRATEGY_NAMES[f.get_strategy(R)], round(ms, 3), "ms, tile rows", f.info().ring_rows, f.info().qring_tile_rows, "same bits:", same)
