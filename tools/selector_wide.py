"""Rows too wide for a 64-row float32 tile (num_cols > 512): the quantised wide form (QRING) against the float32 wide form
(TILERING: its tile form, and its row-streaming form on 16-bit keys where the shape allows it) and what TAHOE_STRATEGY_AUTO
and the create-time form rule pick -> gpurun_out/selector_wide.json"""
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
for (T, D, C, R) in ((500, 8, 3072, 100_000), (2000, 8, 3072, 100_000), (4000, 8, 3072, 50_000), (500, 10, 1024, 100_000), (2000, 10, 1024, 100_000),
                     (200, 6, 768, 200_000), (1000, 12, 768, 100_000), (100, 8, 2048, 50_000), (4000, 8, 1536, 50_000), (300, 9, 600, 300_000),
                     (250, 8, 3072, 100_000), (1000, 6, 2048, 100_000), (500, 7, 1536, 100_000), (60, 5, 1024, 200_000)):
    nodes = ta.synth_forest(T, D, C, seed=5)
    x = torch.from_numpy(ta.synth_data(R, C, seed=6)).cuda()
    out = torch.empty(R, dtype=torch.float32, device="cuda")
    os.environ.pop("TAHOE_WSTREAM", None)
    f = ta.Forest(nodes, T, D, C, missing=-999.0)
    auto = ta.STRATEGY_NAMES[f.get_strategy(R)]
    streams = f.info().stream_slots > 0  # the create-time rule took the row-streaming form for TILERING
    per = {}
    for s in (ta.STRATEGY_QRING, ta.STRATEGY_TILERING):
        try:
            f.set_strategy(s)
        except ta.TahoeError:
            continue
        per[ta.STRATEGY_NAMES[s]] = round(timeit(f, x, out, 10), 4)
    # the other form of TILERING on a second handle
    os.environ["TAHOE_WSTREAM"] = "0" if streams else "1"
    g = ta.Forest(nodes, T, D, C, missing=-999.0)
    other = None
    if g.info().stream_slots > 0 or streams:
        g.set_strategy(ta.STRATEGY_TILERING)
        other = round(timeit(g, x, out, 10), 4)
    g.close()
    os.environ.pop("TAHOE_WSTREAM", None)
    forms = {"stream": per["tilering"] if streams else other, "tile": other if streams else per["tilering"]}
    best = min(per, key=per.get)
    info = f.info()
    res.append({"trees": T, "depth": D, "cols": C, "rows": R, "work_ratio": round(2 * T * D / (13 * C), 3), "auto": auto, "best": best, "ms": per,
                "auto_over_best": round(per[auto] / per[best], 3), "float_tile_rows": info.ring_rows, "u16_tile_rows": info.qring_tile_rows,
                "tilering_forms_ms": forms, "tilering_form_taken": "stream" if streams else "tile",
                "stream_levels": info.stream_levels, "stream_slots": info.stream_slots})
    print(res[-1], flush=True)
    f.close()
    del x
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/selector_wide.json", "w"), indent=1)
