"""Rows too wide for a 64-row float32 tile (num_cols > 512): the quantised wide form (QRING) against the float32 wide form
(TILERING: its tile form, and its row-streaming form on 16-bit keys where the shape allows it) and what TAHOE_STRATEGY_AUTO
and the create-time form rule pick -> gpurun_out/selector_wide.json (stamped with the hash of the kernel sources).  The first 14
shapes are the grid the form rule (3 x trees <= cols, every level above the bottom blocks resident) was fitted on; the HOLDOUT
shapes were not used for it; the last ones come from the histogram-style generator with features on very different scales, where
the key-resolution estimate must send TILERING to its tile form."""
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

import bench

res = []
FITTED = ((500, 8, 3072, 100_000), (2000, 8, 3072, 100_000), (4000, 8, 3072, 50_000), (500, 10, 1024, 100_000), (2000, 10, 1024, 100_000),
                     (200, 6, 768, 200_000), (1000, 12, 768, 100_000), (100, 8, 2048, 50_000), (4000, 8, 1536, 50_000), (300, 9, 600, 300_000),
                     (250, 8, 3072, 100_000), (1000, 6, 2048, 100_000), (500, 7, 1536, 100_000), (60, 5, 1024, 200_000))
HOLDOUT = ((350, 8, 2560, 100_000), (150, 7, 640, 200_000), (700, 9, 2304, 60_000), (90, 8, 1280, 150_000), (600, 8, 1792, 100_000),
           (1000, 8, 3072, 60_000))
SCALES = ((300, 8, 1024, 100_000), (500, 8, 3072, 50_000))  # histogram-style generator, five decades of feature scales
for kind, (T, D, C, R) in [("fitted", s) for s in FITTED] + [("holdout", s) for s in HOLDOUT] + [("scales", s) for s in SCALES]:
    if kind == "scales":
        nodes = ta.synth_forest_hist(T, D, C, seed=5, feature_seed=13, max_bins=255, zipf_s=0.5, leaf_prob=0.0, scale_decades=5.0)
        x = torch.from_numpy(ta.synth_data_hist(R, C, seed=6, feature_seed=13, scale_decades=5.0)).cuda()
    else:
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
    res.append({"set": kind, "stream_key_ties": float(info.stream_key_ties), "trees": T, "depth": D, "cols": C, "rows": R, "work_ratio": round(2 * T * D / (13 * C), 3), "auto": auto, "best": best, "ms": per,
                "auto_over_best": round(per[auto] / per[best], 3), "float_tile_rows": info.ring_rows, "u16_tile_rows": info.qring_tile_rows,
                "tilering_forms_ms": forms, "tilering_form_taken": "stream" if streams else "tile",
                "stream_levels": info.stream_levels, "stream_slots": info.stream_slots})
    print(res[-1], flush=True)
    f.close()
    del x
os.makedirs("gpurun_out", exist_ok=True)
for e in res:  # the faster form of TILERING, and what the create-time rule took
    fm = {k: v for k, v in e["tilering_forms_ms"].items() if v is not None}
    e["tilering_form_best"] = min(fm, key=fm.get) if fm else None
    e["form_taken_over_best"] = round(fm[e["tilering_form_taken"]] / min(fm.values()), 3) if fm and e["tilering_form_taken"] in fm else None
json.dump({"src_hash": bench.kernel_source_hash(), "shapes": res}, open("gpurun_out/selector_wide.json", "w"), indent=1)
