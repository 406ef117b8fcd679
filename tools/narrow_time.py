"""Forests of <= 128 features: the 16-KiB region stride (384-row tiles on u16 codes; 15 walkers and a ring of 24 on u8 codes) against
the 32-KiB stride (TAHOE_QRING_NARROW128=0), pre-pass and walk per predict from the in-library hipEvents.  python tools/narrow_time.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
import bench

SHAPES = [  # trees, depth, cols, rows, max_bins (0 = uniform generator: thousands of thresholds per feature)
    (500, 8, 18, 500_000, 254), (1000, 6, 28, 500_000, 254), (800, 8, 90, 300_000, 255), (1000, 8, 64, 300_000, 0),
    (500, 8, 18, 500_000, 0), (2000, 7, 128, 200_000, 254), (400, 10, 100, 300_000, 0),
]
res = []
for (T, D, C, R, bins) in SHAPES:
    if bins:
        nodes = ta.synth_forest_hist(T, D, C, seed=7, feature_seed=11, max_bins=bins, zipf_s=1.0, leaf_prob=0.02, scale_decades=3.0)
        x = torch.from_numpy(ta.synth_data_hist(R, C, seed=8, feature_seed=11, scale_decades=3.0)).cuda()
    else:
        nodes = ta.synth_forest(T, D, C, seed=7)
        x = torch.from_numpy(ta.synth_data(R, C, seed=8)).cuda()
    out = torch.empty(R, dtype=torch.float32, device="cuda")
    ref, row = None, {"trees": T, "depth": D, "cols": C, "rows": R, "max_bins": bins or "uniform"}
    for narrow in ("0", "1"):
        os.environ["TAHOE_QRING_NARROW128"] = narrow
        f = ta.Forest(nodes, T, D, C, missing=-999.0)
        f.set_strategy(ta.STRATEGY_QRING)
        f.reserve(R)
        for _ in range(3):
            f.predict_raw(x, out)
        f.set_profiling(10)
        for _ in range(10):
            f.predict_raw(x, out)
        torch.cuda.synchronize()
        f.check()
        w, p = f.kernel_times_ms(), f.prepass_times_ms()
        same = True if ref is None else bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        ref = out.clone() if ref is None else ref
        row["stride_16k" if narrow == "1" else "stride_32k"] = {"kernel_form": f.kernel_form(R), "walk_ms": round(float(np.mean(w)), 4),
                                                                "prepass_ms": round(float(np.mean(p)), 4), "same_bits": same}
        f.close()
    row["walk_gain"] = round(1.0 - row["stride_16k"]["walk_ms"] / row["stride_32k"]["walk_ms"], 3)
    res.append(row)
    print(json.dumps(row), flush=True)
os.environ.pop("TAHOE_QRING_NARROW128", None)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"src_hash": bench.kernel_source_hash(), "shapes": res}, open("gpurun_out/narrow_time.json", "w"), indent=1)
