#!/usr/bin/env python3
"""Strategy selector vs brute-force enumeration (what the reference's SetUp does, BaseTahoeTest.h:652-706, and what
its analytic model in main.cu:22-80 tries to predict): for a grid of shapes, time every available strategy and
compare the fastest with TAHOE_STRATEGY_AUTO's choice.  Writes gpurun_out/selector.json (--holdout: shapes the rule was not
fitted on -> selector_holdout.json; --realistic: forests and rows of the histogram-style generator -- <= 255 quantile thresholds
per feature, Zipf-skewed feature usage, early leaves, features on different scales -> selector_realistic.json).  Every file is
stamped with the hash of the kernel sources it was measured on."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tahoe_amd as ta  # noqa: E402

SHAPES = [  # trees, depth, cols, rows
    (100, 4, 16, 200_000), (500, 8, 18, 200_000), (1000, 6, 64, 200_000), (200, 10, 128, 200_000),
    (1000, 12, 256, 200_000), (300, 12, 500, 100_000), (50, 14, 256, 100_000), (500, 8, 1000, 50_000),
    (500, 8, 3072, 50_000), (2000, 3, 32, 200_000),
]
# shapes the AUTO rule was NOT fitted on (`--holdout`): does it generalise?
HOLDOUT = [
    (300, 5, 28, 300_000), (64, 9, 96, 150_000), (1500, 10, 200, 100_000), (20, 12, 512, 100_000),
    (800, 4, 100, 200_000), (400, 7, 400, 100_000), (100, 11, 64, 200_000), (3000, 6, 16, 100_000),
    (150, 8, 2048, 40_000), (30, 3, 8, 500_000),
]


# histogram-style forests (tahoe_synth_forest_hist): shapes of the public GBDT benchmarks the reference is run on
# (run_all_15_examples.sh:51-65: SUSY / HIGGS / covtype / year / epsilon-like widths); trees, depth, cols, rows, max_bins
REALISTIC = [
    (500, 8, 18, 500_000, 254), (1000, 6, 28, 500_000, 254), (300, 10, 54, 300_000, 63), (800, 8, 90, 300_000, 255),
    (400, 12, 256, 300_000, 254), (200, 8, 2000, 50_000, 254), (2000, 7, 128, 200_000, 254),
]


def main():
    import bench

    out = []
    holdout, realistic = "--holdout" in sys.argv, "--realistic" in sys.argv
    for shape in (REALISTIC if realistic else HOLDOUT if holdout else SHAPES):
        T, D, C, R = shape[:4]
        if realistic:
            nodes = ta.synth_forest_hist(T, D, C, seed=7, feature_seed=11, max_bins=shape[4], zipf_s=1.0, leaf_prob=0.02, scale_decades=3.0)
            x = torch.from_numpy(ta.synth_data_hist(R, C, seed=8, feature_seed=11, scale_decades=3.0)).cuda()
        else:
            nodes = ta.synth_forest(T, D, C, seed=7)
            x = torch.from_numpy(ta.synth_data(R, C, seed=8)).cuda()
        f = ta.Forest(nodes, T, D, C, missing=-999.0)
        auto = f.get_strategy(R)
        auto_form = f.kernel_form(R)
        sums = torch.empty(R, dtype=torch.float32, device="cuda")
        times = {}
        for s in range(1, 6):
            try:
                f.set_strategy(s)
            except ta.TahoeError:
                continue
            if s == ta.STRATEGY_DIRECT and T * D * R > 3e10:
                continue  # minutes of divergent gathers; never the selector's choice for these shapes
            f.predict_raw(x, sums)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                f.predict_raw(x, sums)
            torch.cuda.synchronize()
            times[ta.STRATEGY_NAMES[s]] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        best = min(times, key=times.get)
        out.append({"trees": T, "depth": D, "cols": C, "rows": R, "ms": times, "best": best,
                    "auto": ta.STRATEGY_NAMES[auto], "auto_kernel_form": auto_form,
                    "auto_over_best": round(times[ta.STRATEGY_NAMES[auto]] / times[best], 3)})
        if realistic:
            out[-1]["max_bins"] = shape[4]
        print(out[-1], flush=True)
        f.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = "selector_realistic.json" if realistic else "selector_holdout.json" if holdout else "selector.json"
    json.dump({"src_hash": bench.kernel_source_hash(), "generator": "tahoe_synth_forest_hist / tahoe_synth_data_hist" if realistic else
               "tahoe_synth_forest / tahoe_synth_data (uniform)", "auto_is_fastest_on": sum(1 for e in out if e["auto"] == e["best"]),
               "worst_auto_over_best": max(e["auto_over_best"] for e in out), "shapes": out},
              open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)


if __name__ == "__main__":
    main()
