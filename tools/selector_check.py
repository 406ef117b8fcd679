#!/usr/bin/env python3
"""Strategy selector vs brute-force enumeration (what the reference's SetUp does, BaseTahoeTest.h:652-706, and what
its analytic model in main.cu:22-80 tries to predict): for a grid of shapes, time every available strategy and
compare the fastest with TAHOE_STRATEGY_AUTO's choice.  Writes gpurun_out/selector.json."""
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


def main():
    out = []
    holdout = "--holdout" in sys.argv
    for (T, D, C, R) in (HOLDOUT if holdout else SHAPES):
        nodes = ta.synth_forest(T, D, C, seed=7)
        x = torch.from_numpy(ta.synth_data(R, C, seed=8)).cuda()
        f = ta.Forest(nodes, T, D, C, missing=-999.0)
        auto = f.get_strategy(R)
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
                    "auto": ta.STRATEGY_NAMES[auto], "auto_over_best": round(times[ta.STRATEGY_NAMES[auto]] / times[best], 3)})
        print(out[-1], flush=True)
        f.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "selector_holdout.json" if holdout else "selector.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
