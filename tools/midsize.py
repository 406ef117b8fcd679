#!/usr/bin/env python3
"""Times QRING on K3's forest for the per-GPU batch sizes of the strong-scaling run (1 M rows over 1 / 2 / 4 / 8 GPUs) and a few
in between: the plan the launch picks (whole waves of 192-row tiles + a remainder) against one form for the whole batch."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tahoe_amd as ta  # noqa: E402

T, D, C = 1000, 12, 256
nodes = ta.synth_forest(T, D, C, seed=42)
x = torch.from_numpy(ta.synth_data(1_000_000, C, seed=43)).cuda()
res = {}
for rows in (65_536, 100_000, 125_000, 200_000, 250_000, 333_334, 500_000, 1_000_000):
    row = {}
    for label, knob in (("all_128_row_tiles", "2"), ("all_192_row_tiles", "3"), ("picked_plan", None)):
        if knob is None:
            os.environ.pop("TAHOE_QRING_CHAINS", None)
        else:
            os.environ["TAHOE_QRING_CHAINS"] = knob
        f = ta.Forest(nodes, T, D, C, missing=-999.0)
        xs = x[:rows].contiguous()
        out = torch.empty(rows, dtype=torch.float32, device="cuda")
        f.reserve(rows)
        for _ in range(3):
            f.predict_raw(xs, out)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            f.predict_raw(xs, out)
        torch.cuda.synchronize()
        row[label] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        f.check()
        f.close()
    res[rows] = row
    print(rows, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "midsize.json"), "w"), indent=1)
