#!/usr/bin/env python3
"""Times small batches on K1's and K3's forests: QRING with one workgroup per tile (TAHOE_QRING_SLICES=1) against the SPLIT
form the launch picks (tree slices per tile + an ordered per-row sum).  Prints ms per predict and the workgroups launched."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tahoe_amd as ta  # noqa: E402

res = {}
for name, (T, D, C) in {"K1 forest (500 trees, depth 8, 18 features)": (500, 8, 18), "K3 forest (1000 trees, depth 12, 256 features)": (1000, 12, 256)}.items():
    nodes = ta.synth_forest(T, D, C, seed=42)
    x = torch.from_numpy(ta.synth_data(65_536, C, seed=43)).cuda()
    res[name] = {}
    for rows in (1_000, 2_000, 5_000, 10_000, 16_384, 32_768, 65_536):
        row = {}
        for label, knob in (("one_workgroup_per_tile", "1"), ("picked", None)):
            if knob is None:
                os.environ.pop("TAHOE_QRING_SLICES", None)
            else:
                os.environ["TAHOE_QRING_SLICES"] = knob
            f = ta.Forest(nodes, T, D, C, missing=-999.0)
            xs = x[:rows].contiguous()
            out = torch.empty(rows, dtype=torch.float32, device="cuda")
            f.reserve(rows)
            for _ in range(5):
                f.predict_raw(xs, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                f.predict_raw(xs, out)
            torch.cuda.synchronize()
            row[label] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
            f.check()
            f.close()
        tiles = (rows + 127) // 128
        row["tiles_of_128_rows"] = tiles
        res[name][rows] = row
        print(name, rows, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "small_batch.json"), "w"), indent=1)
