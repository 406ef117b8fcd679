"""Times the text loaders (sequential fgets-style reader vs the multi-threaded one) on K3-shaped slices.
Run on the GPU box (its host cores): python tools/loader_bench.py > gpurun_out/loader_bench.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta  # noqa: E402

out = {"cores": os.cpu_count(), "cases": []}
tmp = os.environ.get("TMPDIR", "/tmp")
T, D, C, R = 250, 12, 256, 60_000  # a quarter of K3's model, 6 % of its rows
nodes = ta.synth_forest(T, D, C, seed=42)
data = ta.synth_data(R, C, seed=43)
m, d = os.path.join(tmp, "lb_model.txt"), os.path.join(tmp, "lb_data.txt")
ta.write_model(m, nodes, T, D)
ta.write_data(d, data, -999.0)
for threads in (1, 4, 8, 16):
    os.environ["TAHOE_LOADER_THREADS"] = str(threads)
    t = time.perf_counter()
    n, _, _ = ta.load_model(m)
    tm = time.perf_counter() - t
    t = time.perf_counter()
    x, _ = ta.load_data(d)
    td = time.perf_counter() - t
    assert n.tobytes() == nodes.tobytes() and x.tobytes() == data.tobytes()
    out["cases"].append({"threads": threads, "model_Mlines_per_s": round(T * 8191 * 5 / tm / 1e6, 1),
                         "data_Mlines_per_s": round(R * C / td / 1e6, 1),
                         "model_MB_per_s": round(os.path.getsize(m) / tm / 1e6, 1),
                         "data_MB_per_s": round(os.path.getsize(d) / td / 1e6, 1)})
os.remove(m)
os.remove(d)
print(json.dumps(out))
