"""Full-size parity, every row: K3 (1000 trees) and K4 (8000 trees, tree groups chained) float32 sums of all 1M rows
against the CPU oracle bit for bit, plus leaf indices of a 20 k-row slice.  Minutes of CPU on the GPU box's host cores.
    python tools/full_parity.py > gpurun_out/full_parity.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tahoe_amd as ta  # noqa: E402
from oracle import oracle  # noqa: E402  (a checker script, like tests/)

MISSING = -999.0
C, R, D = 256, 1_000_000, 12
threads = os.cpu_count() or 1
data = ta.synth_data(R, C, seed=43, missing_prob=0.01, missing=MISSING)  # 1 % missing: both compare paths at scale
x = torch.from_numpy(data).cuda()
out = {"rows": R, "cols": C, "depth": D, "missing_prob": 0.01, "cpu_threads": threads, "cases": {}}
for name, T in (("K3", 1000), ("K4", 8000)):
    nodes = ta.synth_forest(T, D, C, seed=42)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    got = f.predict_raw(x).cpu().numpy()
    f.check()
    t = time.perf_counter()
    want = np.empty(R, dtype=np.float32)
    step = 100_000  # in slices, with a progress line each: a silent run is taken for a hung one
    for lo in range(0, R, step):
        want[lo:lo + step] = oracle.predict(nodes, T, D, data[lo:lo + step], MISSING, threads=threads)[0]
        print(f"{name}: oracle rows {lo + step} / {R}, {time.perf_counter() - t:.0f} s", file=sys.stderr, flush=True)
    cpu_s = time.perf_counter() - t
    equal = int(np.count_nonzero(got.view(np.uint32) == want.view(np.uint32)))
    sl = slice(500_000, 520_000)
    _, want_leaf = oracle.predict(nodes, T, D, data[sl], MISSING, want_leaf=True, threads=threads)
    leaf, _ = f.predict_leaf_idx(x[sl].contiguous())
    leaf_equal = bool(np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf))
    out["cases"][name] = {"trees": T, "strategy": ta.STRATEGY_NAMES[f.get_strategy(R)], "tree_groups": f.info().qring_groups,
                          "sums_bitwise_equal_rows": equal, "of_rows": R, "leaf_indices_equal_on_rows_500000_520000": leaf_equal,
                          "cpu_seconds": round(cpu_s, 1)}
    print(name, out["cases"][name], file=sys.stderr, flush=True)
    f.close()
print(json.dumps(out, indent=1))
