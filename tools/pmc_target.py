"""N predicts (TAHOE_STRATEGY_AUTO, resident inputs) of one BASELINE configuration and nothing else on the GPU: the target of
tools/pmc_script.sh for the per-configuration counter profiles (profiles/r04/pmc_k{1..5}.json).
    python tools/pmc_target.py K2 4        env TAHOE_WSTREAM etc. as for any create"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta

import bench  # the workloads are bench.py's (baseline_workload): one definition for the bench legs, the profiles and this script

MISSING = bench.MISSING


def make(cfg):
    """(forest, device rows) of a BASELINE configuration, as bench.py's `configs` legs build them."""
    kind, desc, data = bench.baseline_workload(ta, cfg)
    if kind == "sparse":
        sn, tr, C = desc
        return ta.capi.SparseForest(sn, tr, C, missing=MISSING), torch.from_numpy(data).cuda()
    nodes, T, D, C = desc
    return ta.Forest(nodes, T, D, C, missing=MISSING), torch.from_numpy(data).cuda()


if __name__ == "__main__":
    cfg, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    f, x = make(cfg)
    out = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
    for _ in range(n):
        f.predict_raw(x, out)
    torch.cuda.synchronize()
    f.check()
    print(cfg, "strategy", ta.STRATEGY_NAMES[f.get_strategy(x.shape[0])], n, "predicts")
