"""N predicts (TAHOE_STRATEGY_AUTO, resident inputs) of one BASELINE configuration and nothing else on the GPU: the target of
tools/pmc_script.sh for the per-configuration counter profiles (profiles/r03/pmc_k{1..5}.json).
    python tools/pmc_target.py K2 4        env TAHOE_WSTREAM etc. as for any create"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta

MISSING = -999.0
SHAPES = {"K1": (500, 8, 18, 10_000, 11, 12, 0.05), "K2": (500, 8, 3072, 100_000, 21, 22, 0.0), "K3": (1000, 12, 256, 1_000_000, 42, 43, 0.0),
          "K4": (8000, 12, 256, 1_000_000, 42, 43, 0.0)}


def make(cfg):
    """(forest, device rows) of a BASELINE configuration, as tools/run_configs.py builds them."""
    if cfg == "K5":
        sn, tr = ta.capi.synth_sparse_forest(2000, 256, 4, 24, 0.32, 65535, 44)
        return ta.capi.SparseForest(sn, tr, 256, missing=MISSING), torch.from_numpy(ta.synth_data(200_000, 256, seed=43)).cuda()
    T, D, C, R, fs, ds, lp = SHAPES[cfg]
    f = ta.Forest(ta.synth_forest(T, D, C, seed=fs, leaf_prob=lp), T, D, C, missing=MISSING)
    return f, torch.from_numpy(ta.synth_data(R, C, seed=ds, missing_prob=0.02 if cfg == "K1" else 0.0, missing=MISSING)).cuda()


if __name__ == "__main__":
    cfg, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    f, x = make(cfg)
    out = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
    for _ in range(n):
        f.predict_raw(x, out)
    torch.cuda.synchronize()
    f.check()
    print(cfg, "strategy", ta.STRATEGY_NAMES[f.get_strategy(x.shape[0])], n, "predicts")
