"""Randomised differential test: random dense and sparse forests / batches, every available strategy against the CPU oracle
(leaf indices and float32 sums, bit for bit), for a given number of seconds.  python tests/fuzz_gpu.py [seconds] [seed]  (test infrastructure: it uses the CPU oracle as the checker)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
from oracle import oracle

MISSING = -999.0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # batches of several waves of tiles (mixed tile plans), hundreds of trees
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
t_end = time.time() + budget
cases = checks = 0
while time.time() < t_end:
    sparse = rng.random() < 0.35
    C = int(rng.choice([1, 2, 3, 7, 18, 32, 64, 100, 255, 256, 257, 500, 513, 600, 1000, 1536, 2050, 3072]))
    R = int(rng.choice([1, 5, 63, 64, 65, 127, 129, 191, 193, 384, 500, 1000, 2049, 5000]))
    if BIG:
        C = int(rng.choice([18, 64, 255, 256, 600, 1536, 3072]))
        R = int(rng.choice([20_000, 49_153, 50_000, 66_000, 98_305, 120_000])) if C <= 600 else int(rng.choice([3_000, 9_000, 20_000]))
    mp = float(rng.choice([0.0, 0.0, 0.02, 0.2]))
    data = ta.synth_data(R, C, seed=int(rng.integers(1 << 30)), missing_prob=mp, missing=MISSING, nan_prob=mp / 2)
    x = torch.from_numpy(data).cuda()
    if sparse:
        T = int(rng.choice([1, 3, 17, 40, 150, 400])) if not BIG else int(rng.choice([150, 400, 900]))
        dmin = int(rng.integers(0, 6)); dmax = dmin + int(rng.integers(0, 14))
        sn, tr = ta.capi.synth_sparse_forest(T, C, dmin, dmax, float(rng.choice([0.0, 0.1, 0.35])), 65535, int(rng.integers(1 << 30)))
        want, want_leaf = oracle.sparse_predict(sn, tr, data, MISSING, want_leaf=True, threads=8)
        f = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
        strategies = [ta.STRATEGY_AUTO, ta.STRATEGY_DIRECT, ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK, ta.STRATEGY_QRING]
        desc = f"sparse T={T} depth={dmin}..{dmax} C={C} R={R} missing={mp}"
    else:
        T = int(rng.choice([1, 2, 5, 16, 33, 100, 300])) if not BIG else int(rng.choice([100, 300, 700]))
        D = int(rng.choice([0, 1, 2, 3, 5, 8, 10, 12])) if C <= 600 else int(rng.choice([2, 3, 6, 8, 10]))
        hist = rng.random() < 0.35  # histogram-style forest and rows (few thresholds per feature: QRING's u8 form on large batches)
        if hist:
            fseed, dec, bins = int(rng.integers(1 << 30)), float(rng.choice([0.0, 2.0, 5.0])), int(rng.choice([3, 31, 254, 255, 1000]))
            nodes = ta.synth_forest_hist(T, D, C, seed=int(rng.integers(1 << 30)), feature_seed=fseed, max_bins=bins,
                                         zipf_s=float(rng.choice([0.0, 1.0, 2.0])), leaf_prob=float(rng.choice([0.0, 0.05])), scale_decades=dec)
            data = ta.synth_data_hist(R, C, seed=int(rng.integers(1 << 30)), feature_seed=fseed, scale_decades=dec, missing_prob=mp, missing=MISSING)
            if mp:
                data[::7, ::3] = np.nan
            x = torch.from_numpy(data).cuda()
        else:
            nodes = ta.synth_forest(T, D, C, seed=int(rng.integers(1 << 30)), leaf_prob=float(rng.choice([0.0, 0.1, 0.3])))
        want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
        stream_form = C > 512 and rng.random() < 0.5  # TILERING as the row-streaming kernel (read at create)
        os.environ["TAHOE_WSTREAM"] = "1" if stream_form else "0"
        # the create-time knobs that steer QRING between its forms: a quarter of the cases take a draw of them, so that the forms AUTO
        # would not pick for this shape (u16 codes on few thresholds, the 32-KiB stride on narrow forests, forced tile plans, no slices,
        # the column layout, the pair quantise kernels) meet the same oracle
        knobs = {}
        if rng.random() < 0.25:
            for name, values in (("TAHOE_QRING_CODE8", ["0"]), ("TAHOE_QRING_NARROW128", ["0"]), ("TAHOE_QRING_CHAINS", ["2", "3"]),
                                 ("TAHOE_QRING_SLICES", ["1", "3"]), ("TAHOE_QRING_REGIONS", ["0"]), ("TAHOE_QUANT_MULTI", ["0"]),
                                 ("TAHOE_QUANT_BUCKETS", ["0"])):
                if rng.random() < 0.3:
                    knobs[name] = str(rng.choice(values))
        os.environ.update(knobs)
        f = ta.Forest(nodes, T, D, C, missing=MISSING)
        for name in knobs:
            del os.environ[name]
        strategies = [ta.STRATEGY_AUTO, ta.STRATEGY_DIRECT, ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK, ta.STRATEGY_TILERING, ta.STRATEGY_QRING]
        if knobs:  # the knobs touch QRING only: the float32 strategies of this forest were covered without them
            strategies = [ta.STRATEGY_AUTO, ta.STRATEGY_QRING]
        desc = (f"dense T={T} D={D} C={C} R={R} missing={mp}" + (" wstream" if stream_form else "") + (f" hist bins={bins} decades={dec}" if hist else "")
                + "".join(f" {k[6:]}={v}" for k, v in knobs.items()))
    for s in strategies:
        try:
            f.set_strategy(s)
        except ta.TahoeError:
            continue
        leaf, sums = f.predict_leaf_idx(x)
        raw = f.predict_raw(x)
        f.check()
        ok = (np.array_equal(bits(leaf.cpu().numpy()), want_leaf) and np.array_equal(bits(sums.cpu().numpy()), bits(want))
              and np.array_equal(bits(raw.cpu().numpy()), bits(want)))
        checks += 1
        if not ok:
            print("MISMATCH", desc, "strategy", s, ta.STRATEGY_NAMES[f.get_strategy(R)], flush=True)
            sys.exit(1)
    f.close()
    cases += 1
    if cases % (1 if BIG else 25) == 0:
        print(f"{cases} cases, {checks} strategy runs, last: {desc}", flush=True)
print(f"fuzz ok: {cases} cases, {checks} strategy runs, all bit-exact")
