"""Soak: many back-to-back predicts of the ring kernels (K3 QRING, KR3 QRING on u8 codes, K2 wide forms, K5 sparse: quantised and float32), then tahoe_forest_check
(a bounded LDS wait that ever timed out raises) and a bit-for-bit comparison of the last result with the first."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta

QUICK = len(sys.argv) > 1 and sys.argv[1] == "--quick"  # tests/test_fuzz_gpu.py: 200 K3 predicts, every 20th compared, check() each time
EVERY = 20 if QUICK else 10


def soak(name, f, x, n):
    out = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
    f.predict_raw(x, out)
    torch.cuda.synchronize()
    first = out.cpu().numpy().copy()
    ref = torch.from_numpy(first).cuda()
    t = time.perf_counter()
    bad = 0
    for i in range(n):
        f.predict_raw(x, out)
        if i % EVERY == EVERY - 1:  # every tenth (--quick: twentieth) result is compared on the device, bit for bit, with the first
            bad += int((out.view(torch.int32) != ref.view(torch.int32)).sum().item())
            f.check()
    f.check()
    dt = time.perf_counter() - t
    same = np.array_equal(out.cpu().numpy().view(np.uint32), first.view(np.uint32))
    print(f"{name}: {n} predicts, {dt / n * 1e3:.3f} ms each, error flag clear, last == first: {same}, rows that ever differed: {bad}")
    assert same and bad == 0

x = torch.from_numpy(ta.synth_data(1_000_000, 256, seed=43)).cuda()
f = ta.Forest(ta.synth_forest(1000, 12, 256, seed=42), 1000, 12, 256, missing=-999.0)
soak("K3 qring", f, x, 200 if QUICK else 1000)
f.close()
# KR3: histogram-style forest on u8 codes (384-row tiles, ring of 5 for 14 walkers: the shortest ring of the library)
xh = torch.from_numpy(ta.synth_data_hist(1_000_000, 256, seed=43, feature_seed=7, scale_decades=3.0)).cuda()
f = ta.Forest(ta.synth_forest_hist(1000, 12, 256, seed=42, feature_seed=7, max_bins=254), 1000, 12, 256, missing=-999.0)
assert f.kernel_form(1_000_000) == "qring_region8"
soak("KR3 qring on u8 codes", f, xh, 100 if QUICK else 600)
f.close()
del xh
sn, tr = ta.capi.synth_sparse_forest(2000, 256, 4, 24, 0.32, 65535, 44)
f = ta.capi.SparseForest(sn, tr, 256, missing=-999.0)
soak("K5 sparse, quantised ring", f, x[:200_000].contiguous(), 40 if QUICK else 100)
f.set_strategy(ta.STRATEGY_TILEBLOCK)
soak("K5 sparse, float32 ring", f, x[:200_000].contiguous(), 20 if QUICK else 50)
f.close()
del x
x = torch.from_numpy(ta.synth_data(100_000, 3072, seed=22)).cuda()
os.environ["TAHOE_WSTREAM"] = "0"  # the tile form of wide TILERING (walker slots + LDS ring)
f = ta.Forest(ta.synth_forest(500, 8, 3072, seed=21), 500, 8, 3072, missing=-999.0)
f.set_strategy(ta.STRATEGY_TILERING)
soak("K2 wide, float32 tile form (ring)", f, x, 40 if QUICK else 300)
f.set_strategy(ta.STRATEGY_QRING)
soak("K2 wide, quantised ring", f, x, 40 if QUICK else 200)
f.close()
os.environ["TAHOE_WSTREAM"] = "1"  # the row-streaming wide form (LDS-DMA ring, counters in LDS)
f = ta.Forest(ta.synth_forest(500, 8, 3072, seed=21), 500, 8, 3072, missing=-999.0)
f.set_strategy(ta.STRATEGY_TILERING)
soak("K2 wide, float32 row-streaming form", f, x, 40 if QUICK else 300)
f.close()
print("soak ok")
