"""K4 (8000 trees of depth 12, 256 features, 1 M rows) on one GPU as a function of the number of tree groups (TAHOE_QRING_GROUPS):
a group's busiest feature must stay below 32767 distinct thresholds (4 groups of 2000 trees: ~32 k), but only groups whose feature
PAIRS fit LDS (two sorted tables + start tables) take the one-pass bucketed quantise kernel -- 7 groups and more.  Per group count:
the quantise kernel that ran, pre-pass and walk time per predict (in-library hipEvents; with several groups the first group's
pre-pass is timed apart and the rest counts as walk, so the step total is what compares), bit-equality of the sums across group
counts.  -> gpurun_out/k4_groups.json (stamped).    python tools/k4_groups.py [group counts...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
import bench

T, D, C, R = 8000, 12, 256, 1_000_000
counts = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8]
nodes = ta.synth_forest(T, D, C, seed=42)
x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
out = torch.empty(R, dtype=torch.float32, device="cuda")
res, ref = [], None
for g in counts:
    os.environ["TAHOE_QRING_GROUPS"] = str(g)
    t0 = time.perf_counter()
    f = ta.Forest(nodes, T, D, C, missing=-999.0)
    create_s = time.perf_counter() - t0
    f.reserve(R)
    for _ in range(2):
        f.predict_raw(x, out)
    f.set_profiling(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f.predict_raw(x, out)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    f.check()
    walk, pre = f.kernel_times_ms(), f.prepass_times_ms()
    same = True if ref is None else bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
    ref = out.clone() if ref is None else ref
    info = f.info()
    per_feature = (T // info.qring_groups) * 4095 / C
    res.append({"groups_asked": g, "groups": int(info.qring_groups), "trees_per_group": T // info.qring_groups,
                "thresholds_per_feature_per_group_approx": int(per_feature),
                "pair_tables_bytes_approx": int(2 * per_feature * 4),
                "first_group_prepass_ms": round(float(np.mean(pre)), 3), "rest_ms": round(float(np.mean(walk)), 3),
                "ms_per_predict_events": round(float(np.mean(pre) + np.mean(walk)), 3), "ms_per_predict_wall": round(wall, 3),
                "create_s": round(create_s, 2), "device_MB": round(info.device_bytes / 1e6, 1), "sums_bit_equal_to_first": same})
    print(res[-1], flush=True)
    f.close()
os.environ.pop("TAHOE_QRING_GROUPS", None)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"src_hash": bench.kernel_source_hash(), "workload": "K4: 8000 trees depth 12, 256 features, 1M rows, one GPU", "runs": res},
          open("gpurun_out/k4_groups.json", "w"), indent=1)
