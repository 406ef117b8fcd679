"""What missing values cost the quantised walk: K3 and KR3 with a share of the values replaced by the missing sentinel (every 512-row
quantise chunk then reports "missing seen" and its tiles take the walk with the missing rule).  Pre-pass and walk per predict from the
in-library hipEvents.   python tools/ms_time.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
import bench

MISSING = -999.0
R = 1_000_000
res = []
for name in ("K3", "KR3"):
    kind, (nodes, T, D, C), data = bench.baseline_workload(ta, name)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.reserve(R)
    for mp in (0.0, 0.001, 0.05):
        x = data.copy()
        if mp:
            rng = np.random.default_rng(5)
            x[rng.random(x.shape, dtype=np.float32) < mp] = np.float32(MISSING)
        xd = torch.from_numpy(x).cuda()
        out = torch.empty(R, dtype=torch.float32, device="cuda")
        for _ in range(3):
            f.predict_raw(xd, out)
        f.set_profiling(10)
        for _ in range(10):
            f.predict_raw(xd, out)
        torch.cuda.synchronize()
        f.check()
        w, p = f.kernel_times_ms(), f.prepass_times_ms()
        res.append({"config": name, "kernel_form": f.kernel_form(R), "missing_share": mp, "prepass_ms": round(float(np.mean(p)), 4), "walk_ms": round(float(np.mean(w)), 4)})
        print(json.dumps(res[-1]), flush=True)
        del xd
    f.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"src_hash": bench.kernel_source_hash(), "runs": res}, open("gpurun_out/ms_time.json", "w"), indent=1)
