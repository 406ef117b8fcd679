"""KR3 = K3's shape (1000 trees of depth 12, 256 features, 1 M rows) from the histogram-style generator (<= 254 thresholds per
feature): QRING on u16 codes (TAHOE_QRING_CODE8=0: 192-row tiles) against u8 codes (384-row tiles), pre-pass and walk apart
(in-library hipEvents), bit-equality of the two, and a smaller batch (125 k rows: one of 8 GPUs' share).
    python tools/kr3_time.py [rows...]        -> one JSON line per (rows, form) + gpurun_out/kr3_time.json"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
import bench

kind, (nodes, T, D, C), data = bench.baseline_workload(ta, "KR3")
x_all = torch.from_numpy(data).cuda()
res = []
for R in [int(a) for a in sys.argv[1:]] or [1_000_000, 125_000]:
    x = x_all[:R].contiguous()
    out = torch.empty(R, dtype=torch.float32, device="cuda")
    ref = None
    for code8 in ("0", "1"):
        os.environ["TAHOE_QRING_CODE8"] = code8
        f = ta.Forest(nodes, T, D, C, missing=-999.0)
        f.set_strategy(ta.STRATEGY_QRING)
        f.reserve(R)
        for _ in range(3):
            f.predict_raw(x, out)
        f.set_profiling(10)
        torch.cuda.synchronize()
        for _ in range(10):
            f.predict_raw(x, out)
        torch.cuda.synchronize()
        f.check()
        walk, pre = f.kernel_times_ms(), f.prepass_times_ms()
        same = True if ref is None else bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        ref = out.clone() if ref is None else ref
        res.append({"rows": R, "code8": code8 == "1", "kernel_form": f.kernel_form(R), "tile_rows": int(f.info().qring_tile_rows),
                    "prepass_ms": round(float(np.mean(pre)), 4), "walk_ms": round(float(np.mean(walk)), 4),
                    "ms": round(float(np.mean(pre) + np.mean(walk)), 4), "same_bits_as_u16": same})
        print(json.dumps(res[-1]), flush=True)
        f.close()
os.environ.pop("TAHOE_QRING_CODE8", None)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"src_hash": bench.kernel_source_hash(), "workload": "KR3: tahoe_synth_forest_hist 1000 trees depth 12, 256 features, max_bins 254", "runs": res},
          open("gpurun_out/kr3_time.json", "w"), indent=1)
