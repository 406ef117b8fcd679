"""K3 (1000 trees of depth 12, 256 features, 1 M rows; or the row count given): pre-pass and walk per predict from the in-library
hipEvents, 20 timed predicts after 5 warm-ups.   python tools/k3_time.py [rows]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tahoe_amd as ta
T, D, C = 1000, 12, 256
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
f = ta.Forest(ta.synth_forest(T, D, C, seed=42), T, D, C, missing=-999.0)
x = torch.from_numpy(ta.synth_data(R, C, seed=43)).cuda()
out = torch.empty(R, dtype=torch.float32, device="cuda")
f.reserve(R)
for _ in range(5): f.predict_raw(x, out)
f.set_profiling(20)
for _ in range(20): f.predict_raw(x, out)
torch.cuda.synchronize(); f.check()
w, p = f.kernel_times_ms(), f.prepass_times_ms()
print(f"K3 {R} rows: {f.kernel_form(R)} quantise {np.mean(p):.4f} walk {np.mean(w):.4f} (min {np.min(w):.4f}) total {np.mean(p) + np.mean(w):.4f} ms")
