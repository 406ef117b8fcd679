import sys, numpy as np, torch
sys.path.insert(0, ".")
import tahoe_amd as ta
from oracle import oracle
T, D, C, R = 3, 4, 8, 256
nodes = ta.synth_forest(T, D, C, seed=7)
data = ta.synth_data(R, C, seed=8)
want, wl = oracle.predict(nodes, T, D, data, -999.0, want_leaf=True)
f = ta.Forest(nodes, T, D, C, missing=-999.0); f.set_strategy(5)
leaf, sums = f.predict_leaf_idx(torch.from_numpy(data).cuda()); f.check()
l = leaf.cpu().numpy().view(np.uint32); s = sums.cpu().numpy()
bad = np.argwhere(l != wl)
print("bad leaf count", len(bad), "of", l.size)
print("bad rows (first 40):", sorted(set(bad[:, 0].tolist()))[:40])
print("rows parity of bad:", np.bincount(bad[:, 0] % 2, minlength=2) if len(bad) else None)
print("sum mismatches", int((s.view(np.uint32) != want.view(np.uint32)).sum()))
for r, t in bad[:8]:
    print(r, t, "got", l[r, t], "want", wl[r, t])
