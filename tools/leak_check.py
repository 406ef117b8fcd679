"""Device memory after repeated create / predict / destroy of a wide dense forest and a sparse forest (both forms of each):
the free memory after 5 rounds and after 40 must be equal.  python tools/leak_check.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tahoe_amd as ta
torch.cuda.init()
x1 = torch.from_numpy(ta.synth_data(3000, 3072, seed=1)).cuda()
x2 = torch.from_numpy(ta.synth_data(30000, 64, seed=2)).cuda()
sn, tr = ta.capi.synth_sparse_forest(300, 64, 4, 16, 0.3, 65535, 5)
nodes = ta.synth_forest(100, 8, 3072, seed=3)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
base = None
for i in range(40):
    f = ta.Forest(nodes, 100, 8, 3072, missing=-999.0)
    for s in (ta.STRATEGY_TILERING, ta.STRATEGY_QRING): f.set_strategy(s); f.predict_raw(x1)
    f.check(); f.close()
    g = ta.capi.SparseForest(sn, tr, 64, missing=-999.0)
    for s in (ta.STRATEGY_QRING, ta.STRATEGY_TILEBLOCK): g.set_strategy(s); g.predict_raw(x2)
    g.check(); g.close()
    if i == 4: base = free()
print("free after 5 rounds:", base, "after 40:", free(), "delta MB:", (base - free()) / 1e6)
