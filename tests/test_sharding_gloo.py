"""The N > 1 control flow (tahoe_amd/sharding.py) with world_size 2 on CPU: gloo backend, the per-rank
operator replaced by a stand-in that runs the CPU oracle on the rank's tree slice (the product itself has no
CPU path -- the stand-in lives here, in the test).  Checks the partition, the single all-reduce, the
transform on the total, and the row-sharded variant."""
import os
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T, D, C, R, MISSING = 37, 5, 12, 301, -999.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import tahoe_amd as ta
    from oracle import oracle
    from tahoe_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nodes = ta.synth_forest(T, D, C, seed=3, leaf_prob=0.1)
    data = ta.synth_data(R, C, seed=4, missing_prob=0.05, missing=MISSING)

    class OracleLocal:  # CPU stand-in for tahoe_amd.Forest on this rank's trees
        def __init__(self, my_nodes, my_trees):
            self.nodes, self.trees = my_nodes, my_trees

        def predict_raw(self, x, out=None):
            sums, _ = oracle.predict(self.nodes, self.trees, D, x.numpy(), MISSING)
            return torch.from_numpy(sums.copy())

    def finish(sums):  # AVG then bias, float32, as BaseTahoeTest.h:467-468
        return (sums / np.float32(T) + np.float32(0.125)).to(torch.float32)

    f = sharding.TreeShardedForest(nodes, T, D, OracleLocal, finish)
    lo, hi = f.tree_range
    preds = f.predict(torch.from_numpy(data))
    # row sharding: own block of rows, whole forest, no collective in the data path
    rlo, rhi = sharding.shard_bounds(R, rank, world)
    row_part, _ = oracle.predict(nodes, T, D, data[rlo:rhi], MISSING, output=0x1, global_bias=0.125)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), preds=preds.numpy(), lo=lo, hi=hi, row_part=row_part, rlo=rlo,
             rhi=rhi)
    dist.barrier()
    dist.destroy_process_group()


def test_tree_and_row_sharding_world2(built, tmp_path):
    import torch.multiprocessing as mp

    import tahoe_amd as ta
    from oracle import oracle

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # the tree ranges tile [0, T) without gaps
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == T
    nodes = ta.synth_forest(T, D, C, seed=3, leaf_prob=0.1)
    data = ta.synth_data(R, C, seed=4, missing_prob=0.05, missing=MISSING)
    want, _ = oracle.predict(nodes, T, D, data, MISSING, output=0x1, global_bias=0.125)
    exact = oracle.predict_f64(nodes, T, D, data, MISSING) / T + 0.125
    # both ranks hold the same all-reduced result
    assert np.array_equal(parts[0]["preds"], parts[1]["preds"])
    got = parts[0]["preds"]
    # tree sharding: float32 rounding differs from the sequential sum; bound it against the float64 sum
    assert np.max(np.abs(got - exact)) <= 4e-6
    assert np.max(np.abs(got - want)) <= 4e-6
    # row sharding: bit-identical to the unsharded prediction
    rows = np.concatenate([parts[0]["row_part"], parts[1]["row_part"]])
    assert parts[0]["rlo"] == 0 and parts[0]["rhi"] == parts[1]["rlo"] and parts[1]["rhi"] == R
    assert np.array_equal(rows.view(np.uint32), want.view(np.uint32))


def test_shard_bounds_and_selector():
    from tahoe_amd import sharding

    for n in (0, 1, 7, 1000, 8000):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)
    assert sharding.choose_sharding(1000, 12) == "rows"      # K3: 49 MB forest
    assert sharding.choose_sharding(8000, 12) == "trees"     # K4: 393 MB forest
