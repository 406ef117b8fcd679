"""The N > 1 control flow (tahoe_amd/sharding.py) with world_size 2 on CPU: gloo backend, the per-rank
operator replaced by a stand-in that runs the CPU oracle on the rank's tree slice (the product itself has no
CPU path -- the stand-in lives here, in the test).  Checks the partition, the three ways of combining tree shards
("allreduce64", "allreduce32", "chain"), the transform on the total, and the row-sharded variant -- on a toy forest
and on a 1200-tree forest, where float32 association errors are no longer hidden by the size of the sums."""
import os
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MISSING = -999.0
CASES = {"toy": dict(T=37, D=5, C=12, R=301, chunk=97), "t1200": dict(T=1200, D=6, C=16, R=700, chunk=256)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(case):
    import tahoe_amd as ta

    c = CASES[case]
    nodes = ta.synth_forest(c["T"], c["D"], c["C"], seed=3, leaf_prob=0.1)
    data = ta.synth_data(c["R"], c["C"], seed=4, missing_prob=0.05, missing=MISSING)
    return c, nodes, data


def _worker(rank, world, port, out_dir, case):
    import sys

    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle
    from tahoe_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, nodes, data = _inputs(case)
    T, D, R = c["T"], c["D"], c["R"]

    class OracleLocal:  # CPU stand-in for tahoe_amd.Forest on this rank's trees
        def __init__(self, my_nodes, my_trees):
            self.nodes, self.trees = my_nodes, my_trees

        def predict_raw(self, x, out=None):
            sums, _ = oracle.predict(self.nodes, self.trees, D, x.numpy(), MISSING)
            return torch.from_numpy(sums.copy())

        def predict_accumulate(self, x, sums):  # in place, as tahoe_forest_predict_accumulate
            s = np.ascontiguousarray(sums.numpy().copy())
            oracle.predict_continue(self.nodes, self.trees, D, x.numpy(), MISSING, s)
            sums.copy_(torch.from_numpy(s))
            return sums

    def finish(sums):  # AVG then bias, float32, as BaseTahoeTest.h:467-468
        return (sums / np.float32(T) + np.float32(0.125)).to(torch.float32)

    out = {}
    x = torch.from_numpy(data)
    for mode in sharding.TreeShardedForest.MODES:
        f = sharding.TreeShardedForest(nodes, T, D, OracleLocal, finish, mode=mode, chunk_rows=c["chunk"])
        out[f"sums_{mode}"] = f.predict_sums(x).numpy().copy()
        out[f"preds_{mode}"] = f.predict(x, broadcast=True).numpy().copy()
        out[f"result_rank_{mode}"] = f.result_rank
        lo, hi = f.tree_range
    # row sharding: own block of rows, whole forest, no collective in the data path
    rlo, rhi = sharding.shard_bounds(R, rank, world)
    row_part, _ = oracle.predict(nodes, T, D, data[rlo:rhi], MISSING, output=0x1, global_bias=0.125)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, row_part=row_part, rlo=rlo, rhi=rhi, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", list(CASES))
def test_tree_and_row_sharding_world2(built, tmp_path, case):
    import torch.multiprocessing as mp

    from oracle import oracle
    from tahoe_amd import sharding

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    c, nodes, data = _inputs(case)
    T, D, R = c["T"], c["D"], c["R"]
    # the tree ranges tile [0, T) without gaps
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == T
    want_sums, _ = oracle.predict(nodes, T, D, data, MISSING)
    want, _ = oracle.predict(nodes, T, D, data, MISSING, output=0x1, global_bias=0.125)
    exact = oracle.predict_f64(nodes, T, D, data, MISSING)
    abs_sum = oracle.abs_leaf_sum(nodes, T, D, data, MISSING)
    cpu_err = np.abs(want_sums.astype(np.float64) - exact)

    # chain: the last rank holds THE sequential float32 sum, bit for bit; after the broadcast every rank does
    assert parts[0]["result_rank_chain"] == parts[1]["result_rank_chain"] == world - 1
    assert np.array_equal(parts[1]["sums_chain"].view(np.uint32), want_sums.view(np.uint32))
    for p in parts:
        assert np.array_equal(p["preds_chain"].view(np.uint32), want.view(np.uint32))
    # rank 0 of a chain holds the prefix over its own trees
    prefix, _ = oracle.predict(nodes[: parts[0]["hi"] * oracle.tree_num_nodes(D)], int(parts[0]["hi"]), D, data, MISSING)
    assert np.array_equal(parts[0]["sums_chain"].view(np.uint32), prefix.view(np.uint32))

    # allreduce64: both ranks hold the same total; it obeys the stated bound and is at least as close to the float64
    # sum as the CPU's own sequential float32 sum (max over rows)
    assert np.array_equal(parts[0]["sums_allreduce64"], parts[1]["sums_allreduce64"])
    got64 = parts[0]["sums_allreduce64"].astype(np.float64)
    bound = sharding.sum_error_bound(abs_sum, exact, trees_per_shard=(T + world - 1) // world)
    assert np.all(np.abs(got64 - exact) <= bound)
    assert np.max(np.abs(got64 - exact)) <= np.max(cpu_err) + 1e-12
    # ... and is exactly round(float64(partial_0) + float64(partial_1))
    p0, _ = oracle.predict(nodes[: parts[0]["hi"] * oracle.tree_num_nodes(D)], int(parts[0]["hi"]), D, data, MISSING)
    p1, _ = oracle.predict(nodes[parts[0]["hi"] * oracle.tree_num_nodes(D):], T - int(parts[0]["hi"]), D, data, MISSING)
    assert np.array_equal(parts[0]["sums_allreduce64"], (p0.astype(np.float64) + p1.astype(np.float64)).astype(np.float32))
    # allreduce32: float32 add of the two partials
    assert np.array_equal(parts[0]["sums_allreduce32"], p0 + p1)
    assert np.all(np.abs(parts[0]["sums_allreduce32"].astype(np.float64) - exact) <= bound + sharding.U32 * np.abs(exact))
    # the transform ran on the total
    assert np.array_equal(parts[0]["preds_allreduce64"], (parts[0]["sums_allreduce64"] / np.float32(T) + np.float32(0.125)))

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
    # both BASELINE forests fit one GPU: rows (bit-exact, no collective); only a forest beyond HBM is cut by trees
    assert sharding.choose_sharding(1000, 12) == "rows"      # K3
    assert sharding.choose_sharding(8000, 12) == "rows"      # K4: 2 GB of device layout
    assert sharding.choose_sharding(2_000_000, 12) == "trees"
    assert sharding.gamma(1000) > 1000 * sharding.U32
