"""Sparse (irregular) forests -- the reference's sparse_node_t format (Struct.h:50-54, 2217-2353), BASELINE config 5.
CPU: the converter and the generator against the oracle's restatement.  GPU: the HIP walk against the oracle."""
import numpy as np
import pytest

from oracle import oracle

MISSING = -999.0


@pytest.fixture(scope="module")
def ta(built):
    import tahoe_amd

    return tahoe_amd


def test_dense_to_sparse_matches_the_oracle_converter_and_predicts_like_dense(ta):
    T, D, C, R = 9, 6, 11, 300
    nodes = ta.synth_forest(T, D, C, seed=3, leaf_prob=0.2)
    data = ta.synth_data(R, C, seed=4, missing_prob=0.1, missing=MISSING, nan_prob=0.05)
    sn, tr = ta.capi.dense_to_sparse(nodes, T, D)
    on, ot = oracle.dense_to_sparse(nodes, T, D)
    assert sn.tobytes() == on.tobytes() and tr.tobytes() == ot.tobytes()
    # children are adjacent and after their parent; every tree starts at its root offset
    inner = (sn["bits"].view(np.uint32) >> 31) == 0
    assert (sn["left_idx"][inner] > 0).all()
    dense_pred, _ = oracle.predict(nodes, T, D, data, MISSING)
    sparse_pred, leaf = oracle.sparse_predict(sn, tr, data, MISSING, want_leaf=True)
    assert np.array_equal(dense_pred.view(np.uint32), sparse_pred.view(np.uint32))
    # the leaf a row ends in holds the value the dense walk returned
    t0 = sn["val"][tr[0] + leaf[:, 0]]
    _, dleaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True)
    assert np.array_equal(t0, nodes["val"][dleaf[:, 0]])


def test_irregular_generator_shape(ta):
    sn, tr = ta.capi.synth_sparse_forest(60, 32, min_depth=4, max_depth=24, leaf_prob=0.32, max_tree_nodes=65535, seed=44)
    again, tr2 = ta.capi.synth_sparse_forest(60, 32, min_depth=4, max_depth=24, leaf_prob=0.32, max_tree_nodes=65535, seed=44)
    assert sn.tobytes() == again.tobytes() and tr.tobytes() == tr2.tobytes()
    sizes = np.diff(np.append(tr, sn.size))
    assert sizes.min() >= 31 and sizes.max() <= 65535 and len(set(sizes.tolist())) > 10  # >= depth 4, irregular
    bits = sn["bits"].view(np.uint32)
    inner = (bits >> 31) == 0
    assert ((bits[inner] & ((1 << 30) - 1)) < 32).all()
    # depth of every tree by walking the links
    for t in range(0, 60, 7):
        lo = tr[t]
        depth = {0: 0}
        for i in range(sizes[t]):
            if inner[lo + i]:
                li = sn["left_idx"][lo + i]
                depth[li] = depth[li + 1] = depth[i] + 1
        assert 4 <= max(depth.values()) <= 24


@pytest.mark.gpu
def test_sparse_walk_on_gpu(ta):
    import torch

    assert torch.cuda.is_available()
    cases = []
    # (1) converted dense forest: must reproduce the dense prediction bit for bit
    T, D, C, R = 23, 7, 40, 1500
    nodes = ta.synth_forest(T, D, C, seed=5, leaf_prob=0.15)
    sn, tr = ta.capi.dense_to_sparse(nodes, T, D)
    data = ta.synth_data(R, C, seed=6, missing_prob=0.05, missing=MISSING, nan_prob=0.02)
    dense_pred, _ = oracle.predict(nodes, T, D, data, MISSING)
    cases.append((sn, tr, C, data, dense_pred))
    # (2) irregular K5-like forest, small; (3) wide rows: the 64-row tile does not fit LDS -> features from global
    for (nt, cols, rows, seed) in ((200, 64, 3000, 44), (30, 3072, 200, 45)):
        sn2, tr2 = ta.capi.synth_sparse_forest(nt, cols, 4, 24, 0.32, 65535, seed)
        d2 = ta.synth_data(rows, cols, seed=seed + 1, missing_prob=0.05, missing=MISSING, nan_prob=0.02)
        cases.append((sn2, tr2, cols, d2, None))
    for sn_, tr_, cols, data_, dense_expect in cases:
        want, want_leaf = oracle.sparse_predict(sn_, tr_, data_, MISSING, want_leaf=True, threads=8)
        if dense_expect is not None:
            assert np.array_equal(want.view(np.uint32), dense_expect.view(np.uint32))
        f = ta.capi.SparseForest(sn_, tr_, cols, missing=MISSING)
        assert f.info().is_sparse == 1
        x = torch.from_numpy(data_).cuda()
        strategies = [ta.STRATEGY_AUTO, ta.STRATEGY_DIRECT]
        for extra in (ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK):  # tile in LDS / tile + tree tops in LDS
            try:
                f.set_strategy(extra)
                strategies.append(extra)
            except ta.TahoeError as err:
                assert err.status == 7 and cols > 512  # unavailable only when the tile does not fit
        if cols <= 512:
            assert ta.STRATEGY_TILEBLOCK in strategies
            f.set_strategy(ta.STRATEGY_QRING)  # the walk on quantised codes: num_cols <= 256
            strategies.append(ta.STRATEGY_QRING)
            f.set_strategy(ta.STRATEGY_AUTO)  # QRING when there is enough walking per feature value to pay the quantise pass
            n_trees = len(tr_)
            assert f.get_strategy(len(data_)) == ta.STRATEGY_TILEBLOCK  # a small batch: the 64-row tiles fill more of the chip
            big = 64 * f.info().num_cus
            assert f.get_strategy(big) == (ta.STRATEGY_QRING if 20 * n_trees >= 13 * cols else ta.STRATEGY_TILEBLOCK)
        else:
            with pytest.raises(ta.TahoeError):
                f.set_strategy(ta.STRATEGY_QRING)
        for strategy in strategies:
            f.set_strategy(strategy)
            leaf, sums = f.predict_leaf_idx(x)
            raw = f.predict_raw(x)
            f.check()
            assert np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf)
            assert np.array_equal(sums.cpu().numpy().view(np.uint32), want.view(np.uint32))
            assert np.array_equal(raw.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # hand-made: a root that is a leaf, a 3-node tree, an orphan pair nobody links to, fewer trees than waves
    nd = np.zeros(8, dtype=sn.dtype)
    LEAF = np.int32(-2**31)
    nd["val"][:] = [7.0, 0.5, -1.0, 2.0, 100.0, 200.0, 3.0, 0.0]
    nd["bits"][:] = [LEAF, 1, LEAF, LEAF, LEAF, LEAF, LEAF, LEAF]  # tree 1 splits on feature 1 at 0.5
    nd["left_idx"][:] = [0, 1, 0, 0, 0, 0, 0, 0]
    roots = np.array([0, 1, 6], dtype=np.int32)  # tree 1 = nodes 1..5 (4, 5 unreachable), tree 2 = nodes 6..7
    xs = np.array([[0.0, 0.4], [0.0, 0.5], [0.0, np.nan], [0.0, MISSING]], dtype=np.float32)
    want, want_leaf = oracle.sparse_predict(nd, roots, xs, MISSING, want_leaf=True)
    assert want.tolist() == [9.0, 12.0, 9.0, 12.0]  # 7 + (x1 >= 0.5 or missing&!def_left ? 2 : -1) + 3
    f = ta.capi.SparseForest(nd, roots, 2, missing=MISSING)
    for strategy in (ta.STRATEGY_QRING, ta.STRATEGY_TILEBLOCK, ta.STRATEGY_ROWTILE, ta.STRATEGY_DIRECT):
        f.set_strategy(strategy)
        leaf, sums = f.predict_leaf_idx(torch.from_numpy(xs).cuda())
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf)
    # a tree with more than 65536 reachable nodes does not fit the compact 16-bit links: the 12-byte-node kernels serve
    big, big_tr = ta.capi.synth_sparse_forest(3, 8, 16, 17, 0.0, 300_000, 77)
    assert np.diff(np.append(big_tr, big.size)).max() > 65536
    bx = ta.synth_data(500, 8, seed=78, missing_prob=0.05, missing=MISSING)
    want, want_leaf = oracle.sparse_predict(big, big_tr, bx, MISSING, want_leaf=True, threads=8)
    f = ta.capi.SparseForest(big, big_tr, 8, missing=MISSING)
    assert f.get_strategy(500) == ta.STRATEGY_ROWTILE
    for unavailable in (ta.STRATEGY_TILEBLOCK, ta.STRATEGY_QRING):
        with pytest.raises(ta.TahoeError):
            f.set_strategy(unavailable)
    leaf, sums = f.predict_leaf_idx(torch.from_numpy(bx).cuda())
    assert np.array_equal(sums.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf)
    # malformed forests are rejected (the reference would walk out of the arrays or spin)
    bad = sn.copy()
    first_inner = int(np.flatnonzero((bad["bits"].view(np.uint32) >> 31) == 0)[0])
    bad["left_idx"][first_inner] = 0  # child link pointing backwards / at itself
    with pytest.raises(ta.TahoeError) as e:
        ta.capi.SparseForest(bad, tr, C)
    assert e.value.status == 6


@pytest.mark.gpu
def test_sparse_quantised_walk_tree_groups_and_tile_plans(ta):
    """QRING on a sparse handle: a feature with more than 32767 distinct thresholds cuts the forest into tree groups whose
    float32 sums are chained; batches that end in a partly filled tile, 192- and 128-row tile forms, missing / NaN / inf
    inputs, continued sums, and a batch after a larger one (workspace re-use)."""
    import torch

    cols = 2
    sn, tr = ta.capi.synth_sparse_forest(48, cols, 10, 14, 0.05, 65535, 91)
    inner = (sn["bits"].view(np.uint32) >> 31) == 0
    per_feature = np.bincount(sn["bits"][inner] & ((1 << 30) - 1), minlength=cols)
    assert per_feature.max() > 40_000  # more than one group's worth of thresholds on a feature
    rows = 5000
    data = ta.synth_data(rows, cols, seed=92, missing_prob=0.1, missing=MISSING, nan_prob=0.05)
    data[7, 0], data[8, 1], data[9, 0] = np.inf, -np.inf, -0.0
    want, want_leaf = oracle.sparse_predict(sn, tr, data, MISSING, want_leaf=True, threads=8)
    f = ta.capi.SparseForest(sn, tr, cols, missing=MISSING)
    f.set_strategy(ta.STRATEGY_QRING)
    x = torch.from_numpy(data).cuda()
    for n in (rows, 1, 63, 64, 65, 127, 129, 191, 193, 385, 4097):
        leaf, sums = f.predict_leaf_idx(x[:n].contiguous())
        raw = f.predict_raw(x[:n].contiguous())
        f.check()
        assert np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf[:n]), n
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), want[:n].view(np.uint32)), n
        assert np.array_equal(raw.cpu().numpy().view(np.uint32), want[:n].view(np.uint32)), n
    # continued sums: the forest cut in two handles, the second continues the first one's float32 sums
    cut = 24
    fa = ta.capi.SparseForest(sn[: tr[cut]], tr[:cut], cols, missing=MISSING)
    fb = ta.capi.SparseForest(sn[tr[cut]:], tr[cut:] - tr[cut], cols, missing=MISSING)
    for h in (fa, fb):
        h.set_strategy(ta.STRATEGY_QRING)
    acc = fa.predict_raw(x)
    fb.predict_accumulate(x, acc)
    fa.check()
    fb.check()
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # a wider forest, without missing values (the single-compare fast path), both tile forms forced
    import os
    sn2, tr2 = ta.capi.synth_sparse_forest(300, 256, 4, 24, 0.32, 65535, 93)
    d2 = ta.synth_data(3000, 256, seed=94)
    want2, leaf2 = oracle.sparse_predict(sn2, tr2, d2, MISSING, want_leaf=True, threads=8)
    x2 = torch.from_numpy(d2).cuda()
    for chains in ("2", "3", None):
        if chains is None:
            os.environ.pop("TAHOE_QRING_CHAINS", None)
        else:
            os.environ["TAHOE_QRING_CHAINS"] = chains
        try:
            f2 = ta.capi.SparseForest(sn2, tr2, 256, missing=MISSING)
        finally:
            os.environ.pop("TAHOE_QRING_CHAINS", None)
        assert f2.get_strategy(1_000_000) == ta.STRATEGY_QRING
        f2.set_strategy(ta.STRATEGY_QRING)
        leaf, sums = f2.predict_leaf_idx(x2)
        f2.check()
        assert np.array_equal(leaf.cpu().numpy().view(np.uint32), leaf2), chains
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), want2.view(np.uint32)), chains
        f2.close()


@pytest.mark.gpu
def test_sparse_quantised_walk_vines_odd_widths_host_batches(ta):
    """Shapes that stress the layout of the quantised sparse walk: vines (every inner node has one leaf child: 40 levels, i.e.
    a long chain of two-level blocks below the 9-level top, early-leaf padding inside top and blocks), a tree that is one
    leaf, odd num_cols (single-feature quantise kernels) and num_cols = 1; predictions through the host-batch pipeline."""
    import torch

    rng = np.random.default_rng(5)
    LEAF = np.int32(-2**31)

    def vine(depth, cols, right_goes_on):
        """`depth` inner nodes in a chain: one child of each is a leaf, the other the next inner node (children adjacent, after
        their parent, left_idx relative to the root)"""
        n = np.zeros(2 * depth + 1, dtype=ta.capi.SPARSE_NODE_DTYPE)
        pos, nxt = 0, 1
        for d in range(depth):
            n["val"][pos] = rng.choice([-0.5, 0.0, -0.0, 0.25, np.nan, np.inf]) if d % 7 == 3 else rng.uniform(-1, 1)
            n["bits"][pos] = int(rng.integers(0, cols)) | (int(rng.integers(0, 2)) << 30)
            n["left_idx"][pos] = nxt
            go, stop = (nxt + 1, nxt) if right_goes_on else (nxt, nxt + 1)
            n["val"][stop], n["bits"][stop] = rng.uniform(-2, 2), LEAF
            if d == depth - 1:
                n["val"][go], n["bits"][go] = rng.uniform(-2, 2), LEAF
            pos, nxt = go, nxt + 2
        return n

    for cols in (1, 7, 255):
        parts = [vine(40, cols, True), vine(33, cols, False), vine(9, cols, True), vine(10, cols, False), vine(1, cols, True)]
        one_leaf = np.zeros(1, dtype=ta.capi.SPARSE_NODE_DTYPE)
        one_leaf["val"], one_leaf["bits"] = 4.5, LEAF
        parts.insert(2, one_leaf)
        bushy, btr = ta.capi.synth_sparse_forest(12, cols, 3, 13, 0.25, 65535, 300 + cols)
        roots = np.cumsum([0] + [len(p) for p in parts]).astype(np.int32)
        nodes = np.concatenate(parts + [bushy])
        trees = np.concatenate([roots[:-1], btr + roots[-1]]).astype(np.int32)
        rows = 777
        data = ta.synth_data(rows, cols, seed=400 + cols, missing_prob=0.08, missing=MISSING, nan_prob=0.04)
        want, want_leaf = oracle.sparse_predict(nodes, trees, data, MISSING, want_leaf=True, threads=4)
        f = ta.capi.SparseForest(nodes, trees, cols, missing=MISSING)
        x = torch.from_numpy(data).cuda()
        for strategy in (ta.STRATEGY_QRING, ta.STRATEGY_TILEBLOCK):
            f.set_strategy(strategy)
            leaf, sums = f.predict_leaf_idx(x)
            f.check()
            assert np.array_equal(sums.cpu().numpy().view(np.uint32), want.view(np.uint32)), (cols, strategy)
            assert np.array_equal(leaf.cpu().numpy().view(np.uint32), want_leaf), (cols, strategy)
        f.set_strategy(ta.STRATEGY_QRING)
        host = f.predict_host(data, chunk_rows=256)  # chunked upload + traversal
        f.check()
        assert np.array_equal(host.view(np.uint32), want.view(np.uint32)), cols
        f.close()
