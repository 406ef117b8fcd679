"""Every BASELINE.json configuration at its stated size, through the C ABI, on the GPU (K3 lives in
test_gpu_parity.py::test_k3_full_size_properties).  For each: the CPU oracle on a strided sample of rows (leaf indices and
float32 sums bit for bit; all rows where the oracle is cheap enough), and size-independent properties on every row --
two independent kernels agree, row-permutation equivariance, a ragged prefix equals the prefix of the full batch.

  K1  SUSY-like: 500 trees x depth 8, 18 features, 10 k rows, through the reference's text formats
  K2  SVHN-like: 500 trees x depth 8, 3072 features, 100 k rows (wide rows: 16-row tiles, several trees per wave)
  K4  8000 trees x depth 12, 256 features, 1 M rows on ONE GPU (tree groups), and as 8 tree shards combined the three ways
  K5  irregular sparse forest: 2000 trees of depth 4..24, 256 features, 200 k rows
  KR3 K3's shape from the histogram-style generator (<= 254 thresholds per feature): QRING on u8 codes, every row
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MISSING = -999.0


@pytest.fixture(scope="module")
def env(built):
    import torch

    import tahoe_amd as ta
    from oracle import oracle

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.cuda.set_device(0)
    return ta, oracle, torch


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def strided(R, step, extra=()):
    return np.unique(np.concatenate([np.arange(0, R, step), [R - 1, R - 2, max(R - 63, 0), max(R - 64, 0), max(R - 65, 0)],
                                     np.asarray(extra, dtype=np.int64)])).astype(np.int64)


def properties(torch, forest, x, got, ragged):
    """Permutation equivariance and a ragged prefix, on every row."""
    R = x.shape[0]
    perm = torch.randperm(R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    p = forest.predict_raw(x[perm].contiguous()).cpu().numpy()
    assert np.array_equal(bits(p), bits(got[perm.cpu().numpy()]))
    r = forest.predict_raw(x[:ragged].contiguous()).cpu().numpy()
    assert np.array_equal(bits(r), bits(got[:ragged]))


def test_k1_susy_like_through_text_files(env, tmp_path):
    ta, oracle, torch = env
    T, D, C, R = 500, 8, 18, 10_000
    nodes = ta.synth_forest(T, D, C, seed=11, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=12, missing_prob=0.02, missing=MISSING)
    ta.write_model(str(tmp_path / "m.txt"), nodes, T, D)
    ta.write_data(str(tmp_path / "d.txt"), data, MISSING)
    n2, T2, D2 = ta.load_model(str(tmp_path / "m.txt"))
    x2, miss = ta.load_data(str(tmp_path / "d.txt"))
    assert (T2, D2) == (T, D) and n2.tobytes() == nodes.tobytes() and x2.tobytes() == data.tobytes() and miss == MISSING
    want, want_leaf = oracle.predict(n2, T2, D2, x2, miss, want_leaf=True, threads=8)  # every row
    x = torch.from_numpy(x2).cuda()
    forest = ta.Forest(n2, T2, D2, C, missing=miss)
    for s in (ta.STRATEGY_AUTO, ta.STRATEGY_QRING, ta.STRATEGY_TILERING, ta.STRATEGY_DIRECT):
        forest.set_strategy(s)
        leaf, sums = forest.predict_leaf_idx(x)
        forest.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf), s
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want)), s
    forest.set_strategy(ta.STRATEGY_AUTO)
    properties(torch, forest, x, want, ragged=7_777)


def test_k2_svhn_like_wide_rows(env):
    ta, oracle, torch = env
    T, D, C, R = 500, 8, 3072, 100_000
    nodes = ta.synth_forest(T, D, C, seed=21)
    data = ta.synth_data(R, C, seed=22, missing_prob=0.001, missing=MISSING)
    x = torch.from_numpy(data).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=MISSING)
    # 2 * 500 * 8 lookups per row against 3072 features: too little walking to pay a quantise pass -> the float32 wide form
    assert forest.get_strategy(R) == ta.STRATEGY_TILERING and forest.info().ring_rows == 8
    assert forest.info().qring_tile_rows in (16, 32, 64)  # the quantised wide form stays available
    got = forest.predict_raw(x).cpu().numpy()
    forest.check()
    forest.set_strategy(ta.STRATEGY_QRING)
    assert np.array_equal(bits(forest.predict_raw(x).cpu().numpy()), bits(got))  # every row, the other wide form
    forest.set_strategy(ta.STRATEGY_AUTO)
    assert forest.info().stream_slots >= 4  # ... as the row-streaming kernel on 16-bit keys (wkey.hip)
    want_all, _ = oracle.predict(nodes, T, D, data, MISSING, threads=16)  # every row against the oracle (4e8 node visits)
    assert np.array_equal(bits(got), bits(want_all))
    idx = strided(R, 50)
    want, want_leaf = oracle.predict(nodes, T, D, data[idx], MISSING, want_leaf=True, threads=8)
    assert np.array_equal(bits(got[idx]), bits(want))
    leaf, _ = forest.predict_leaf_idx(x[torch.from_numpy(idx).cuda()].contiguous(), want_sums=False)
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    forest.set_strategy(ta.STRATEGY_DIRECT)  # an independent kernel (float32 compares, no quantisation) on every row
    assert np.array_equal(bits(forest.predict_raw(x).cpu().numpy()), bits(got))
    forest.set_strategy(ta.STRATEGY_AUTO)
    properties(torch, forest, x, got, ragged=33_333)


def test_k4_8000_trees_one_gpu_and_tree_shards(env):
    ta, oracle, torch = env
    from tahoe_amd import sharding

    T, D, C, R = 8000, 12, 256, 1_000_000
    per_tree = ta.capi.tree_num_nodes(D)
    nodes = ta.synth_forest(T, D, C, seed=45)
    data = ta.synth_data(R, C, seed=43, missing_prob=0.001, missing=MISSING)
    x = torch.from_numpy(data).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert forest.get_strategy(R) == ta.STRATEGY_QRING and forest.info().qring_groups >= 2
    got = forest.predict_raw(x).cpu().numpy()
    forest.check()
    idx = strided(R, 997)[:1100]
    want, want_leaf = oracle.predict(nodes, T, D, data[idx], MISSING, want_leaf=True, threads=8)
    assert np.array_equal(bits(got[idx]), bits(want))
    # ... and every row of the first and the last 30 k (5.8e9 node visits on the host)
    for lo, hi in ((0, 30_000), (R - 30_000, R)):
        w, _ = oracle.predict(nodes, T, D, data[lo:hi], MISSING, threads=32)
        assert np.array_equal(bits(got[lo:hi]), bits(w)), (lo, hi)
    xs = x[torch.from_numpy(idx).cuda()].contiguous()
    leaf, _ = forest.predict_leaf_idx(xs[:256].contiguous(), want_sums=False)
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf[:256])
    # an independent kernel on a slice (DIRECT at 8000 trees: ~1.6 s per 1M rows)
    forest.set_strategy(ta.STRATEGY_DIRECT)
    assert np.array_equal(bits(forest.predict_raw(x[:50_000].contiguous()).cpu().numpy()), bits(got[:50_000]))
    forest.set_strategy(ta.STRATEGY_AUTO)
    properties(torch, forest, x, got, ragged=100_003)
    # row shards are bit-exact by construction: rank 3 of 8
    lo, hi = sharding.shard_bounds(R, 3, 8)
    assert np.array_equal(bits(forest.predict_raw(x[lo:hi].contiguous()).cpu().numpy()), bits(got[lo:hi]))
    forest.close()

    # 8 tree shards on a 200 k-row slice, the three ways of combining them (what N ranks compute, run here shard by shard)
    Rs = 200_000
    xsl = x[:Rs].contiguous()
    exact = oracle.predict_f64_mt(nodes, T, D, data[idx[idx < Rs]], MISSING, threads=8)
    a = oracle.abs_leaf_sum(nodes, T, D, data[idx[idx < Rs]], MISSING, threads=8)
    sel = idx[idx < Rs]
    chain = torch.zeros(Rs, dtype=torch.float32, device="cuda")
    acc64 = torch.zeros(Rs, dtype=torch.float64, device="cuda")
    acc32 = torch.zeros(Rs, dtype=torch.float32, device="cuda")
    for k in range(8):
        t_lo, t_hi = sharding.shard_bounds(T, k, 8)
        fk = ta.Forest(nodes[t_lo * per_tree: t_hi * per_tree], t_hi - t_lo, D, C, missing=MISSING)
        part = fk.predict_raw(xsl)
        acc64 += part.double()
        acc32 += part
        for c_lo in range(0, Rs, 65_536):  # chunked hand-over, as the pipeline does
            c_hi = min(c_lo + 65_536, Rs)
            fk.predict_accumulate(xsl[c_lo:c_hi], chain[c_lo:c_hi])
        fk.check()
        fk.close()
    # chain: THE sequential float32 sum, every row
    assert np.array_equal(bits(chain.cpu().numpy()), bits(got[:Rs]))
    # all-reduce of float64 partials: within the stated bound of the float64 sum, and no further from it than the CPU's
    # own float32 sum; the float32 all-reduce is looser.  Neither is within 1e-6 relative of the CPU float32 sum.
    g64 = acc64.float().cpu().numpy()[sel].astype(np.float64)
    g32 = acc32.cpu().numpy()[sel].astype(np.float64)
    cpu32 = got[sel].astype(np.float64)
    bound = sharding.sum_error_bound(a, exact, trees_per_shard=1000)
    assert np.all(np.abs(g64 - exact) <= bound)
    assert np.max(np.abs(g64 - exact)) <= np.max(np.abs(cpu32 - exact))
    assert np.all(np.abs(g32 - exact) <= bound + 8 * sharding.U32 * np.abs(exact))
    rel = np.abs(g64 - cpu32) / np.maximum(np.abs(cpu32), 1e-30)
    assert rel.max() > 1e-6, "if this ever holds, say so in DESIGN.md 7: the all-reduce would meet north_star's bar"


def test_k5_irregular_sparse_forest(env):
    ta, oracle, torch = env
    T, C, R = 2000, 256, 200_000
    sn, tr = ta.capi.synth_sparse_forest(T, C, 4, 24, 0.32, 65535, 44)
    sizes = np.diff(np.append(tr, sn.size))
    assert sizes.max() > 20_000 and sizes.min() < 200  # depth 4 .. 24: small and large trees mixed
    data = ta.synth_data(R, C, seed=43, missing_prob=0.001, missing=MISSING)
    x = torch.from_numpy(data).cuda()
    forest = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
    got = forest.predict_raw(x).cpu().numpy()
    forest.check()
    want_all, _ = oracle.sparse_predict(sn, tr, data, MISSING, threads=32)  # every row against the oracle
    assert np.array_equal(bits(got), bits(want_all))
    idx = strided(R, 400)
    want, want_leaf = oracle.sparse_predict(sn, tr, data[idx], MISSING, want_leaf=True, threads=8)
    assert np.array_equal(bits(got[idx]), bits(want))
    leaf, _ = forest.predict_leaf_idx(x[torch.from_numpy(idx).cuda()].contiguous(), want_sums=False)
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    assert forest.get_strategy(R) == ta.STRATEGY_QRING  # the walk on quantised codes
    for s in (ta.STRATEGY_DIRECT, ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK):  # the float32 sparse kernels, every row
        forest.set_strategy(s)
        assert np.array_equal(bits(forest.predict_raw(x).cpu().numpy()), bits(got)), s
    forest.set_strategy(ta.STRATEGY_AUTO)
    properties(torch, forest, x, got, ragged=77_777)
    # continued sums on a sparse handle too (chained shards of a sparse forest)
    base = torch.full((1000,), 0.5, dtype=torch.float32, device="cuda")
    cont = forest.predict_accumulate(x[:1000].contiguous(), base.clone()).cpu().numpy()
    one, _ = oracle.sparse_predict(sn, tr, data[:1000], MISSING)
    assert not np.array_equal(bits(cont), bits(one))  # it did start from 0.5 ...
    assert np.allclose(cont, one + 0.5, rtol=0, atol=2e-4)  # ... and is the same sum shifted (bit-exact check: test_gpu_parity)


def test_kr3_histogram_style_forest_at_k3_size(env, monkeypatch):
    """KR3: K3's shape (1000 trees of depth 12, 256 features, 1 M rows) from the histogram-style generator -- 254 quantile
    thresholds per feature, Zipf-skewed feature usage, skewed branch probabilities, early leaves, 0.1 % missing values: what
    trained models look like (run_all_15_examples.sh:51-65).  AUTO walks it on u8 rank codes in 384-row tiles, ONE tree group
    (round 3 cut such forests by node counts); EVERY row against the oracle, the u16 form and DIRECT on every row / a slice."""
    ta, oracle, torch = env
    monkeypatch.delenv("TAHOE_QRING_CODE8", raising=False)
    T, D, C, R = 1000, 12, 256, 1_000_000
    nodes = ta.synth_forest_hist(T, D, C, seed=42, feature_seed=7, max_bins=254, zipf_s=1.0, leaf_prob=0.02, scale_decades=3.0)
    data = ta.synth_data_hist(R, C, seed=43, feature_seed=7, scale_decades=3.0, missing_prob=0.001, missing=MISSING)
    x = torch.from_numpy(data).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=MISSING)
    info = forest.info()
    assert forest.get_strategy(R) == ta.STRATEGY_QRING and forest.kernel_form(R) == "qring_region8"
    assert info.qring_groups == 1 and info.qring_tile_rows == 384
    got = forest.predict_raw(x).cpu().numpy()
    forest.check()
    want_all, _ = oracle.predict(nodes, T, D, data, MISSING, threads=min(os.cpu_count() or 8, 256))  # every row: 1.2e10 visits at most
    assert np.array_equal(bits(got), bits(want_all))
    idx = strided(R, 4001)
    want, want_leaf = oracle.predict(nodes, T, D, data[idx], MISSING, want_leaf=True, threads=8)
    leaf, sums = forest.predict_leaf_idx(x[torch.from_numpy(idx).cuda()].contiguous())
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf) and np.array_equal(bits(sums.cpu().numpy()), bits(want))
    forest.set_strategy(ta.STRATEGY_DIRECT)  # an independent kernel (float32 compares) on a slice
    assert np.array_equal(bits(forest.predict_raw(x[:100_000].contiguous()).cpu().numpy()), bits(got[:100_000]))
    forest.set_strategy(ta.STRATEGY_AUTO)
    properties(torch, forest, x, got, ragged=333_333)
    forest.close()
    monkeypatch.setenv("TAHOE_QRING_CODE8", "0")  # the u16 form of the same forest, every row
    f16 = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f16.kernel_form(R).startswith("qring_region") and f16.kernel_form(R) != "qring_region8"
    assert np.array_equal(bits(f16.predict_raw(x).cpu().numpy()), bits(got))
    f16.check()
    f16.close()
