"""Parity of the HIP traversal (through the C ABI) against the CPU oracle.  Needs an MI355X.

Bars: leaf indices bit-exact; raw float32 sums bit-exact (the kernels add in the CPU's tree order);
outputs with a sigmoid within 1e-6 relative (device expf vs glibc expf), everything else bit-exact.
"""
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


def run_case(env, nodes, T, D, C, data, strategies=None, missing=MISSING, **out):
    ta, oracle, torch = env
    want, want_leaf = oracle.predict(nodes, T, D, data, missing, want_leaf=True)
    x = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=missing)
    info = forest.info()
    if strategies is None:
        strategies = [ta.STRATEGY_DIRECT] + ([ta.STRATEGY_ROWTILE] if info.lds_bytes_per_block > 0 else []) + (
            [ta.STRATEGY_TILEBLOCK] if info.tile_rows > 0 else []) + (
            [ta.STRATEGY_TILERING] if info.ring_rows > 0 else []) + (
            [ta.STRATEGY_QRING] if info.qring_walkers > 0 else []) + [ta.STRATEGY_AUTO]
    for s in strategies:
        forest.set_strategy(s)
        leaf, sums = forest.predict_leaf_idx(x)
        raw = forest.predict_raw(x)
        forest.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf), f"leaf index mismatch, strategy {s}"
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want)), f"sum (leaf pass) mismatch, strategy {s}"
        assert np.array_equal(bits(raw.cpu().numpy()), bits(want)), f"sum mismatch, strategy {s}"
    forest.close()
    return want, want_leaf


SHAPES = [
    # T, D, C, R, leaf_prob
    (1, 0, 1, 1, 0.0),
    (3, 1, 2, 5, 0.0),
    (4, 2, 3, 64, 0.3),
    (7, 3, 18, 100, 0.2),
    (50, 6, 32, 1000, 0.1),
    (5, 8, 17, 129, 0.0),       # cols not a multiple of 4 -> scalar tile staging
    (130, 8, 256, 777, 0.05),
    (21, 9, 64, 300, 0.0),      # one level below the LDS-staged top
    (20, 12, 256, 2000, 0.0),   # K3 shape, small
    (9, 12, 256, 511, 0.15),
    (33, 10, 500, 129, 0.0),    # 125 KiB tile: one workgroup per CU
    (6, 7, 3072, 200, 0.0),     # K2-like width: no 64-row float32 tile -> 8-row float32 tiles (TILERING), 16-row u16 tiles (QRING)
    (3, 14, 40, 97, 0.02),
]


@pytest.mark.parametrize("T,D,C,R,leaf_prob", SHAPES)
def test_random_forests(env, T, D, C, R, leaf_prob):
    ta = env[0]
    nodes = ta.synth_forest(T, D, C, seed=100 + T, leaf_prob=leaf_prob)
    data = ta.synth_data(R, C, seed=200 + R, missing_prob=0.05, missing=MISSING, nan_prob=0.02)
    run_case(env, nodes, T, D, C, data)


def test_threshold_ties_and_missing_band(env):
    """x == threshold goes right; |x - missing| <= 1e-6 (float32) takes the default branch, both signs of
    def_left; values just outside the band compare normally; NaN and +-inf features."""
    ta = env[0]
    T, D, C, R = 16, 5, 8, 512
    nodes = ta.synth_forest(T, D, C, seed=5)
    rng = np.random.default_rng(0)
    thr = nodes["val"][rng.integers(0, nodes.size, size=(R, C))]  # rows made of actual thresholds
    data = thr.astype(np.float32)
    m = np.float32(0.25)  # a missing sentinel in the data range so the band is resolvable in float32
    band = np.array([m, m + np.float32(5e-7), m - np.float32(5e-7), m + np.float32(1e-6), m - np.float32(1e-6),
                     m + np.float32(2e-6), m - np.float32(2e-6), np.nextafter(m + np.float32(1e-6), np.float32(1))],
                    dtype=np.float32)
    data[:64, :] = band[rng.integers(0, band.size, size=(64, C))]
    data[64:80, ::2] = np.nan
    data[80:96, 1::3] = np.inf
    data[96:112, ::3] = -np.inf
    run_case(env, nodes, T, D, C, data, missing=float(m))


def test_nan_missing_sentinel(env):
    """missing = NaN: the band test is never true (NaN compares false), NaN features go left."""
    ta = env[0]
    T, D, C, R = 8, 4, 6, 200
    nodes = ta.synth_forest(T, D, C, seed=9)
    data = ta.synth_data(R, C, seed=10, nan_prob=0.2)
    run_case(env, nodes, T, D, C, data, missing=float("nan"))


def test_early_leaves_with_garbage_below(env):
    """A leaf above the bottom level ends the walk; whatever the file holds below it is never read."""
    ta = env[0]
    T, D, C, R = 12, 7, 10, 400
    nodes = ta.synth_forest(T, D, C, seed=11, leaf_prob=0.5)  # root itself is a leaf in about half the trees
    data = ta.synth_data(R, C, seed=12, missing_prob=0.1, missing=MISSING)
    want, leaf = run_case(env, nodes, T, D, C, data)
    n = ta.capi.tree_num_nodes(D)
    assert (leaf < (n >> 1)).any(), "case must contain leaves above the bottom level"


def test_hand_built_tree(env):
    """Known answers derived by hand: depth-2 tree over 2 features.
         node0: f0 >= 0.5 ? right : left        (def_left = 1)
         node1: f1 >= -1.0                      (def_left = 0)   node2: leaf 10.0 (early leaf)
         node3: leaf 1.0   node4: leaf 2.0      nodes 5,6 garbage (below the leaf)"""
    ta, oracle, torch = env
    enc = ta.capi.encode_nodes
    nodes = enc(fid=[0, 1, 0, 0, 0, 1, 1], value=[0.5, -1.0, 10.0, 1.0, 2.0, 77.0, 88.0],
                def_left=[1, 0, 0, 0, 0, 0, 0], weight=[0] * 7, is_leaf=[0, 0, 1, 1, 1, 0, 0])
    data = np.array([[0.5, 0.0],      # tie at node0 -> right -> leaf node2
                     [0.4, -1.0],     # left; tie at node1 -> right -> node4
                     [0.4, -1.5],     # left; left -> node3
                     [MISSING, 5.0],  # missing at node0, def_left -> left; node1 right -> node4
                     [0.0, MISSING],  # left; missing at node1, def_left=0 -> right -> node4
                     [np.nan, np.nan]], dtype=np.float32)  # NaN fails >= -> left, left -> node3
    want_leaf = np.array([[2], [4], [3], [4], [4], [3]], dtype=np.uint32)
    want = np.array([10.0, 2.0, 1.0, 2.0, 2.0, 1.0], dtype=np.float32)
    o_pred, o_leaf = oracle.predict(nodes, 1, 2, data, MISSING, want_leaf=True)
    assert np.array_equal(o_leaf, want_leaf) and np.array_equal(o_pred, want)
    got, got_leaf = run_case(env, nodes, 1, 2, 2, data)
    assert np.array_equal(got_leaf, want_leaf) and np.array_equal(got, want)


@pytest.mark.parametrize("output,threshold,bias", [
    (0x1, 0.0, 0.0), (0x0, 0.0, 0.75), (0x1, 0.0, -0.5), (0x10, 0.0, 0.0), (0x11, 0.0, 0.1),
    (0x100, 0.3, 0.0), (0x101, 0.01, 0.0), (0x111, 0.5, 0.05),
])
def test_output_transforms(env, output, threshold, bias):
    ta, oracle, torch = env
    T, D, C, R = 37, 5, 20, 999
    nodes = ta.synth_forest(T, D, C, seed=21)
    data = ta.synth_data(R, C, seed=22, missing_prob=0.02, missing=MISSING)
    want, _ = oracle.predict(nodes, T, D, data, MISSING, output=output, threshold=threshold, global_bias=bias)
    forest = ta.Forest(nodes, T, D, C, missing=MISSING, output=output, threshold=threshold, global_bias=bias)
    got = forest.predict(torch.from_numpy(data).cuda()).cpu().numpy()
    if output & 0x10 and not output & 0x100:
        # sigmoid: device expf vs glibc expf, tolerance 1e-6 relative
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)
    elif output & 0x10:
        # thresholded sigmoid: may only differ where the sigmoid is within 1e-6 relative of the threshold
        raw, _ = oracle.predict(nodes, T, D, data, MISSING, output=output & ~0x100, global_bias=bias)
        differ = got != want
        assert np.all(np.abs(raw[differ] - threshold) <= 1e-6 * abs(threshold))
    else:
        assert np.array_equal(bits(got), bits(want))


def test_empty_and_degenerate(env):
    ta, oracle, torch = env
    nodes = ta.synth_forest(3, 2, 4, seed=1)
    forest = ta.Forest(nodes, 3, 2, 4, missing=MISSING)
    empty = torch.empty((0, 4), dtype=torch.float32, device="cuda")
    assert forest.predict(empty).numel() == 0
    leaf, sums = forest.predict_leaf_idx(empty)
    assert leaf.shape == (0, 3)
    # a forest with no trees predicts 0 (+ bias) for every row
    f0 = ta.Forest(np.empty(0, dtype=ta.NODE_DTYPE), 0, 3, 4, missing=MISSING, global_bias=0.5)
    x = torch.from_numpy(ta.synth_data(10, 4, seed=2)).cuda()
    assert np.array_equal(f0.predict(x).cpu().numpy(), np.full(10, 0.5, dtype=np.float32))


def test_invalid_forests_are_rejected(env):
    ta = env[0]
    enc = ta.capi.encode_nodes
    # reachable bottom-level node that is not a leaf: the reference would walk out of the tree
    nodes = enc([0, 0, 0], [0.0, 1.0, 2.0], [0, 0, 0], [0, 0, 0], [0, 1, 0])
    with pytest.raises(ta.TahoeError) as e:
        ta.Forest(nodes, 1, 1, 1)
    assert e.value.status == 6
    # fid out of range
    nodes = enc([5, 0, 0], [0.0, 1.0, 2.0], [0, 0, 0], [0, 0, 0], [0, 1, 1])
    with pytest.raises(ta.TahoeError) as e:
        ta.Forest(nodes, 1, 1, 3)
    assert e.value.status == 6
    # ROWTILE cannot hold a 3072-column tile
    f = ta.Forest(ta.synth_forest(2, 3, 3072, seed=3), 2, 3, 3072)
    with pytest.raises(ta.TahoeError) as e:
        f.set_strategy(ta.STRATEGY_ROWTILE)
    assert e.value.status == 7
    # too wide for a 64-row float32 tile: TILERING is the wide-row form (8-row float32 tiles, eight trees per wave), which AUTO
    # takes for so little walking per feature value; the quantised walk (16-row u16 tiles) stays available
    assert f.get_strategy(100) == ta.STRATEGY_TILERING and f.info().ring_rows == 8 and f.info().qring_walkers == 15


def test_golden_fixtures(env):
    """Committed model/data text files -> expected sums and leaf indices (tests/golden/README.md)."""
    import glob
    import os

    ta, oracle, torch = env
    here = os.path.join(os.path.dirname(__file__), "golden")
    cases = sorted(glob.glob(os.path.join(here, "*.model.txt")))
    assert cases, "no golden fixtures"
    for model in cases:
        stem = model[: -len(".model.txt")]
        nodes, T, D = ta.load_model(model)
        data, missing = ta.load_data(stem + ".data.txt")
        exp = np.load(stem + ".expected.npz")
        forest = ta.Forest(nodes, T, D, data.shape[1], missing=missing)
        for s in (ta.STRATEGY_DIRECT, ta.STRATEGY_AUTO):
            forest.set_strategy(s)
            leaf, sums = forest.predict_leaf_idx(torch.from_numpy(data).cuda())
            assert np.array_equal(bits(leaf.cpu().numpy()), exp["leaf_idx"]), stem
            assert np.array_equal(bits(sums.cpu().numpy()), exp["sums_bits"]), stem


def test_k3_full_size_properties(env):
    """BASELINE config 3 at full size (1000 trees, depth 12, 256 features, 1M rows): every sum against the oracle, leaf
    indices on a strided sample of rows, and size-independent properties on all rows."""
    ta, oracle, torch = env
    T, D, C, R = 1000, 12, 256, 1_000_000
    nodes = ta.synth_forest(T, D, C, seed=42)
    data = ta.synth_data(R, C, seed=43)
    x = torch.from_numpy(data).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=MISSING)
    sums = forest.predict_raw(x)
    torch.cuda.synchronize()
    got = sums.cpu().numpy()
    # (1) oracle on 1500 strided rows (first, last, ragged tail of the last 64-row tile included)
    idx = np.unique(np.concatenate([np.arange(0, R, 997), [R - 1, R - 2, R - 63, R - 64, R - 65]]))[:1500]
    want, want_leaf = oracle.predict(nodes, T, D, data[idx], MISSING, want_leaf=True, threads=8)
    assert np.array_equal(bits(got[idx]), bits(want))
    leaf, _ = forest.predict_leaf_idx(x[torch.from_numpy(idx).cuda()].contiguous(), want_sums=False)
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    # (1b) EVERY row of the batch against the oracle (1e6 rows x 1000 trees x 12 levels = 1.2e10 node visits on the host)
    want_all, _ = oracle.predict(nodes, T, D, data, MISSING, threads=32)
    assert np.array_equal(bits(got), bits(want_all))
    # (2) two independent kernels agree on every row
    forest.set_strategy(ta.STRATEGY_DIRECT)
    d_sums = forest.predict_raw(x[: 200_000].contiguous()).cpu().numpy()
    assert np.array_equal(bits(d_sums), bits(got[:200_000]))
    forest.set_strategy(ta.STRATEGY_AUTO)
    # (3) row-permutation equivariance: predict(x[perm]) == predict(x)[perm]
    perm = torch.randperm(R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    p_sums = forest.predict_raw(x[perm].contiguous()).cpu().numpy()
    assert np.array_equal(bits(p_sums), bits(got[perm.cpu().numpy()]))
    # (4) a ragged batch (not a multiple of the 64-row tile) matches the prefix of the full batch
    r_sums = forest.predict_raw(x[: 100_003].contiguous()).cpu().numpy()
    assert np.array_equal(bits(r_sums), bits(got[:100_003]))


def test_predict_accumulate_continues_the_sum_on_every_strategy(env):
    """tahoe_forest_predict_accumulate: the per-row accumulator starts from the given running sums, so two shards
    chained in tree order reproduce the single sequential float32 sum bit for bit -- dense strategies, tree groups,
    wide rows, and a sparse handle."""
    ta, oracle, torch = env
    rng = np.random.default_rng(5)
    for (T, D, C, R, lp) in [(50, 6, 32, 1000, 0.1), (700, 8, 4, 500, 0.0), (40, 7, 600, 300, 0.0), (9, 3, 5, 77, 0.2)]:
        nodes = ta.synth_forest(T, D, C, seed=7, leaf_prob=lp)
        data = ta.synth_data(R, C, seed=8, missing_prob=0.05, missing=MISSING)
        x = torch.from_numpy(data).cuda()
        start = rng.standard_normal(R).astype(np.float32)
        want = oracle.predict_continue(nodes, T, D, data, MISSING, start.copy())
        forest = ta.Forest(nodes, T, D, C, missing=MISSING)
        info = forest.info()
        strategies = [ta.STRATEGY_DIRECT, ta.STRATEGY_AUTO] + ([ta.STRATEGY_ROWTILE] if info.lds_bytes_per_block > 0 else []) + (
            [ta.STRATEGY_TILEBLOCK] if info.tile_rows > 0 else []) + ([ta.STRATEGY_TILERING] if info.ring_rows > 0 else []) + (
            [ta.STRATEGY_QRING] if info.qring_walkers > 0 else [])
        for s in strategies:
            forest.set_strategy(s)
            got = forest.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
            forest.check()
            assert np.array_equal(bits(got.cpu().numpy()), bits(want)), (T, D, C, s)
        # two shards chained == the whole forest
        per = ta.capi.tree_num_nodes(D)
        cut = T // 3
        a = ta.Forest(nodes[: cut * per], cut, D, C, missing=MISSING)
        b = ta.Forest(nodes[cut * per:], T - cut, D, C, missing=MISSING)
        run = torch.zeros(R, dtype=torch.float32, device="cuda")
        a.predict_accumulate(x, run)
        b.predict_accumulate(x, run)
        whole, _ = oracle.predict(nodes, T, D, data, MISSING)
        assert np.array_equal(bits(run.cpu().numpy()), bits(whole))
    # sparse handle
    T, D, C, R = 30, 7, 20, 400
    nodes = ta.synth_forest(T, D, C, seed=17, leaf_prob=0.15)
    data = ta.synth_data(R, C, seed=18, missing_prob=0.05, missing=MISSING)
    sn, tr = ta.capi.dense_to_sparse(nodes, T, D)
    start = rng.standard_normal(R).astype(np.float32)
    want = oracle.predict_continue(nodes, T, D, data, MISSING, start.copy())
    f = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
    for s in (ta.STRATEGY_AUTO, ta.STRATEGY_ROWTILE, ta.STRATEGY_DIRECT):
        f.set_strategy(s)
        got = f.predict_accumulate(torch.from_numpy(data).cuda(), torch.from_numpy(start.copy()).cuda())
        f.check()
        assert np.array_equal(bits(got.cpu().numpy()), bits(want)), s


def test_region_form_two_and_three_chains(env, monkeypatch):
    """QRING's region layout (num_cols <= 256): tiles of three 64-row regions (14 walkers x 3 chains) or two (15 x 2), picked
    per batch; both forced here, plus the 128-slot column layout it replaces, on ragged row counts, with missing values
    confined to a few rows so that a 192-row tile straddles a quantise chunk that saw none and one that did."""
    ta, oracle, torch = env
    T, D, C, R = 45, 9, 200, 3000
    nodes = ta.synth_forest(T, D, C, seed=81, leaf_prob=0.03)
    data = ta.synth_data(R, C, seed=82)
    data[500:530, ::7] = MISSING      # quantise chunks are >= 512 rows: rows 384..575 form a 192-row tile across chunks 0 and 1
    data[2000, 3] = MISSING
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=4)
    x = torch.from_numpy(data).cuda()
    # (this forest has < 255 thresholds per feature: left alone, large tiles would be walked on u8 codes -- the last three settings;
    # TAHOE_QRING_CODE8=0 keeps the u16 forms this test is about)
    u16 = {"TAHOE_QRING_CODE8": "0"}
    for env_set in (dict(u16, TAHOE_QRING_CHAINS="3"), dict(u16, TAHOE_QRING_CHAINS="2"), dict(u16, TAHOE_QRING_REGIONS="0"), u16,
                    {"TAHOE_QRING_CHAINS": "3"}, {"TAHOE_QRING_CHAINS": "2"}, {}):
        for k, v in env_set.items():
            monkeypatch.setenv(k, v)
        forest = ta.Forest(nodes, T, D, C, missing=MISSING)
        forest.set_strategy(ta.STRATEGY_QRING)
        assert forest.info().qring_tile_rows == (128 if "TAHOE_QRING_REGIONS" in env_set else 192 if "TAHOE_QRING_CODE8" in env_set else 384)
        for rows in (1, 63, 64, 65, 191, 192, 193, 385, 577, 1000, R):
            leaf, sums = forest.predict_leaf_idx(x[:rows].contiguous())
            raw = forest.predict_raw(x[:rows].contiguous())
            forest.check()
            assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf[:rows]), (env_set, rows)
            assert np.array_equal(bits(sums.cpu().numpy()), bits(want[:rows])), (env_set, rows)
            assert np.array_equal(bits(raw.cpu().numpy()), bits(want[:rows])), (env_set, rows)
        cont = forest.predict_accumulate(x, torch.full((R,), 0.25, dtype=torch.float32, device="cuda"))
        assert np.array_equal(bits(cont.cpu().numpy()),
                              bits(oracle.predict_continue(nodes, T, D, data, MISSING, np.full(R, 0.25, dtype=np.float32))))
        forest.close()
        for k in env_set:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("code8", ["0", "1"])
def test_batches_walked_as_waves_of_192_row_tiles_plus_a_128_row_remainder(env, monkeypatch, code8):
    """Mid-size batches of the region form run as whole waves of 192-row tiles followed by a remainder of 128-row tiles
    (two launches over disjoint row ranges; 125 k rows on 256 CUs: 98,304 + 26,696) -- on u8 codes (code8 = 1; this forest has
    ~120 thresholds per feature) as whole waves of 384-row tiles + a remainder of 128-row tiles (125 k rows: the same cut).
    Every row against the oracle, leaf indices on the rows around the cut, running sums continued across it."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_QRING_CODE8", code8)
    T, D, C, R = 30, 8, 64, 300_000
    nodes = ta.synth_forest(T, D, C, seed=95, leaf_prob=0.02)
    data = ta.synth_data(R, C, seed=96, missing_prob=0.0002, missing=MISSING)
    want, _ = oracle.predict(nodes, T, D, data, MISSING, threads=8)
    x = torch.from_numpy(data).cuda()
    forest = ta.Forest(nodes, T, D, C, missing=MISSING)
    forest.set_strategy(ta.STRATEGY_QRING)
    # (64 features: regions at a 16-KiB stride -- six chains per lane on u16 codes too, the same 384 + 128 plan)
    assert forest.kernel_form(125_000) == ("qring_region8" if code8 == "1" else "qring_region6")
    for rows in (110_000, 125_000, 147_457, 250_000, R):
        got = forest.predict_raw(x[:rows].contiguous())
        forest.check()
        assert np.array_equal(bits(got.cpu().numpy()), bits(want[:rows])), rows
    lo, hi = 98_304 - 300, 98_304 + 300  # around the cut of the 125 k-row plan
    _, want_leaf = oracle.predict(nodes, T, D, data[lo:hi], MISSING, want_leaf=True)
    leaf, _ = forest.predict_leaf_idx(x[:125_000].contiguous(), want_sums=False)
    assert np.array_equal(bits(leaf[lo:hi].cpu().numpy()), want_leaf)
    start = np.linspace(-1, 1, 125_000).astype(np.float32)
    cont = oracle.predict_continue(nodes, T, D, data[:125_000], MISSING, start.copy(), threads=8)
    got = forest.predict_accumulate(x[:125_000].contiguous(), torch.from_numpy(start.copy()).cuda())
    assert np.array_equal(bits(got.cpu().numpy()), bits(cont))


def test_small_batches_walked_in_tree_slices(env, monkeypatch):
    """QRING's SPLIT form (the counterpart of the reference's split-forest strategy idx 4): a batch with fewer tiles than CUs
    gives every tile to several workgroups, each a slice of the trees, and a second kernel adds the leaf values per row in
    tree order -- same bits as the sequential sum.  Forced slice counts and the automatic choice, tree groups, continued sums."""
    ta, oracle, torch = env
    rng = np.random.default_rng(9)
    for (T, D, C, R, lp) in [(500, 8, 18, 10_000, 0.05), (130, 6, 64, 777, 0.1), (700, 8, 4, 500, 0.0), (61, 3, 7, 129, 0.2)]:
        nodes = ta.synth_forest(T, D, C, seed=91, leaf_prob=lp)
        data = ta.synth_data(R, C, seed=92, missing_prob=0.02, missing=MISSING)
        want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
        start = rng.standard_normal(R).astype(np.float32)
        cont = oracle.predict_continue(nodes, T, D, data, MISSING, start.copy(), threads=8)
        x = torch.from_numpy(data).cuda()
        for slices in (None, "1", "2", "3", "8", "64"):
            if slices is None:
                monkeypatch.delenv("TAHOE_QRING_SLICES", raising=False)
            else:
                monkeypatch.setenv("TAHOE_QRING_SLICES", slices)
            forest = ta.Forest(nodes, T, D, C, missing=MISSING)
            forest.set_strategy(ta.STRATEGY_QRING)
            assert forest.info().qring_tile_rows in (192, 384)  # (384: few thresholds per feature, large batches on u8 codes)
            for rows in sorted({1, 65, 128, 129, R}):
                if rows > R:
                    continue
                leaf, sums = forest.predict_leaf_idx(x[:rows].contiguous())
                raw = forest.predict_raw(x[:rows].contiguous())
                forest.check()
                assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf[:rows]), (T, C, slices, rows)
                assert np.array_equal(bits(sums.cpu().numpy()), bits(want[:rows])), (T, C, slices, rows)
                assert np.array_equal(bits(raw.cpu().numpy()), bits(want[:rows])), (T, C, slices, rows)
            got = forest.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
            assert np.array_equal(bits(got.cpu().numpy()), bits(cont)), (T, C, slices)
            forest.close()
    monkeypatch.delenv("TAHOE_QRING_SLICES", raising=False)


def test_probability_relayout_keeps_every_result(env):
    """SURVEY 8f N3 (Struct.h:1775-1825): with TAHOE_CREATE_PROB_RELAYOUT the subtrees are ordered by dense_node_t.weight and
    marked nodes invert their condition; leaf indices (original heap numbering) and float32 sums must not change."""
    ta, oracle, torch = env
    cases = [(60, 8, 32, 1500, 0.1, True), (40, 12, 256, 700, 0.0, True), (25, 5, 7, 300, 0.3, False), (12, 2, 3, 200, 0.0, False),
             (30, 7, 600, 257, 0.05, True), (3, 0, 2, 10, 0.0, False), (5, 1, 4, 65, 0.0, False)]
    for (T, D, C, R, lp, prob_weights) in cases:
        nodes = ta.synth_forest(T, D, C, seed=51, leaf_prob=lp)  # weight = a random number per node
        if prob_weights:
            ta.capi.set_probability_weights(nodes, T, D)          # weight = probability of reaching the node
        data = ta.synth_data(R, C, seed=52, missing_prob=0.05, missing=MISSING, nan_prob=0.01)
        want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True)
        x = torch.from_numpy(data).cuda()
        forest = ta.Forest(nodes, T, D, C, missing=MISSING, relayout=True)
        info = forest.info()
        assert info.relayout == 1 and (D == 0 or info.relayout_swaps > 0)
        assert info.tile_rows == 0 and info.ring_rows == 0  # the float32 block views have no room for the exchange bit
        if C <= 256:
            assert info.qring_walkers == 15
        else:
            assert info.qring_walkers == 0
            with pytest.raises(ta.TahoeError):
                forest.set_strategy(ta.STRATEGY_QRING)
        with pytest.raises(ta.TahoeError):
            forest.set_strategy(ta.STRATEGY_TILEBLOCK)
        strategies = [ta.STRATEGY_DIRECT, ta.STRATEGY_AUTO] + ([ta.STRATEGY_ROWTILE] if info.lds_bytes_per_block > 0 else []) + (
            [ta.STRATEGY_QRING] if info.qring_walkers > 0 else [])
        for s in strategies:
            forest.set_strategy(s)
            leaf, sums = forest.predict_leaf_idx(x)
            raw = forest.predict_raw(x)
            forest.check()
            assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf), (T, D, C, s)
            assert np.array_equal(bits(sums.cpu().numpy()), bits(want)), (T, D, C, s)
            assert np.array_equal(bits(raw.cpu().numpy()), bits(want)), (T, D, C, s)
        # without missing values the walk kernel takes its fast path (MS = false): check that one too
        clean = ta.synth_data(R, C, seed=53)
        w2, _ = oracle.predict(nodes, T, D, clean, MISSING)
        forest.set_strategy(ta.STRATEGY_AUTO)
        assert np.array_equal(bits(forest.predict_raw(torch.from_numpy(clean).cuda()).cpu().numpy()), bits(w2))
        forest.close()
    # the likelier child is the left one afterwards: with probability weights, rows concentrate on low leaf positions
    T, D, C, R = 8, 10, 16, 4000
    nodes = ta.capi.set_probability_weights(ta.synth_forest(T, D, C, seed=61), T, D)
    data = ta.synth_data(R, C, seed=62)
    plain, relaid = ta.Forest(nodes, T, D, C, missing=MISSING), ta.Forest(nodes, T, D, C, missing=MISSING, relayout=True)
    x = torch.from_numpy(data).cuda()
    a, _ = plain.predict_leaf_idx(x, want_sums=False)
    b, _ = relaid.predict_leaf_idx(x, want_sums=False)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())  # same leaves, in the original numbering


# ---- the rank-quantised path (QRING): exactness at the edges of the float order ----

def test_quantised_threshold_edge_values(env):
    """Thresholds and features drawn from {+-0, +-inf, NaN, denormals, duplicates, neighbours in float order};
    the missing sentinel is itself one of the thresholds.  Codes must reproduce x >= thr for every pair."""
    ta, oracle, torch = env
    rng = np.random.default_rng(3)
    T, D, C, R = 24, 6, 6, 1500
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 1.0, np.nextafter(np.float32(1), np.float32(2)),
                        np.nextafter(np.float32(1), np.float32(0)), -1.0, 0.5, 0.5, 0.5, 3.4028235e38, -3.4028235e38, 0.25],
                       dtype=np.float32)
    nodes = ta.synth_forest(T, D, C, seed=31, leaf_prob=0.05)
    inner = (nodes["bits"].view(np.uint32) >> 31) == 0
    nodes["val"][inner] = special[rng.integers(0, special.size, size=int(inner.sum()))]
    data = special[rng.integers(0, special.size, size=(R, C))]
    info_missing = 0.25  # a threshold value that is also the missing sentinel
    want, want_leaf = run_case(env, nodes, T, D, C, data, missing=info_missing)
    f = ta.Forest(nodes, T, D, C, missing=info_missing)
    assert f.info().qring_walkers > 0 and f.get_strategy(R) == ta.STRATEGY_QRING


def test_qring_tree_groups_and_unavailable(env):
    """A forest whose features see more than 32767 distinct thresholds is quantised in groups of consecutive
    trees; the running float32 sums are chained through the groups, so sums and leaf indices stay bit-exact.
    A single tree that alone exceeds the limit makes the strategy step aside (AUTO falls back to float32)."""
    ta, oracle, torch = env
    T, D, C, R = 5, 15, 1, 300  # 32767 inner nodes per tree, all on feature 0, all distinct -> one tree per group
    nodes = ta.synth_forest(T, D, C, seed=41)
    data = ta.synth_data(R, C, seed=42, missing_prob=0.05, missing=MISSING)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.info().qring_walkers > 0 and f.info().qring_groups == T
    run_case(env, nodes, T, D, C, data)
    # a few hundred trees per group
    T, D, C, R = 700, 8, 4, 500
    nodes = ta.synth_forest(T, D, C, seed=43)
    data = ta.synth_data(R, C, seed=44, missing_prob=0.05, missing=MISSING)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert 2 <= f.info().qring_groups <= 4 and f.get_strategy(R) == ta.STRATEGY_QRING
    run_case(env, nodes, T, D, C, data, strategies=[ta.STRATEGY_QRING, ta.STRATEGY_TILERING])
    # one tree with 65535 distinct thresholds on one feature: unavailable
    T, D, C, R = 2, 16, 1, 100
    nodes = ta.synth_forest(T, D, C, seed=45)
    data = ta.synth_data(R, C, seed=46)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.info().qring_walkers == 0
    with pytest.raises(ta.TahoeError) as e:
        f.set_strategy(ta.STRATEGY_QRING)
    assert e.value.status == 7
    assert f.get_strategy(R) in (ta.STRATEGY_TILERING, ta.STRATEGY_TILEBLOCK, ta.STRATEGY_ROWTILE)
    run_case(env, nodes, T, D, C, data)


def test_qring_missing_only_in_a_late_chunk_and_workspace_growth(env):
    """Rows are quantised in chunks of 32768; a chunk without missing values takes the single-compare fast
    path.  Missing values only in the third chunk exercise both paths in one launch; batches of growing and
    shrinking size exercise the grow-only workspace."""
    ta, oracle, torch = env
    T, D, C, R = 10, 7, 12, 80_000
    nodes = ta.synth_forest(T, D, C, seed=51, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=52)
    late = ta.synth_data(R - 70_000, C, seed=53, missing_prob=0.2, missing=MISSING, nan_prob=0.05)
    data[70_000:] = late
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.set_strategy(ta.STRATEGY_QRING)
    for n in (1000, R, 129, 40_000):
        x = torch.from_numpy(data[R - n:].copy()).cuda()
        want, want_leaf = oracle.predict(nodes, T, D, data[R - n:], MISSING, want_leaf=True, threads=8)
        leaf, sums = f.predict_leaf_idx(x)
        f.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf), n
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want)), n


def test_handles_of_different_shapes_coexist(env):
    """The dynamic-LDS attribute is per kernel and process-wide: a small forest created after a large one must not
    lower it under the large one's launch size.  Both handles stay usable in any order, on every strategy."""
    ta, oracle, torch = env
    big_n = ta.synth_forest(12, 12, 256, seed=61)
    big_x = ta.synth_data(700, 256, seed=62, missing_prob=0.05, missing=MISSING)
    small_n = ta.synth_forest(5, 4, 3, seed=63)
    small_x = ta.synth_data(300, 3, seed=64, missing_prob=0.05, missing=MISSING)
    big = ta.Forest(big_n, 12, 12, 256, missing=MISSING)
    small = ta.Forest(small_n, 5, 4, 3, missing=MISSING)
    want_big = oracle.predict(big_n, 12, 12, big_x, MISSING)[0]
    want_small = oracle.predict(small_n, 5, 4, small_x, MISSING)[0]
    bx, sx = torch.from_numpy(big_x).cuda(), torch.from_numpy(small_x).cuda()
    for s in (ta.STRATEGY_AUTO, ta.STRATEGY_ROWTILE, ta.STRATEGY_TILEBLOCK, ta.STRATEGY_TILERING, ta.STRATEGY_QRING):
        big.set_strategy(s)
        small.set_strategy(s)
        got_s = small.predict_raw(sx)
        got_b = big.predict_raw(bx)
        big.check()
        small.check()
        assert np.array_equal(bits(got_b.cpu().numpy()), bits(want_big)), s
        assert np.array_equal(bits(got_s.cpu().numpy()), bits(want_small)), s
    big.close()
    small.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_host_batches_through_the_upload_pipeline(env, pinned):
    """tahoe_forest_predict_host: chunked upload overlapped with the traversal gives the same bits as one resident
    batch -- several chunks with a ragged last one, a single chunk, chunk > rows, pageable and pinned sources,
    and an output transform (applied per chunk on the device)."""
    ta, oracle, torch = env
    T, D, C, R = 40, 9, 24, 70_001
    nodes = ta.synth_forest(T, D, C, seed=71, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=72, missing_prob=0.03, missing=MISSING, nan_prob=0.01)
    want = oracle.predict(nodes, T, D, data, MISSING, threads=8)[0]
    want_avg = oracle.predict(nodes, T, D, data, MISSING, output=ta.OUT_AVG, global_bias=0.25, threads=8)[0]
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    g = ta.Forest(nodes, T, D, C, missing=MISSING, output=ta.OUT_AVG, global_bias=0.25)
    pin = None
    if pinned:
        pin = ta.PinnedArray(R, C)
        pin.array[:] = data
        src = pin.array
    else:
        src = data
    for chunk in (0, 32768, 10_000, R, 3 * R):
        got = f.predict_host(src, chunk_rows=chunk)
        assert np.array_equal(bits(got), bits(want)), chunk
    for n in (1, 129, 20_000):  # shrinking batches reuse the buffers
        got = f.predict_host(np.ascontiguousarray(src[:n]), chunk_rows=4096)
        assert np.array_equal(bits(got), bits(want[:n])), n
    got = g.predict_host(src, chunk_rows=16384)
    assert np.array_equal(bits(got), bits(want_avg))
    assert f.predict_host(np.empty((0, C), dtype=np.float32)).shape == (0,)
    f.close()
    g.close()
    if pin is not None:
        pin.close()


@pytest.mark.parametrize("form", ["buckets", "tree", "multi"])
def test_quantiser_on_skewed_thresholds(env, monkeypatch, form):
    """The bucketed quantiser maps x to a run of the sorted thresholds with a monotone linear map; a skewed
    distribution (almost everything in a sliver of the range, a few huge outliers, one constant feature, one
    feature with only infinite thresholds) makes some runs long but may not change a single code.  Feature values
    sit on thresholds, one ulp either side of them, and far outside.  All quantiser forms against the oracle."""
    ta, oracle, torch = env
    # the three quantise kernels: bucketed pair form, search-tree pair form, many-features-per-workgroup form
    monkeypatch.setenv("TAHOE_QUANT_MULTI", "1" if form == "multi" else "0")
    monkeypatch.setenv("TAHOE_QUANT_BUCKETS", "0" if form == "tree" else "1")
    rng = np.random.default_rng(11)
    T, D, C, R = 60, 9, 8, 6000
    nodes = ta.synth_forest(T, D, C, seed=81)
    bits_u = nodes["bits"].view(np.uint32)
    inner = (bits_u >> 31) == 0
    fid = bits_u & ((1 << 30) - 1)
    n_inner = int(inner.sum())
    thr = np.empty(n_inner, dtype=np.float32)
    f_in = fid[inner]
    u = rng.random(n_inner)
    thr[:] = (1.0 + 1e-4 * rng.standard_normal(n_inner)).astype(np.float32)          # a sliver around 1
    thr[u < 0.02] = (rng.standard_normal(int((u < 0.02).sum())) * 1e30).astype(np.float32)  # outliers
    thr[f_in == 1] = np.float32(0.75)                                                 # constant feature
    thr[f_in == 2] = np.where(rng.random(int((f_in == 2).sum())) < 0.5, np.inf, -np.inf).astype(np.float32)
    thr[f_in == 3] = np.exp(rng.uniform(-80, 80, int((f_in == 3).sum()))).astype(np.float32)  # log-uniform
    thr[f_in == 4] = (rng.integers(0, 5, int((f_in == 4).sum())) * 1e-42).astype(np.float32)  # denormals, ties
    nodes["val"][inner] = thr
    pool = np.concatenate([thr, np.nextafter(thr, np.float32(np.inf)), np.nextafter(thr, np.float32(-np.inf)),
                           np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 3.4e38, -3.4e38, MISSING], dtype=np.float32)])
    data = pool[rng.integers(0, pool.size, size=(R, C))].astype(np.float32)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.get_strategy(R) == ta.STRATEGY_QRING
    f.close()
    run_case(env, nodes, T, D, C, data, strategies=[ta.STRATEGY_QRING, ta.STRATEGY_TILERING])


@pytest.mark.parametrize("T,D,C,R", [(120, 8, 18, 10_000),    # K1-like: memset + quantise + walk (tree slices + ordered sum) + transform
                                     (200, 7, 2048, 5_000),   # wide rows, the row-streaming form: its leaf-value workspace is reserved
                                     (150, 8, 64, 60_000)])   # histogram-style forest, a large batch: u8 codes, 384-row tiles + remainder
def test_predict_is_capturable_in_a_hip_graph(env, T, D, C, R):
    """A predict on a reserved handle is a fixed sequence of stream operations (a memset, the quantise and walk
    kernels, the output transform): no allocation, no synchronisation.  It can therefore be captured into a
    hipGraph once and replayed -- the way a launch-bound caller (K1: 10 k rows, three launches per batch) removes
    the per-launch host cost.  Replays must give the oracle's bits, also after the input buffer changes."""
    ta, oracle, torch = env
    nodes = (ta.synth_forest_hist(T, D, C, seed=91, feature_seed=3, max_bins=200, scale_decades=0.0) if R > 50_000 else
             ta.synth_forest(T, D, C, seed=91, leaf_prob=0.05))
    f = ta.Forest(nodes, T, D, C, missing=MISSING, output=ta.OUT_AVG | ta.OUT_SIGMOID, global_bias=0.1)
    if C > 512:
        assert f.info().stream_slots > 0 and f.get_strategy(R) == ta.STRATEGY_TILERING
    if R > 50_000:
        assert f.kernel_form(R) == "qring_region8"
    f.reserve(R)
    x = torch.empty((R, C), dtype=torch.float32, device="cuda")
    out = torch.zeros(R, dtype=torch.float32, device="cuda")
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        f.predict(x, out, stream=torch.cuda.current_stream())
    for seed in (92, 93):
        data = ta.synth_data(R, C, seed=seed, missing_prob=0.05, missing=MISSING)
        x.copy_(torch.from_numpy(data))
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        f.check()
        want = oracle.predict(nodes, T, D, data, MISSING, output=ta.OUT_AVG | ta.OUT_SIGMOID, global_bias=0.1, threads=8)[0]
        got = out.cpu().numpy()
        assert np.allclose(got, want, rtol=1e-6, atol=0), seed  # sigmoid: device expf vs glibc expf
        eager = f.predict(x).cpu().numpy()
        assert np.array_equal(bits(got), bits(eager)), seed     # replay == eager launch, bit for bit
    f.close()


def test_fuzz_shapes_against_the_oracle(env):
    """Seeded random shapes (trees, depth, columns on both sides of 256 / 512 and odd, rows around tile and chunk
    multiples, early-leaf and missing / NaN rates), every available strategy, dense and converted-to-sparse:
    leaf indices and float32 sums must equal the oracle's bits in every case."""
    ta, oracle, torch = env
    rng = np.random.default_rng(2026)
    for case in range(40):
        T = int(rng.integers(1, 90))
        D = int(rng.integers(0, 12))
        C = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 255, 256, 257, 300, 511, 512, 513, 700]))
        R = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 255, 1000, 4097]))
        leaf_prob = float(rng.choice([0.0, 0.0, 0.1, 0.4]))
        miss = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
        nanp = float(rng.choice([0.0, 0.02]))
        nodes = ta.synth_forest(T, D, C, seed=1000 + case, leaf_prob=leaf_prob)
        data = ta.synth_data(R, C, seed=2000 + case, missing_prob=miss, missing=MISSING, nan_prob=nanp)
        want, want_leaf = run_case(env, nodes, T, D, C, data)
        if case % 4 == 0:  # the same forest through the sparse format
            sn, tr = ta.capi.dense_to_sparse(nodes, T, D)
            sw, sl = oracle.sparse_predict(sn, tr, data, MISSING, want_leaf=True)
            assert np.array_equal(bits(sw), bits(want)), case
            f = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
            leaf, sums = f.predict_leaf_idx(torch.from_numpy(data).cuda())
            assert np.array_equal(bits(sums.cpu().numpy()), bits(want)), case
            assert np.array_equal(bits(leaf.cpu().numpy()), sl), case
            f.close()


WIDE_SHAPES = [
    # T, D, C, R, leaf_prob: rows too wide for a 128-row u16 tile -> 64 / 32 / 16-row tiles, several trees per wave
    (6, 7, 3072, 200, 0.0),     # K2 width: 16-row tiles, 4 trees per wave
    (7, 12, 1200, 300, 0.0),    # 32-row tiles; the LDS slot holds 9 of the 10 top levels, level 10 from the heap
    (5, 13, 3072, 77, 0.1),     # 16-row tiles; levels 9..11 from the quantised heap in global memory
    (9, 3, 700, 130, 0.0),      # 64-row tiles, one tree per wave
    (3, 0, 2000, 50, 0.0),      # depth 0 -> De = 2: no top levels at all
    (130, 6, 1024, 1000, 0.2),  # more trees than two ring rounds
    (1, 2, 5000, 33, 0.0),      # not even 16 rows fit: features from the quantised tile in L2 (GX form)
    (67, 9, 2048, 4097, 0.05),  # tree count not a multiple of the trees per wave, rows not of the tile
]


@pytest.mark.parametrize("T,D,C,R,leaf_prob", WIDE_SHAPES)
def test_wide_rows(env, T, D, C, R, leaf_prob):
    ta = env[0]
    nodes = ta.synth_forest(T, D, C, seed=300 + T, leaf_prob=leaf_prob)
    data = ta.synth_data(R, C, seed=400 + R, missing_prob=0.05, missing=MISSING, nan_prob=0.02)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    want_rows = {3072: 16, 1200: 32, 700: 64, 2000: 32, 1024: 64, 5000: 0, 2048: 32}[C]
    assert f.info().qring_tile_rows == want_rows, (C, f.info().qring_tile_rows)
    f.close()
    run_case(env, nodes, T, D, C, data, strategies=[ta.STRATEGY_QRING, ta.STRATEGY_DIRECT, ta.STRATEGY_AUTO])
    # no missing values at all: the single-compare fast path
    clean = ta.synth_data(R, C, seed=500 + R)
    run_case(env, nodes, T, D, C, clean, strategies=[ta.STRATEGY_QRING])


def test_two_handles_on_two_streams_concurrently(env):
    """Handles share no mutable global state (the reference keeps its strategy and device buffers in globals,
    Struct.h:9-11): two forests of different shapes predicting back to back on two streams, several rounds in
    flight before anything synchronises, each give their oracle's bits."""
    ta, oracle, torch = env
    a_nodes = ta.synth_forest(200, 10, 64, seed=101, leaf_prob=0.05)
    b_nodes = ta.synth_forest(40, 6, 700, seed=102)
    a_data = ta.synth_data(20_000, 64, seed=103, missing_prob=0.05, missing=MISSING)
    b_data = ta.synth_data(3_000, 700, seed=104, missing_prob=0.05, missing=MISSING)
    a, b = ta.Forest(a_nodes, 200, 10, 64, missing=MISSING), ta.Forest(b_nodes, 40, 6, 700, missing=MISSING)
    a.reserve(20_000)
    b.reserve(3_000)
    ax, bx = torch.from_numpy(a_data).cuda(), torch.from_numpy(b_data).cuda()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs_a = [torch.empty(20_000, dtype=torch.float32, device="cuda") for _ in range(4)]
    outs_b = [torch.empty(3_000, dtype=torch.float32, device="cuda") for _ in range(4)]
    for i in range(4):
        a.predict_raw(ax, outs_a[i], stream=sa)
        b.predict_raw(bx, outs_b[i], stream=sb)
    a.check(sa)
    b.check(sb)
    want_a = oracle.predict(a_nodes, 200, 10, a_data, MISSING, threads=8)[0]
    want_b = oracle.predict(b_nodes, 40, 6, b_data, MISSING, threads=8)[0]
    for i in range(4):
        assert np.array_equal(bits(outs_a[i].cpu().numpy()), bits(want_a)), i
        assert np.array_equal(bits(outs_b[i].cpu().numpy()), bits(want_b)), i
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,C,R,missing_prob", [
    (300, 8, 3072, 1003, 0.0),    # K2's shape: 8-row tiles, eight trees per wave; more trees than ring entries; ragged last tile
    (300, 8, 3072, 1003, 0.03),   # ... with missing values and NaN: the full branch rule
    (37, 10, 2050, 333, 0.02),    # cols not a multiple of 4 (scalar staging); levels 6..7 from the heap in global memory
    (5, 2, 1000, 70, 0.05),       # two levels: no LDS top at all, one block per tree
    (9, 3, 700, 64, 0.0),         # 32-row tiles, one top level
    (66, 12, 600, 500, 0.01),     # deep trees on 32-row tiles: eight levels in LDS, two from the heap, two in the blocks
    (129, 9, 1500, 257, 0.0),     # 16-row tiles
])
def test_wide_rows_float32_form(env, monkeypatch, T, D, C, R, missing_prob):
    """TILERING on rows too wide for a 64-row float32 tile, in its tile form (widef.hip; TAHOE_WSTREAM=0 keeps the shapes the
    row-streaming form would take on it): leaf indices and sums against the oracle, continued sums, and the same bits as the
    quantised wide form and DIRECT."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_WSTREAM", "0")
    nodes = ta.synth_forest(T, D, C, seed=900 + T, leaf_prob=0.1 if D > 4 else 0.0)
    data = ta.synth_data(R, C, seed=901 + R, missing_prob=missing_prob, missing=MISSING, nan_prob=missing_prob / 2)
    if missing_prob:
        data[3, :5] = [np.inf, -np.inf, -0.0, MISSING + 5e-7, MISSING - 2e-6]
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.info().ring_rows in (8, 16, 32)
    f.close()
    want, _ = run_case(env, nodes, T, D, C, data, strategies=[ta.STRATEGY_TILERING, ta.STRATEGY_QRING, ta.STRATEGY_DIRECT, ta.STRATEGY_AUTO])
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.set_strategy(ta.STRATEGY_TILERING)
    x = torch.from_numpy(data).cuda()
    start = np.linspace(-2.0, 2.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    cont = oracle.predict_continue(nodes, T, D, data, MISSING, start)
    assert np.array_equal(bits(acc.cpu().numpy()), bits(cont))
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,C,R,missing_prob", [
    (300, 8, 3072, 1003, 0.03),   # K2's shape: six levels of all trees resident, 4 row slots; missing values and NaN
    (300, 8, 3072, 5, 0.0),       # fewer rows than CUs: one-row workgroups
    (37, 10, 2052, 333, 0.02),    # rows of 8 pieces + a 16-byte tail piece (a third chunk with three pieces past the end)
    (1000, 9, 1024, 700, 0.01),   # many trees: five resident levels, two levels in between on float32 values from global memory
    (65, 3, 640, 4097, 0.0),      # depth 3: one resident level; 65 trees = two chunks, the second one lane
    (9, 2, 700, 64, 0.05),        # depth 2: no resident level at all, the whole tree is its one bottom block
])
def test_wide_rows_streaming_form(env, monkeypatch, T, D, C, R, missing_prob):
    """TAHOE_WSTREAM=1: TILERING on wide rows as the row-streaming kernel on 16-bit keys whatever the shape (wkey.hip: tops of all
    trees resident in LDS, rows turned into keys on their way into a ring of LDS slots, lane = tree, equal keys decided on the
    float32 values): leaf indices, sums and continued sums against the oracle."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_WSTREAM", "1")
    nodes = ta.synth_forest(T, D, C, seed=950 + T, leaf_prob=0.1 if D > 4 else 0.0)
    data = ta.synth_data(R, C, seed=951 + R, missing_prob=missing_prob, missing=MISSING, nan_prob=missing_prob / 2)
    if missing_prob and R > 3:
        data[3, :5] = [np.inf, -np.inf, -0.0, MISSING + 5e-7, MISSING - 2e-6]
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    info = f.info()
    assert info.stream_slots >= 4 and 0 <= info.stream_levels <= max(D, 2) - 2
    assert info.tilering_lds_bytes <= 160 * 1024
    f.set_strategy(ta.STRATEGY_TILERING)
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    for _ in range(2):  # twice: the handle keeps no state between predicts
        leaf, sums = f.predict_leaf_idx(x)
        preds = f.predict_raw(x)
        f.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want))
        assert np.array_equal(bits(preds.cpu().numpy()), bits(want))
    start = np.linspace(-2.0, 2.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(oracle.predict_continue(nodes, T, D, data, MISSING, start)))
    # an unaligned row pointer (rows start 4 bytes into a buffer) cannot be DMA'd in 16-byte pieces: the tile form serves it
    if R > 1:
        flat = torch.empty(R * C + 1, dtype=torch.float32, device="cuda")
        flat[1:] = x.reshape(-1)
        shifted = flat[1:].view(R, C)
        assert shifted.data_ptr() % 16 != 0
        assert np.array_equal(bits(f.predict_raw(shifted).cpu().numpy()), bits(want))
        f.check()
    f.close()


@pytest.mark.gpu
def test_wide_rows_streaming_form_equal_keys(env, monkeypatch):
    """The 16-bit keys of the row-streaming form decide a compare only when they differ; equal keys fall back to the float32
    values.  Feature values and thresholds on a coarse grid, nudged by a few ulps either way (and exactly equal: ties go right),
    make most compares of this case equal-key compares; integer-valued features (pixels) against half-integer thresholds too."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_WSTREAM", "1")
    T, D, C, R = 150, 8, 1024, 3000
    rng = np.random.default_rng(77)
    nodes = ta.synth_forest(T, D, C, seed=970, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=971, missing_prob=0.01, missing=MISSING, nan_prob=0.005)
    grid = lambda a: (np.round(a * 32.0) / 32.0).astype(np.float32)
    nudge = lambda a: np.nextafter(a, np.float32(np.inf) * rng.choice([-1.0, 1.0], size=a.shape).astype(np.float32)).astype(np.float32)
    keep = np.isnan(data) | (data == MISSING)
    coarse = grid(data)
    mix = rng.random(data.shape)
    data2 = np.where(mix < 0.4, coarse, np.where(mix < 0.8, nudge(coarse), data)).astype(np.float32)
    data2[keep] = data[keep]
    nodes = nodes.copy()
    thr = grid(nodes["val"])
    tmix = rng.random(thr.shape)
    nodes["val"] = np.where(tmix < 0.5, thr, nudge(thr))  # leaf values change too: they are only added
    data2[:, :8] = np.floor(rng.random((R, 8)) * 256.0).astype(np.float32)  # "pixels"
    for sel in range(0, nodes.size, 97):  # some thresholds at half-integers, whatever feature they test
        nodes["val"][sel] = np.float32(rng.integers(0, 256)) + np.float32(0.5)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.info().stream_slots >= 4
    f.set_strategy(ta.STRATEGY_TILERING)
    want, want_leaf = oracle.predict(nodes, T, D, data2, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data2).cuda()
    leaf, sums = f.predict_leaf_idx(x)
    f.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    assert np.array_equal(bits(sums.cpu().numpy()), bits(want))
    f.set_strategy(ta.STRATEGY_DIRECT)
    assert np.array_equal(bits(f.predict_raw(x).cpu().numpy()), bits(want))
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["constant_thresholds", "infinite_and_nan_thresholds", "one_tree", "huge_values"])
def test_wide_rows_streaming_form_degenerate_key_maps(env, monkeypatch, case):
    """The key map of the row-streaming form is built from the range of the forest's finite thresholds: all thresholds equal
    (scale 0: every compare ties and is decided on the float32 values), +-inf / NaN thresholds (keys at the ends / 0xFFFF),
    a single tree, features far outside the thresholds' range (clamped keys)."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_WSTREAM", "1")
    T, D, C, R = (1 if case == "one_tree" else 70), 5, 768, 900
    nodes = ta.synth_forest(T, D, C, seed=980, leaf_prob=0.1).copy()
    data = ta.synth_data(R, C, seed=981, missing_prob=0.02, missing=MISSING, nan_prob=0.01)
    inner = nodes["bits"] >= 0  # is_leaf is the sign bit
    if case == "constant_thresholds":
        nodes["val"][inner] = np.float32(0.25)
        data[::3, ::5] = np.float32(0.25)
    elif case == "infinite_and_nan_thresholds":
        idx = np.flatnonzero(inner)
        nodes["val"][idx[::7]] = np.float32(np.inf)
        nodes["val"][idx[3::11]] = np.float32(-np.inf)
        nodes["val"][idx[5::13]] = np.float32(np.nan)
        data[::4, ::9] = np.float32(np.inf)
        data[1::4, ::9] = np.float32(-np.inf)
    elif case == "huge_values":
        data[::2] *= np.float32(1e30)
        data[1::5, ::3] = np.float32(3.0e38)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    assert f.info().stream_slots >= 4
    f.set_strategy(ta.STRATEGY_TILERING)
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    leaf, sums = f.predict_leaf_idx(x)
    f.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    assert np.array_equal(bits(sums.cpu().numpy()), bits(want))
    f.close()


@pytest.mark.gpu
def test_wide_rows_streaming_form_walks_large_batches_in_slabs(env, monkeypatch):
    """The leaf-value workspace of the row-streaming form is capped (1 GiB; here 1 MiB -> the floor of 64 rows per CU): a larger
    batch is walked slab by slab on the stream, also with continued sums and leaf indices."""
    ta, oracle, torch = env
    monkeypatch.setenv("TAHOE_WSTREAM", "1")
    monkeypatch.setenv("TAHOE_WSTREAM_SLAB_MB", "1")
    T, D, C = 40, 6, 640
    nodes = ta.synth_forest(T, D, C, seed=990, leaf_prob=0.1)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    R = f.info().num_cus * 64 * 2 + 777  # two full slabs and a ragged third
    data = ta.synth_data(R, C, seed=991, missing_prob=0.01, missing=MISSING)
    f.set_strategy(ta.STRATEGY_TILERING)
    assert f.info().stream_slots >= 4
    x = torch.from_numpy(data).cuda()
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    leaf, sums = f.predict_leaf_idx(x)
    f.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    assert np.array_equal(bits(sums.cpu().numpy()), bits(want))
    start = np.linspace(-1.0, 1.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(oracle.predict_continue(nodes, T, D, data, MISSING, start)))
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["two_scales", "one_outlier_threshold", "uniform"])
def test_wide_rows_form_rule_looks_at_key_resolution(env, monkeypatch, case):
    """The row-streaming form maps every feature with ONE affine 16-bit key map over the range of all thresholds.  A forest
    whose features live on different scales (or with one outlier threshold) would tie on most compares of its small-scale
    features -- bit-exact still, but on the slow float32 path.  The create-time rule estimates that (stream_key_ties) and
    leaves such forests to the tile form; forced with TAHOE_WSTREAM=1 the streaming form still reproduces the oracle."""
    ta, oracle, torch = env
    monkeypatch.delenv("TAHOE_WSTREAM", raising=False)
    T, D, C, R = 200, 8, 1024, 1500   # 3 T <= C, six resident levels: the shape rule alone would take the streaming form
    nodes = ta.synth_forest(T, D, C, seed=1201).copy()
    data = ta.synth_data(R, C, seed=1202, missing_prob=0.01, missing=MISSING)
    inner = nodes["bits"] >= 0
    fid = nodes["bits"] & ((1 << 30) - 1)
    if case == "two_scales":          # odd features: thresholds and values 1000 x larger
        big = inner & (fid % 2 == 1)
        nodes["val"][big] *= np.float32(1000.0)
        keep = (data == np.float32(MISSING))
        data[:, 1::2] *= np.float32(1000.0)
        data[keep] = np.float32(MISSING)
    elif case == "one_outlier_threshold":
        nodes["val"][np.flatnonzero(inner)[5]] = np.float32(1.0e6)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    info = f.info()
    if case == "uniform":
        assert info.stream_slots >= 4 and info.stream_key_ties < 1e-4 and f.kernel_form(R) != "tilering_wide_tile"
    else:
        assert info.stream_slots == 0 and info.stream_key_ties > 1e-4, (info.stream_slots, info.stream_key_ties)
    f.set_strategy(ta.STRATEGY_TILERING)
    assert f.kernel_form(R) == ("tilering_wide_stream" if case == "uniform" else "tilering_wide_tile")
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    leaf, sums = f.predict_leaf_idx(x)
    f.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf) and np.array_equal(bits(sums.cpu().numpy()), bits(want))
    f.close()
    monkeypatch.setenv("TAHOE_WSTREAM", "1")     # the streaming form on the poor key map: slow, exact
    g = ta.Forest(nodes, T, D, C, missing=MISSING)
    g.set_strategy(ta.STRATEGY_TILERING)
    assert g.info().stream_slots >= 4 and g.kernel_form(R) == "tilering_wide_stream"
    leaf, sums = g.predict_leaf_idx(x)
    g.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf) and np.array_equal(bits(sums.cpu().numpy()), bits(want))
    g.close()


HIST_SHAPES = [
    # T, D, C, R, max_bins, zipf_s, scale_decades
    (200, 8, 28, 4000, 255, 1.0, 3.0),     # HIGGS-like width
    (120, 6, 128, 3001, 63, 1.2, 4.0),     # few bins, strongly skewed usage
    (60, 10, 256, 2500, 255, 0.8, 2.0),    # deep trees on K3's width: early leaves everywhere
    (40, 7, 1024, 900, 255, 1.0, 5.0),     # wide rows with features on very different scales
    (500, 5, 54, 2000, 32, 1.5, 0.0),      # covtype-like: many shallow trees, one scale
]


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,C,R,bins,zipf,decades", HIST_SHAPES)
def test_histogram_style_forests(env, T, D, C, R, bins, zipf, decades):
    """Forests in the style of histogram-trained GBDT models (tahoe_synth_forest_hist: <= 255 quantile thresholds per feature,
    Zipf-skewed feature usage, skewed branch probabilities, early leaves, features on different scales and of different
    shapes incl. integer counts against half-integer thresholds), rows drawn from the same feature distributions: every
    strategy against the oracle, leaf indices and sums bit for bit."""
    ta = env[0]
    nodes = ta.synth_forest_hist(T, D, C, seed=300 + T, feature_seed=17, max_bins=bins, zipf_s=zipf, leaf_prob=0.02, scale_decades=decades)
    data = ta.synth_data_hist(R, C, seed=400 + R, feature_seed=17, scale_decades=decades, missing_prob=0.02, missing=MISSING)
    data[::97, ::5] = np.nan
    want, leaf = run_case(env, nodes, T, D, C, data)
    level = np.floor(np.log2(leaf.astype(np.float64) + 1.0))
    assert level.min() < D, "case must contain leaves above the bottom level"


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,C,R,bins,missing_prob", [
    (200, 8, 64, 40_003, 254, 0.02),     # several waves' worth of 384-row tiles + a ragged remainder, missing values (MS walk)
    (120, 10, 256, 33_000, 255, 0.0),    # K3's width; the generator caps at 255 edges, the group's busiest feature decides the form
    (300, 6, 28, 100_000, 63, 0.0),      # HIGGS-like width: most of every region is empty
    (40, 12, 128, 20_000, 254, 0.01),    # deep trees: 10-level tops + bottom blocks, early leaves
])
def test_qring_on_8bit_codes(env, monkeypatch, T, D, C, R, bins, missing_prob):
    """Forests whose features each see <= 254 distinct thresholds (histogram-trained models) are quantised to u8 rank codes for
    batches large enough for whole tiles: 128-row regions, 384-row tiles, six chains per lane (qring_kernel<..., CODE8>).  Leaf
    indices, sums and continued sums against the oracle; the u16 form (TAHOE_QRING_CODE8=0) gives the same bits."""
    ta, oracle, torch = env
    monkeypatch.delenv("TAHOE_QRING_CODE8", raising=False)
    nodes = ta.synth_forest_hist(T, D, C, seed=500 + T, feature_seed=23, max_bins=bins, zipf_s=1.0, leaf_prob=0.02, scale_decades=3.0)
    data = ta.synth_data_hist(R, C, seed=600 + T, feature_seed=23, scale_decades=3.0, missing_prob=missing_prob, missing=MISSING)
    if missing_prob:
        data[5::211, ::3] = np.nan
    inner = nodes["bits"] >= 0
    fid = nodes["bits"] & ((1 << 30) - 1)
    most = max(np.unique(nodes["val"][inner & (fid == f)]).size for f in range(C))
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.set_strategy(ta.STRATEGY_QRING)
    want8 = most <= 254
    assert (f.kernel_form(R) == "qring_region8") == want8, (most, f.kernel_form(R))
    assert f.info().qring_tile_rows == (384 if (want8 or C <= 128) else 192)
    # small batches of forests with enough trees keep the u16 tree slices (a slice must still give every walker a few trees)
    assert f.kernel_form(1000) == ("qring_split" if T >= 120 else "qring_region8" if want8 else "qring_region6" if C <= 128 else "qring_region2")
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    for _ in range(2):
        leaf, sums = f.predict_leaf_idx(x)
        raw = f.predict_raw(x)
        f.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want)) and np.array_equal(bits(raw.cpu().numpy()), bits(want))
    start = np.linspace(-1.0, 1.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(oracle.predict_continue(nodes, T, D, data, MISSING, start)))
    small = f.predict_raw(x[:1000].contiguous())  # then a small batch on the same handle (u16 codes in the same workspace), then large again
    assert np.array_equal(bits(small.cpu().numpy()), bits(want[:1000]))
    assert np.array_equal(bits(f.predict_raw(x).cpu().numpy()), bits(want))
    f.check()
    f.close()
    monkeypatch.setenv("TAHOE_QRING_CODE8", "0")
    g = ta.Forest(nodes, T, D, C, missing=MISSING)
    g.set_strategy(ta.STRATEGY_QRING)
    assert g.kernel_form(R) in (("qring_region6",) if C <= 128 else ("qring_region_mixed", "qring_region3", "qring_region2"))
    assert np.array_equal(bits(g.predict_raw(x).cpu().numpy()), bits(want))
    g.check()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,C,R,missing_prob", [
    (300, 8, 18, 60_001, 0.01),    # SUSY-like width, random thresholds (~4 k per feature): u16 codes, six 16-KiB regions
    (150, 10, 128, 45_000, 0.0),   # the widest forest of the form
    (300, 8, 100, 30_000, 0.02),
])
def test_narrow_forests_walk_384_row_tiles_on_u16_codes(env, monkeypatch, T, D, C, R, missing_prob):
    """num_cols <= 128: a 64-row region of u16 codes is at most 16 KiB, so six of them fit where three 32-KiB regions did -- 384
    rows per staged top with any number of thresholds per feature (qring_kernel<..., K = 6, REGB = 16 KiB>).  Leaf indices, sums
    and continued sums against the oracle; TAHOE_QRING_NARROW128=0 (the 192-row tiles) gives the same bits."""
    ta, oracle, torch = env
    monkeypatch.delenv("TAHOE_QRING_NARROW128", raising=False)
    nodes = ta.synth_forest(T, D, C, seed=700 + T, leaf_prob=0.03)
    data = ta.synth_data(R, C, seed=800 + T, missing_prob=missing_prob, missing=MISSING, nan_prob=missing_prob / 2)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.set_strategy(ta.STRATEGY_QRING)
    assert f.kernel_form(R) == "qring_region6" and f.info().qring_tile_rows == 384
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    for rows in (R, 384 * 256 + 5, 777):
        rows = min(rows, R)
        leaf, sums = f.predict_leaf_idx(x[:rows].contiguous())
        raw = f.predict_raw(x[:rows].contiguous())
        f.check()
        assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf[:rows]), rows
        assert np.array_equal(bits(sums.cpu().numpy()), bits(want[:rows])) and np.array_equal(bits(raw.cpu().numpy()), bits(want[:rows])), rows
    start = np.linspace(-1.0, 1.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(oracle.predict_continue(nodes, T, D, data, MISSING, start)))
    f.close()
    monkeypatch.setenv("TAHOE_QRING_NARROW128", "0")
    g = ta.Forest(nodes, T, D, C, missing=MISSING)
    g.set_strategy(ta.STRATEGY_QRING)
    assert g.kernel_form(R) in ("qring_region_mixed", "qring_region3", "qring_region2") and g.info().qring_tile_rows == 192
    assert np.array_equal(bits(g.predict_raw(x).cpu().numpy()), bits(want))
    g.check()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u16_wide", "u16_narrow", "u8"])
def test_small_remainder_behind_whole_waves_is_walked_in_tree_slices(env, monkeypatch, kind):
    """A batch of one whole wave of large tiles + a few hundred rows: the remainder's few 128-row tiles are given to several
    workgroups each (tree slices) and an ordered sum over those rows, so that it does not cost a whole tile time on a mostly idle
    chip.  Same bits as the oracle (sums, continued sums, leaf indices) and as the plain remainder (TAHOE_QRING_SLICES=1)."""
    ta, oracle, torch = env
    monkeypatch.delenv("TAHOE_QRING_SLICES", raising=False)
    T, D = 200, (7 if kind == "u8" else 9)  # (depth 9: > 254 thresholds per feature -> u16 codes)
    if kind == "u8":
        C = 200
        nodes = ta.synth_forest_hist(T, D, C, seed=611, feature_seed=29, max_bins=200, scale_decades=2.0)
    else:
        C = 200 if kind == "u16_wide" else 64
        nodes = ta.synth_forest(T, D, C, seed=612, leaf_prob=0.02)
    f = ta.Forest(nodes, T, D, C, missing=MISSING)
    f.set_strategy(ta.STRATEGY_QRING)
    big = 192 if kind == "u16_wide" else 384
    R = f.info().num_cus * big + 777
    assert f.kernel_form(R) == {"u16_wide": "qring_region_mixed", "u16_narrow": "qring_region6", "u8": "qring_region8"}[kind]
    data = (ta.synth_data_hist(R, C, seed=613, feature_seed=29, scale_decades=2.0, missing_prob=0.01, missing=MISSING) if kind == "u8" else
            ta.synth_data(R, C, seed=613, missing_prob=0.01, missing=MISSING))
    want, want_leaf = oracle.predict(nodes, T, D, data, MISSING, want_leaf=True, threads=8)
    x = torch.from_numpy(data).cuda()
    f.reserve(R)
    leaf, sums = f.predict_leaf_idx(x)
    raw = f.predict_raw(x)
    f.check()
    assert np.array_equal(bits(leaf.cpu().numpy()), want_leaf)
    assert np.array_equal(bits(sums.cpu().numpy()), bits(want)) and np.array_equal(bits(raw.cpu().numpy()), bits(want))
    start = np.linspace(-1.0, 1.0, R).astype(np.float32)
    acc = f.predict_accumulate(x, torch.from_numpy(start.copy()).cuda())
    f.check()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(oracle.predict_continue(nodes, T, D, data, MISSING, start)))
    f.close()
    monkeypatch.setenv("TAHOE_QRING_SLICES", "1")  # the remainder as plain 128-row tiles
    g = ta.Forest(nodes, T, D, C, missing=MISSING)
    g.set_strategy(ta.STRATEGY_QRING)
    assert np.array_equal(bits(g.predict_raw(x).cpu().numpy()), bits(want))
    g.check()
    g.close()
