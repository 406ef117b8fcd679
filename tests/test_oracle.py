"""The CPU oracle (oracle/tahoe_oracle.c) against hand-derived known answers, an independent pure-Python
restatement of the same reference lines, and the committed golden fixtures.  CPU only.

The reference holds no fixtures for this path, so "golden" here means: inputs in the
reference's file formats + outputs that were derived by hand or by the oracle and are cross-checked below by a
second, independently written implementation ("parity unpinned")."""
import glob
import math
import os
import struct

import numpy as np
import pytest

from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
FID_MASK, DEF_LEFT, IS_LEAF = (1 << 30) - 1, 1 << 30, 1 << 31  # Struct.h:57-59


def enc(fid, value, def_left, weight, is_leaf):
    n = np.zeros(len(fid), dtype=oracle.NODE_DTYPE)
    n["weight"], n["val"] = weight, value
    bits = (np.asarray(fid, np.int64) & FID_MASK) | np.asarray(def_left, np.int64) * DEF_LEFT | np.asarray(
        is_leaf, np.int64) * IS_LEAF
    n["bits"] = bits.astype(np.uint32).view(np.int32)
    return n


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def py_predict(nodes, T, D, data, missing, output=0, threshold=0.0, bias=0.0):
    """Independent restatement of BaseTahoeTest.h:440-474 with Python scalars (float32 emulated)."""
    n_per = (1 << (D + 1)) - 1
    missing = np.float32(missing)
    preds, leaves = [], []
    for row in data:
        pred = np.float32(0.0)
        lv = []
        for t in range(T):
            curr = 0
            while True:
                node = nodes[t * n_per + curr]
                bits = int(node["bits"]) & 0xFFFFFFFF
                if bits & IS_LEAF:
                    break
                x = np.float32(row[bits & FID_MASK])
                with np.errstate(invalid="ignore", over="ignore"):
                    is_missing = bool(np.abs(np.float32(x - missing)) <= np.float32(1e-6))
                    cond = (not (bits & DEF_LEFT)) if is_missing else bool(x >= node["val"])
                curr = 2 * curr + 1 + (1 if cond else 0)
            pred = np.float32(pred + node["val"])
            lv.append(curr)
        if output & 0x1:
            pred = np.float32(pred / np.float32(T))
        pred = np.float32(pred + np.float32(bias))
        if output & 0x10:
            pred = np.float32(np.float32(1.0) / np.float32(np.float32(1.0) + np.float32(math.exp(-float(pred)))))
        if output & 0x100:
            pred = np.float32(1.0 if pred > np.float32(threshold) else 0.0)
        preds.append(pred)
        leaves.append(lv)
    return np.array(preds, dtype=np.float32), np.array(leaves, dtype=np.uint32).reshape(len(data), T)


def test_hand_derived_tree():
    """depth-2 tree, answers worked out on paper (ties go right, missing follows def_left, NaN goes left,
    an early leaf ends the walk)."""
    M = -999.0
    nodes = enc([0, 1, 0, 0, 0, 1, 1], [0.5, -1.0, 10.0, 1.0, 2.0, 77.0, 88.0], [1, 0, 0, 0, 0, 0, 0], [0] * 7,
                [0, 0, 1, 1, 1, 0, 0])
    data = np.array([[0.5, 0.0], [0.4, -1.0], [0.4, -1.5], [M, 5.0], [0.0, M], [np.nan, np.nan]], dtype=np.float32)
    preds, leaf = oracle.predict(nodes, 1, 2, data, M, want_leaf=True)
    assert leaf[:, 0].tolist() == [2, 4, 3, 4, 4, 3]
    assert preds.tolist() == [10.0, 2.0, 1.0, 2.0, 2.0, 1.0]


def test_known_answers_every_branch_of_the_walk():
    """Every branch of infer_one_tree (BaseTahoeTest.h:440-456), expected values written by hand -- none of them produced
    by the code under test.  One depth-1 tree: root on feature 0 with threshold 0.5, left leaf -3, right leaf +4."""
    M = -999.0

    def stump(thr, def_left):
        return enc([0, 0, 0], [thr, -3.0, 4.0], [def_left, 0, 0], [0, 0, 0], [0, 1, 1])

    below = np.nextafter(np.float32(0.5), np.float32(0))  # 0.49999997
    xs = np.array([0.5, below, 0.6, np.nan, np.inf, -np.inf, M, M + 5e-4, -0.0], dtype=np.float32).reshape(-1, 1)
    #            tie->R  <->L  >->R  NaN->L  +inf->R  -inf->L  missing  next float after M: outside the band -> compare -> L
    for def_left, at_missing in ((1, -3.0), (0, 4.0)):  # :452  missing ? !def_left : x >= thr
        preds, leaf = oracle.predict(stump(0.5, def_left), 1, 1, xs, M, want_leaf=True)
        assert preds.tolist() == [4.0, -3.0, 4.0, -3.0, 4.0, -3.0, at_missing, -3.0, -3.0]
        assert leaf[:, 0].tolist() == [2, 1, 2, 1, 2, 1, 1 if def_left else 2, 1, 1]  # :453  curr = 2*curr + 1 + cond
    # a NaN threshold is never reached by >= : everything that is not missing goes left
    preds, _ = oracle.predict(stump(np.nan, 0), 1, 1, xs, M)
    assert preds.tolist() == [-3.0, -3.0, -3.0, -3.0, -3.0, -3.0, 4.0, -3.0, -3.0]
    # a NaN sentinel never matches (fabs(x - NaN) <= eps is false), not even a NaN feature
    preds, _ = oracle.predict(stump(0.5, 0), 1, 1, xs, float("nan"))
    assert preds.tolist() == [4.0, -3.0, 4.0, -3.0, 4.0, -3.0, -3.0, -3.0, -3.0]
    # -0.0 == 0.0: a threshold of +0.0 sends -0.0 right (ties go right)
    preds, _ = oracle.predict(stump(0.0, 0), 1, 1, np.array([[-0.0], [0.0], [-1e-45]], dtype=np.float32), M)
    assert preds.tolist() == [4.0, 4.0, -3.0]
    # a leaf at the root (:449 before any feature is read): depth 0, leaf index 0, features irrelevant
    preds, leaf = oracle.predict(enc([5], [7.25], [0], [0], [1]), 1, 0, np.array([[np.nan]], dtype=np.float32), M, want_leaf=True)
    assert preds.tolist() == [7.25] and leaf.tolist() == [[0]]
    # the feature index comes from bits & FID_MASK with DEF_LEFT and IS_LEAF stripped (Struct.h:110-117): fid 3 of 4
    nodes = enc([3, 0, 0], [0.5, -3.0, 4.0], [1, 0, 0], [0, 0, 0], [0, 1, 1])
    preds, _ = oracle.predict(nodes, 1, 1, np.array([[9, 9, 9, 0.4], [0, 0, 0, 0.6]], dtype=np.float32), M)
    assert preds.tolist() == [-3.0, 4.0]


def test_known_answers_every_output_mode():
    """predict_on_cpu's epilogue (BaseTahoeTest.h:467-472) in its order -- AVG (a DIVISION by num_trees), + bias, sigmoid,
    threshold (strict >) -- with expectations written by hand.  Three single-leaf trees summing to 5.0."""
    nodes = enc([0, 0, 0], [1.0, 1.5, 2.5], [0, 0, 0], [0, 0, 0], [1, 1, 1])
    x = np.zeros((1, 1), np.float32)
    RAW, AVG, SIG, THR = 0x0, 0x1, 0x10, 0x100

    def run(output, threshold=0.0, bias=0.0):
        return oracle.predict(nodes, 3, 0, x, -999.0, output, threshold, bias)[0]

    assert run(RAW)[0] == 5.0
    assert run(RAW, bias=0.25)[0] == 5.25                      # :468 the bias is added in RAW mode too
    # 5/3 in float32 is 0x3FD55555 (1.6666666); 5 * fl(1/3) would be 0x3FD55556 -- the CPU predictor divides
    assert run(AVG).view(np.uint32)[0] == 0x3FD55555
    assert run(AVG, bias=-0.5).view(np.uint32)[0] == np.float32(np.float32(5.0) / np.float32(3.0) - np.float32(0.5)).view(np.uint32)
    zero = enc([0, 0], [2.0, -2.0], [0, 0], [0, 0], [1, 1])  # two trees summing to 0
    z = lambda output, threshold=0.0, bias=0.0: oracle.predict(zero, 2, 0, x, -999.0, output, threshold, bias)[0][0]  # noqa: E731
    assert z(SIG) == 0.5                                        # 1 / (1 + exp(-0)) exactly
    assert z(SIG | THR, threshold=0.5) == 0.0                   # strict: 0.5 > 0.5 is false
    assert z(SIG | THR, threshold=0.49) == 1.0
    assert z(THR, threshold=-0.1) == 1.0 and z(THR, threshold=0.0) == 0.0
    assert z(AVG | SIG, bias=0.0) == 0.5
    assert z(SIG, bias=200.0) == 1.0                            # exp(-200) underflows: 1 / (1 + 0)
    assert z(SIG, bias=-200.0) == 0.0                           # expf(200) = inf: 1 / inf
    assert z(AVG | SIG | THR, threshold=0.7, bias=1.0) == 1.0   # sigmoid(0/2 + 1) = 0.7310586 > 0.7
    assert z(AVG | SIG | THR, threshold=0.74, bias=1.0) == 0.0


def test_missing_band_is_on_the_float32_difference():
    # |x - missing| <= 1e-6f with the subtraction rounded to float32 (BaseTahoeTest.h:451-452)
    m = np.float32(0.25)
    nodes = enc([0, 0, 0], [100.0, 1.0, 2.0], [1, 0, 0], [0, 0, 0], [0, 1, 1])  # def_left: missing -> leaf 1.0
    xs = np.array([m, m + np.float32(5e-7), m - np.float32(9e-7), m + np.float32(2e-6), m - np.float32(2e-6)],
                  dtype=np.float32).reshape(-1, 1)
    preds, _ = oracle.predict(nodes, 1, 1, xs, float(m))
    # inside the band -> default (left, 1.0); outside -> x >= 100 is false -> left as well (1.0)
    assert preds.tolist() == [1.0, 1.0, 1.0, 1.0, 1.0]
    nodes = enc([0, 0, 0], [-100.0, 1.0, 2.0], [1, 0, 0], [0, 0, 0], [0, 1, 1])  # outside the band -> right
    preds, _ = oracle.predict(nodes, 1, 1, xs, float(m))
    assert preds.tolist() == [1.0, 1.0, 1.0, 2.0, 2.0]


def test_float32_sum_is_sequential_in_tree_order():
    # three single-leaf trees: (1e8 + 1) - 1e8 in float32 is 0, not 1: the order of the adds is visible
    nodes = enc([0, 0, 0], [1e8, 1.0, -1e8], [0, 0, 0], [0, 0, 0], [1, 1, 1])
    preds, _ = oracle.predict(nodes, 3, 0, np.zeros((1, 1), np.float32), 0.5)
    assert preds[0] == 0.0
    nodes = enc([0, 0, 0], [1e8, -1e8, 1.0], [0, 0, 0], [0, 0, 0], [1, 1, 1])
    preds, _ = oracle.predict(nodes, 3, 0, np.zeros((1, 1), np.float32), 0.5)
    assert preds[0] == 1.0


@pytest.mark.parametrize("T,D,C,R,seed", [(5, 3, 4, 50, 0), (9, 5, 18, 40, 1), (3, 7, 30, 25, 2), (2, 0, 1, 3, 3)])
@pytest.mark.parametrize("output,threshold,bias", [(0, 0.0, 0.0), (0x1, 0.0, 0.25), (0x11, 0.0, 0.0), (0x101, 0.05, 0.0)])
def test_oracle_matches_independent_python(T, D, C, R, seed, output, threshold, bias):
    rng = np.random.default_rng(seed)
    n_per = (1 << (D + 1)) - 1
    first_bottom = (1 << D) - 1
    is_leaf = np.zeros((T, n_per), np.int64)
    is_leaf[:, first_bottom:] = 1
    is_leaf[:, :first_bottom] = rng.random((T, first_bottom)) < 0.15
    nodes = enc(rng.integers(0, C, T * n_per), rng.uniform(-1, 1, T * n_per).astype(np.float32),
                rng.integers(0, 2, T * n_per), rng.random(T * n_per).astype(np.float32), is_leaf.reshape(-1))
    data = rng.uniform(-1, 1, (R, C)).astype(np.float32)
    data[rng.random((R, C)) < 0.1] = -999.0
    data[rng.random((R, C)) < 0.05] = np.nan
    data[0, :] = nodes["val"][:C] if T * n_per >= C else data[0, :]  # threshold ties
    want, want_leaf = py_predict(nodes, T, D, data, -999.0, output, threshold, bias)
    got, got_leaf = oracle.predict(nodes, T, D, data, -999.0, output, threshold, bias, want_leaf=True)
    assert np.array_equal(got_leaf, want_leaf)
    if output & 0x10:
        np.testing.assert_allclose(got, want, rtol=2e-7)  # expf (C) vs math.exp rounded
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_threads_do_not_change_results():
    rng = np.random.default_rng(5)
    T, D, C, R = 20, 6, 12, 1000
    n_per = (1 << (D + 1)) - 1
    is_leaf = np.zeros((T, n_per), np.int64)
    is_leaf[:, (1 << D) - 1:] = 1
    nodes = enc(rng.integers(0, C, T * n_per), rng.uniform(-1, 1, T * n_per).astype(np.float32),
                rng.integers(0, 2, T * n_per), np.zeros(T * n_per, np.float32), is_leaf.reshape(-1))
    data = rng.uniform(-1, 1, (R, C)).astype(np.float32)
    a, la = oracle.predict(nodes, T, D, data, -999.0, want_leaf=True, threads=1)
    b, lb = oracle.predict(nodes, T, D, data, -999.0, want_leaf=True, threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(la, lb)
    s64 = oracle.predict_f64(nodes, T, D, data, -999.0)
    assert np.max(np.abs(s64 - a)) < 1e-4


def test_golden_fixtures_oracle_side():
    cases = sorted(glob.glob(os.path.join(HERE, "golden", "*.model.txt")))
    assert len(cases) >= 4
    for model in cases:
        stem = model[: -len(".model.txt")]
        nodes, T, D = oracle.load_model(model)
        data, missing = oracle.load_data(stem + ".data.txt")
        exp = np.load(stem + ".expected.npz")
        assert np.array_equal(nodes.view(np.uint32).reshape(-1, 3), exp["nodes_bits"]), stem
        assert np.array_equal(data.view(np.uint32), exp["data_bits"]), stem
        sums, leaf = oracle.predict(nodes, T, D, data, missing, want_leaf=True)
        assert np.array_equal(sums.view(np.uint32), exp["sums_bits"]), stem
        assert np.array_equal(leaf, exp["leaf_idx"]), stem
        # and the independent Python restatement agrees with the stored expectations
        p_sums, p_leaf = py_predict(nodes, T, D, data, missing)
        assert np.array_equal(p_sums.view(np.uint32), exp["sums_bits"]), stem
        assert np.array_equal(p_leaf, exp["leaf_idx"]), stem
