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
