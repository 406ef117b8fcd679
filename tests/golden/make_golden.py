"""Generates the fixtures in this directory.  Run from the repo root: python tests/golden/make_golden.py

Inputs are written in the reference's text formats (BaseTahoeTest.h:267-402) with the product's writer;
expected outputs come from the CPU oracle (oracle/tahoe_oracle.c).  The reference itself holds no
fixtures for this path and cannot be built in this image, so these are ORACLE-generated known answers
("parity unpinned"): they pin the HIP path and the loaders to the oracle and guard
against regressions; they are not outputs of the reference binary.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import tahoe_amd as ta  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def emit(name, nodes, T, D, data, missing):
    stem = os.path.join(HERE, name)
    ta.write_model(stem + ".model.txt", nodes, T, D)
    ta.write_data(stem + ".data.txt", data, missing)
    # expected values are computed from what the ORACLE's loader reads back from the files
    n2, T2, D2 = oracle.load_model(stem + ".model.txt")
    d2, m2 = oracle.load_data(stem + ".data.txt")
    sums, leaf = oracle.predict(n2, T2, D2, d2, m2, want_leaf=True)
    np.savez(stem + ".expected.npz", sums_bits=sums.view(np.uint32), leaf_idx=leaf,
             nodes_bits=n2.view(np.uint32).reshape(-1, 3), data_bits=d2.view(np.uint32))


def main():
    # char-width features (C=18 < 32), missing hits with both def_left values, early leaves
    T, D, C, R = 6, 4, 18, 40
    emit("susy_like_c18", ta.synth_forest(T, D, C, seed=1, leaf_prob=0.2), T, D,
         ta.synth_data(R, C, seed=2, missing_prob=0.15, missing=-999.0, nan_prob=0.05), -999.0)
    # short-width features (C=256), depth 5
    T, D, C, R = 4, 5, 256, 12
    emit("k3_like_c256", ta.synth_forest(T, D, C, seed=3), T, D, ta.synth_data(R, C, seed=4), -999.0)
    # depth 0 (single leaf per tree) and depth 1
    T, D, C, R = 3, 0, 2, 5
    emit("depth0", ta.synth_forest(T, D, C, seed=5), T, D, ta.synth_data(R, C, seed=6), 0.0)
    T, D, C, R = 5, 1, 3, 9
    nodes = ta.synth_forest(T, D, C, seed=7)
    data = ta.synth_data(R, C, seed=8, missing_prob=0.3, missing=0.5)
    data[0, :] = nodes["val"][0]  # threshold tie at the first root
    emit("depth1_ties", nodes, T, D, data, 0.5)


if __name__ == "__main__":
    main()
