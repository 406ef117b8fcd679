"""ctypes loader for oracle/liboracle.so (the CPU restatement of the reference's predictor).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the tahoe_amd package.  PARITY UNPINNED (see tahoe_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

NODE_DTYPE = np.dtype([("weight", "<f4"), ("val", "<f4"), ("bits", "<i4")])  # dense_node_t, Struct.h:44-48


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    lib.oracle_tree_num_nodes.restype = i
    lib.oracle_tree_num_nodes.argtypes = [i]
    lib.oracle_load_model.restype = i
    lib.oracle_load_model.argtypes = [C.c_char_p, C.POINTER(i), C.POINTER(i), C.POINTER(vp)]
    lib.oracle_load_data.restype = i
    lib.oracle_load_data.argtypes = [C.c_char_p, C.POINTER(i), C.POINTER(i), C.POINTER(f), C.POINTER(vp)]
    lib.oracle_free.restype = None
    lib.oracle_free.argtypes = [vp]
    lib.oracle_predict.restype = None
    lib.oracle_predict.argtypes = [vp, i, i, vp, sz, sz, i, f, i, f, f, vp, vp]
    lib.oracle_predict_f64.restype = None
    lib.oracle_predict_f64.argtypes = [vp, i, i, vp, sz, sz, i, f, vp]
    lib.oracle_predict_continue.restype = None
    lib.oracle_predict_continue.argtypes = [vp, i, i, vp, sz, sz, i, f, vp]
    lib.oracle_abs_leaf_sum.restype = None
    lib.oracle_abs_leaf_sum.argtypes = [vp, i, i, vp, sz, sz, i, f, vp]
    lib.oracle_encode_node.restype = None
    lib.oracle_encode_node.argtypes = [vp, i, f, i, f, i]
    return lib


lib = _load()


def tree_num_nodes(depth: int) -> int:
    return lib.oracle_tree_num_nodes(depth)


def load_model(path: str, num_trees: int = 10, depth: int = 20):
    nt, d, ptr = C.c_int(num_trees), C.c_int(depth), C.c_void_p()
    rc = lib.oracle_load_model(os.fsencode(path), C.byref(nt), C.byref(d), C.byref(ptr))
    if rc != 0:
        raise OSError(f"oracle_load_model({path}) -> {rc}")
    n = nt.value * tree_num_nodes(d.value)
    try:
        buf = (C.c_char * (n * NODE_DTYPE.itemsize)).from_address(ptr.value) if n else b""
        nodes = np.frombuffer(buf, dtype=NODE_DTYPE, count=n).copy()
    finally:
        lib.oracle_free(ptr)
    return nodes, nt.value, d.value


def load_data(path: str, num_rows: int = 1000, num_cols: int = 500, missing: float = 0.0):
    nr, nc, ms, ptr = C.c_int(num_rows), C.c_int(num_cols), C.c_float(missing), C.c_void_p()
    rc = lib.oracle_load_data(os.fsencode(path), C.byref(nr), C.byref(nc), C.byref(ms), C.byref(ptr))
    if rc != 0:
        raise OSError(f"oracle_load_data({path}) -> {rc}")
    n = nr.value * nc.value
    try:
        buf = (C.c_char * (n * 4)).from_address(ptr.value) if n else b""
        data = np.frombuffer(buf, dtype=np.float32, count=n).copy().reshape(nr.value, nc.value)
    finally:
        lib.oracle_free(ptr)
    return data, ms.value


def predict(nodes: np.ndarray, num_trees: int, depth: int, data: np.ndarray, missing: float, output: int = 0,
            threshold: float = 0.0, global_bias: float = 0.0, want_leaf: bool = False, threads: int = 1):
    """-> (preds float32 [rows], leaf_idx uint32 [rows, trees] or None).

    threads > 1 splits the rows into contiguous blocks (each row is still summed sequentially in
    tree order by one thread, so results do not depend on `threads`)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    assert nodes.size == num_trees * tree_num_nodes(depth)
    preds = np.empty(rows, dtype=np.float32)
    leaf = np.empty((rows, num_trees), dtype=np.uint32) if want_leaf else None

    def run(lo, hi):
        lib.oracle_predict(nodes.ctypes.data, num_trees, depth, data.ctypes.data, lo, hi, cols, missing, output,
                           threshold, global_bias, preds.ctypes.data, leaf.ctypes.data if want_leaf else None)

    if threads <= 1 or rows < 2 * threads:
        run(0, rows)
    else:
        bounds = np.linspace(0, rows, threads + 1).astype(np.int64)
        with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL during the call
            list(ex.map(lambda k: run(int(bounds[k]), int(bounds[k + 1])), range(threads)))
    return preds, leaf


def predict_f64(nodes: np.ndarray, num_trees: int, depth: int, data: np.ndarray, missing: float) -> np.ndarray:
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    sums = np.empty(rows, dtype=np.float64)
    lib.oracle_predict_f64(nodes.ctypes.data, num_trees, depth, data.ctypes.data, 0, rows, cols, missing,
                           sums.ctypes.data)
    return sums


def _rows_in_threads(fn, rows: int, threads: int) -> None:
    if threads <= 1 or rows < 2 * threads:
        fn(0, rows)
        return
    bounds = np.linspace(0, rows, threads + 1).astype(np.int64)
    with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL during the call
        list(ex.map(lambda k: fn(int(bounds[k]), int(bounds[k + 1])), range(threads)))


def predict_f64_mt(nodes: np.ndarray, num_trees: int, depth: int, data: np.ndarray, missing: float, threads: int = 1) -> np.ndarray:
    """predict_f64 with the rows split over `threads` host threads."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    sums = np.empty(rows, dtype=np.float64)
    _rows_in_threads(lambda lo, hi: lib.oracle_predict_f64(nodes.ctypes.data, num_trees, depth, data.ctypes.data, lo, hi, cols,
                                                           missing, sums.ctypes.data), rows, threads)
    return sums


def predict_continue(nodes: np.ndarray, num_trees: int, depth: int, data: np.ndarray, missing: float, sums: np.ndarray,
                     threads: int = 1) -> np.ndarray:
    """In place: sums[i] <- sums[i] continued through the trees in order, float32 (a chained tree shard's step)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    assert sums.dtype == np.float32 and sums.shape == (rows,) and sums.flags.c_contiguous
    _rows_in_threads(lambda lo, hi: lib.oracle_predict_continue(nodes.ctypes.data, num_trees, depth, data.ctypes.data, lo, hi, cols,
                                                                missing, sums.ctypes.data), rows, threads)
    return sums


def abs_leaf_sum(nodes: np.ndarray, num_trees: int, depth: int, data: np.ndarray, missing: float, threads: int = 1) -> np.ndarray:
    """float64 sum over trees of |leaf value| per row: the A of the float32 summation error bound gamma(n) * A."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    out = np.empty(rows, dtype=np.float64)
    _rows_in_threads(lambda lo, hi: lib.oracle_abs_leaf_sum(nodes.ctypes.data, num_trees, depth, data.ctypes.data, lo, hi, cols,
                                                            missing, out.ctypes.data), rows, threads)
    return out


# ---- sparse forests (Struct.h:50-54, 2217-2250; BaseTahoeTest.h:728-764) ----
SPARSE_NODE_DTYPE = np.dtype([("val", "<f4"), ("bits", "<i4"), ("left_idx", "<i4")])
lib.oracle_sparse_predict.restype = None
lib.oracle_sparse_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                      C.c_float, C.c_void_p, C.c_void_p]
lib.oracle_dense_to_sparse.restype = C.c_size_t
lib.oracle_dense_to_sparse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]


def sparse_predict(nodes: np.ndarray, trees: np.ndarray, data: np.ndarray, missing: float, want_leaf: bool = False,
                   threads: int = 1):
    nodes = np.ascontiguousarray(nodes, dtype=SPARSE_NODE_DTYPE)
    trees = np.ascontiguousarray(trees, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, cols = data.shape
    preds = np.empty(rows, dtype=np.float32)
    leaf = np.empty((rows, trees.size), dtype=np.uint32) if want_leaf else None

    def run(lo, hi):
        lib.oracle_sparse_predict(nodes.ctypes.data, trees.ctypes.data, trees.size, data.ctypes.data, lo, hi, cols, missing,
                                  preds.ctypes.data, leaf.ctypes.data if want_leaf else None)

    if threads <= 1 or rows < 2 * threads:
        run(0, rows)
    else:
        bounds = np.linspace(0, rows, threads + 1).astype(np.int64)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda k: run(int(bounds[k]), int(bounds[k + 1])), range(threads)))
    return preds, leaf


def dense_to_sparse(nodes: np.ndarray, num_trees: int, depth: int):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pn, pt = C.c_void_p(), C.c_void_p()
    n = lib.oracle_dense_to_sparse(nodes.ctypes.data, num_trees, depth, C.byref(pn), C.byref(pt))
    try:
        sn = np.frombuffer((C.c_char * (n * 12)).from_address(pn.value), dtype=SPARSE_NODE_DTYPE, count=n).copy()
        tr = np.frombuffer((C.c_char * (num_trees * 4)).from_address(pt.value), dtype=np.int32, count=num_trees).copy()
    finally:
        lib.oracle_free(pn)
        lib.oracle_free(pt)
    return sn, tr
