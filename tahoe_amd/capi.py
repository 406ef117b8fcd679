"""ctypes binding of include/tahoe_amd.h.

Plumbing only: torch (or anything else) owns device memory and streams; this module passes raw
device pointers through the C ABI.  If libtahoe_amd.so has not been built the import fails --
there is deliberately no pure-Python or CPU substitute.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtahoe_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C tahoe_amd/csrc`.  tahoe_amd has no fallback path."
    )

# torch bundles its own libamdhip64.so.7.  Two HIP runtimes in one process do not share a device
# context (the second one sees no GPU), so load torch's first: libtahoe_amd.so then binds to the
# already-loaded runtime by SONAME.  Without torch (the C++ CLI) the system runtime is used.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

lib = C.CDLL(LIB_PATH)

# ---- constants mirrored from the header ----
OUT_RAW, OUT_AVG, OUT_SIGMOID, OUT_THRESHOLD = 0x0, 0x1, 0x10, 0x100
STRATEGY_AUTO, STRATEGY_DIRECT, STRATEGY_ROWTILE, STRATEGY_TILEBLOCK, STRATEGY_TILERING, STRATEGY_QRING = range(6)
CREATE_PROB_RELAYOUT = 0x1
STRATEGY_NAMES = {1: "direct", 2: "rowtile", 3: "tileblock", 4: "tilering", 5: "qring"}
STATUS_NAMES = {
    0: "TAHOE_OK",
    1: "TAHOE_ERR_INVALID_ARG",
    2: "TAHOE_ERR_IO",
    3: "TAHOE_ERR_NO_MEMORY",
    4: "TAHOE_ERR_NO_DEVICE",
    5: "TAHOE_ERR_HIP",
    6: "TAHOE_ERR_INVALID_FOREST",
    7: "TAHOE_ERR_UNSUPPORTED",
}

# dense_node_t (Struct.h:44-48): weight, val, bits
NODE_DTYPE = np.dtype([("weight", "<f4"), ("val", "<f4"), ("bits", "<i4")])


class ForestParams(C.Structure):
    """tahoe_forest_params == forest_params_t (Struct.h:166-189)."""

    _fields_ = [
        ("num_nodes", C.c_int),
        ("depth", C.c_int),
        ("num_trees", C.c_int),
        ("num_cols", C.c_int),
        ("algo", C.c_int),
        ("output", C.c_int),
        ("threshold", C.c_float),
        ("global_bias", C.c_float),
        ("strategy", C.c_int),
        ("missing", C.c_float),
    ]


class ForestInfo(C.Structure):
    _fields_ = [
        ("num_trees", C.c_int),
        ("depth", C.c_int),
        ("num_cols", C.c_int),
        ("bits_bytes", C.c_int),
        ("lds_levels", C.c_int),
        ("device_bytes", C.c_size_t),
        ("lds_bytes_per_block", C.c_int),
        ("device_id", C.c_int),
        ("num_cus", C.c_int),
        ("top_levels", C.c_int),
        ("tile_rows", C.c_int),
        ("tileblock_lds_bytes", C.c_int),
        ("qring_walkers", C.c_int),
        ("qring_lds_bytes", C.c_int),
        ("qring_groups", C.c_int),
        ("is_sparse", C.c_int),
        ("ring_rows", C.c_int),
        ("tilering_lds_bytes", C.c_int),
        ("qring_tile_rows", C.c_int),
        ("relayout", C.c_int),
        ("relayout_swaps", C.c_size_t),
        ("stream_slots", C.c_int),
        ("stream_levels", C.c_int),
        ("stream_key_ties", C.c_float),
    ]


class TahoeError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = lib.tahoe_last_error().decode(errors="replace")
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}: {msg}")


def _check(status: int, where: str) -> None:
    if status != 0:
        raise TahoeError(status, where)


_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
_PROTOS = {
    "tahoe_last_error": (C.c_char_p, []),
    "tahoe_abi_version": (_i, []),
    "tahoe_encode_node": (None, [_vp, _i, _f, _i, _f, _i]),
    "tahoe_decode_node": (None, [_vp] + [_vp] * 5),
    "tahoe_tree_num_nodes": (_i, [_i]),
    "tahoe_forest_create": (_i, [C.POINTER(_vp), _vp, C.POINTER(ForestParams)]),
    "tahoe_forest_create_ex": (_i, [C.POINTER(_vp), _vp, C.POINTER(ForestParams), C.c_uint]),
    "tahoe_forest_destroy": (None, [_vp]),
    "tahoe_sparse_forest_create": (_i, [C.POINTER(_vp), _vp, _vp, C.POINTER(ForestParams)]),
    "tahoe_dense_to_sparse": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz)]),
    "tahoe_synth_sparse_forest": (_i, [_vp, _vp, C.POINTER(_sz), _i, _i, _i, _i, _f, _i, C.c_uint64]),
    "tahoe_forest_predict": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tahoe_forest_predict_raw": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tahoe_forest_predict_accumulate": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tahoe_forest_predict_leaf_idx": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "tahoe_transform_preds": (_i, [_vp, _sz, _i, _i, _f, _f, _vp]),
    "tahoe_forest_set_strategy": (_i, [_vp, _i]),
    "tahoe_forest_get_strategy": (_i, [_vp, _sz]),
    "tahoe_forest_check": (_i, [_vp, _vp]),
    "tahoe_forest_reserve": (_i, [_vp, _sz]),
    "tahoe_forest_predict_host": (_i, [_vp, _vp, _vp, _sz, _sz]),
    "tahoe_host_alloc": (_i, [C.POINTER(_vp), _sz]),
    "tahoe_host_free": (_i, [_vp]),
    "tahoe_forest_get_info": (_i, [_vp, C.POINTER(ForestInfo)]),
    "tahoe_forest_get_kernel_form": (_i, [_vp, _sz]),
    "tahoe_kernel_form_name": (C.c_char_p, [_i]),
    "tahoe_forest_set_profiling": (_i, [_vp, _i]),
    "tahoe_forest_kernel_times": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "tahoe_forest_prepass_times": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "tahoe_load_model": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "tahoe_load_data": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_vp)]),
    "tahoe_write_model": (_i, [C.c_char_p, _i, _i, _vp]),
    "tahoe_write_data": (_i, [C.c_char_p, _i, _i, _f, _vp]),
    "tahoe_save_model_bin": (_i, [C.c_char_p, _i, _i, _vp]),
    "tahoe_load_model_bin": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "tahoe_save_data_bin": (_i, [C.c_char_p, _i, _i, _f, _vp]),
    "tahoe_load_data_bin": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_vp)]),
    "tahoe_load_model_cached": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i)]),
    "tahoe_load_data_cached": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_vp), C.POINTER(_i)]),
    "tahoe_free_host": (None, [_vp]),
    "tahoe_synth_forest": (None, [_vp, _i, _i, _i, C.c_uint64, _f]),
    "tahoe_synth_data": (None, [_vp, _sz, _sz, _i, C.c_uint64, _f, _f, _f]),
    "tahoe_synth_forest_hist": (_i, [_vp, _i, _i, _i, C.c_uint64, C.c_uint64, _i, _f, _f, _f]),
    "tahoe_synth_data_hist": (_i, [_vp, _sz, _sz, _i, C.c_uint64, C.c_uint64, _f, _f, _f]),
    "tahoe_device_count": (_i, [C.POINTER(_i)]),
    "tahoe_device_set": (_i, [_i]),
    "tahoe_device_alloc": (_i, [C.POINTER(_vp), _sz, _i]),
    "tahoe_device_free": (_i, [_vp]),
    "tahoe_device_memset": (_i, [_vp, _i, _sz, _vp]),
    "tahoe_widen_f32_to_f64": (_i, [_vp, _vp, _sz, _vp]),
    "tahoe_narrow_f64_to_f32": (_i, [_vp, _vp, _sz, _vp]),
    "tahoe_copy_to_device": (_i, [_vp, _vp, _sz, _vp]),
    "tahoe_copy_to_host": (_i, [_vp, _vp, _sz, _vp]),
    "tahoe_stream_create": (_i, [C.POINTER(_vp)]),
    "tahoe_stream_destroy": (_i, [_vp]),
    "tahoe_stream_synchronize": (_i, [_vp]),
    "tahoe_event_create": (_i, [C.POINTER(_vp)]),
    "tahoe_event_destroy": (_i, [_vp]),
    "tahoe_event_record": (_i, [_vp, _vp]),
    "tahoe_stream_wait_event": (_i, [_vp, _vp]),
    "tahoe_copy_peer": (_i, [_vp, _i, _vp, _i, _sz, _vp]),
    "tahoe_device_synchronize": (_i, []),
    "tahoe_device_lds_bytes": (_i, [C.POINTER(_i)]),
    "tahoe_compare_device": (_i, [_vp, _vp, _sz, _f, C.POINTER(_sz), _vp]),
}
for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED_SYMBOLS = tuple(_PROTOS)


# ---- host-side helpers (formats, synthetic inputs) ----
class PinnedArray:
    """float32 [rows, cols] array in pinned host memory (tahoe_host_alloc); `.array` is the numpy view."""

    def __init__(self, rows: int, cols: int):
        self._p = _vp()
        _check(lib.tahoe_host_alloc(C.byref(self._p), max(rows * cols * 4, 4)), "tahoe_host_alloc")
        buf = (C.c_float * (rows * cols)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(rows, cols)

    def close(self) -> None:
        if self._p:
            self.array = None
            _check(lib.tahoe_host_free(self._p), "tahoe_host_free")
            self._p = None


def tree_num_nodes(depth: int) -> int:
    return lib.tahoe_tree_num_nodes(depth)


def _take_nodes(ptr, n):
    try:
        buf = (C.c_char * (n * NODE_DTYPE.itemsize)).from_address(ptr.value) if n else b""
        return np.frombuffer(buf, dtype=NODE_DTYPE, count=n).copy()
    finally:
        lib.tahoe_free_host(ptr)


def _take_floats(ptr, rows, cols):
    n = rows * cols
    try:
        buf = (C.c_char * (n * 4)).from_address(ptr.value) if n else b""
        return np.frombuffer(buf, dtype=np.float32, count=n).copy().reshape(rows, cols)
    finally:
        lib.tahoe_free_host(ptr)


def load_model(path: str, num_trees: int = 10, depth: int = 20, cached: bool = False):
    """-> (nodes[NODE_DTYPE], num_trees, depth).  Defaults are the BaseTahoeTest ctor defaults.
    cached=True: use / refresh the binary cache "<path>.tbin" (tahoe_load_model_cached)."""
    nt, d, ptr = _i(num_trees), _i(depth), _vp()
    if cached:
        _check(lib.tahoe_load_model_cached(os.fsencode(path), C.byref(nt), C.byref(d), C.byref(ptr), None),
               "tahoe_load_model_cached")
    else:
        _check(lib.tahoe_load_model(os.fsencode(path), C.byref(nt), C.byref(d), C.byref(ptr)), "tahoe_load_model")
    return _take_nodes(ptr, nt.value * tree_num_nodes(d.value)), nt.value, d.value


def load_data(path: str, num_rows: int = 1000, num_cols: int = 500, missing: float = 0.0, cached: bool = False):
    """-> (data[rows, cols] float32, missing)."""
    nr, nc, ms, ptr = _i(num_rows), _i(num_cols), _f(missing), _vp()
    if cached:
        _check(lib.tahoe_load_data_cached(os.fsencode(path), C.byref(nr), C.byref(nc), C.byref(ms), C.byref(ptr), None),
               "tahoe_load_data_cached")
    else:
        _check(lib.tahoe_load_data(os.fsencode(path), C.byref(nr), C.byref(nc), C.byref(ms), C.byref(ptr)),
               "tahoe_load_data")
    return _take_floats(ptr, nr.value, nc.value), ms.value


def save_model_bin(path: str, nodes: np.ndarray, num_trees: int, depth: int) -> None:
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    assert nodes.size == num_trees * tree_num_nodes(depth)
    _check(lib.tahoe_save_model_bin(os.fsencode(path), num_trees, depth, nodes.ctypes.data), "tahoe_save_model_bin")


def load_model_bin(path: str):
    nt, d, ptr = _i(0), _i(0), _vp()
    _check(lib.tahoe_load_model_bin(os.fsencode(path), C.byref(nt), C.byref(d), C.byref(ptr)), "tahoe_load_model_bin")
    return _take_nodes(ptr, nt.value * tree_num_nodes(d.value)), nt.value, d.value


def save_data_bin(path: str, data: np.ndarray, missing: float) -> None:
    data = np.ascontiguousarray(data, dtype=np.float32)
    _check(lib.tahoe_save_data_bin(os.fsencode(path), data.shape[0], data.shape[1], missing, data.ctypes.data),
           "tahoe_save_data_bin")


def load_data_bin(path: str):
    nr, nc, ms, ptr = _i(0), _i(0), _f(0.0), _vp()
    _check(lib.tahoe_load_data_bin(os.fsencode(path), C.byref(nr), C.byref(nc), C.byref(ms), C.byref(ptr)),
           "tahoe_load_data_bin")
    return _take_floats(ptr, nr.value, nc.value), ms.value


def write_model(path: str, nodes: np.ndarray, num_trees: int, depth: int) -> None:
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    assert nodes.size == num_trees * tree_num_nodes(depth)
    _check(lib.tahoe_write_model(os.fsencode(path), num_trees, depth, nodes.ctypes.data), "tahoe_write_model")


def write_data(path: str, data: np.ndarray, missing: float) -> None:
    data = np.ascontiguousarray(data, dtype=np.float32)
    _check(lib.tahoe_write_data(os.fsencode(path), data.shape[0], data.shape[1], missing, data.ctypes.data),
           "tahoe_write_data")


def synth_forest(num_trees: int, depth: int, num_cols: int, seed: int = 42, leaf_prob: float = 0.0) -> np.ndarray:
    nodes = np.empty(num_trees * tree_num_nodes(depth), dtype=NODE_DTYPE)
    lib.tahoe_synth_forest(nodes.ctypes.data, num_trees, depth, num_cols, seed, leaf_prob)
    return nodes


def synth_data(rows: int, num_cols: int, seed: int = 43, missing_prob: float = 0.0, missing: float = -999.0,
               nan_prob: float = 0.0, first_row: int = 0) -> np.ndarray:
    data = np.empty((rows, num_cols), dtype=np.float32)
    lib.tahoe_synth_data(data.ctypes.data, first_row, rows, num_cols, seed, missing_prob, missing, nan_prob)
    return data


def synth_forest_hist(num_trees: int, depth: int, num_cols: int, seed: int = 42, feature_seed: int = 7, max_bins: int = 255,
                      zipf_s: float = 1.0, leaf_prob: float = 0.02, scale_decades: float = 3.0) -> np.ndarray:
    """Forest in the style of histogram-trained GBDT models (tahoe_synth_forest_hist)."""
    nodes = np.zeros(num_trees * tree_num_nodes(depth), dtype=NODE_DTYPE)
    _check(lib.tahoe_synth_forest_hist(nodes.ctypes.data, num_trees, depth, num_cols, seed, feature_seed, max_bins, zipf_s, leaf_prob,
                                       scale_decades), "tahoe_synth_forest_hist")
    return nodes


def synth_data_hist(rows: int, num_cols: int, seed: int = 43, feature_seed: int = 7, scale_decades: float = 3.0,
                    missing_prob: float = 0.0, missing: float = -999.0, first_row: int = 0) -> np.ndarray:
    """Rows drawn from the per-feature distributions of synth_forest_hist (same feature_seed / scale_decades)."""
    out = np.empty((rows, num_cols), dtype=np.float32)
    _check(lib.tahoe_synth_data_hist(out.ctypes.data, first_row, rows, num_cols, seed, feature_seed, scale_decades, missing_prob,
                                     missing), "tahoe_synth_data_hist")
    return out


def set_probability_weights(nodes: np.ndarray, num_trees: int, depth: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """Fills dense_node_t.weight with the probability of reaching each node for features uniform in [lo, hi) and
    independent (what synth_data draws): root 1, left child p * P(x < thr), right child p * P(x >= thr).  Input
    preparation for the probability-guided re-layout (a trained model carries such weights from its training data)."""
    per = tree_num_nodes(depth)
    n = nodes.reshape(num_trees, per)
    w = np.zeros((num_trees, per), dtype=np.float64)
    w[:, 0] = 1.0
    thr = n["val"].astype(np.float64)
    is_leaf = (n["bits"].view(np.uint32) >> 31) != 0
    for level in range(depth):
        a, b = (1 << level) - 1, (2 << level) - 1
        p_right = np.clip((hi - thr[:, a:b]) / (hi - lo), 0.0, 1.0)
        p_right = np.where(np.isnan(p_right), 0.0, p_right)
        live = np.where(is_leaf[:, a:b], 0.0, w[:, a:b])  # nothing is reached below a leaf
        w[:, 2 * a + 1: 2 * b + 1: 2] = live * (1.0 - p_right)
        w[:, 2 * a + 2: 2 * b + 2: 2] = live * p_right
    n["weight"] = w.astype(np.float32)
    return nodes


def encode_nodes(fid, value, def_left, weight, is_leaf) -> np.ndarray:
    """Vector form of encode_node (Struct.h:103-108) for building test forests by hand."""
    fid = np.asarray(fid, dtype=np.int64)
    nodes = np.empty(fid.shape, dtype=NODE_DTYPE)
    nodes["weight"] = np.asarray(weight, dtype=np.float32)
    nodes["val"] = np.asarray(value, dtype=np.float32)
    bits = (fid & ((1 << 30) - 1)) | (np.asarray(def_left, dtype=np.int64) != 0) * (1 << 30) | (
        np.asarray(is_leaf, dtype=np.int64) != 0) * (1 << 31)
    nodes["bits"] = bits.astype(np.uint32).view(np.int32)
    return nodes


# ---- the forest operator ----
def _ptr(t) -> int:
    """Device pointer of a torch tensor (or a raw int address)."""
    return t if isinstance(t, int) else t.data_ptr()


def _stream(stream) -> int:
    if stream is None:
        import torch

        return torch.cuda.current_stream().cuda_stream
    return stream if isinstance(stream, int) else stream.cuda_stream


class Forest:
    """Handle on a device forest: tahoe_forest_create / predict / destroy.

    Mirrors the reference's init_dense* + predict_dense* pair (BaseTahoeTest.h:519-547, :599-611).
    Tensors are torch CUDA tensors (float32 data [rows, cols] contiguous, float32 preds [rows])."""

    def __init__(self, nodes: np.ndarray, num_trees: int, depth: int, num_cols: int, missing: float = 0.0,
                 output: int = OUT_RAW, threshold: float = 0.0, global_bias: float = 0.0, algo: int = 0,
                 strategy: int = 0, relayout: bool = False):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        if nodes.size != num_trees * tree_num_nodes(depth):
            raise ValueError("nodes.size != num_trees * tree_num_nodes(depth)")
        self.params = ForestParams(0, depth, num_trees, num_cols, algo, output, threshold, global_bias, strategy,
                                   missing)
        self._h = _vp()
        if relayout:  # TAHOE_CREATE_PROB_RELAYOUT: subtrees ordered by dense_node_t.weight (Struct.h:1775-1825)
            _check(lib.tahoe_forest_create_ex(C.byref(self._h), nodes.ctypes.data if nodes.size else None,
                                              C.byref(self.params), CREATE_PROB_RELAYOUT), "tahoe_forest_create_ex")
        else:
            _check(lib.tahoe_forest_create(C.byref(self._h), nodes.ctypes.data if nodes.size else None,
                                           C.byref(self.params)), "tahoe_forest_create")
        self.num_trees, self.depth, self.num_cols = num_trees, depth, num_cols

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.tahoe_forest_destroy(self._h)
            self._h = _vp()

    __del__ = close

    def _check_data(self, data):
        assert data.is_cuda and data.is_contiguous() and data.dtype.is_floating_point and data.element_size() == 4
        assert data.dim() == 2 and data.shape[1] == self.num_cols, (tuple(data.shape), self.num_cols)

    def predict(self, data, preds=None, stream=None):
        import torch

        self._check_data(data)
        rows = data.shape[0]
        if preds is None:
            preds = torch.empty(rows, dtype=torch.float32, device=data.device)
        _check(lib.tahoe_forest_predict(self._h, _ptr(preds), _ptr(data), rows, _stream(stream)),
               "tahoe_forest_predict")
        return preds

    def predict_raw(self, data, sums=None, stream=None):
        import torch

        self._check_data(data)
        rows = data.shape[0]
        if sums is None:
            sums = torch.empty(rows, dtype=torch.float32, device=data.device)
        _check(lib.tahoe_forest_predict_raw(self._h, _ptr(sums), _ptr(data), rows, _stream(stream)),
               "tahoe_forest_predict_raw")
        return sums

    def predict_accumulate(self, data, sums, stream=None):
        """Continues the running float32 sums in `sums` (the trees before this forest) through this forest's trees,
        in place (tahoe_forest_predict_accumulate): the step of a chained, bit-exact tree-sharded predict."""
        self._check_data(data)
        assert sums.is_cuda and sums.is_contiguous() and sums.dtype.is_floating_point and sums.element_size() == 4
        assert sums.numel() == data.shape[0]
        _check(lib.tahoe_forest_predict_accumulate(self._h, _ptr(sums), _ptr(data), data.shape[0], _stream(stream)),
               "tahoe_forest_predict_accumulate")
        return sums

    def predict_leaf_idx(self, data, want_sums: bool = True, stream=None):
        import torch

        self._check_data(data)
        rows = data.shape[0]
        leaf = torch.empty((rows, self.num_trees), dtype=torch.int32, device=data.device)
        sums = torch.empty(rows, dtype=torch.float32, device=data.device) if want_sums else None
        _check(lib.tahoe_forest_predict_leaf_idx(self._h, _ptr(leaf), _ptr(sums) if want_sums else None,
                                                 _ptr(data), rows, _stream(stream)),
               "tahoe_forest_predict_leaf_idx")
        return leaf, sums

    def set_strategy(self, strategy: int) -> None:
        _check(lib.tahoe_forest_set_strategy(self._h, strategy), "tahoe_forest_set_strategy")

    def get_strategy(self, rows: int) -> int:
        return lib.tahoe_forest_get_strategy(self._h, rows)

    def kernel_form(self, rows: int) -> str:
        """Name of the kernel form a predict of `rows` rows launches (TAHOE_FORM_*, include/tahoe_amd.h)."""
        return lib.tahoe_kernel_form_name(lib.tahoe_forest_get_kernel_form(self._h, rows)).decode()

    def reserve(self, rows: int) -> None:
        _check(lib.tahoe_forest_reserve(self._h, rows), "tahoe_forest_reserve")

    def predict_host(self, data: np.ndarray, preds: np.ndarray = None, chunk_rows: int = 0) -> np.ndarray:
        """Host-resident batch: chunked upload overlapped with the traversal (tahoe_forest_predict_host)."""
        if data.dtype != np.float32 or data.ndim != 2 or data.shape[1] != self.num_cols or not data.flags.c_contiguous:
            raise ValueError(f"data must be C-contiguous float32 [rows, {self.num_cols}]")
        if preds is None:
            preds = np.empty(data.shape[0], dtype=np.float32)
        if (not isinstance(preds, np.ndarray) or preds.dtype != np.float32 or preds.shape != (data.shape[0],)
                or not preds.flags.c_contiguous or not preds.flags.writeable):
            raise ValueError(f"preds must be a writeable C-contiguous float32 array of shape ({data.shape[0]},)")
        _check(lib.tahoe_forest_predict_host(self._h, preds.ctypes.data, data.ctypes.data, data.shape[0], chunk_rows),
               "tahoe_forest_predict_host")
        return preds

    def check(self, stream=None) -> None:
        """Waits for the stream; raises if a kernel flagged an internal error."""
        _check(lib.tahoe_forest_check(self._h, _stream(stream)), "tahoe_forest_check")

    def info(self) -> ForestInfo:
        info = ForestInfo()
        _check(lib.tahoe_forest_get_info(self._h, C.byref(info)), "tahoe_forest_get_info")
        return info

    def set_profiling(self, max_launches: int) -> None:
        _check(lib.tahoe_forest_set_profiling(self._h, int(max_launches)), "tahoe_forest_set_profiling")

    def kernel_times_ms(self, capacity: int = 4096) -> np.ndarray:
        """Durations (ms) of the traversal kernels launched since set_profiling; waits for them."""
        out = np.empty(capacity, dtype=np.float32)
        n = _i()
        _check(lib.tahoe_forest_kernel_times(self._h, out.ctypes.data, capacity, C.byref(n)),
               "tahoe_forest_kernel_times")
        return out[: n.value].copy()

    def prepass_times_ms(self, capacity: int = 4096) -> np.ndarray:
        """Durations (ms) of the pre-pass kernel (QRING's quantise kernel) of the same launches."""
        out = np.empty(capacity, dtype=np.float32)
        n = _i()
        _check(lib.tahoe_forest_prepass_times(self._h, out.ctypes.data, capacity, C.byref(n)),
               "tahoe_forest_prepass_times")
        return out[: n.value].copy()


# ---- sparse (irregular) forests ----
SPARSE_NODE_DTYPE = np.dtype([("val", "<f4"), ("bits", "<i4"), ("left_idx", "<i4")])  # sparse_node_t, Struct.h:50-54


def dense_to_sparse(nodes: np.ndarray, num_trees: int, depth: int):
    """dense2sparse (BaseTahoeTest.h:728-764) -> (sparse nodes, root offsets int32[num_trees])."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pn, pt, n = _vp(), _vp(), _sz()
    _check(lib.tahoe_dense_to_sparse(nodes.ctypes.data, num_trees, depth, C.byref(pn), C.byref(pt), C.byref(n)),
           "tahoe_dense_to_sparse")
    try:
        sn = np.frombuffer((C.c_char * (n.value * 12)).from_address(pn.value), dtype=SPARSE_NODE_DTYPE, count=n.value).copy()
        tr = np.frombuffer((C.c_char * (num_trees * 4)).from_address(pt.value), dtype=np.int32, count=num_trees).copy() \
            if num_trees else np.empty(0, np.int32)
    finally:
        lib.tahoe_free_host(pn)
        lib.tahoe_free_host(pt)
    return sn, tr


def synth_sparse_forest(num_trees: int, num_cols: int, min_depth: int = 4, max_depth: int = 24, leaf_prob: float = 0.32,
                        max_tree_nodes: int = 65535, seed: int = 44):
    """BASELINE config 5 generator -> (sparse nodes, root offsets)."""
    n = _sz()
    _check(lib.tahoe_synth_sparse_forest(None, None, C.byref(n), num_trees, num_cols, min_depth, max_depth, leaf_prob,
                                         max_tree_nodes, seed), "tahoe_synth_sparse_forest")
    nodes = np.empty(n.value, dtype=SPARSE_NODE_DTYPE)
    trees = np.empty(num_trees, dtype=np.int32)
    _check(lib.tahoe_synth_sparse_forest(nodes.ctypes.data, trees.ctypes.data, C.byref(n), num_trees, num_cols, min_depth,
                                         max_depth, leaf_prob, max_tree_nodes, seed), "tahoe_synth_sparse_forest")
    return nodes, trees


class SparseForest(Forest):
    """tahoe_sparse_forest_create: nodes[SPARSE_NODE_DTYPE] + root offsets; predict* as Forest."""

    def __init__(self, nodes: np.ndarray, trees: np.ndarray, num_cols: int, missing: float = 0.0, output: int = OUT_RAW,
                 threshold: float = 0.0, global_bias: float = 0.0):
        nodes = np.ascontiguousarray(nodes, dtype=SPARSE_NODE_DTYPE)
        trees = np.ascontiguousarray(trees, dtype=np.int32)
        self.params = ForestParams(int(nodes.size), 0, int(trees.size), num_cols, 0, output, threshold, global_bias, 0,
                                   missing)
        self._h = _vp()
        _check(lib.tahoe_sparse_forest_create(C.byref(self._h), trees.ctypes.data if trees.size else None,
                                              nodes.ctypes.data if nodes.size else None, C.byref(self.params)),
               "tahoe_sparse_forest_create")
        self.num_trees, self.depth, self.num_cols = int(trees.size), 0, num_cols


def transform_preds(preds, output: int, num_trees_total: int, threshold: float, global_bias: float, stream=None):
    _check(lib.tahoe_transform_preds(_ptr(preds), preds.numel(), output, num_trees_total, threshold, global_bias,
                                     _stream(stream)), "tahoe_transform_preds")
    return preds
