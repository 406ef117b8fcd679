"""tahoe_amd -- Python binding (ctypes) of libtahoe_amd.so, the MI355X implementation of Tahoe's
batched tree-ensemble traversal.  The product is the C-ABI library (include/tahoe_amd.h); this
package only loads it for tests and bench.py.  It never computes anything itself and has no CPU
fallback: without the built library, importing `tahoe_amd.capi` raises."""

from .capi import (  # noqa: F401
    Forest,
    PinnedArray,
    ForestParams,
    TahoeError,
    lib,
    load_data,
    load_data_bin,
    load_model,
    load_model_bin,
    save_data_bin,
    save_model_bin,
    synth_data,
    synth_data_hist,
    synth_forest,
    synth_forest_hist,
    write_data,
    write_model,
    NODE_DTYPE,
    OUT_RAW,
    OUT_AVG,
    OUT_SIGMOID,
    OUT_THRESHOLD,
    STRATEGY_AUTO,
    STRATEGY_DIRECT,
    STRATEGY_ROWTILE,
    STRATEGY_TILEBLOCK,
    STRATEGY_TILERING,
    STRATEGY_QRING,
    STRATEGY_NAMES,
)
