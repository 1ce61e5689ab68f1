"""dashing_amd -- MI355X-native (gfx950) implementation of dashing's HLL sketch-and-compare
hot path.  The product is libdashing_hip.so (C-ABI in include/dashing_hip.h) and the
dashing-amd CLI; this package is the thin ctypes binding used by tests and bench.py."""
from .api import (  # noqa: F401
    Context,
    PinnedArray,
    DshError,
    ESTIM_ORIGINAL,
    ESTIM_ERTL_IMPROVED,
    ESTIM_ERTL_MLE,
    MASH_DIST,
    JI,
    FULL_MASH_DIST,
    SIZES,
    FULL_CONTAINMENT_DIST,
    CONTAINMENT_INDEX,
    CONTAINMENT_DIST,
    SYMMETRIC_CONTAINMENT_INDEX,
    SYMMETRIC_CONTAINMENT_DIST,
    device_count,
    comm_unique_id,
    backend_name,
    lib_path,
    load_library,
    partition_rows,
    balance_rows,
    tri_index,
    tri_span,
)
