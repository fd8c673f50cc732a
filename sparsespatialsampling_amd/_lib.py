"""
ctypes bindings of the two native libraries of this package:

* ``libs3hip.so``  -- the MI355X (gfx950) kernels behind the C ABI declared in ``include/s3hip.h``
* ``libs3topo.so`` -- the host-side topology engine (``csrc/topology.cpp``; plain C++, no GPU)

Both are built in-tree by ``__graft_entry__.build()`` (or ``python -m sparsespatialsampling_amd.build``).  There is no
fallback: if a library is missing, or no HIP device is present when a compute entry point is reached, the caller gets a
``HipUnavailableError`` / ``S3HipError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_HERE, "libs3hip.so")
TOPO_SO = os.environ.get("S3_TOPO_SO") or os.path.join(_HERE, "libs3topo.so")      # (S3_TOPO_SO: the sanitizer build of the CPU test job)


class HipUnavailableError(RuntimeError):
    """libs3hip.so is missing/unloadable or no HIP device is visible.  There is no CPU fallback."""


class S3HipError(RuntimeError):
    """A libs3hip entry point returned an error code."""


_hip = None
_topo = None
ABI_VERSION = 6          # S3_ABI_VERSION of include/s3hip.h this file was written against
_REBUILD = "python -c 'import __graft_entry__ as g; g.build()'"

c_i64, c_i32, c_int, c_dbl, c_vp = C.c_int64, C.c_int32, C.c_int, C.c_double, C.c_void_p

# name -> (restype, argtypes); must list every symbol include/s3hip.h declares (tests/test_abi.py checks this)
HIP_SIGNATURES = {
    "s3_last_error": (C.c_char_p, []),
    "s3_abi_version": (c_int, []),
    "s3_shutdown": (c_int, []),
    "s3_debug_abort_backtrace": (c_int, []),
    "s3_device_count": (c_int, [C.POINTER(c_int)]),
    "s3_set_device": (c_int, [c_int]),
    "s3_malloc": (c_int, [C.POINTER(c_vp), C.c_size_t]),
    "s3_free": (c_int, [c_vp]),
    "s3_memcpy_h2d": (c_int, [c_vp, c_vp, C.c_size_t, c_vp]),
    "s3_memcpy_d2h": (c_int, [c_vp, c_vp, C.c_size_t, c_vp]),
    "s3_download": (c_int, [c_vp, c_vp, C.c_size_t, c_vp]),
    "s3_row_moments": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp]),
    "s3_row_abs_moments": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp]),
    "s3_centered_gemm": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "s3_host_register": (c_int, [c_vp, C.c_size_t, C.POINTER(c_vp)]),
    "s3_host_unregister": (c_int, [c_vp]),
    "s3_snapshot_major_rows": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "s3_upload_rows_indexed": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "s3_upload_row_pieces": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "s3_upload_rows": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "s3_stream_synchronize": (c_int, [c_vp]),
    "s3_knn_create": (c_int, [c_vp, c_i64, c_int, c_dbl, c_vp, C.POINTER(c_vp)]),
    "s3_knn_destroy": (None, [c_vp]),
    "s3_knn_info": (c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "s3_knn_set_values": (c_int, [c_vp, c_vp, c_vp]),
    "s3_knn_query": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "s3_idw_predict": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "s3_make_children": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp]),
    "s3_child_gain": (c_int, [c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp]),
    "s3_child_gain_reuse": (c_int, [c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp,
                                    c_i64, c_vp, c_vp]),
    "s3_mask_box": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "s3_mask_sphere": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp, c_dbl, c_int, c_int, c_vp, c_vp]),
    "s3_mask_cylinder": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_vp, c_dbl, c_dbl, c_dbl, c_int, c_int,
                                 c_int, c_vp, c_vp]),
    "s3_mask_polygon": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "s3_mask_triangle": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_int, c_int, c_vp, c_vp]),
    "s3_mask_prism": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_vp, c_dbl, c_vp, c_vp, c_int, c_int, c_vp,
                              c_vp]),
    "s3_mask_tetrahedra": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "s3_commit_batch": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "s3_sumsq_leaf": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "s3_topn_scratch_bytes": (C.c_size_t, [c_i64, c_i64]),
    "s3_topn_leaf": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, C.POINTER(c_i64), c_vp, c_vp]),
    "s3_idw_weights": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "s3_interp": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_i64, c_i64, c_vp, c_vp]),
    "s3_snapshot_major": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_vp]),
    "s3_interp_plan_create": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_int, c_int, c_vp, C.POINTER(c_vp)]),
    "s3_interp_plan_destroy": (None, [c_vp]),
    "s3_interp_plan_info": (c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "s3_interp_plan_partition": (c_int, [c_vp, c_int, c_vp, C.POINTER(c_i64), c_vp]),
    "s3_interp_plan_set_weights": (c_int, [c_vp, c_vp, c_vp]),
    "s3_interp_planned": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp]),
    "s3_interp_plan_set_source_ids": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "s3_interp_planned_src": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "s3_yard_stream": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "s3_yard_plan_loads": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, C.POINTER(c_i64)]),
    "s3_debug_reload_env": (c_int, []),
    "s3_comm_available": (c_int, []),
    "s3_comm_gather_to_root": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "s3_interp_plan_cost_profile": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "s3_comm_unique_id": (c_int, [c_vp, C.c_size_t]),
    "s3_comm_init": (c_int, [c_vp, C.c_size_t, c_int, c_int, C.POINTER(c_vp)]),
    "s3_comm_destroy": (None, [c_vp]),
    "s3_comm_rank": (c_int, [c_vp, C.POINTER(c_int), C.POINTER(c_int)]),
    "s3_comm_allgather_inplace": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "s3_comm_allreduce_f64": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "s3_sumsq_blocks": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "s3_sum_ordered": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "s3_weighted_gram_scratch_bytes": (C.c_size_t, [c_i64, c_i64]),
    "s3_weighted_gram": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "s3_sym_eig_available": (c_int, []),
    "s3_sym_eig_scratch_bytes": (C.c_size_t, [c_i64]),
    "s3_sym_eig": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "s3_mark_rows": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "s3_compact_rows": (c_int, [c_vp, c_i64, c_vp, C.POINTER(c_i64), c_vp]),
    "s3_remap_indices": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "s3_spatial_order": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "s3_exclusive_scan": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "s3_sort_pairs": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "s3_positions_of": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "s3_gather_rows": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "s3_topo_create": (c_int, [c_int, c_dbl, c_vp, C.POINTER(c_vp)]),
    "s3_topo_destroy": (None, [c_vp]),
    "s3_topo_refine": (c_int, [c_vp, c_vp, c_i64, c_int, C.POINTER(c_i64)]),
    "s3_topo_relink_parent_of": (c_int, [c_vp, c_vp, c_i64]),
    "s3_topo_mark_invalid": (c_int, [c_vp, c_vp, c_i64]),
    "s3_topo_sync": (c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_int)]),
    "s3_topo_table": (c_int, [c_vp, c_int, C.POINTER(c_vp)]),
    "s3_topo_finalize": (c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "s3_topo_export_grid": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "s3_topo_gather_cells": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
}

TOPO_SIGNATURES = {
    "s3t_create": (c_vp, [c_int, c_dbl, c_vp]),
    "s3t_destroy": (None, [c_vp]),
    "s3t_n_cells": (c_i64, [c_vp]),
    "s3t_n_nodes": (c_i64, [c_vp]),
    "s3t_level": (c_vp, [c_vp]),
    "s3t_parent": (c_vp, [c_vp]),
    "s3t_first_child": (c_vp, [c_vp]),
    "s3t_nb": (c_vp, [c_vp]),
    "s3t_node_idx": (c_vp, [c_vp]),
    "s3t_center": (c_vp, [c_vp]),
    "s3t_nodes": (c_vp, [c_vp]),
    "s3t_refine": (c_i64, [c_vp, c_vp, c_i64, c_int]),
    "s3t_relink_parent_of": (None, [c_vp, c_vp, c_i64]),
    "s3t_mark_invalid": (None, [c_vp, c_vp, c_i64]),
    "s3t_check_nb": (c_int, [c_vp, c_i64, c_vp]),
    "s3t_finalize": (c_i64, [c_vp, C.POINTER(c_i64)]),
    "s3t_export_grid": (None, [c_vp, c_vp, c_int, c_vp]),
    "s3t_gather_cells": (None, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "s3t_selfcheck": (c_int, [c_int]),
    "s3t_submit": (c_int, [c_vp, c_int, c_vp, c_i64, c_int]),
    "s3t_sync": (c_int, [c_vp]),
    "s3t_stats": (None, [c_vp, c_vp]),
    "s3set_create": (c_vp, []),
    "s3set_destroy": (None, [c_vp]),
    "s3set_len": (c_i64, [c_vp]),
    "s3set_mask": (c_i64, [c_vp]),
    "s3set_fill": (c_i64, [c_vp]),
    "s3set_table": (c_vp, [c_vp]),
    "s3set_contains": (c_int, [c_vp, c_i64]),
    "s3set_add": (c_int, [c_vp, c_i64]),
    "s3set_discard": (None, [c_vp, c_i64]),
    "s3set_update_ids": (c_int, [c_vp, c_vp, c_i64]),
    "s3set_update_range": (c_int, [c_vp, c_i64, c_i64]),
    "s3set_update_rangeset": (c_int, [c_vp, c_i64, c_i64]),
    "s3set_range_mask": (c_i64, [c_i64]),
    "s3set_difference_update_ids": (c_int, [c_vp, c_vp, c_i64]),
    "s3set_update_rangeset_async": (c_int, [c_vp, c_i64, c_i64]),
    "s3set_difference_update_ids_async": (c_int, [c_vp, c_vp, c_i64]),
    "s3set_wait": (c_int, [c_vp]),
    "s3set_update_set": (c_int, [c_vp, c_vp]),
    "s3set_difference_update": (c_int, [c_vp, c_vp]),
    "s3set_to_array": (None, [c_vp, c_vp]),
    "s3set_update_flagged": (c_int, [c_vp, c_vp, c_vp, c_i64]),
}


def _bind(lib, signatures, path):
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:                 # a library older than these bindings
            raise HipUnavailableError(f"{path} does not export {name}: stale build -- rebuild it with `{_REBUILD}`.") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_lib():
    """Load libs3hip.so (no device is touched by loading)."""
    global _hip
    if _hip is None:
        # torch bundles its own HIP runtime (soname libamdhip64.so.7, the same soname libs3hip.so needs).  It must be
        # in the process first so that the loader binds libs3hip.so to that one copy: two HIP/HSA runtimes in one
        # process cannot both own the device, and device pointers / streams must come from the runtime torch uses.
        import torch  # noqa: F401
        if not os.path.exists(HIP_SO):
            raise HipUnavailableError(f"{HIP_SO} not found -- build it with `python -c 'import __graft_entry__ as g; "
                                      f"g.build()'`.  This package has no CPU fallback.")
        try:
            lib = _bind(C.CDLL(HIP_SO), HIP_SIGNATURES, HIP_SO)
        except OSError as e:
            raise HipUnavailableError(f"cannot load {HIP_SO}: {e}") from e
        if lib.s3_abi_version() != ABI_VERSION:
            raise HipUnavailableError(f"{HIP_SO} reports ABI version {lib.s3_abi_version()}, these bindings need "
                                      f"{ABI_VERSION}: stale build -- rebuild it with `{_REBUILD}`.")
        _hip = lib
        # the library's transfer lanes are joined before the interpreter and the HIP runtime go down.  atexit runs its hooks in
        # reverse order of registration and torch (imported above) has registered its own already: this one runs before them.
        import atexit
        atexit.register(_shutdown)
        if os.environ.get("S3_ABORT_BACKTRACE"):                 # (debugging aid: native frames of a thread that calls abort();
                                                                 #  "1" -> stderr, an absolute path -> appended to that file)
            lib.s3_debug_abort_backtrace()
    return _hip


def _shutdown():
    if _hip is not None:
        try:
            _hip.s3_shutdown()
        except Exception:       # interpreter already half gone
            pass


def topo_lib():
    global _topo
    if _topo is None:
        if not os.path.exists(TOPO_SO):
            raise HipUnavailableError(f"{TOPO_SO} not found -- build it with `__graft_entry__.build()`.")
        _topo = _bind(C.CDLL(TOPO_SO), TOPO_SIGNATURES, TOPO_SO)
    return _topo


def check(rc, what=""):
    if rc != 0:
        msg = hip_lib().s3_last_error().decode(errors="replace")
        if rc == -2:
            raise HipUnavailableError(f"{what}: no usable HIP device ({msg})")
        raise S3HipError(f"{what} failed with code {rc}: {msg}")


def require_device():
    """Raise unless at least one HIP device is visible to libs3hip.so."""
    n = c_int(0)
    rc = hip_lib().s3_device_count(C.byref(n))
    if rc != 0 or n.value < 1:
        raise HipUnavailableError("no HIP device visible to libs3hip.so -- the S^3 hot path runs on MI355X only "
                                  "(there is no CPU fallback).")
    return n.value
