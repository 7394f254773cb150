"""ctypes binding of libgraphpope_hip.so (C ABI: include/graphpope_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgraphpope_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_INDEX, ERR_HOP_OVERFLOW, ERR_WORKSPACE, ERR_NO_DEVICE, ERR_UNSORTED = 0, -1, -2, -3, -4, -5, -6, -7
METRIC = {"distance": 0, "similarity": 1, "euclidean": 2}
KNOB_LIVE_MODE, KNOB_FINALIZE_VARIANT, KNOB_FINALIZE_BLOCKS, KNOB_GEMM_TILE = 0, 1, 2, 3
KNOB_PAIRWISE_KERNEL, KNOB_COPY_BATCHES, KNOB_FAIL_HOST_REGISTER = 4, 5, 7
KNOB_SAGE_FORWARD_OVERLAP, KNOB_GEMM_SMALL_TILE16 = 14, 15
KNOB_GEMM_TILE16_BUFFERS = 18
KNOB_PREPARE_MERGE = 19
KNOB_STREAMK_XCD = 20

# name -> (restype, argtypes); exactly the symbols include/graphpope_hip.h declares
SIGNATURES = {
    "pope_last_error": (c_char_p, []),
    "pope_version": (c_char_p, []),
    "pope_require_device": (c_int, [POINTER(c_int32)]),
    "pope_debug_set": (c_int, [c_int32, c_int32]),
    "pope_host_copy_2d": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32]),
    "pope_copy_2d_to_host": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "pope_assemble_host_result": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32,
                                          c_int32, c_void_p]),
    "pope_assemble_begin": (c_void_p, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32]),
    "pope_assemble_finish": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "pope_assemble_abort": (None, [c_void_p]),
    "pope_assemble_prepare": (None, [c_int32]),
    "pope_assemble_ring_ready": (c_int32, []),
    "pope_assemble_finish_codes": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "pope_geodesic_hop_codes": (c_int, [c_void_p, c_int32, c_int32, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "pope_host_pin": (c_int, [c_void_p, c_size_t]),
    "pope_host_unpin": (c_int, [c_void_p]),
    "pope_copy_to_device": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "pope_csr_scratch_bytes": (c_size_t, [c_int64, c_int64]),
    "pope_csr_aux_elems": (c_size_t, [c_int64]),
    "pope_csr_build": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_int32, c_void_p]),
    "pope_csr_build_canonical": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pope_pagerank_weights": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pope_pagerank_step": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p]),
    "pope_kmeans_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "pope_column_moments": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pope_shift_columns": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "pope_kmeans_plusplus": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "pope_kmeans_lloyd_step": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "pope_words": (c_int32, [c_int32]),
    "pope_plane_bytes": (c_size_t, [c_int64, c_int32]),
    "pope_bfs_scratch_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
    "pope_geodesic_bfs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                  c_void_p, c_size_t, POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "pope_geodesic_bfs_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                        c_void_p, c_size_t, c_void_p]),
    "pope_geodesic_bfs_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                         c_void_p, c_size_t, POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "pope_geodesic_finalize": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int64,
                                       c_int32, c_void_p]),
    "pope_geodesic_run_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "pope_geodesic_run_planes": (c_void_p, [c_void_p, c_int64, c_int64, c_int32, c_int32]),
    "pope_geodesic_run": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int64,
                                  c_int32, c_void_p, c_size_t, POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "pope_profile_levels": (None, [c_int32]),
    "pope_profile_read": (c_int32, [c_void_p, c_void_p, c_int32]),
    "pope_geodesic_finalize_shards": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_int64, c_int32, c_void_p, c_int32,
                                              c_void_p, c_int64, c_void_p]),
    "pope_finalize_kernel_name": (c_int, [c_int64, c_int32, c_int32, c_int32, c_int32, c_char_p, c_size_t]),
    "pope_level_kernel_name": (c_int, [c_int64, c_int32, c_char_p, c_size_t]),
    "pope_column_stats_scratch_bytes": (c_size_t, [c_int32]),
    "pope_geodesic_column_stats": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pope_geodesic_hops": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
    "pope_pairwise_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "pope_pairwise_minmax": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                     c_int32, c_void_p, c_size_t, c_void_p]),
    "pope_pairwise_by_id_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "pope_pairwise_features_by_id": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int32,
                                             c_void_p, c_size_t, c_void_p]),
    "pope_pairwise_features": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                       c_int32, c_void_p, c_size_t, c_void_p]),
    "pope_concat": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    "sage_conv_scratch_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int32, c_int32]),
    "sage_conv_forward_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sage_conv_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p,
                                  c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "sage_gather_mean": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    "sage_conv_forward_indexed": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                          c_void_p, c_void_p]),
    "sage_conv_forward_stats": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int32,
                                        c_void_p, c_void_p]),
    "sage_conv_forward_indexed_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32,
                                                c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "sage_conv_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int32,
                                   c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p, c_void_p]),
    "sage_conv_forward_indexed_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sage_conv_backward_indexed_scratch_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int32, c_int32]),
    "sage_conv_backward_indexed": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p,
                                           c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "sage_bn_scratch_bytes": (c_size_t, [c_int32]),
    "sage_bn_relu_dropout_forward": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                             c_int32, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                             c_void_p, c_void_p, c_void_p]),
    "sage_bn_relu_dropout_forward_stats": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                                   c_int32, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sage_bn_relu_dropout_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int32, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p, c_void_p, c_void_p]),
    "sage_bn_relu_dropout_backward_bias": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_int32, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "sage_cross_entropy_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int32, c_void_p]),
    "sage_cross_entropy_backward": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sage_adam_step": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                               c_double, c_int64, c_void_p, c_void_p]),
    "sage_adam_step_loss": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                    c_double, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "sage_advance_counters": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "sage_copy_segments": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sage_sample_batch_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_uint64, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sage_sample_epoch_batch_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_uint64,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sage_sample_scratch_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "sage_sample_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_uint64, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sage_sample_hop": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_uint64, c_int32, c_void_p,
                                c_void_p, c_int64, c_void_p, POINTER(c_int64), POINTER(c_int64), c_void_p, c_size_t, c_void_p]),
}


class PopeError(RuntimeError):
    """A libgraphpope_hip call returned a negative code (shim raises RuntimeError)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libgraphpope_hip error {code}: {message}")
        self.code = code


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C graphpope_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). graphpope_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so is stale: also loud
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code: int) -> None:
    if code != OK:
        raise PopeError(code, load().pope_last_error().decode())


def ptr(t) -> c_void_p:
    """Device (or host) address of a torch tensor / None as a void*."""
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """``with on_device(dev):`` -- make `dev` the current HIP device for the library call, as ``torch.cuda.device(dev)``
    does, but free when it already is (the common case: a context-manager round trip through torch costs ~4 us per op
    in a launch-bound training step)."""
    import torch
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return _NO_GUARD if idx == torch.cuda.current_device() else torch.cuda.device(idx)

