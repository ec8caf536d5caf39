"""ctypes binding of libslic_hip.so (C ABI: include/slic_hip.h).  Fails loudly — no fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLIC_LIB_PATH") or os.path.join(_HERE, "csrc", "libslic_hip.so")     # override: kernel experiments

c_void_p, c_int, c_int64, c_size_t, c_float, c_double = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float, ctypes.c_double)
P, I, L, F, Dbl = c_void_p, c_int, c_int64, c_float, c_double

# name -> (restype, argtypes).  Must list every symbol include/slic_hip.h declares
# (tests/test_abi.py parses the header and checks).
SIGNATURES = {
    "slic_version": (I, []),
    "slic_last_error": (ctypes.c_char_p, []),
    "slic_device_check": (I, []),
    # k-means
    "slic_kmeans_cnorm": (I, [P, I, I, I, P, P]),
    "slic_kmeans_assign_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeans_assign": (I, [P, L, I, I, P, I, I, P, P, P, P, P, P, P]),
    "slic_kmeans_accumulate_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeans_accumulate": (I, [P, L, I, I, P, I, P, P, P, P]),
    "slic_kmeans_combine_shards": (I, [P, P, L, I, I, I, P, P, P]),
    "slic_kmeans_dist_to_assigned": (I, [P, L, I, I, P, I, P, P, P]),
    "slic_sum_f32_to_f64_workspace_bytes": (c_size_t, [L]),
    "slic_sum_f32_to_f64": (I, [P, L, P, P, P]),
    "slic_kmeans_select_far": (I, [P, L, I, P, P, P]),
    "slic_kmeans_apply_relocation": (I, [P, I, P, P, I, I, P, P, P]),
    "slic_kmeans_finalize": (I, [P, P, P, I, I, P, P, P, P, I, P, P, P]),
    "slic_kmeans_permute_k8": (I, [P, L, I, I, P, I, P]),
    "slic_kmeans_lloyd_step_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeans_lloyd_step": (I, [P, P, L, I, I, P, P, P, I, P, P, P, P, P, P, P, P, P, I, P, P, P]),
    "slic_kmeans_assign_perm": (I, [P, L, I, I, P, I, I, P, P, P, P, P, P, P]),
    "slic_kmeans_lloyd_local_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeans_lloyd_local": (I, [P, P, L, I, I, P, P, I, P, P, P, I, P, P]),
    "slic_kmeans_lloyd_global": (I, [P, I, L, I, P, I, I, P, P, P, P, P, P, I, P, P]),
    "slic_comm_unique_id": (I, [P]),
    "slic_comm_create": (I, [P, I, I, P]),
    "slic_comm_create_timeout": (I, [P, I, I, I, P]),
    "slic_comm_wait": (I, [P, P, I]),
    "slic_comm_wait_event": (I, [P, P, I]),
    "slic_comm_abort": (I, [P]),
    "slic_allreduce_f32": (I, [P, P, L, P]),
    "slic_allreduce_f64": (I, [P, P, L, P]),
    "slic_comm_destroy": (I, [P]),
    "slic_oneshot_create": (I, [I, I, c_int64, I, P, P]),
    "slic_oneshot_connect": (I, [P, P]),
    "slic_allreduce_oneshot_f64": (I, [P, P, c_int64, P]),
    "slic_oneshot_check": (I, [P]),
    "slic_oneshot_info": (I, [P, P]),
    "slic_oneshot_destroy": (I, [P]),
    "slic_col_stats_workspace_bytes": (c_size_t, [L, I]),
    "slic_col_stats": (I, [P, L, I, I, P, P, P, P]),
    "slic_sub_rowvec": (I, [P, L, I, I, P, P, I, P]),
    "slic_l2norm_rows": (I, [P, L, I, I, P, I, P]),
    "slic_kmeanspp_run_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeanspp_run": (I, [P, L, I, I, I, I, I, P, P, P, P, P, P]),
    "slic_kmeanspp_run_batch_workspace_bytes": (c_size_t, [L, I, I]),
    "slic_kmeanspp_run_batch": (I, [P, P, L, I, I, I, P, I, I, P, P, P, P]),
    "slic_kmeanspp_step_workspace_bytes": (c_size_t, [L, I]),
    "slic_kmeanspp_step": (I, [P, L, I, I, P, I, P, P, P, P, P]),
    "slic_cumsum_search_workspace_bytes": (c_size_t, [L]),
    "slic_cumsum_search": (I, [P, L, P, I, P, P, P]),
    # conv / linear
    "slic_conv_tile_m": (I, [P, I]),
    "slic_conv_gemm": (I, [P, I, P]),
    "slic_conv_gemm_multi": (I, [P, I, I, P]),
    "slic_conv_gemm_tailsplit_workspace_bytes": (c_size_t, [P, I, I, I]),
    "slic_conv_gemm_tailsplit": (I, [P, I, I, I, P, P]),
    "slic_conv_wgrad_workspace_bytes": (c_size_t, [P, I]),
    "slic_conv_row_table": (I, [P, P, P]),
    "slic_conv_wgrad": (I, [P, P, I, I, I, I, P, P, P]),
    "slic_pack_weight_fwd": (I, [P, I, I, I, I, I, P, P]),
    "slic_pack_weight_fwd_runs": (I, [P, I, I, I, I, I, I, P, P]),
    "slic_pack_weight_dgrad": (I, [P, I, I, I, I, I, P, P]),
    "slic_pack_weight_wino": (I, [P, I, I, I, P, P]),
    "slic_pack_weight_wino2": (I, [P, I, I, I, P, P]),
    "slic_conv_wgrad_wino2_workspace_bytes": (c_size_t, [P, I]),
    "slic_conv_wino2_tile_table": (I, [P, P, P]),
    "slic_conv_wgrad_wino2": (I, [P, P, I, P, P, P, P]),
    "slic_conv_wgrad_wino_workspace_bytes": (c_size_t, [P, I]),
    "slic_conv_wino_tile_table": (I, [P, P, P]),
    "slic_conv_wgrad_wino": (I, [P, P, I, P, P, P, P]),
    "slic_ncdhw_to_ndhwc": (I, [P, I, I, L, I, P, P]),
    "slic_ncdhw_to_ndhwc_wpad": (I, [P, I, I, L, I, I, I, P, P]),
    # batch norm / pool
    "slic_bn_finalize_workspace_bytes": (c_size_t, [I, I]),
    "slic_bn_finalize": (I, [P, I, I, I, L, F, F, P, P, P, P, P, P, P, P, P, P]),
    "slic_bn_eval_affine": (I, [P, P, P, P, F, I, P, P, P]),
    "slic_bn_merge_stats": (I, [P, I, I, I, L, P, P, P]),
    "slic_bn_finalize_sync": (I, [P, I, I, F, F, P, P, P, P, P, P, P, P, P]),
    "slic_bn_bwd_sums_workspace_bytes": (c_size_t, [L, I, I]),
    "slic_bn_bwd_sums": (I, [P, I, P, P, P, P, P, L, I, P, P, P, P, P, P]),
    "slic_bn_bwd_apply": (I, [P, P, P, P, P, P, P, L, I, P, P]),
    "slic_bn_apply": (I, [P, P, P, P, I, L, I, P, P]),
    "slic_bn_bwd_workspace_bytes": (c_size_t, [L, I, I]),
    "slic_bn_bwd_rows_per_partial": (I, []),
    "slic_bn_bwd": (I, [P, P, P, P, P, P, L, I, P, P, P, P, P, P]),
    "slic_bn_bwd_fused_workspace_bytes": (c_size_t, [I, I]),
    "slic_bn_bwd_fused": (I, [P, I, P, P, P, P, P, L, I, P, P, P, P, P]),
    "slic_avgpool_fwd": (I, [P, I, I, I, P, P]),
    "slic_avgpool_bwd": (I, [P, I, I, I, P, P]),
    "slic_maxpool3d_fwd": (I, [P, I, I, I, I, I, P, P, P]),
    "slic_maxpool3d_bwd": (I, [P, P, I, I, I, I, I, P, P]),
    "slic_shortcut_a": (I, [P, I, I, I, I, I, I, I, P, P]),
    "slic_colsum": (I, [P, L, I, P, P]),
    # losses
    "slic_ntxent_workspace_bytes": (c_size_t, [I, I]),
    "slic_ntxent_fwd": (I, [P, I, I, I, F, P, P, P]),
    "slic_ntxent_bwd": (I, [P, I, I, F, P, P, I, P]),
    "slic_pair_distance": (I, [P, P, I, I, I, P, P]),
    "slic_pair_distance_bwd": (I, [P, P, P, I, I, I, P, P, P]),
    "slic_margin_cos_fwd": (I, [P, P, P, I, I, F, P, P, P, P]),
    "slic_margin_cos_bwd": (I, [P, P, P, P, I, I, P, P, P, P, P]),
    "slic_margin_euclid_fwd": (I, [P, P, P, I, I, F, P, P, P, P]),
    "slic_margin_euclid_bwd": (I, [P, P, P, P, I, I, P, P, P, P, P]),
    "slic_ntxent_euclid_fwd": (I, [P, I, I, F, P, P, P, P, P]),
    "slic_ntxent_euclid_bwd": (I, [P, P, P, I, I, P, P, P]),
    "slic_triplet_select": (I, [P, P, I, P, P, I, F, I, P, P, P]),
    "slic_triplet_select_k": (I, [P, P, I, P, P, I, F, I, P, P, P, P]),
    "slic_triplet_select_cross": (I, [P, P, I, P, P, P, I, F, I, P, P, P]),
    "slic_pdist": (I, [P, I, I, F, I, P, P]),
    "slic_pdist2": (I, [P, I, P, I, I, F, I, P, P]),
    "slic_infonce_rows_fwd": (I, [P, P, I, I, I, F, P, P, P, P]),
    "slic_infonce_rows_bwd": (I, [P, P, P, I, I, I, F, P, P, P, P]),
    # retrieval
    "slic_normalize_rows": (I, [P, L, I, I, P, P]),
    "slic_cosine_topk_workspace_bytes": (c_size_t, [I, I, I]),
    "slic_cosine_topk_plan": (I, [I, I, I, I, P]),
    "slic_cosine_topk": (I, [P, I, P, I, I, I, I, P, P, P, P]),
    "slic_topk_merge_lists": (I, [P, P, I, I, I, P, P, P]),
    "slic_pairwise_euclidean": (I, [P, I, P, I, I, P, P]),
    # memory-bank NCE
    "slic_nce_scores_fwd": (I, [P, P, P, I, I, I, F, P, P, P]),
    "slic_nce_scores_bwd": (I, [P, P, P, I, I, I, F, P, P]),
    "slic_nce_bank_update": (I, [P, P, P, I, I, F, P]),
    "slic_nce_fused_fwd": (I, [P, P, P, P, P, I, I, I, F, P, P, P, P]),
    "slic_nce_fused_update": (I, [P, P, P, P, P, I, I, F, P, P, I, P, P, P, P]),
    "slic_nce_fused_bwd": (I, [P, P, P, I, I, I, F, P, P, P]),
    "slic_softmax_ce0_fwd": (I, [P, I, I, P, P, P, P]),
    "slic_softmax_ce0_bwd": (I, [P, P, I, I, P, P, P]),
}


class SlicConvArgs(ctypes.Structure):
    """mirror of `struct SlicConvArgs` in include/slic_hip.h"""
    _fields_ = [
        ("src", P), ("wgt", P), ("dst", P), ("tab", P), ("tap_tab", P), ("bias", P), ("scale", P), ("shift", P),
        ("addend", P), ("stat_partial", P),
        ("M", L),
        ("src_bytes", ctypes.c_uint32), ("wgt_bytes", ctypes.c_uint32),
        ("N", I), ("nchunks", I), ("Cs", I), ("Ts", I), ("Hs", I), ("Ws", I),
        ("Ga", I), ("Gb", I), ("Gc", I), ("sa", I), ("sb", I), ("sc", I),
        ("ldw", I), ("ldo", I),
        ("dst_strided", I), ("Da", I), ("Db", I), ("Dc", I), ("da", I), ("db", I), ("dc", I),
        ("ea", I), ("eb", I), ("ec", I),
        ("relu", I),
        ("mask_src", P), ("bwd_z", P), ("bwd_mean", P), ("bwd_invstd", P), ("bwd_partial", P),
        ("row_tab", P),
        ("k_run_len", I), ("k_run_px", I),
    ]


class SlicError(RuntimeError):
    pass


_lib = None


def load():
    """Load libslic_hip.so (built by `make -C video_similarity_search_amd/csrc` or
    __graft_entry__.build()).  Raises if it is missing: there is no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlicError(
                f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C video_similarity_search_amd/csrc`; this package has no CPU/PyTorch fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().slic_last_error().decode("utf-8", "replace")
        raise SlicError(f"{what} failed (rc={rc}): {msg}")


def call(name, *args):
    """Call an int-returning entry point and raise SlicError on a non-zero return."""
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """torch's current HIP stream as the void* the C ABI takes.  Through torch's raw-handle entry points: the public
    torch.cuda.current_stream() builds a Stream object and resolves the device index in Python — 9 us a call, 300-400 calls per
    training step (scripts/r5/host_floor.py: 1.3 ms of a 7 ms host step in the forward alone)"""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(*tensors):
    """every tensor must live on a HIP device, be fp32/int32/float64 as the ABI expects and contiguous
    in its last dimension; the current device must be gfx950"""
    if not torch.cuda.is_available():
        raise SlicError("no HIP device: video_similarity_search_amd needs an MI355X (gfx950); no CPU fallback")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SlicError("expected a device tensor, got a CPU tensor (no CPU fallback)")


_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    """grow-only scratch buffer per (device, tag, CURRENT STREAM) — the C ABI never allocates.  Keyed by stream so that
    growing a buffer never frees memory another stream's queued kernel may still read: within one stream the caching
    allocator's reuse is ordered behind the kernels already queued there; across streams it is not."""
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


MAX_BUFFER_BYTES = 0xFFFFFF00       # buffer resources carry a 32-bit byte range (SlicConvArgs.src_bytes / wgt_bytes)


def u32_bytes(t, what):
    """byte size of a tensor for a uint32 range field; raises instead of wrapping modulo 2^32 (a wrapped range would make
    the range-checked buffer loads return zeros past it, silently)"""
    n = t.numel() * t.element_size()
    if n >= MAX_BUFFER_BYTES:
        raise SlicError(f"{what}: {n} bytes does not fit the 32-bit buffer range of one launch — split the batch")
    return n
