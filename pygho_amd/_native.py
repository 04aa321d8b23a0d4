"""
ctypes binding of the C ABI declared in ``include/pygho_hip.h``.

The shared object ``pygho_amd/_lib/libpygho_hip.so`` is built in-tree by
``__graft_entry__.build()`` / ``python -m pygho_amd.build`` (hipcc, gfx950).  There is
NO fallback: if the library is missing every compute entry point raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_uint64, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

# PYGHO_AMD_LIB: an A/B build of the same sources (python -m pygho_amd.build --variant NAME -D...), for timing two kernel
# versions on one box; the product default is the in-tree build
LIB_PATH = os.environ.get("PYGHO_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libpygho_hip.so")

# enum pygho_dtype / pygho_aggr (include/pygho_hip.h)
F32, BF16, F16, F64, I64, I32 = 0, 1, 2, 3, 4, 5
DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, torch.float64: F64,
              torch.int64: I64, torch.int32: I32}
AGGR_CODE = {"sum": 0, "mean": 1, "max": 2, "min": 3, "amax": 2, "amin": 3}

P, I, L, Z, D = c_void_p, c_int, c_int64, c_size_t, c_double

# name -> (restype, argtypes); mirrors the header one to one (checked by tests/test_cabi.py)
PROTOTYPES = {
    "pygho_abi_version": (I, []),
    "pygho_last_error": (c_char_p, []),
    "pygho_seg_gather_mul_reduce": (I, [P, P, P, P, P, P, P, L, L, L, L, L, L, I, I, P]),
    "pygho_seg_gather_mul_reduce_add": (I, [P, P, P, P, P, P, P, P, L, L, L, L, L, L, I, I, P]),
    "pygho_seg_gather_mul_reduce_window": (I, [P, P, P, P, P, P, P, P, L, L, L, L, I, I, P]),
    "pygho_seg_tile_chunk": (I, []),
    "pygho_seg_tile_plan": (I, [P, P, P, P, L, L, P]),
    "pygho_seg_gather_mul_reduce_tiled": (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, L, L, I, I, P]),
    "pygho_table_grad_supported": (I, [L, L]),
    "pygho_table_grad_blocks": (I, [L, L, L]),
    "pygho_table_grad": (I, [P, P, P, L, L, L, I, P, P]),
    "pygho_pair_bwd_types": (I, []),
    "pygho_pair_bwd_blocks": (I, [L, L, I]),
    "pygho_pair_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, L, I, P]),
    "pygho_seg_gather_mul_reduce_act": (I, [P, P, P, P, P, P, P, P, P, P, I, I, L, L, L, L, I, I, P]),
    "pygho_seg_triple_product": (I, [P, P, P, P, P, P, P, P, L, L, L, L, L, I, I, P]),
    "pygho_seg_sum_f32out": (I, [P, P, P, P, L, L, L, I, P]),
    "pygho_seg_prod": (I, [P, P, P, P, L, L, I, P]),
    "pygho_seg_prod_bwd": (I, [P, P, P, P, P, P, L, L, I, P]),
    "pygho_seg_extremum_ties": (I, [P, P, P, P, P, P, P, L, L, I, P]),
    "pygho_seg_extremum_bwd": (I, [P, P, P, P, P, P, P, P, P, L, L, I, P]),
    "pygho_seg_scatter_limits": (I, [P, P, P]),
    "pygho_seg_scatter_count": (I, [P, P, P, P, P, P, P, L, P]),
    "pygho_seg_scatter_write": (I, [P, P, P, P, P, P, P, P, L, L, L, P]),
    "pygho_seg_scatter_mul_reduce": (I, [P, P, P, P, P, P, P, P, L, L, L, L, L, L, L, L, I, P]),
    "pygho_seg_scatter_count_aligned": (I, [P, P, P, P, P, P, P, P, P, L, P]),
    "pygho_seg_scatter_write_aligned": (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, L, P]),
    "pygho_seg_dual_limits": (I, [P, P, P]),
    "pygho_seg_dual_tg_blocks": (I, [L, L, L, I]),
    "pygho_seg_dual_tg": (I, [P, P, P, P, P, L, P, P, P, P, P, P, P, L, L, L, L, L, I, P, P]),
    "pygho_seg_dual": (I, [P, P, P, P, P, P, L, P, P, P, P, P, P, P, P, L, L, L, L, L, L, L, L, I, P]),
    "pygho_seg_fused_limits": (I, [P, P, P, P]),
    "pygho_seg_fused_count": (I, [P, P, P, P, P, L, P]),
    "pygho_seg_fused_write": (I, [P, P, P, P, P, P, P, L, L, L, P]),
    "pygho_seg_fused_fwd": (I, [P, P, P, P, P, P, P, P, L, I, P, P, P, P, P, L, L, L, L, I, I, I, P]),
    "pygho_seg_gather_mul_reduce_ties": (I, [P, P, P, P, P, P, P, L, L, L, L, I, I, P]),
    "pygho_seg_extremum_share": (I, [P, P, P, P, P, P, P, P, L, L, L, L, L, I, P]),
    "pygho_seg_extremum_bwd_shared": (I, [P, P, P, P, P, P, P, P, L, L, L, L, L, I, P]),
    "pygho_row_gather": (I, [P, P, P, P, L, L, I, P]),
    "pygho_row_gather_mean": (I, [P, P, P, P, L, L, I, P]),
    "pygho_xcc_ids": (I, [P, L, P]),
    "pygho_narrow_i64_i32": (I, [P, P, L, P, P]),
    "pygho_csr_from_sorted": (I, [P, P, L, L, P, P]),
    "pygho_group_by_key_workspace": (Z, [L, L]),
    "pygho_group_by_key": (I, [P, P, P, L, L, P, Z, P, P]),
    "pygho_gather_i32": (I, [P, P, P, L, P]),
    "pygho_scatter_i32": (I, [P, P, P, L, P]),
    "pygho_hash_pack": (I, [P, P, L, L, L, P, P]),
    "pygho_hash_unpack": (I, [P, P, L, L, P]),
    "pygho_sorted_match": (I, [P, P, L, P, L, P]),
    "pygho_search_bounds": (I, [P, P, P, L, P, L, P]),
    "pygho_sort_pairs_i64_workspace": (Z, [L]),
    "pygho_sort_pairs_i64": (I, [P, P, P, L, I, P, Z, P]),
    "pygho_run_ids_workspace": (Z, [L]),
    "pygho_run_ids": (I, [P, P, P, L, P, Z, P]),
    "pygho_expand_pairs": (I, [P, P, P, P, L, L, P]),
    "pygho_exclusive_scan_i64_workspace": (Z, [L]),
    "pygho_exclusive_scan_i64": (I, [P, P, L, P, Z, P]),
    "pygho_product_hash": (I, [P, P, L, L, L, P, L, L, L, P, P, L, P, P]),
    "pygho_gather_cols_i64": (I, [P, P, L, L, P, I, L, P]),
    "pygho_gather_i32_to_i64": (I, [P, P, P, L, P]),
    "pygho_plan_triples": (I, [P, P, P, P, P, L, P]),
    "pygho_collate_rows": (I, [P, P, L, L, L, P, P, P, L, L, P]),
    "pygho_collate_rows_i32": (I, [P, P, L, L, L, P, P, P, L, L, I, P]),
    "pygho_pad_stack": (I, [P, P, P, P, P, L, I, L, L, L, L, L, P]),
    "pygho_dense_adj": (I, [P, P, P, P, P, P, L, L, L, L, c_uint64, I, P]),
    "pygho_flag_scan_nonneg": (I, [P, P, P, L, P, Z, P]),
    "pygho_compact_positions": (I, [P, P, L, P]),
    "pygho_masked_bmm": (I, [P, P, P, P, P, P, L, L, L, L, L, I, I, I, P]),
    "pygho_mask_extents": (I, [P, P, P, P, L, L, L, L, I, I, P]),
    "pygho_masked_bmm_clipped": (I, [P, P, P, P, P, P, P, L, L, L, L, L, I, I, I, P]),
    "pygho_mask_lists": (I, [P, P, P, L, L, L, I, P]),
    "pygho_masked_bmm_lists": (I, [P, P, P, P, P, P, P, I, L, L, L, L, L, I, I, I, P]),
    "pygho_masked_bmm_outlists": (I, [P, P, P, P, P, P, L, L, L, L, L, L, I, I, I, P]),
    "pygho_masked_fill": (I, [P, P, P, D, L, L, I, P]),
    "pygho_masked_reduce": (I, [P, P, P, P, L, L, L, L, I, I, P]),
    "pygho_masked_reduce_bwd": (I, [P, P, P, P, P, L, L, L, L, I, I, P]),
    "pygho_masked_broadcast": (I, [P, P, P, D, L, L, L, L, I, P]),
    "pygho_masked_pair_combine": (I, [P, P, P, P, P, I, P, L, L, L, L, I, P]),
    "pygho_pair_gather_combine": (I, [P, P, P, P, P, I, P, P, L, L, I, P]),
    "pygho_bn_workspace": (Z, [L, L, I]),
    "pygho_bn_stats": (I, [P, P, P, L, L, P, I, P]),
    "pygho_bn_prepare": (I, [P, P, P, P, P, P, L, L, P, P, D, P, P, D, P, I, P]),
    "pygho_bn_finalize": (I, [P, P, P, P, P, P, L, P, L, L, P, P, D, P, P, D, P]),
    "pygho_bn_act_fwd": (I, [P, P, P, P, L, L, I, I, P]),
    "pygho_bn_act_fwd_add": (I, [P, P, P, P, P, L, L, I, I, P]),
    "pygho_bn_act_bwd": (I, [P, P, P, P, P, P, P, P, P, L, L, I, I, P, I, P, P]),
    "pygho_rowblock_linear_blocks": (I, [L]),
    "pygho_rowblock_linear_slots": (I, [L, L]),
    "pygho_rowblock_linear_bwd_apply": (I, [P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, P, I, P]),
    "pygho_rowblock_linear": (I, [P, P, P, P, P, P, P, L, L, I, P]),
    "pygho_rowblock_linear_autoshift": (I, [P, P, P, P, P, P, P, L, L, I, P]),
    "pygho_bn_bwd_linear": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, P]),
    "pygho_bn_bwd_linear_dw_blocks": (I, [L]),
    "pygho_bn_bwd_linear_dw": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, L, P]),
    "pygho_weight_grad": (I, [P, P, P, P, L, L, L, I, L, P]),
    "pygho_sum_blocks": (I, [P, P, L, L, P]),
    "pygho_sum_blocks_pad": (I, [P, P, L, L, L, P]),
    "pygho_bn_act_bwd_sums": (I, [P, P, P, P, P, P, P, P, L, L, I, P, I, P]),
    "pygho_rowblock_linear_bn_act": (I, [P, P, P, P, P, P, P, L, L, I, I, P]),
    "pygho_rowblock_linear_bwd_sums": (I, [P, P, P, P, P, P, P, P, P, P, L, L, I, P, I, P]),
    "pygho_bn_bwd_linear_dw_recompute": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, L, P]),
    "pygho_bn_bwd_fold_sums": (I, [P, P, P, L, L, P]),
    "pygho_narrow_i64_i32_bounded": (I, [P, P, L, L, P, P]),
    "pygho_block_cuts_workspace": (Z, [L]),
    "pygho_block_cuts": (I, [P, P, P, L, P, Z, P]),
    "pygho_rowblock_linear_autoshift_dyn": (I, [P, P, P, P, P, P, P, L, P, L, I, P]),
    "pygho_rowblock_linear_bwd_sums_dyn": (I, [P, P, P, P, P, P, P, P, P, P, L, P, L, I, P, I, P]),
    "pygho_bn_bwd_linear_dw_dyn": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, I, L, P]),
    "pygho_bn_bwd_linear_dw_recompute_dyn": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, I, L, P]),
    "pygho_weight_grad_strided": (I, [P, P, P, L, P, L, L, P, L, I, L, P]),
    "pygho_weight_grad_dyn": (I, [P, P, P, P, L, L, P, L, I, L, P]),
    "pygho_bn_prepare_dyn": (I, [P, P, P, P, P, P, L, P, L, P, P, D, P, P, D, P, I, P]),
    "pygho_bn_finalize_dyn": (I, [P, P, P, P, P, P, L, P, L, P, L, P, P, D, P, P, D, P]),
    "pygho_bn_act_bwd_dyn": (I, [P, P, P, P, P, P, P, P, P, L, P, L, I, I, P, I, P, P]),
    "pygho_bn_act_bwd_sums_dyn": (I, [P, P, P, P, P, P, P, P, L, P, L, I, P, I, P]),
    "pygho_table_grad_dyn": (I, [P, P, P, L, P, L, L, I, P, P]),
    "pygho_collate_desc_bytes": (Z, []),
    "pygho_collate_batch": (I, [P, L, L, L, P]),
    "pygho_graph_bfs_dist": (I, [P, P, P, P, P, L, L, I, P]),
    "pygho_khop_count": (I, [P, P, P, P, P, L, I, P]),
    "pygho_khop_emit": (I, [P, P, P, L, P, P, P, P, L, I, P]),
    "pygho_pair_count": (I, [P, P, P, L, P, P, P, P, I, P]),
    "pygho_pair_emit": (I, [P, P, P, L, P, P, L, P, P, P, P, I, P]),
}

_lib: Optional[ctypes.CDLL] = None


class BackendUnavailable(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """load (once) the HIP extension; loud failure when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendUnavailable(
                f"pygho_amd: HIP extension not built ({LIB_PATH} missing). Run `python -m pygho_amd.build` "
                "(needs hipcc); there is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.pygho_abi_version() != 1:
            raise BackendUnavailable("pygho_amd: ABI version mismatch, rebuild the extension")
        if os.environ.get("PYGHO_ROCTX", "0") not in ("", "0"):
            handle = _RangedLib(handle)
        _lib = handle
    return _lib


class _RangedLib:
    """PYGHO_ROCTX=1: every C-ABI launch is bracketed by a roctx range named after its entry point (torch.cuda.nvtx maps to
    roctx on ROCm), so `rocprofv3 --marker-trace` groups the kernels by operator family.  Off by default (two extra calls per
    launch)."""

    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name not in PROTOTYPES or name.endswith("_workspace") or PROTOTYPES[name][0] is not I or not PROTOTYPES[name][1]:
            return fn

        def ranged(*args):
            torch.cuda.nvtx.range_push(name)
            try:
                return fn(*args)
            finally:
                torch.cuda.nvtx.range_pop()
        setattr(self, name, ranged)
        return ranged


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().pygho_last_error()
        raise RuntimeError(f"pygho_amd.{what} failed (code {rc}): {msg.decode() if msg else ''}")


def require_device(*tensors) -> torch.device:
    """every compute entry point works on ROCm device memory only."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pygho_amd: the HIP backend needs tensors on a ROCm device (got a CPU tensor); "
                               "there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"pygho_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        # kernels are launched on the CURRENT device with the tensor device's stream: one process drives one GPU
        # (torch.cuda.set_device(local_rank), as bench.py does); another device needs `with torch.cuda.device(dev):`
        raise RuntimeError(f"pygho_amd: tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(...) or wrap the call in `with torch.cuda.device(...)`")
    return dev


def ptr(t: Optional[torch.Tensor]):
    """device address for a `void*` parameter: a plain int (ctypes converts ints and None for c_void_p argtypes itself; wrapping
    every address in a c_void_p object first cost 0.2 us x ~260 addresses per training step)"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device):
    """torch's current stream on `device` as the raw hipStream_t the C ABI takes (the private raw getter when this torch has it:
    0.2 us instead of the 1.9 us it takes to build a torch.cuda.Stream object per launch)"""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def dtype_code(t: torch.Tensor) -> int:
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise TypeError(f"pygho_amd: unsupported dtype {t.dtype}") from None
