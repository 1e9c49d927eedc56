"""ctypes binding of libcvnets_hip.so (C ABI declared in include/cvnets_hip.h).

There is NO CPU fallback: if the library is missing every op raises.  The .so is built in-tree by
``ml-cvnets_amd/build.py`` (``__graft_entry__.build()``), never pip-installed.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVNETS_HIP_LIB: developer override used for in-process/same-box A/B runs of two builds of the SAME ABI (box-to-box variance on the
# pool is larger than most kernel changes); the shipped path is always the in-tree library.
LIB_PATH = os.environ.get("CVNETS_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libcvnets_hip.so")

P = c_void_p
I = c_int
F = c_float
L = c_longlong
U = c_uint
D = c_double

# name -> argtypes ; mirrors include/cvnets_hip.h one-to-one (checked by tests/test_abi.py)
SIGNATURES = {
    "cvh_nchw_to_nhwc": [I, P, P, I, I, I, I, I, P],
    "cvh_nhwc_to_nchw": [I, P, P, I, I, I, I, I, P],
    "cvh_weight_pack": [I, P, P, I, I, I, I, P],
    "cvh_weight_pack_multi": [I, P, I, L, P, P],
    "cvh_cast_from_f32": [I, P, P, L, P],
    "cvh_cast_to_f32": [I, P, P, L, P],
    "cvh_conv_gemm": [I, P, P, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P, I, P, P, I, P, F, P, U, P, P],
    "cvh_conv_gemm_takes_gelu_d": [L, I, I],
    "cvh_conv_gemm_grid_rows": [I, I],
    "cvh_stream_counters": [I, P],  # out = long long[4]
    "cvh_family_counters": [I, I, P],  # out = long long[2]
    "cvh_stem_rows": [I, I, I, I],
    "cvh_ir_exp_bwd_rows": [L, I, I],
    "cvh_dwx_rows": [I, I, I, I, I],
    "cvh_dwx_fwd_rows": [I, I, I, I, I, I],
    "cvh_axpb": [P, F, P, P, I, P],
    "cvh_gram_bn_stats": [P, P, P, P, I, I, I, P],
    "cvh_dwx_fwd": [I, P, P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, P],
    "cvh_dwx_bwd": [I, P, P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "cvh_ir_pb_rows": [I, I, I],
    "cvh_ir_pb": [I, P, P, P, P, P, I, P, P, P, P, I, I, I, P],
    "cvh_ir_red_fwd_rows": [L, I, I],
    "cvh_ir_red_fwd": [I, P, P, P, I, P, P, P, L, I, I, P],
    "cvh_ir_exp_bwd": [I, P, P, P, P, P, P, P, L, I, I, P],
    "cvh_ir_exp_bwd_s": [I, P, P, P, P, P, P, P, P, L, I, I, P],
    "cvh_stem_conv_fwd": [I, P, I, P, P, P, I, I, I, I, P],
    "cvh_stem_conv_dw": [I, P, P, P, I, I, I, I, P],
    "cvh_gemm_dw": [I, P, P, P, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    "cvh_gemm_dw_bias": [I, P, P, P, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    "cvh_gemm_dw_folds_bias": [I, I, I, I],
    "cvh_gemm_dw_scratch_elems": [I, I, I],
    "cvh_gemm_dw_scratch_elems_conv": [I, I, I, I, I, I, I, I, I, I, I, I, I, I, I],
    "cvh_dwconv_fwd": [I, P, P, P, I, I, I, I, I, I, I, I, I, I, P, P],
    "cvh_dwconv_rows": [I, I, I, I, I, I, I, I],
    "cvh_dwconv_bwd_x": [I, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cvh_dwconv_bwd_w": [I, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cvh_dwconv_bwd_w_rows": [I, I, I, I, I, I, I, I],
    "cvh_colreduce_rows": [L, I],
    "cvh_bn_stats": [I, P, L, I, P, P],
    "cvh_bn_finalize": [P, I, I, D, P, P, P, P, F, F, P, P, P, P, P],
    "cvh_bn_eval_coeff": [P, P, P, P, F, I, P, P, P, P, P],
    "cvh_bn_apply": [I, P, P, P, I, P, P, L, I, P],
    "cvh_bn_apply_gram_rows": [L, I],
    "cvh_bn_apply_gram": [I, P, P, P, I, P, P, L, I, P, I, P, P],
    "cvh_bn_bwd_reduce": [I, P, P, P, P, P, P, I, L, I, P, P],
    "cvh_bn_bwd_finalize": [P, I, I, D, P, P, P, I, I, P, P, P, P, P, P],
    "cvh_bn_bwd_finalize_out": [P, I, I, D, P, P, P, P, I, I, P, P, P, P, P, P],
    "cvh_bn_bwd_apply": [I, P, P, P, P, I, P, P, P, P, L, I, P],
    "cvh_colsum": [I, P, L, I, P, P, F, I, P],
    "cvh_reduce_multi": [P, I, P],
    "cvh_sum_partials": [P, I, I, I, P, F, I, P],
    "cvh_pool_fwd": [I, P, P, I, I, I, P],
    "cvh_pool_bwd": [I, P, P, I, I, I, P],
    "cvh_adaptive_pool_fwd": [I, P, P, I, I, I, I, I, P],
    "cvh_adaptive_pool_bwd": [I, P, P, I, I, I, I, I, P],
    "cvh_dropout": [I, P, P, L, F, P, U, P],
    "cvh_seed_advance": [P, P],
    "cvh_drop_path": [I, P, P, P, L, I, I, I, I, I, F, P, U, P],
    "cvh_mix_batch": [I, P, P, I, I, I, I, I, F, I, I, I, I, P],
    "cvh_add": [I, P, P, P, L, P],
    "cvh_dropout2d": [I, P, P, I, I, I, F, P, U, P],
    "cvh_cat_channels": [I, P, P, I, P, L, I, P],
    "cvh_rows_gather_idx": [I, P, P, P, I, I, I, P],
    "cvh_l2norm_fwd": [I, P, P, P, I, I, F, P],
    "cvh_l2norm_bwd": [I, P, P, P, P, I, I, F, P],
    "cvh_scaled_ce_fwd": [I, P, P, P, P, I, I, I, P],
    "cvh_scaled_ce_bwd": [I, P, P, P, P, P, P, I, I, I, P],
    "cvh_ln_seq_fwd": [I, P, P, P, P, P, I, I, I, I, I, I, I, I, F, P],
    "cvh_ln_seq_bwd": [I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, P],
    "cvh_adamw_multi": [P, I, L, P, P, P, F, F, F, P, P, F, P],
    "cvh_lerp_multi": [P, I, L, F, P],
    "cvh_ce_fwd": [I, P, P, F, L, P, P, I, I, P],
    "cvh_ce_bwd": [I, P, P, P, P, F, L, P, I, I, P],
    "cvh_ce_mean": [P, P, L, P, I, P],
    "cvh_ce_soft_fwd": [I, P, P, F, P, P, P, I, I, P],
    "cvh_ce_soft_bwd": [I, P, P, P, P, P, F, P, I, I, P],
    "cvh_gn_chunks": [I, I, I],
    "cvh_gn_fwd": [I, P, P, P, P, P, P, I, I, I, F, P],
    "cvh_gn_bwd": [I, P, P, P, P, P, P, P, I, I, I, P],
    "cvh_linattn_fwd": [I, P, P, P, I, I, I, I, I, I, P],
    "cvh_linattn_bwd": [I, P, P, P, P, I, I, I, I, I, I, P],
    "cvh_resize_bilinear_fwd": [I, P, P, I, I, I, I, I, I, I, P],
    "cvh_resize_bilinear_bwd": [I, P, P, I, I, I, I, I, I, I, P],
    "cvh_layernorm_fwd": [I, P, P, P, P, P, P, L, I, F, P],
    "cvh_layernorm_bwd": [I, P, P, P, P, P, P, P, L, I, P],
    "cvh_layernorm_bwd_res": [I, P, P, P, P, P, P, P, L, I, P, P],
    "cvh_layernorm_cf_fwd": [I, P, P, P, P, P, P, L, I, F, P],
    "cvh_layernorm_cf_bwd": [I, P, P, P, P, P, P, P, L, I, F, P],
    "cvh_ln_bwd_rows": [L],
    "cvh_ln_bwd_drop_ok": [I],
    "cvh_layernorm_bwd_res_drop": [I, P, P, P, P, P, P, P, L, I, P, P, F, P, U, P],
    "cvh_set_tuning": [I, I],
    "cvh_conv_dx_patch": [I, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cvh_vit_embed_fwd": [I, P, P, P, P, I, I, I, P],
    "cvh_vit_embed_bwd": [I, P, P, I, I, I, I, P],
    "cvh_batch_sum": [I, P, P, I, L, I, P],
    "cvh_rows_copy": [I, P, P, L, I, L, L, P],
    "cvh_embed_lookup_fwd": [I, P, P, P, P, L, I, I, P],
    "cvh_embed_lookup_bwd": [I, P, P, P, L, I, L, P],
    "cvh_pw_gemm_bn": [I, P, P, I, P, P, L, I, P, I, P, P, I, P, P],
    "cvh_pw_gemm_dw_bn": [I, P, P, P, P, P, L, I, I, I, P, L, I, P],
    "cvh_bn_dx_weights": [I, P, P, P, P, I, I, P],
    "cvh_bn_dw_combine": [P, P, P, P, P, P, I, I, I, P],
    "cvh_dwconv_bn_rows": [I, I, I, I, I],
    "cvh_dwconv_bn_fwd": [I, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "cvh_dwconv_bn_bwd": [I, P, P, P, P, I, P, P, P, P, I, I, I, I, I, I, I, P],
    "cvh_attn_fwd": [I, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P],
    "cvh_attn_bwd": [I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P],
    "cvh_attn_fwd_drop": [I, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, F, P, U, P],
    "cvh_attn_bwd_drop": [I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, F, P, U, P],
    "cvh_attn_fwd_mask": [I, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, F, I, F, P, U, P],
    "cvh_attn_bwd_mask": [I, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, F, I, F, P, U, P],
    "cvh_comm_available": [],
    "cvh_comm_unique_id": [P],
    "cvh_comm_init": [P, I, I, P],
    "cvh_comm_destroy": [P],
    "cvh_comm_world": [P],
    "cvh_comm_rank": [P],
    "cvh_comm_allreduce": [P, P, L, I, I, P],
    "cvh_comm_broadcast": [P, P, L, I, I, P],
    "cvh_comm_allgather": [P, P, P, L, I, P],
    "cvh_comm_reducescatter": [P, P, P, L, I, P],
    "cvh_comm_counters": [I, P],
}

class OperandXf(ctypes.Structure):
    """cvh_operand_xf (include/cvnets_hip.h): transform applied to a kernel operand while it is loaded."""
    _fields_ = [("mode", c_int), ("src2", c_void_p), ("c0", c_void_p), ("c1", c_void_p), ("c2", c_void_p), ("act", c_int)]


class ReduceDesc(ctypes.Structure):
    """cvh_reduce_desc (include/cvnets_hip.h)."""
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("row_stride", c_longlong), ("n_out", c_longlong), ("rows", c_int), ("kind", c_int),
                ("N", c_int), ("Ktot", c_int), ("Cin", c_int), ("Cin_real", c_int), ("khw", c_int), ("scale", c_float),
                ("accumulate", c_int), ("pad_", c_int)]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python ml-cvnets_amd/build.py` (hipcc, gfx950). "
            "cvnets_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            if os.environ.get("CVNETS_HIP_LIB"):  # developer A/B against an OLDER build: entry points added since are simply absent
                continue
            raise
        fn.argtypes = argtypes
        fn.restype = c_longlong if "_elems" in name else c_int
    for kv in filter(None, os.environ.get("CVH_TUNE", "").split(",")):  # developer A/B knob overrides, e.g. CVH_TUNE="6=512,2=1024"
        k, v = kv.split("=")
        lib.cvh_set_tuning(int(k), int(v))
    _lib = lib
    return lib


_TRACE_CALLS = os.environ.get("CVH_TRACE_CALLS", "0") == "1"


def call(name: str, *args) -> int:
    """Invoke a status-returning entry point; raise RuntimeError on any non-zero status."""
    if _TRACE_CALLS:  # debugging aid (CVH_TRACE_CALLS=1): name every launch before it runs and wait for it — an asynchronous GPU fault is
        import sys     # then reported right after the line of the entry point that caused it
        import torch
        sys.stderr.write(f"[cvh] {name} " + " ".join(hex(a) if isinstance(a, int) and a > 0xFFFFFFFF else (repr(a) if isinstance(a, (int, float, type(None))) else type(a).__name__) for a in args) + "\n")
        sys.stderr.flush()
        rc = getattr(load(), name)(*args)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}" + (" (HIP out of memory)" if rc == 2 else ""))
    return rc


def query(name: str, *args) -> int:
    """Invoke a size-query entry point (returns a count, negative = rejected arguments)."""
    rc = getattr(load(), name)(*args)
    if rc < 0:
        raise RuntimeError(f"{name}{args} rejected its arguments (status {rc})")
    return rc
