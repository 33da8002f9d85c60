"""ctypes binding of libmmhip.so (the C ABI declared in include/mmhip.h).

There is NO fallback: if the library is missing, or a tensor is not on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MM_LIB") or os.path.join(_HERE, "lib", "libmmhip.so")   # MM_LIB: kernel A/B experiments

MM_F32, MM_BF16 = 0, 1
LAYOUT_BLC, LAYOUT_BCL = 0, 1
ATTN_SOFTMAX, ATTN_STABLEMAX = 0, 1

_lib: Optional[C.CDLL] = None

vp, i64, f32, i32, sz = C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("out", vp),
        ("M", i64), ("N", i64), ("K", i64),
        ("lda", i64), ("ldb", i64), ("ldo", i64),
        ("a_kmajor", i32), ("b_kmajor", i32),
        ("dtype", i32), ("out_dtype", i32),
        ("bias", vp), ("pre", vp), ("mul", vp),
        ("act", i32),
        ("gamma", vp), ("rowscale", vp), ("rows_per_scale", i64),
        ("res", vp),
        ("split_k", i32), ("workspace", vp), ("workspace_bytes", sz),
    ]


_SIGNATURES = {
    "mm_version": (i32, []),
    "mm_last_error": (C.c_char_p, []),
    "mm_dwconv7_ln_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, f32, i32, i32, vp]),
    "mm_dwconv7_ln_bwd_workspace": (sz, [i64, i64, i64]),
    "mm_dwconv7_ln_bwd": (i32, [vp] * 13 + [vp, sz, i64, i64, i64, i32, vp]),
    "mm_stem_ln_fwd": (i32, [vp] * 8 + [i64, i64, i64, f32, i32, vp]),
    "mm_stem_ln_bwd_workspace": (sz, [i64, i64, i64]),
    "mm_stem_ln_bwd": (i32, [vp] * 12 + [vp, sz, i64, i64, i64, i32, vp]),
    "mm_layernorm_fwd": (i32, [vp] * 6 + [i64, i64, i64, i64, f32, i32, vp]),
    "mm_layernorm_bwd_workspace": (sz, [i64, i64]),
    "mm_layernorm_bwd": (i32, [vp] * 9 + [vp, sz, i64, i64, i64, i64, i32, vp]),
    "mm_layernorm_bwd_sums": (i32, [vp] * 11 + [vp, sz, i64, i64, i32, vp]),
    "mm_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "mm_colsum_workspace": (sz, [i64, i64]),
    "mm_colsum": (i32, [vp, vp, vp, sz, i64, i64, i64, i32, vp]),
    "mm_colsum_noalloc": (i32, [vp, vp, i64, i64, i64, i32, vp]),
    "mm_prefix_attn_fwd": (i32, [vp] * 7 + [i64] * 7 + [i32, i32, vp]),
    "mm_prefix_attn_bwd_workspace": (sz, [i64, i64, i64]),
    "mm_prefix_attn_bwd": (i32, [vp] * 11 + [vp, sz] + [i64] * 8 + [i32, i32, vp]),
    "mm_rope_inplace": (i32, [vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, vp]),
    "mm_decode_attn": (i32, [vp] * 8 + [i64] * 7 + [i32, i32, vp]),
    "mm_decode_attn_dev": (i32, [vp] * 8 + [i64] * 4 + [vp] + [i64] * 2 + [i32, i32, vp]),
    "mm_kv_cache_fill": (i32, [vp] * 6 + [i64] * 7 + [i32, vp]),
    "mm_layerscale_bwd_workspace": (sz, [i64, i64]),
    "mm_layerscale_bwd": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, vp, sz, i64, i64, i32, vp]),
    "mm_argmax_rows": (i32, [vp, vp, i64, i64, i64, i32, vp]),
    "mm_relu2_fwd": (i32, [vp, vp, i64, i32, vp]),
    "mm_relu2_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "mm_droppath_scale": (i32, [vp, vp, i64, f32, vp]),
    "mm_relu2_bwd_colsum": (i32, [vp, vp, vp, vp, vp, sz, i64, i64, i32, vp]),
    "mm_cast_transpose": (i32, [vp, vp, i64, i64, i64, i32, vp]),
    "mm_cast_transpose_multi": (i32, [vp, vp, i64, vp]),
    "mm_orthograd_adamw_workspace": (sz, [i64, i64]),
    "mm_orthograd_adamw": (i32, [vp, vp, vp, vp, i64, i64, i64, f32, f32, f32, f32, f32, i64, f32, i32, i32, vp, sz, vp]),
    "mm_orthograd_adamw_dev": (i32, [vp, vp, vp, vp, i64, i64, i64, vp, f32, f32, f32, f32, f32, i32, i32, vp, sz, vp]),
}


def exported_symbols():
    return list(_SIGNATURES.keys())


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libmmhip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU or PyTorch fallback for the hot path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_TRACE = bool(os.environ.get("MM_TRACE_OPS"))


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mm_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
    if _TRACE:                    # debugging aid: name every launch and wait for it, so a hung kernel is the last line
        import sys
        print(f"[mm] {what} ...", end="", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(" done", file=sys.stderr, flush=True)


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return MM_F32
    if t.dtype == torch.bfloat16:
        return MM_BF16
    raise TypeError(f"libmmhip: unsupported dtype {t.dtype} (float32 or bfloat16)")


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmmhip: tensor is not on the GPU; the hot path has no CPU fallback")
    return t.data_ptr()


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream
