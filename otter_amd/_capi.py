"""ctypes binding of libotter_hip.so (include/otter_hip.h).  No fallback: if the library is missing or a call fails,
this raises -- the product path never silently degrades to PyTorch ops or to the CPU oracle."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTTER_LIB_PATH") or os.path.join(_HERE, "lib", "libotter_hip.so")  # override: diagnostics builds only



def _declared_abi_version() -> int:
    """OTTER_ABI_VERSION as include/otter_hip.h declares it -- the one place the number lives, so the loader's check cannot lag behind
    a signature change (a library built from an older header exported every symbol and was called with shifted arguments)."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "otter_hip.h")
    if not os.path.exists(hdr):
        # the package was copied / installed without its sibling include/ directory: fall back to the mirrored constant
        # (tests/test_capi_symbols.py::test_abi_version_mirror_matches_header keeps the two equal)
        return ABI_VERSION_MIRROR
    with open(hdr) as f:
        m = re.search(r"^#define\s+OTTER_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise RuntimeError("include/otter_hip.h does not define OTTER_ABI_VERSION")
    return int(m.group(1))


ABI_VERSION_MIRROR = 3
ABI_VERSION = _declared_abi_version()
F32, BF16 = 0, 1
GRID_DEFAULT, GRID_PERSISTENT, GRID_PER_TILE = 0, 1, 2    # otter_grid_mode
EPI_STORE, EPI_GELU, EPI_SCALE_RES, EPI_GATE_BWD = 0, 1, 2, 3
MASK_NONE, MASK_EQ, MASK_GE = 0, 1, 2


class RowMap(C.Structure):
    _fields_ = [("grp_rows", C.c_int64), ("grp_stride", C.c_int64), ("row_off", C.c_int64)]


class EpilogueArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("accumulate", C.c_int),
        ("gate", C.c_void_p),
        ("R", C.c_void_p),
        ("ldr", C.c_int64),
        ("r_dtype", C.c_int),
        ("C2", C.c_void_p),
        ("ldc2", C.c_int64),
        ("aux", C.c_void_p),
        ("ldaux", C.c_int64),
        ("aux_dtype", C.c_int),
        ("aux_is_gelu_input", C.c_int),
        ("partial", C.c_void_p),
        ("grid_mode", C.c_int),
    ]


class FlashView(C.Structure):
    _fields_ = [("batch_stride", C.c_int64), ("seq_stride", C.c_int64), ("head_stride", C.c_int64)]


class FlashDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("qv", FlashView), ("kv", FlashView), ("vv", FlashView),
        ("o", C.c_void_p), ("ov", FlashView),
        ("lse", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("key_valid", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int), ("head_dim", C.c_int), ("causal", C.c_int),
        ("scale", C.c_float),
        ("dout", C.c_void_p), ("dov", FlashView),
        ("delta", C.c_void_p),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("dqv", FlashView), ("dkv", FlashView), ("dvv", FlashView),
    ]


_i64, _int, _f32, _vp = C.c_int64, C.c_int, C.c_float, C.c_void_p

# name -> (restype, argtypes): exactly the declarations of include/otter_hip.h (tests/test_capi_symbols.py checks this)
SIGNATURES = {
    "otter_abi_version": (_int, []),
    "otter_last_error": (C.c_char_p, []),
    "otter_device_check": (_int, []),
    "otter_debug_occupy_cus": (_int, [_int, _vp, C.c_ulonglong, _vp]),
    "otter_probe_mfma": (_int, [_vp, _vp, _int, _int, _vp]),
    "otter_layernorm_fwd": (_int, [_vp, _int, _vp, _vp, _int, _vp, _int, RowMap, _vp, _vp, _vp, _i64, _i64, _f32, _vp]),
    "otter_add_layernorm_fwd": (_int, [_vp, _int, _vp, _int, _vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _i64, _i64, _f32, _vp]),
    "otter_layernorm_bwd_workspace_bytes": (_i64, [_i64, _i64]),
    "otter_layernorm_bwd": (_int, [_vp, _int, RowMap, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _vp,
                                   _i64, _i64, _vp]),
    "otter_colsum": (_int, [_vp, _int, RowMap, _vp, _int, _vp, _i64, _i64, _vp]),
    "otter_rmsnorm_fwd": (_int, [_vp, _int, _vp, _int, _vp, _vp, _i64, _i64, _f32, _vp]),
    "otter_rmsnorm_bwd": (_int, [_vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _int, _vp, _i64, _i64, _vp]),
    "otter_add_rmsnorm_fwd": (_int, [_vp, _int, _vp, _int, _vp, _vp, _int, _vp, _int, _vp, _i64, _i64, _f32, _vp]),
    "otter_rmsnorm_bwd_ex": (_int, [_vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _int, _vp, _vp, _int, _vp, _i64, _i64, _vp]),
    "otter_gemm_num_partials": (_i64, [_i64, _i64, _int]),
    "otter_gemm_nt": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _int, C.POINTER(EpilogueArgs), _vp]),
    "otter_gemm": (_int, [_vp, _i64, _int, _vp, _i64, _int, _vp, _i64, _i64, _i64, _i64, _int, _int, C.POINTER(EpilogueArgs), _vp]),
    "otter_gemm_kmajor_supported": (_int, [_i64, _i64, _i64, _i64, _i64, _int, _int, _int]),
    "otter_gemm_set_variant": (_int, [_int]),
    "otter_gemm_variant_available": (_int, [_int]),
    "otter_gemm_set_cu_budget": (_int, [_int]),
    "otter_gemm_set_persistent": (_int, [_int]),
    "otter_gemm_set_debug": (_int, [_int]),
    "otter_gemm_read_timeline": (_int, [_vp, _int]),
    "otter_reduce_partials": (_int, [_vp, _i64, _vp, _vp, _int, _vp]),
    "otter_transpose": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _i64, _int, _i64, _i64, _vp]),
    "otter_cast": (_int, [_vp, _int, _vp, _int, _i64, _vp]),
    "otter_text_time": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "otter_attn_fwd": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _f32, _int,
                              _vp]),
    "otter_attn_set_variant": (_int, [_int]),
    "otter_attn_bwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "otter_attn_bwd": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64,
                              _i64, _i64, _i64, _int, _f32, _int, _vp]),
    "otter_rope": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp]),
    "otter_rope_strided": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _i64, _i64, _vp]),
    "otter_quick_gelu": (_int, [_vp, _vp, _i64, _int, _vp]),
    "otter_gelu_fwd": (_int, [_vp, _vp, _i64, _int, _vp]),
    "otter_gelu_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "otter_swiglu_fwd": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "otter_swiglu_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "otter_add_frame_embs": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _i64, _vp]),
    "otter_add_rows": (_int, [_vp, _vp, RowMap, _i64, _i64, _int, _vp]),
    "otter_flash_attn_fwd": (_int, [C.POINTER(FlashDesc), _vp]),
    "otter_flash_attn_bwd": (_int, [C.POINTER(FlashDesc), _vp]),
    "otter_flash_set_variant": (_int, [_int]),
    "otter_decode_attn": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                 _i64, _f32, _vp]),
    "otter_qk_norm_rope_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i64, _vp]),
    "otter_qk_norm_rope_bwd_blocks": (_i64, [_i64, _i64]),
    "otter_qk_norm_rope_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "otter_sqrelu_fwd": (_int, [_vp, _vp, _i64, _vp]),
    "otter_sqrelu_bwd": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "otter_scatter_rows": (_int, [_vp, _int, _vp, _int, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "otter_cross_entropy_fwd": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp]),
    "otter_cross_entropy_bwd": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "otter_adamw_chunk": (_int, []),
    "otter_grad_sumsq": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "otter_clip_coef": (_int, [_vp, _i64, _f32, _vp, _vp]),
    "otter_adamw_step": (_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "otter_prof_arm_gemm": (_int, [_i64, _i64, _i64, _int]),
    "otter_prof_disarm": (_int, []),
    "otter_prof_collect": (_int, [C.POINTER(_int), C.POINTER(C.c_double)]),
    "otter_prof_collect_split": (_int, [C.POINTER(_int), C.POINTER(C.c_double), C.POINTER(_int), C.POINTER(C.c_double)]),
}

_lib = None


class OtterHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the C-ABI library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and os.environ.get("OTTER_NO_AUTOBUILD") != "1":
        # build-on-demand (hipcc, gfx950) -- still no alternative compute path: if this fails we raise below
        try:
            from . import build as _build

            _build.build(verbose=False)
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise OtterHipError(
            f"{LIB_PATH} is missing: build it with `python -m otter_amd.build` (hipcc --offload-arch=gfx950). "
            "otter_amd has no PyTorch/CPU fallback for the fusion hot path.")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if l.otter_abi_version() != ABI_VERSION:
        raise OtterHipError("%s reports ABI version %d, include/otter_hip.h declares %d: rebuild it (python -m otter_amd.build --force)"
                            % (LIB_PATH, l.otter_abi_version(), ABI_VERSION))
    _lib = l
    return l


_gemm_debug_word = 0


def gemm_set_debug(flags: int) -> int:
    """otter_gemm_set_debug through a host-side mirror of the word (the C ABI has a setter only): returns the PREVIOUS value so that a caller
    that borrows a bit -- bench.in_step_gemm_clock's tile stamps -- can put back what the session had set (tile-order / K-order overrides of
    the A/B tools) instead of clearing it."""
    global _gemm_debug_word
    prev = _gemm_debug_word
    check(lib().otter_gemm_set_debug(int(flags)), "gemm_set_debug")
    _gemm_debug_word = int(flags)
    return prev


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().otter_last_error().decode(errors="replace")
        raise OtterHipError(f"{what or 'otter_hip'} failed ({rc}): {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise OtterHipError(f"unsupported dtype {t.dtype} (f32 / bf16 only)")


def dt_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise OtterHipError(f"unsupported dtype {dtype} (f32 / bf16 only)")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream() -> int:
    """The raw hipStream_t of torch's current stream: every kernel is enqueued there, so torch ops and ours interleave
    in program order without extra synchronisation."""
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise OtterHipError("otter_amd kernels run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)
