"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/otter_hip.h).

PyTorch is used here for device memory (torch.empty), the current HIP stream and nothing else: all arithmetic
happens in libotter_hip.so.  Every wrapper validates devices/dtypes/contiguity and raises on error."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _capi as K
from ._capi import BF16, EPI_GATE_BWD, EPI_GELU, EPI_SCALE_RES, EPI_STORE, F32, MASK_EQ, MASK_GE, MASK_NONE, RowMap

_IDENT = RowMap(0, 0, 0)


class _Workspace:
    """Grow-only scratch buffer per (device, stream): reuse is ordered by the stream the kernels are launched on (torch's current stream).
    Keyed by the stream since round 5: the fusion modules launch on side streams too (functional._SideStream), and two kernels on two
    streams must never share one scratch buffer."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = b
        return b


_ws = _Workspace()


# OTTER_GEMM_LOG=1: count the GEMM shapes of a run and print them at exit (tuning aid)
_GEMM_LOG = None
if os.environ.get("OTTER_GEMM_LOG") == "1":
    import atexit
    import collections

    _GEMM_LOG = collections.Counter()
    atexit.register(lambda: print("\n".join("gemm M=%d N=%d K=%d epi=%d %s->%s  x%d" % (k + (v,)) for k, v in sorted(_GEMM_LOG.items()))))


def _c2d(t: torch.Tensor) -> torch.Tensor:
    if t.dim() != 2 or t.stride(1) != 1:
        raise K.OtterHipError(f"expected a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


# ----------------------------------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------------------------------


def layernorm_fwd(x2d, gamma, beta, out_dtype, eps=1e-5, y=None, ymap: Optional[RowMap] = None, y2=None, need_stats=True):
    """x2d [rows, D] contiguous.  Returns (y, mean, rstd)."""
    K.require_cuda(x2d, gamma, beta)
    x2d = _c2d(x2d)
    assert x2d.is_contiguous()
    rows, D = x2d.shape
    if y is None:
        y = torch.empty((rows, D), dtype=out_dtype, device=x2d.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device) if need_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if need_stats else None
    wdt = K.dt(gamma) if gamma is not None else F32
    if gamma is not None and beta is not None and gamma.dtype != beta.dtype:
        raise K.OtterHipError("layernorm: gamma/beta dtype mismatch")
    K.check(K.lib().otter_layernorm_fwd(x2d.data_ptr(), K.dt(x2d), K.ptr(gamma), K.ptr(beta), wdt, y.data_ptr(), K.dt(y),
                                        ymap or _IDENT, K.ptr(y2), K.ptr(mean), K.ptr(rstd), rows, D, float(eps), K.stream()),
            "layernorm_fwd")
    return y, mean, rstd


def add_layernorm_fwd(x2d, delta2d, gamma, beta, out_dtype, eps=1e-5, need_stats=True):
    """(xsum, y, mean, rstd) with xsum = x + delta (x's dtype) and y = LN(xsum)."""
    K.require_cuda(x2d, delta2d, gamma, beta)
    assert x2d.is_contiguous() and delta2d.is_contiguous() and x2d.shape == delta2d.shape
    rows, D = x2d.shape
    xsum = torch.empty_like(x2d)
    y = torch.empty((rows, D), dtype=out_dtype, device=x2d.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device) if need_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if need_stats else None
    wdt = K.dt(gamma) if gamma is not None else F32
    K.check(K.lib().otter_add_layernorm_fwd(x2d.data_ptr(), K.dt(x2d), delta2d.data_ptr(), K.dt(delta2d), xsum.data_ptr(),
                                            K.ptr(gamma), K.ptr(beta), wdt, y.data_ptr(), K.dt(y), K.ptr(mean), K.ptr(rstd), rows, D,
                                            float(eps), K.stream()), "add_layernorm_fwd")
    return xsum, y, mean, rstd


def layernorm_bwd(dy, x2d, gamma, mean, rstd, dx_dtype, dres=None, dymap: Optional[RowMap] = None, need_dw=True,
                  need_dbeta=True, need_dx=True, dx_bf16=None):
    """Returns (dx, dgamma, dbeta) -- dgamma/dbeta fp32.  dx_bf16: optional preallocated bf16 [rows, D] tensor that
    receives a bf16 copy of dx in the same pass."""
    K.require_cuda(dy, x2d)
    rows, D = x2d.shape
    dev = x2d.device
    dx = torch.empty((rows, D), dtype=dx_dtype, device=dev) if need_dx else None
    if dres is not None and dres.dtype != dx_dtype:
        raise K.OtterHipError("layernorm_bwd: dres must have the dx dtype")
    if dx_bf16 is not None and (dx_bf16.dtype != torch.bfloat16 or not dx_bf16.is_contiguous() or tuple(dx_bf16.shape) != (rows, D)):
        raise K.OtterHipError("layernorm_bwd: dx_bf16 must be a contiguous bf16 [rows, D] tensor")
    dg = torch.empty(D, dtype=torch.float32, device=dev) if need_dw else None
    db = torch.empty(D, dtype=torch.float32, device=dev) if (need_dw and need_dbeta) else None
    ws = None
    if need_dw:
        ws = _ws.get(K.lib().otter_layernorm_bwd_workspace_bytes(rows, D), dev)
    wdt = K.dt(gamma) if gamma is not None else F32
    K.check(K.lib().otter_layernorm_bwd(dy.data_ptr(), K.dt(dy), dymap or _IDENT, x2d.data_ptr(), K.dt(x2d), K.ptr(gamma), wdt,
                                        mean.data_ptr(), rstd.data_ptr(), K.ptr(dres), K.ptr(dx), K.dt_of(dx_dtype), K.ptr(dx_bf16),
                                        K.ptr(dg), K.ptr(db), 0, K.ptr(ws), rows, D, K.stream()), "layernorm_bwd")
    return dx, dg, db


def colsum(src2d, src_map: Optional[RowMap], rows: int, out=None, accumulate=False):
    """out[c] (+)= sum over r < rows of src2d[map(r)][c]; fp32 [D]."""
    D = src2d.shape[1]
    if out is None:
        out = torch.empty(D, dtype=torch.float32, device=src2d.device)
    ws = _ws.get(K.lib().otter_layernorm_bwd_workspace_bytes(rows, D), src2d.device)
    K.check(K.lib().otter_colsum(src2d.data_ptr(), K.dt(src2d), src_map or _IDENT, out.data_ptr(), 1 if accumulate else 0,
                                 ws.data_ptr(), rows, D, K.stream()), "colsum")
    return out


def rmsnorm_fwd(x2d, w, eps=1e-6):
    K.require_cuda(x2d, w)
    rows, D = x2d.shape
    y = torch.empty_like(x2d)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    K.check(K.lib().otter_rmsnorm_fwd(x2d.data_ptr(), K.dt(x2d), w.data_ptr(), K.dt(w), y.data_ptr(), rstd.data_ptr(), rows, D,
                                      float(eps), K.stream()), "rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy, x2d, w, rstd):
    rows, D = x2d.shape
    dx = torch.empty_like(x2d)
    dw = torch.empty(D, dtype=torch.float32, device=x2d.device)
    ws = _ws.get(K.lib().otter_layernorm_bwd_workspace_bytes(rows, D), x2d.device)
    K.check(K.lib().otter_rmsnorm_bwd(dy.data_ptr(), x2d.data_ptr(), K.dt(x2d), w.data_ptr(), K.dt(w), rstd.data_ptr(),
                                      dx.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), rows, D, K.stream()), "rmsnorm_bwd")
    return dx, dw


def add_rmsnorm_fwd(x2d, delta2d, w, out_dtype, eps=1e-6):
    """LLaMA host: (xsum, y, rstd) with xsum = x + delta (x's dtype; None when delta is None) and y = RMSNorm(xsum) in
    `out_dtype` (bf16 of the fp32 result when the stream is fp32)."""
    K.require_cuda(x2d, delta2d, w)
    assert x2d.is_contiguous() and (delta2d is None or (delta2d.is_contiguous() and delta2d.shape == x2d.shape))
    rows, D = x2d.shape
    xsum = torch.empty_like(x2d) if delta2d is not None else None
    y = torch.empty((rows, D), dtype=out_dtype, device=x2d.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    K.check(K.lib().otter_add_rmsnorm_fwd(x2d.data_ptr(), K.dt(x2d), K.ptr(delta2d), K.dt(delta2d) if delta2d is not None else F32, K.ptr(xsum),
                                          w.data_ptr(), K.dt(w), y.data_ptr(), K.dt(y), rstd.data_ptr(), rows, D, float(eps), K.stream()),
            "add_rmsnorm_fwd")
    return xsum, y, rstd


def rmsnorm_bwd_ex(dy, x2d, w, rstd, dx_dtype, dres=None, need_dw=False, dx_bf16=None):
    """(dx, dw): dx = RMSNorm'(dy) (+ dres), optional bf16 copy of dx into dx_bf16, dw fp32 [D] only when need_dw."""
    K.require_cuda(dy, x2d, w, rstd, dres, dx_bf16)
    rows, D = x2d.shape
    dev = x2d.device
    assert dy.is_contiguous() and tuple(dy.shape) == (rows, D)
    if dres is not None and (dres.dtype != dx_dtype or not dres.is_contiguous()):
        raise K.OtterHipError("rmsnorm_bwd_ex: dres must be contiguous and have the dx dtype")
    dx = torch.empty((rows, D), dtype=dx_dtype, device=dev)
    dw = torch.empty(D, dtype=torch.float32, device=dev) if need_dw else None
    ws = _ws.get(K.lib().otter_layernorm_bwd_workspace_bytes(rows, D), dev) if need_dw else None
    K.check(K.lib().otter_rmsnorm_bwd_ex(dy.data_ptr(), K.dt(dy), x2d.data_ptr(), K.dt(x2d), w.data_ptr(), K.dt(w), rstd.data_ptr(), K.ptr(dres),
                                         dx.data_ptr(), K.dt(dx), K.ptr(dx_bf16), K.ptr(dw), 0, K.ptr(ws), rows, D, K.stream()), "rmsnorm_bwd_ex")
    return dx, dw


# ----------------------------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------------------------


def gemm_nt(A, B, out_dtype=None, out=None, kind=EPI_STORE, gate=None, R=None, C2=None, aux=None, aux_gelu=False,
            partial=None, accumulate=False, k=None, a_kmajor=False, b_kmajor=False):
    """C[M,N] = epilogue(A[M,K] . B[N,K]^T).  A, B row-major 2-D (row stride free), same dtype.
    `k` restricts the reduction to the first k columns (used with zero-padded transposed operands).
    a_kmajor / b_kmajor (see `gemm`): the operand is given as [K, M] / [K, N] instead."""
    K.require_cuda(A, B)
    A, B = _c2d(A), _c2d(B)
    if A.dtype != B.dtype:
        raise K.OtterHipError(f"gemm: operand dtypes differ ({A.dtype} vs {B.dtype})")
    M, Ka = (A.shape[1], A.shape[0]) if a_kmajor else A.shape
    N, Kb = (B.shape[1], B.shape[0]) if b_kmajor else B.shape
    Kd = k if k is not None else Ka
    if Kd > Ka or Kd > Kb or (k is None and Ka != Kb):
        raise K.OtterHipError(f"gemm: K mismatch A{tuple(A.shape)} B{tuple(B.shape)} k={k}")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    out = _c2d(out)
    e = K.EpilogueArgs()
    e.kind = kind
    e.accumulate = 1 if accumulate else 0
    e.gate = K.ptr(gate)
    if R is not None:
        R = _c2d(R)
        e.R, e.ldr, e.r_dtype = R.data_ptr(), R.stride(0), K.dt(R)
    if C2 is not None:
        C2 = _c2d(C2)
        if C2.dtype != out.dtype:
            raise K.OtterHipError("gemm: C2 must have the dtype of C")
        e.C2, e.ldc2 = C2.data_ptr(), C2.stride(0)
    if aux is not None:
        aux = _c2d(aux)
        e.aux, e.ldaux, e.aux_dtype = aux.data_ptr(), aux.stride(0), K.dt(aux)
    # activation whose derivative the GATE_BWD tail applies to aux; "stash": the derivative itself travels (GELU launch: C2 = GELU'(acc),
    # GATE_BWD launch: aux holds it)
    e.aux_is_gelu_input = 3 if aux_gelu == "stash" else (2 if aux_gelu == "sqrelu" else (1 if aux_gelu else 0))
    e.partial = K.ptr(partial)
    e.grid_mode = _grid_mode
    if _GEMM_LOG is not None:
        _GEMM_LOG[(M, N, Kd, kind, str(A.dtype).split(".")[-1], str(out.dtype).split(".")[-1])] += 1
    if a_kmajor or b_kmajor:
        K.check(K.lib().otter_gemm(A.data_ptr(), A.stride(0), 1 if a_kmajor else 0, B.data_ptr(), B.stride(0), 1 if b_kmajor else 0, out.data_ptr(),
                                   out.stride(0), M, N, Kd, K.dt(A), K.dt(out), C.byref(e), K.stream()), "gemm")
        return out
    K.check(K.lib().otter_gemm_nt(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0), M, N, Kd,
                                  K.dt(A), K.dt(out), C.byref(e), K.stream()), "gemm_nt")
    return out


def gemm_kmajor_supported(M, N, Kd, lda, ldb, a_kmajor, b_kmajor, dtype) -> bool:
    """True when otter_gemm takes these operands in place (K-major = [K rows][M or N columns]); else transpose + gemm_nt."""
    if dtype != torch.bfloat16:
        return False
    return bool(K.lib().otter_gemm_kmajor_supported(int(M), int(N), int(Kd), int(lda), int(ldb), 1 if a_kmajor else 0, 1 if b_kmajor else 0, BF16))


def gemm(A, B, a_kmajor=False, b_kmajor=False, **kw):
    """C = epilogue(op(A) . op(B)^T) with K-MAJOR operands read in place: a_kmajor -> A is [K, M] (the reduction index is the row
    index), b_kmajor -> B is [K, N].  dW = dy^T x is gemm(dy, x, True, True); dx = dy W (W stored [out, in]) is gemm(dy, W, False, True)."""
    return gemm_nt(A, B, a_kmajor=a_kmajor, b_kmajor=b_kmajor, **kw)


def gemm_num_partials(M, N, dtype) -> int:
    return int(K.lib().otter_gemm_num_partials(M, N, K.dt_of(dtype)))


def reduce_partials(partial, gate=None, out=None, accumulate=False):
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=partial.device)
    K.check(K.lib().otter_reduce_partials(partial.data_ptr(), partial.numel(), K.ptr(gate), out.data_ptr(), 1 if accumulate else 0,
                                          K.stream()), "reduce_partials")
    return out


def transpose(src, out_dtype, want_same=False, pad_to=8):
    """src [rows, cols] -> (src^T as [cols, rows_padded] with zero tail columns, optional same-layout cast copy).
    rows_padded = roundup(rows, pad_to) so the result can feed gemm_nt as a K-contiguous operand."""
    K.require_cuda(src)
    src = _c2d(src)
    rows, cols = src.shape
    rp = (rows + pad_to - 1) // pad_to * pad_to
    if rp != rows:
        dst_t = torch.zeros((cols, rp), dtype=out_dtype, device=src.device)
    else:
        dst_t = torch.empty((cols, rp), dtype=out_dtype, device=src.device)
    same = torch.empty((rows, cols), dtype=out_dtype, device=src.device) if want_same else None
    K.check(K.lib().otter_transpose(src.data_ptr(), src.stride(0), K.dt(src), dst_t.data_ptr(), dst_t.stride(0), K.ptr(same),
                                    cols if want_same else 0, K.dt_of(out_dtype), rows, cols, K.stream()), "transpose")
    return (dst_t, same) if want_same else dst_t


def cast(src, out_dtype):
    K.require_cuda(src)
    if src.dtype == out_dtype:
        return src
    src = src.contiguous()
    out = torch.empty(src.shape, dtype=out_dtype, device=src.device)
    K.check(K.lib().otter_cast(src.data_ptr(), K.dt(src), out.data_ptr(), K.dt(out), src.numel(), K.stream()), "cast")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# attention + mask
# ----------------------------------------------------------------------------------------------------------------------


def text_time(media_locations: torch.Tensor, attend_previous: bool = True) -> torch.Tensor:
    """bool/uint8 [B,T] -> int32 [B,T] (modeling_otter.py:298-311)."""
    K.require_cuda(media_locations)
    ml = media_locations.to(torch.uint8).contiguous()
    B, T = ml.shape
    tt = torch.empty((B, T), dtype=torch.int32, device=ml.device)
    K.check(K.lib().otter_text_time(ml.data_ptr(), tt.data_ptr(), B, T, 1 if attend_previous else 0, K.stream()), "text_time")
    return tt


def attn_fwd(q, k, v, H, tt, n_per_media, mask_mode, scale, need_lse=True):
    """q [B,Tq,H*64]; k, v [B,M,H*64] (may be strided views of one [B,M,2*H*64] buffer).  Returns (o, lse)."""
    K.require_cuda(q, k, v)
    B, Tq, HD = q.shape
    M = k.shape[1]
    if HD != H * 64:
        raise K.OtterHipError(f"attention core supports head_dim 64 only (got inner={HD}, heads={H})")
    if q.stride(2) != 1 or k.stride(2) != 1 or v.stride(2) != 1 or k.stride(1) != v.stride(1):
        raise K.OtterHipError("attn: bad strides")
    if q.stride(0) != Tq * q.stride(1) or k.stride(0) != M * k.stride(1) or v.stride(0) != M * v.stride(1):
        raise K.OtterHipError("attn: batch stride must equal rows*row_stride")
    o = torch.empty((B, Tq, HD), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if need_lse else None
    K.check(K.lib().otter_attn_fwd(q.data_ptr(), q.stride(1), k.data_ptr(), v.data_ptr(), k.stride(1), o.data_ptr(), o.stride(1),
                                   K.ptr(lse), K.ptr(tt), B, H, Tq, M, n_per_media, mask_mode, float(scale), K.dt(q), K.stream()),
            "attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, do, lse, H, tt, n_per_media, mask_mode, scale, dkv_out=None):
    """Returns (dq [B,Tq,H*64], dkv [B,M,2*H*64]) -- dk in the first half of the last axis, dv in the second."""
    B, Tq, HD = q.shape
    M = k.shape[1]
    dq = torch.empty((B, Tq, HD), dtype=q.dtype, device=q.device)
    dkv = dkv_out if dkv_out is not None else torch.empty((B, M, 2 * HD), dtype=q.dtype, device=q.device)
    dk, dv = dkv[..., :HD], dkv[..., HD:]
    ws = _ws.get(K.lib().otter_attn_bwd_workspace_bytes(B, H, Tq, M), q.device)
    do = do.contiguous()
    K.check(K.lib().otter_attn_bwd(q.data_ptr(), q.stride(1), k.data_ptr(), v.data_ptr(), k.stride(1), o.data_ptr(), do.data_ptr(),
                                   o.stride(1), lse.data_ptr(), K.ptr(tt), dq.data_ptr(), dq.stride(1), dk.data_ptr(),
                                   dv.data_ptr(), dkv.stride(1), ws.data_ptr(), B, H, Tq, M, n_per_media, mask_mode, float(scale),
                                   K.dt(q), K.stream()), "attn_bwd")
    return dq, dkv


# ----------------------------------------------------------------------------------------------------------------------
# misc
# ----------------------------------------------------------------------------------------------------------------------


def _flash_view(t: torch.Tensor, hd: int = 128) -> K.FlashView:
    """t: a [B, S, H, head_dim] bf16 view with unit stride along the head dim."""
    if t.dim() != 4 or t.shape[3] != hd or t.stride(3) != 1 or t.dtype != torch.bfloat16:
        raise K.OtterHipError(f"flash attention wants [B,S,H,{hd}] bf16 views (got {tuple(t.shape)}, {t.dtype}, strides {t.stride()})")
    return K.FlashView(t.stride(0), t.stride(1), t.stride(2))


def _flash_desc(q, k, v, o, lse, slopes, key_valid, scale, causal) -> K.FlashDesc:
    K.require_cuda(q, k, v, o, lse, slopes, key_valid)
    B, Sq, H, hd = q.shape
    Sk = k.shape[1]
    if hd not in (64, 128) or (hd == 64 and H % 2):
        raise K.OtterHipError(f"flash attention: head_dim 128, or 64 with an even number of heads (got head_dim {hd}, {H} heads)")
    if key_valid is not None and (key_valid.dtype != torch.uint8 or not key_valid.is_contiguous() or key_valid.shape != (B, Sk)):
        raise K.OtterHipError("flash attention: key_valid must be a contiguous uint8 [B, Sk] tensor")
    if slopes is not None and (slopes.dtype != torch.float32 or not slopes.is_contiguous() or slopes.numel() != H):
        raise K.OtterHipError("flash attention: alibi slopes must be a contiguous fp32 [H] tensor")
    d = K.FlashDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    d.qv, d.kv, d.vv, d.ov = _flash_view(q, hd), _flash_view(k, hd), _flash_view(v, hd), _flash_view(o, hd)
    d.lse = lse.data_ptr()
    d.alibi_slopes = K.ptr(slopes)
    d.key_valid = K.ptr(key_valid)
    d.B, d.H, d.Sq, d.Sk, d.head_dim, d.causal = B, H, Sq, Sk, hd, int(bool(causal))
    d.scale = float(scale)
    return d


def flash_attn_fwd(q, k, v, slopes, key_valid, scale, causal=True):
    """Decoder-host attention (mpt/attention.py:22-84 + ALiBi :447-464) on [B,S,H,128] bf16 views, or Persimmon's
    (fuyu/modeling_persimmon.py:310) on [B,S,H,64] views with H even.  Returns (o [B,Sq,H,head_dim] contiguous, lse [B,H,Sq] fp32)."""
    B, Sq, H, hd = q.shape
    o = torch.empty((B, Sq, H, hd), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    d = _flash_desc(q, k, v, o, lse, slopes, key_valid, scale, causal)
    K.check(K.lib().otter_flash_attn_fwd(C.byref(d), K.stream()), "flash_attn_fwd")
    return o, lse


def flash_attn_bwd(q, k, v, o, lse, dout, dq, dk, dv, slopes, key_valid, scale, causal=True):
    """Writes dq / dk / dv (caller-provided [B,S,H,head_dim] views, e.g. the three slices of one dqkv buffer)."""
    K.require_cuda(dout, dq, dk, dv)
    hd = q.shape[3]
    d = _flash_desc(q, k, v, o, lse, slopes, key_valid, scale, causal)
    delta = torch.empty((2,) + tuple(lse.shape), dtype=torch.float32, device=lse.device)  # row dots + log2-domain lse
    d.dout, d.dov = dout.data_ptr(), _flash_view(dout, hd)
    d.delta = delta.data_ptr()
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.dqv, d.dkv, d.dvv = _flash_view(dq, hd), _flash_view(dk, hd), _flash_view(dv, hd)
    K.check(K.lib().otter_flash_attn_bwd(C.byref(d), K.stream()), "flash_attn_bwd")


def rope(x, cos, sin, rot_dim=None, inverse=False, out=None):
    """x [B,S,H,d] contiguous, cos/sin fp32 [S,rot]."""
    K.require_cuda(x, cos, sin)
    x = x.contiguous()
    B, S, H, d = x.shape
    rot = rot_dim or d
    y = out if out is not None else torch.empty_like(x)
    K.check(K.lib().otter_rope(x.data_ptr(), y.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, S, H, d, rot, 1 if inverse else 0,
                               K.dt(x), K.stream()), "rope")
    return y


def rope_strided(x, y, cos, sin, tokens, S, H, d, x_token_stride, y_token_stride, inverse=False):
    """bf16 full-rotary RoPE on `tokens` = B*S tokens of H contiguous heads each, tokens x_/y_token_stride elements apart
    (x / y are any tensors whose data_ptr is the first rotated element; in place when y is x).  cos/sin fp32 [S, d]."""
    K.require_cuda(x, y, cos, sin)
    if x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16 or cos.dtype != torch.float32 or sin.dtype != torch.float32:
        raise K.OtterHipError("rope_strided: bf16 data, fp32 tables")
    if not cos.is_contiguous() or not sin.is_contiguous() or tuple(cos.shape) != (S, d) or tuple(sin.shape) != (S, d):
        raise K.OtterHipError("rope_strided: cos/sin must be contiguous [S, d]")
    K.check(K.lib().otter_rope_strided(x.data_ptr(), y.data_ptr(), cos.data_ptr(), sin.data_ptr(), tokens, S, H, d, 1 if inverse else 0,
                                       x_token_stride, y_token_stride, K.stream()), "rope_strided")
    return y


def quick_gelu_(x):
    """In place x * sigmoid(1.702 x) on a contiguous bf16 / f32 tensor (CLIP MLP activation)."""
    K.require_cuda(x)
    if not x.is_contiguous() or x.numel() % 8:
        raise K.OtterHipError("quick_gelu: contiguous tensor with numel % 8 == 0")
    K.check(K.lib().otter_quick_gelu(x.data_ptr(), x.data_ptr(), x.numel(), K.dt(x), K.stream()), "quick_gelu")
    return x


def gelu_fwd(x):
    """Exact-erf GELU (nn.GELU() default) of a contiguous bf16 / f32 tensor, numel % 8 == 0 (MPT MLP activation)."""
    K.require_cuda(x)
    if not x.is_contiguous() or x.numel() % 8:
        raise K.OtterHipError("gelu_fwd: contiguous tensor with numel % 8 == 0")
    y = torch.empty_like(x)
    K.check(K.lib().otter_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), K.dt(x), K.stream()), "gelu_fwd")
    return y


def gelu_bwd(x, dy):
    """dx = dy * gelu'(x) (exact-erf form), same layout rules as gelu_fwd."""
    K.require_cuda(x, dy)
    if not x.is_contiguous() or not dy.is_contiguous() or x.numel() % 8 or dy.shape != x.shape or dy.dtype != x.dtype:
        raise K.OtterHipError("gelu_bwd: x and dy must be contiguous, same shape / dtype, numel % 8 == 0")
    dx = torch.empty_like(x)
    K.check(K.lib().otter_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), K.dt(x), K.stream()), "gelu_bwd")
    return dx


def swiglu_fwd(gu2d):
    """gu2d [rows, 2*I] bf16 contiguous (gate | up) -> h [rows, I] = silu(gate) * up."""
    K.require_cuda(gu2d)
    rows, I2 = gu2d.shape
    if gu2d.dtype != torch.bfloat16 or not gu2d.is_contiguous() or I2 % 16:
        raise K.OtterHipError("swiglu: contiguous bf16 [rows, 2*I] with I % 8 == 0")
    h = torch.empty((rows, I2 // 2), dtype=torch.bfloat16, device=gu2d.device)
    K.check(K.lib().otter_swiglu_fwd(gu2d.data_ptr(), h.data_ptr(), rows, I2 // 2, K.stream()), "swiglu_fwd")
    return h


def swiglu_bwd(gu2d, dh2d):
    K.require_cuda(gu2d, dh2d)
    rows, I2 = gu2d.shape
    if dh2d.dtype != torch.bfloat16 or not dh2d.is_contiguous() or tuple(dh2d.shape) != (rows, I2 // 2):
        raise K.OtterHipError("swiglu_bwd: dh must be contiguous bf16 [rows, I]")
    dgu = torch.empty_like(gu2d)
    K.check(K.lib().otter_swiglu_bwd(gu2d.data_ptr(), dh2d.data_ptr(), dgu.data_ptr(), rows, I2 // 2, K.stream()), "swiglu_bwd")
    return dgu


def decode_attn(q, k, v, slopes, key_valid, scale):
    """Single-query attention over a KV cache.  q [B,H,128] bf16; k, v: 4-d bf16 tensors indexed [b, h, key, dim] through their own
    strides (pass `k_cache.transpose(2, 3)` for the MPT layout [B,H,d,S]); slopes fp32 [H] or None; key_valid uint8 [B,Sk] or None.
    Returns o [B,H,128]."""
    K.require_cuda(q, k, v, slopes, key_valid)
    B, H, d = q.shape
    Sk = k.shape[2]
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16 or d != 128 or q.stride(2) != 1:
        raise K.OtterHipError("decode_attn: bf16, head_dim 128, unit-stride q")
    if tuple(k.shape) != (B, H, Sk, d) or tuple(v.shape) != (B, H, Sk, d):
        raise K.OtterHipError("decode_attn: k, v must be indexable as [B,H,Sk,128]")
    if key_valid is not None and (key_valid.dtype != torch.uint8 or not key_valid.is_contiguous() or tuple(key_valid.shape) != (B, Sk)):
        raise K.OtterHipError("decode_attn: key_valid must be contiguous uint8 [B,Sk]")
    o = torch.empty((B, H, d), dtype=torch.bfloat16, device=q.device)
    K.check(K.lib().otter_decode_attn(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2), k.stride(3),
                                      v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), v.stride(3), o.data_ptr(), o.stride(0), o.stride(1),
                                      K.ptr(slopes), K.ptr(key_valid), B, H, Sk, d, float(scale), K.stream()), "decode_attn")
    return o


def qk_norm_rope_fwd(qkv, gq, bq, gk, bk, cos, sin, H, rot, eps, width=128, copy_v=True):
    """qkv [B,S,H*3*64] bf16 (per head q|k|v) -> q', k', v as [B,S,H,width] bf16, stats [B*S,H,2,2].  width 128: upper 64 columns zero
    (the 128-wide flash kernels); width 64: compact heads (the head-pair kernels), and with copy_v=False v is returned as the strided
    [B,S,H,64] view of qkv's v slots that those kernels read in place."""
    K.require_cuda(qkv, gq, bq, gk, bk, cos, sin)
    B, S, W = qkv.shape
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous() or W != H * 3 * 64:
        raise K.OtterHipError("qk_norm_rope: contiguous bf16 [B,S,H*3*64]")
    if width not in (64, 128):
        raise K.OtterHipError("qk_norm_rope: width 64 or 128")
    for t in (gq, bq, gk, bk):
        if t.dtype != torch.float32 or t.numel() != 64 or not t.is_contiguous():
            raise K.OtterHipError("qk_norm_rope: gamma / beta must be contiguous fp32 [64]")
    if cos.dtype != torch.float32 or not cos.is_contiguous() or tuple(cos.shape) != (S, rot) or tuple(sin.shape) != (S, rot) or not sin.is_contiguous():
        raise K.OtterHipError("qk_norm_rope: cos / sin must be contiguous fp32 [S, rot]")
    q = torch.empty((B, S, H, width), dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty_like(q)
    v = torch.empty_like(q) if copy_v else None
    stats = torch.empty((B * S, H, 2, 2), dtype=torch.float32, device=qkv.device)
    K.check(K.lib().otter_qk_norm_rope_fwd(qkv.data_ptr(), gq.data_ptr(), bq.data_ptr(), gk.data_ptr(), bk.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                           q.data_ptr(), k.data_ptr(), K.ptr(v), stats.data_ptr(), B * S, S, H, rot, float(eps), width, K.stream()),
            "qk_norm_rope_fwd")
    if v is None:
        v = qkv.view(B, S, H, 3, 64)[:, :, :, 2]
    return q, k, v, stats


def qk_norm_rope_bwd(dq, dk, dv, qkv, stats, gq, gk, cos, sin, H, rot, dqkv=None):
    """-> (dqkv like qkv, dgamma_q, dbeta_q, dgamma_k, dbeta_k fp32 [64]).  dq / dk (/ dv) contiguous [B,S,H,64] or [B,S,H,128]; dv None:
    the caller's dqkv already holds the v gradients in its v slots (written in place by the attention backward)."""
    K.require_cuda(dq, dk, dv, qkv, stats, dqkv)
    B, S, _ = qkv.shape
    width = dq.shape[-1]
    for t in (dq, dk, dv):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or tuple(t.shape) != (B, S, H, width) or width not in (64, 128)):
            raise K.OtterHipError("qk_norm_rope_bwd: dq / dk / dv must be contiguous bf16 [B,S,H,64] or [B,S,H,128]")
    if dv is None and dqkv is None:
        raise K.OtterHipError("qk_norm_rope_bwd: dv=None needs the dqkv buffer whose v slots hold the v gradients")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    elif dqkv.dtype != torch.bfloat16 or not dqkv.is_contiguous() or dqkv.shape != qkv.shape:
        raise K.OtterHipError("qk_norm_rope_bwd: dqkv must be contiguous bf16 like qkv")
    nb = int(K.lib().otter_qk_norm_rope_bwd_blocks(B * S, H))
    partial = torch.empty((nb, 4, 64), dtype=torch.float32, device=qkv.device)
    K.check(K.lib().otter_qk_norm_rope_bwd(dq.data_ptr(), dk.data_ptr(), K.ptr(dv), qkv.data_ptr(), stats.data_ptr(), gq.data_ptr(), gk.data_ptr(),
                                           cos.data_ptr(), sin.data_ptr(), dqkv.data_ptr(), partial.data_ptr(), B * S, S, H, rot, width, K.stream()),
            "qk_norm_rope_bwd")
    p = partial.sum(0)
    return dqkv, p[0], p[1], p[2], p[3]


def sqrelu_fwd(x):
    K.require_cuda(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.numel() % 8:
        raise K.OtterHipError("sqrelu: contiguous bf16 tensor with numel % 8 == 0")
    y = torch.empty_like(x)
    K.check(K.lib().otter_sqrelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), K.stream()), "sqrelu_fwd")
    return y


def sqrelu_bwd(x, dy):
    K.require_cuda(x, dy)
    if dy.dtype != torch.bfloat16 or not dy.is_contiguous() or dy.shape != x.shape:
        raise K.OtterHipError("sqrelu_bwd: dy must be contiguous bf16 like x")
    dx = torch.empty_like(x)
    K.check(K.lib().otter_sqrelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), K.stream()), "sqrelu_bwd")
    return dx


def scatter_rows(word, patch, idx):
    """word [B,S,D], patch [B,P,D] (f32 or bf16 each), idx int64 [B,S] -> out like word."""
    K.require_cuda(word, patch, idx)
    B, S, D = word.shape
    P = patch.shape[1]
    if not (word.is_contiguous() and patch.is_contiguous() and idx.is_contiguous()) or idx.dtype != torch.int64 or tuple(idx.shape) != (B, S) \
            or patch.shape[0] != B or patch.shape[2] != D:
        raise K.OtterHipError("scatter_rows: contiguous word [B,S,D], patch [B,P,D], int64 idx [B,S]")
    out = torch.empty_like(word)
    K.check(K.lib().otter_scatter_rows(word.data_ptr(), K.dt(word), patch.data_ptr(), K.dt(patch), idx.data_ptr(), out.data_ptr(), B, S, P, D, K.stream()),
            "scatter_rows")
    return out


def add_frame_embs_(x5, emb):
    """x5 [b,T,F,v,D] (contiguous, modified in place) += emb[:F] broadcast (modeling_otter.py:224-226)."""
    b, T, F, v, D = x5.shape
    assert x5.is_contiguous() and emb.dtype == torch.float32 and emb.is_contiguous()
    K.check(K.lib().otter_add_frame_embs(x5.data_ptr(), K.dt(x5), emb.data_ptr(), b * T, F, v, D, K.stream()), "add_frame_embs")
    return x5


def add_rows_(dst2d, src2d, src_map: RowMap):
    """dst2d[r] += src2d[map(r)] (same dtype, same D)."""
    rows, D = dst2d.shape
    assert dst2d.is_contiguous() and src2d.is_contiguous() and dst2d.dtype == src2d.dtype
    K.check(K.lib().otter_add_rows(dst2d.data_ptr(), src2d.data_ptr(), src_map, rows, D, K.dt(dst2d), K.stream()), "add_rows")
    return dst2d


def set_attn_variant(v: int):
    """0 = MFMA attention cores for bf16 (default), 1 = fp32 VALU kernels (A/B hook)."""
    K.check(K.lib().otter_attn_set_variant(int(v)), "attn_set_variant")


def set_flash_variant(v: int):
    """0 = default (LDS-DMA tiles, longest-first block order, persistent dK/dV where it applies), 1 = register-staged tiles, 2 = LDS-DMA tiles
    on the plain grid, 3-5 = dK/dV occupancy / order variants, 6 = forward version 3, 7 = per-key-block dK/dV, 8 = persistent dK/dV wherever
    it can run (A/B hook of the decoder-host flash attention; csrc/flash.hip, include/otter_hip.h)."""
    K.check(K.lib().otter_flash_set_variant(int(v)), "flash_set_variant")


def set_gemm_variant(v: int):
    K.check(K.lib().otter_gemm_set_variant(int(v)), "gemm_set_variant")


def gemm_variant_available(v: int) -> bool:
    """True when schedule `v` is compiled into the loaded library (the product build carries 0-3, 13, 25, 26; the rest live in the
    tools-only experimental build: OTTER_LIB_PATH=otter_amd/lib/libotter_hip_experimental.so)."""
    return bool(K.lib().otter_gemm_variant_available(int(v)))


# Grid shape of the large-grid GEMM launches issued while a `gemm_grid_mode(...)` scope is open (otter_grid_mode in include/otter_hip.h).
# The scope is a module variable, not a thread-local: the backward products are launched from autograd's device thread, not from the thread
# that called TrainStep.__call__.  It is set for the duration of ONE training step by the TrainStep that owns a DP reducer and restored
# when the step returns -- nothing is left behind in the library.  It is PROCESS-WIDE while that step runs:
# any other thread that launches GEMMs during the window (an eval thread, a second TrainStep) inherits the mode, and overlapping scopes
# from two threads restore in LIFO order only if they nest.  The mode affects the grid shape (performance), never results; one training
# thread per process (the reference's own model: one rank = one Python thread) is what the scope is designed for.
_grid_mode = K.GRID_DEFAULT


class gemm_grid_mode:
    """with ops.gemm_grid_mode(K.GRID_PER_TILE): ...  -- every otter_gemm / otter_gemm_nt launch inside asks for that grid shape."""

    def __init__(self, mode: int):
        if mode not in (K.GRID_DEFAULT, K.GRID_PERSISTENT, K.GRID_PER_TILE):
            raise ValueError("gemm_grid_mode: %r" % (mode,))
        self.mode, self.prev = mode, None

    def __enter__(self):
        global _grid_mode
        self.prev, _grid_mode = _grid_mode, self.mode
        return self

    def __exit__(self, *exc):
        global _grid_mode
        _grid_mode = self.prev
        return False


def set_gemm_persistent(on: bool) -> None:
    """False: variant 26 runs one workgroup per tile (robust when RCCL kernels hold CUs during the backward GEMMs)."""
    K.check(K.lib().otter_gemm_set_persistent(1 if on else 0), "gemm_set_persistent")


def set_gemm_cu_budget(cus: int) -> int:
    """Cap the persistent GEMM grids at `cus` workgroups (0 = all CUs); returns the grid size in effect."""
    n = K.lib().otter_gemm_set_cu_budget(int(cus))
    if n < 0:
        K.check(n, "gemm_set_cu_budget")
    return int(n)


def prof_arm_gemm(M, N, Kd, max_events=4096):
    K.check(K.lib().otter_prof_arm_gemm(M, N, Kd, max_events), "prof_arm")


def prof_collect():
    n = C.c_int(0)
    ms = C.c_double(0.0)
    K.check(K.lib().otter_prof_collect(C.byref(n), C.byref(ms)), "prof_collect")
    return n.value, ms.value


def prof_collect_split():
    """(launches, total ms, launches with a K-major operand, their ms)."""
    n, nk = C.c_int(0), C.c_int(0)
    ms, mk = C.c_double(0.0), C.c_double(0.0)
    K.check(K.lib().otter_prof_collect_split(C.byref(n), C.byref(ms), C.byref(nk), C.byref(mk)), "prof_collect_split")
    return n.value, ms.value, nk.value, mk.value


def prof_disarm():
    K.check(K.lib().otter_prof_disarm(), "prof_disarm")
