"""Autograd functions of the fusion hot path: one torch.autograd.Function per reference module forward
(OtterPerceiverBlock, OtterGatedCrossAttentionBlock, LayerNorm), each a fixed schedule of libotter_hip.so calls with a
hand-derived backward.  No ATen arithmetic happens here; torch supplies tensors (memory) and the autograd graph.

Precision contract (mirrors what accelerate's bf16 autocast does to the reference, SURVEY.md section 7 "autocast dtype
contract"): `cd` = compute dtype (bf16 under autocast / bf16 models, f32 otherwise) is the storage type of GEMM operands
and intermediate activations; the residual stream keeps the dtype of the incoming hidden states; parameters keep their
master dtype, with per-step cd shadows (W and W^T) cached on the parameter version counter; all accumulation is fp32.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn.functional as F

from . import ops
from ._capi import EPI_GATE_BWD, EPI_GELU, EPI_SCALE_RES, EPI_STORE, MASK_EQ, MASK_GE, MASK_NONE, RowMap

HEAD_DIM = 64


def compute_dtype_for(x: torch.Tensor) -> torch.dtype:
    if x.dtype == torch.bfloat16:
        return torch.bfloat16
    if torch.is_autocast_enabled():
        adt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
        if adt == torch.bfloat16:
            return torch.bfloat16
    return torch.float32


class _Shadows:
    """cd-typed copies of weight matrices (same layout, and transposed for the dgrad GEMMs), refreshed when the
    parameter's version counter moves (optimizer.step / load_state_dict bump it)."""

    def __init__(self):
        self._w = {}
        self._wt = {}

    @staticmethod
    def _key(p, cd):
        return (id(p), cd)

    def _valid(self, store, p, cd):
        ent = store.get(self._key(p, cd))
        if ent is None:
            return None
        ref, ver, ptr_, t = ent
        if ref() is not p or ver != p._version or ptr_ != p.data_ptr():
            return None
        return t

    def w(self, p: torch.Tensor, cd) -> torch.Tensor:
        if p.dtype == cd:
            return p.detach()
        t = self._valid(self._w, p, cd)
        if t is None:
            t = ops.cast(p.detach(), cd)
            self._w[self._key(p, cd)] = (weakref.ref(p), p._version, p.data_ptr(), t)
        return t

    def wt(self, p: torch.Tensor, cd) -> torch.Tensor:
        t = self._valid(self._wt, p, cd)
        if t is None:
            # transposing the cd-typed copy (when it is current) reads half the bytes and gives the same bits
            src = self._valid(self._w, p, cd) if p.dtype != cd else None
            t = ops.transpose(src if src is not None else p.detach(), cd)
            self._wt[self._key(p, cd)] = (weakref.ref(p), p._version, p.data_ptr(), t)
        return t

    # ---- hooks for an optimizer that refreshes the cd-typed copy inside its own update kernel (otter_amd/optim.py) ----
    def stale_w(self, p: torch.Tensor, cd):
        """The cached copy's storage (whatever its version), so the optimizer can overwrite it in place; None if absent."""
        ent = self._w.get(self._key(p, cd))
        if ent is None or ent[0]() is not p or ent[2] != p.data_ptr() or not ent[3].is_contiguous():
            return None
        return ent[3]

    def mark_w(self, p: torch.Tensor, cd, t: torch.Tensor):
        self._w[self._key(p, cd)] = (weakref.ref(p), p._version, p.data_ptr(), t)

    def clear(self):
        self._w.clear()
        self._wt.clear()


shadows = _Shadows()


# Set by dp.GradReducer: lets a weight gradient be written by its GEMM straight into the parameter's slice of the flat
# communication bucket (no zero-fill + accumulate pass, no copy).  Interface: take(param) -> fp32 view or None,
# ready(param).
grad_sink = None


# Set by train.TrainStep: collects the SPARSE part of a tied embedding's gradient (the rows the input lookup touched) on the side, so
# that the parameter's dense gradient -- the un-embedding product, the first thing backward computes -- is final, and reducible, right
# away instead of after the very last node of the graph.  Interface: add(param, ids, drows); anchor (a 0-d tensor that requires grad).
embed_sink = None


# Set by train.TrainStep on a single rank (round 6b): the clip_grad_norm_ reduction of a weight gradient is taken INSIDE the GEMM that
# produces it (otter_epilogue_args::partial with a plain fp32 store: sum(dW^2) per output tile while the values are in registers) instead
# of by the optimizer's 4-byte-per-parameter sweep over every gradient.  Interface: slot(param, n) -> fp32 [n] buffer for the launch's
# partials, commit(param, grad_tensor).  The optimizer (optim.FusedAdamW) takes a tensor's fused partials only if the parameter's .grad
# still IS that launch's output, untouched (data pointer and version counter) -- anything else falls back to the sweep.
norm_sink = None


class GradNormSink:
    """Per-step registry of weight gradients whose sum of squares was produced by their own GEMM launch."""

    def __init__(self):
        self.buf = None
        self.used = 0
        self.entries = {}     # id(param) -> (offset, n, data_ptr, version, numel)

    def begin(self):
        self.used = 0
        self.entries.clear()

    def slot(self, param, n: int):
        if id(param) in self.entries:          # a second weight-gradient launch for the same parameter (shared weight): not fusable
            self.entries[id(param)] = None
            return None
        if self.buf is None or self.used + n > self.buf.numel():
            if self.used:                      # a live step never moves its buffer: launches already hold pointers into it
                return None
            self.buf = torch.empty(max(1 << 16, 2 * n), dtype=torch.float32, device=param.device)
        off = self.used
        self.used += n
        self.entries[id(param)] = (off, n, 0, 0, 0)
        return self.buf[off:off + n]

    def commit(self, param, grad: torch.Tensor):
        e = self.entries.get(id(param))
        if e is not None:
            self.entries[id(param)] = (e[0], e[1], grad.data_ptr(), grad._version, grad.numel())

    def fused(self, param):
        """(offset, n) if `param.grad` is still exactly what the committed launch wrote, else None."""
        e = self.entries.get(id(param))
        g = param.grad
        if e is None or g is None or e[2] == 0:
            return None
        if g.data_ptr() != e[2] or g._version != e[3] or g.numel() != e[4] or g.dtype != torch.float32 or not g.is_contiguous():
            return None
        return e[0], e[1]


def _wgrad_fused_norm(launch, M, N, dtype, param):
    """Run a weight-gradient launch (`launch(partial)` -> fp32 tensor) with its sum-of-squares partials handed to norm_sink."""
    part = norm_sink.slot(param, ops.gemm_num_partials(M, N, dtype))
    res = launch(part)
    if part is not None:
        norm_sink.commit(param, res)
    return res


class EmbedRowsFn(torch.autograd.Function):
    """rows = weight[ids] with the gradient routed to `embed_sink` instead of to the parameter's autograd leaf (the weight enters detached;
    the 0-d `anchor` only makes the output differentiable so that backward runs)."""

    @staticmethod
    def forward(ctx, anchor, ids, weight_data, holder):
        ctx.ids, ctx.holder = ids, holder
        holder[0].expect(ids.numel())      # the row count of the gradient this lookup will hand to the sink: known now, needed across ranks before apply()
        return F.embedding(ids, weight_data)

    @staticmethod
    def backward(ctx, dout):
        sink, param = ctx.holder
        sink.add(param, ctx.ids, dout)
        return None, None, None, None


def embedding_rows(ids: torch.Tensor, weight: torch.nn.Parameter) -> torch.Tensor:
    """F.embedding(ids, weight); with a live `embed_sink` (otter_amd's own TrainStep) the lookup's gradient is kept as (ids, rows) and
    applied after backward -- see train.SparseEmbedSink.  Plain autograd (the reference's loop through the shim) is untouched."""
    sink = embed_sink
    if sink is not None and weight.requires_grad and torch.is_grad_enabled():
        return EmbedRowsFn.apply(sink.anchor(weight.device), ids, weight.detach(), (sink, weight))
    return F.embedding(ids, weight)


def _wgrad(dyT: torch.Tensor, xT: torch.Tensor, gate=None, param=None):
    """dW[out,in] = (s *) dy^T . x   from the two transposed, zero-padded operands (fp32 result).  With a grad sink and
    `param`, the result lands in the sink's buffer and None is returned (autograd then has nothing to accumulate)."""
    out = grad_sink.take(param) if (grad_sink is not None and param is not None) else None
    if out is not None:
        ops.gemm_nt(dyT, xT, out=out, kind=EPI_STORE, gate=gate)
        grad_sink.ready(param)
        return None
    if norm_sink is not None and param is not None:
        return _wgrad_fused_norm(lambda part: ops.gemm_nt(dyT, xT, out_dtype=torch.float32, kind=EPI_STORE, gate=gate, partial=part),
                                 dyT.shape[0], xT.shape[0], dyT.dtype, param)
    return ops.gemm_nt(dyT, xT, out_dtype=torch.float32, kind=EPI_STORE, gate=gate)


class _SideStream:
    """A second HIP stream per device for the HALF-CHIP launches of the fusion modules (round 4).  The skinny projections of a gated block
    (to_q, dO, dWo, dWq: 4096 x 512 x 4096 in some order) are 128 workgroups of one workgroup per CU: half of the MI355X's 256 CUs idle for
    ~40 us each.  Two of them that do not depend on each other are issued on two streams and share the chip.  Protocol (all inside one
    autograd Function call): fork() makes the side stream wait for everything issued so far on the current stream; work is launched inside
    `with side.ctx():` (ops.* read torch's current stream); join() makes the current stream wait for the side stream.  Tensors handed
    from one stream to the other are kept alive by the caller until after join(); outputs consumed on the main stream are allocated on
    the main stream BEFORE the fork.  OTTER_NO_SIDE_STREAM=1: everything on one stream (A/B switch)."""

    _streams = {}

    def __init__(self, device):
        self.enabled = device.type == "cuda" and os.environ.get("OTTER_NO_SIDE_STREAM") != "1"
        self.forked = False
        if self.enabled:
            key = device.index if device.index is not None else torch.cuda.current_device()
            st = _SideStream._streams.get(key)
            if st is None:
                st = _SideStream._streams[key] = torch.cuda.Stream(device=device)
            self.side = st

    def fork(self):
        if self.enabled:
            self.side.wait_stream(torch.cuda.current_stream())
            self.forked = True
        return self

    def ctx(self):
        import contextlib

        return torch.cuda.stream(self.side) if (self.enabled and self.forked) else contextlib.nullcontext()

    def join(self):
        if self.enabled and self.forked:
            torch.cuda.current_stream().wait_stream(self.side)
            self.forked = False


def _wgrad_rows(dy_rows: torch.Tensor, x_rows: torch.Tensor, gate=None, param=None):
    """dW[out,in] = (s *) dy^T . x from the operands AS THEY LIE in HBM -- dy [rows, out], x [rows, in], compute dtype -- through the
    K-major GEMM (csrc/gemm.hip, transpose reads inside the kernel) when the shape qualifies; otherwise the operands are transposed
    first (otter_transpose) and the K-contiguous kernels run.  Same sink protocol as _wgrad."""
    rows, n_out = dy_rows.shape
    n_in = x_rows.shape[1]
    if (os.environ.get("OTTER_NO_KMAJOR") != "1"
            and ops.gemm_kmajor_supported(n_out, n_in, rows, dy_rows.stride(0), x_rows.stride(0), True, True, dy_rows.dtype)):
        out = grad_sink.take(param) if (grad_sink is not None and param is not None) else None
        if out is not None:
            ops.gemm(dy_rows, x_rows, True, True, out=out, kind=EPI_STORE, gate=gate)
            grad_sink.ready(param)
            return None
        if norm_sink is not None and param is not None:
            return _wgrad_fused_norm(lambda part: ops.gemm(dy_rows, x_rows, True, True, out_dtype=torch.float32, kind=EPI_STORE, gate=gate,
                                                           partial=part), n_out, n_in, dy_rows.dtype, param)
        return ops.gemm(dy_rows, x_rows, True, True, out_dtype=torch.float32, kind=EPI_STORE, gate=gate)
    cd = dy_rows.dtype
    return _wgrad(ops.transpose(dy_rows, cd), ops.transpose(x_rows, cd), gate=gate, param=param)


def _dgrad(dy_rows: torch.Tensor, W: torch.Tensor, cd, **epi):
    """dx = epilogue(dy . W) for y = x W^T with W stored [out, in]: W is the K-major B operand of the product, read in place (no
    transposed shadow of the weight: for the gated blocks' FFN matrices that is 2 x 128 MB per block rebuilt every optimizer step);
    small / irregular shapes go through the transposed shadow and the K-contiguous kernels."""
    rows, n_out = dy_rows.shape
    n_in = W.shape[1]
    Wc = shadows.w(W, cd)
    if (os.environ.get("OTTER_NO_KMAJOR") != "1"
            and ops.gemm_kmajor_supported(rows, n_in, n_out, dy_rows.stride(0), Wc.stride(0), False, True, dy_rows.dtype)):
        return ops.gemm(dy_rows, Wc, False, True, **epi)
    return ops.gemm_nt(dy_rows, shadows.wt(W, cd), **epi)


def _flat_gate(g):
    if g.dtype != torch.float32:
        raise RuntimeError("gate parameters must be fp32 (1-element) tensors")
    return g.detach()


# ----------------------------------------------------------------------------------------------------------------------
# LayerNorm (perceiver.norm, and the frozen MPT LPLayerNorm)
# ----------------------------------------------------------------------------------------------------------------------


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        need = x.requires_grad or (weight is not None and weight.requires_grad)
        y, mean, rstd = ops.layernorm_fwd(x2, weight.detach() if weight is not None else None,
                                          bias.detach() if bias is not None else None, out_dtype, eps, need_stats=need)
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.shp = shp
        ctx.xdtype = x.dtype
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        need_dw = weight is not None and weight.requires_grad
        dy2 = dy.reshape(-1, ctx.shp[-1]).contiguous()
        dx, dg, db = ops.layernorm_bwd(dy2, x2, weight.detach() if weight is not None else None, mean, rstd, ctx.xdtype,
                                       need_dw=need_dw, need_dbeta=ctx.has_bias, need_dx=ctx.needs_input_grad[0])
        return (dx.view(ctx.shp) if dx is not None else None, dg.to(weight.dtype) if dg is not None else None,
                db.to(weight.dtype) if (db is not None and ctx.has_bias) else None, None, None)


class AddLayerNormFn(torch.autograd.Function):
    """(xsum, y) = (x + delta, LN(x + delta)) in one pass over the residual stream (mpt/blocks.py:83-84)."""

    @staticmethod
    def forward(ctx, x, delta, weight, bias, eps, out_dtype):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        d2 = delta.reshape(-1, shp[-1]).contiguous()
        xsum, y, mean, rstd = ops.add_layernorm_fwd(x2, d2, weight.detach() if weight is not None else None,
                                                    bias.detach() if bias is not None else None, out_dtype, eps)
        ctx.save_for_backward(xsum, weight, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.shp = shp
        ctx.ddtype = delta.dtype
        return xsum.view(shp), y.view(shp)

    @staticmethod
    def backward(ctx, d_xsum, dy):
        xsum, weight, mean, rstd = ctx.saved_tensors
        need_dw = weight is not None and weight.requires_grad
        D = ctx.shp[-1]
        dres = d_xsum.reshape(-1, D).contiguous() if d_xsum is not None else None
        # the gradient of the (bf16) branch output is the same tensor in its dtype: written by the same pass
        fused = ctx.needs_input_grad[1] and ctx.ddtype == torch.bfloat16 and xsum.dtype != torch.bfloat16
        ddelta = torch.empty(xsum.shape, dtype=torch.bfloat16, device=xsum.device) if fused else None
        dx, dg, db = ops.layernorm_bwd(dy.reshape(-1, D).contiguous(), xsum, weight.detach() if weight is not None else None, mean,
                                       rstd, xsum.dtype, dres=dres, need_dw=need_dw, need_dbeta=ctx.has_bias, dx_bf16=ddelta)
        if not fused:
            ddelta = ops.cast(dx, ctx.ddtype) if ctx.needs_input_grad[1] else None
        return (dx.view(ctx.shp), ddelta.view(ctx.shp) if ddelta is not None else None,
                dg.to(weight.dtype) if dg is not None else None, db.to(weight.dtype) if (db is not None and ctx.has_bias) else None,
                None, None)


def add_layer_norm(x, delta, weight, bias, eps=1e-5, out_dtype=None):
    return AddLayerNormFn.apply(x, delta, weight, bias, eps, out_dtype or x.dtype)


class ForkLayerNormFn(torch.autograd.Function):
    """(x, y) = (x, LN(x)) for a residual stream that is read by the norm AND carried on (mpt/blocks.py:77-84 `a = norm_1(x); ...; x = x + b`).
    As two autograd nodes the stream's gradient arrives twice and the engine adds the two [rows, D] fp32 tensors in a separate pass
    (29 us per decoder layer that follows a gated block at C2); here the carried-on gradient enters the LayerNorm backward as its
    residual term (`dres`): one pass.  The first output is the input itself (same storage)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, weight.detach() if weight is not None else None,
                                          bias.detach() if bias is not None else None, out_dtype, eps)
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.shp = shp
        return x.view(shp), y.view(shp)

    @staticmethod
    def backward(ctx, d_x, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        need_dw = weight is not None and weight.requires_grad
        D = ctx.shp[-1]
        dres = d_x.reshape(-1, D).contiguous() if d_x is not None else None
        if dres is not None and dres.dtype != x2.dtype:
            dres = dres.to(x2.dtype)
        dx, dg, db = ops.layernorm_bwd(dy.reshape(-1, D).contiguous(), x2, weight.detach() if weight is not None else None, mean, rstd,
                                       x2.dtype, dres=dres, need_dw=need_dw, need_dbeta=ctx.has_bias)
        return (dx.view(ctx.shp), dg.to(weight.dtype) if dg is not None else None,
                db.to(weight.dtype) if (db is not None and ctx.has_bias) else None, None, None)


def fork_layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    return ForkLayerNormFn.apply(x, weight, bias, eps, out_dtype or x.dtype)


class RMSNormFn(torch.autograd.Function):
    """LlamaRMSNorm (xformers_model/llama.py:95-112 / HF): y = w * (x * rsqrt(mean(x^2) + eps)).to(x.dtype), config C4."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        y, rstd = ops.rmsnorm_fwd(x2, weight.detach(), eps)
        ctx.save_for_backward(x2, weight, rstd)
        ctx.shp = shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, ctx.shp[-1]).contiguous()
        if dy2.dtype != x2.dtype:
            dy2 = ops.cast(dy2, x2.dtype)
        dx, dw = ops.rmsnorm_bwd(dy2, x2, weight.detach(), rstd)
        return dx.view(ctx.shp), (dw.to(weight.dtype) if weight.requires_grad else None), None


def rms_norm(x, weight, eps=1e-6):
    return RMSNormFn.apply(x, weight, eps)


class AddRMSNormFn(torch.autograd.Function):
    """LLaMA host (config C4): y = RMSNorm(x [+ delta]) with its own output dtype, the residual add fused into the same pass
    (xformers_model/llama.py:95-112 and the `residual + hidden_states` of :311-318).  Returns (xsum, y); xsum is x itself when
    there is no delta.  Backward: one pass producing d(x) (+ the incoming d(xsum)) and, for a bf16 branch, its bf16 copy."""

    @staticmethod
    def forward(ctx, x, delta, weight, eps, out_dtype):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        d2 = delta.reshape(-1, shp[-1]).contiguous() if delta is not None else None
        xsum, y, rstd = ops.add_rmsnorm_fwd(x2, d2, weight.detach(), out_dtype, eps)
        xs = xsum if xsum is not None else x2
        ctx.save_for_backward(xs, weight, rstd)
        ctx.shp = shp
        ctx.has_delta = delta is not None
        ctx.ddtype = delta.dtype if delta is not None else None
        if delta is None:
            return y.view(shp)          # (an input must not be handed back as an output: no xsum without a delta)
        return xsum.view(shp), y.view(shp)

    @staticmethod
    def backward(ctx, *grads):
        d_xsum, dy = grads if ctx.has_delta else (None, grads[0])
        xs, weight, rstd = ctx.saved_tensors
        D = ctx.shp[-1]
        need_dw = weight.requires_grad
        dres = d_xsum.reshape(-1, D).contiguous() if d_xsum is not None else None
        if dres is not None and dres.dtype != xs.dtype:
            dres = dres.to(xs.dtype)
        dy2 = dy.reshape(-1, D).contiguous()
        fused = ctx.has_delta and ctx.needs_input_grad[1] and ctx.ddtype == torch.bfloat16 and xs.dtype != torch.bfloat16
        ddelta = torch.empty(xs.shape, dtype=torch.bfloat16, device=xs.device) if fused else None
        dx, dw = ops.rmsnorm_bwd_ex(dy2, xs, weight.detach(), rstd, xs.dtype, dres=dres, need_dw=need_dw, dx_bf16=ddelta)
        if ctx.has_delta and ctx.needs_input_grad[1] and not fused:
            ddelta = dx if ctx.ddtype == dx.dtype else ops.cast(dx, ctx.ddtype)
        return (dx.view(ctx.shp), ddelta.view(ctx.shp) if ddelta is not None else None,
                dw.to(weight.dtype) if dw is not None else None, None, None)


def add_rms_norm(x, delta, weight, eps=1e-6, out_dtype=None):
    """y (delta is None) or (x + delta, y)."""
    return AddRMSNormFn.apply(x, delta, weight, eps, out_dtype or x.dtype)


class GeluFn(torch.autograd.Function):
    """Exact-erf GELU of the decoder MLP (mpt/blocks.py:37-49) on csrc/elementwise.hip's one-pass kernels (forward 2 x, backward 3 x the
    activation bytes at ~6 TB/s; replaces torch's GeluCUDAKernelImpl / GeluBackwardCUDAKernelImpl: 72 + 89 -> 40 + 60 us per layer at C2)."""

    @staticmethod
    def forward(ctx, u):
        u = u.contiguous()
        ctx.save_for_backward(u)
        return ops.gelu_fwd(u)

    @staticmethod
    def backward(ctx, dh):
        (u,) = ctx.saved_tensors
        dh = dh.contiguous() if dh.dtype == u.dtype else dh.to(u.dtype).contiguous()
        return ops.gelu_bwd(u, dh)


def gelu(u):
    """nn.GELU() (approximate='none'): HIP kernels for GPU bf16 / f32 tensors with numel % 8 == 0, torch otherwise (CPU parity mode)."""
    if u.is_cuda and u.dtype in (torch.bfloat16, torch.float32) and u.numel() % 8 == 0 and u.numel() > 0:
        return GeluFn.apply(u)
    return torch.nn.functional.gelu(u)


class FrozenMLPFn(torch.autograd.Function):
    """The frozen decoder MLP (mpt/blocks.py:37-49: down_proj(gelu(up_proj(x))), no biases, frozen weights) on csrc/gemm.hip with the
    fusions a library GEMM cannot give: GELU in the up-projection's tail (the pre-activation u is stored beside
    it for the backward; no separate GELU pass, no 128 MB round trip), GELU' in the tail of down_proj's input-gradient GEMM, and both
    input-gradient products read the weights AS STORED through the K-major kernel (no transposed copies: 8.6 GB for MPT-7B).  Frozen
    weights: there is no weight gradient, so h is not kept.  x2 [rows, D], weights in the compute dtype (bf16)."""

    @staticmethod
    def forward(ctx, x2, Wu, Wd, Wu_t=None, Wd_t=None):
        need = ctx.needs_input_grad[0]
        u = torch.empty((x2.shape[0], Wu.shape[0]), dtype=x2.dtype, device=x2.device) if need else None
        h = ops.gemm_nt(x2, Wu, kind=EPI_GELU, C2=u)
        y = ops.gemm_nt(h, Wd)
        if need:
            ctx.have_t = Wu_t is not None and Wd_t is not None
            if ctx.have_t:
                ctx.save_for_backward(u, Wu_t, Wd_t)
            else:
                ctx.save_for_backward(u, Wu, Wd)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, Wu, Wd = ctx.saved_tensors
        dy = dy.contiguous() if dy.dtype == u.dtype else dy.to(u.dtype).contiguous()
        rows = dy.shape[0]
        if ctx.have_t:   # stored transposed copies (mode "1t"): every operand K-contiguous
            du = ops.gemm_nt(dy, Wd, kind=EPI_GATE_BWD, aux=u, aux_gelu=True)
            return ops.gemm_nt(du, Wu), None, None, None, None
        if ops.gemm_kmajor_supported(rows, Wd.shape[1], Wd.shape[0], dy.stride(0), Wd.stride(0), False, True, dy.dtype):
            du = ops.gemm(dy, Wd, False, True, kind=EPI_GATE_BWD, aux=u, aux_gelu=True)
        else:
            du = ops.gemm_nt(dy, ops.transpose(Wd, Wd.dtype), kind=EPI_GATE_BWD, aux=u, aux_gelu=True)
        if ops.gemm_kmajor_supported(rows, Wu.shape[1], Wu.shape[0], du.stride(0), Wu.stride(0), False, True, du.dtype):
            dx = ops.gemm(du, Wu, False, True)
        else:
            dx = ops.gemm_nt(du, ops.transpose(Wu, Wu.dtype))
        return dx, None, None, None, None


def mlp_stash_dgelu() -> bool:
    """OTTER_MLP_STASH_DGELU=1 (round 6c, opt-in A/B switch): the frozen MLP keeps GELU'(u) instead of u between forward and backward."""
    return os.environ.get("OTTER_MLP_STASH_DGELU", "0") == "1"


class FrozenMLPFusedLegsFn(torch.autograd.Function):
    """The frozen decoder MLP with ONLY its two fusable products on csrc/gemm.hip -- up_proj with GELU in the tail (u kept beside it) and
    down_proj's input gradient with GELU' in the tail, read against the weight as stored (K-major kernel: no transposed copy of down_proj) --
    and the two plain products (down_proj forward, up_proj input gradient against its transposed copy) on hipBLASLt.  Per-shape A/B at C2
    (tools/decoder_gemm_ab.py, profiles/r04_decoder_gemm_ab.txt): fused up 396.6 us vs library + gelu_fwd 398.8, fused down-dgrad 414.4 vs
    library + gelu_bwd 419.7; the plain products are 2-7 % faster on the library."""

    @staticmethod
    def forward(ctx, x2, Wu, Wd, Wu_t, Wd_t=None):
        need = ctx.needs_input_grad[0]
        u = torch.empty((x2.shape[0], Wu.shape[0]), dtype=x2.dtype, device=x2.device) if need else None
        # derivative stash (round 6c): with frozen weights the backward needs GELU'(u) and nothing else of u, so the forward tail -- bound by its
        # two outputs' stores, with arithmetic to spare -- writes g = GELU'(u) in u's place and the backward tail only multiplies
        ctx.stash = need and mlp_stash_dgelu()
        h = ops.gemm_nt(x2, Wu, kind=EPI_GELU, C2=u, aux_gelu="stash" if ctx.stash else False)
        y = torch.nn.functional.linear(h, Wd)
        if need:
            ctx.have_wdt = Wd_t is not None
            ctx.save_for_backward(u, Wd_t if Wd_t is not None else Wd, Wu_t)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, Wd, Wu_t = ctx.saved_tensors
        dy = dy.contiguous() if dy.dtype == u.dtype else dy.to(u.dtype).contiguous()
        act = "stash" if ctx.stash else True
        if ctx.have_wdt:   # Wd here is the stored transposed copy [in, out]: both operands K-contiguous (the cross-tile form of variant 26)
            du = ops.gemm_nt(dy, Wd, kind=EPI_GATE_BWD, aux=u, aux_gelu=act)
        elif ops.gemm_kmajor_supported(dy.shape[0], Wd.shape[1], Wd.shape[0], dy.stride(0), Wd.stride(0), False, True, dy.dtype):
            du = ops.gemm(dy, Wd, False, True, kind=EPI_GATE_BWD, aux=u, aux_gelu=act)
        else:
            du = ops.gemm_nt(dy, ops.transpose(Wd, Wd.dtype), kind=EPI_GATE_BWD, aux=u, aux_gelu=act)
        return torch.nn.functional.linear(du, Wu_t), None, None, None, None


def frozen_mlp_fused_legs(x, Wu, Wd, Wu_t, Wd_t=None):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    return FrozenMLPFusedLegsFn.apply(x2 if x2.is_contiguous() else x2.contiguous(), Wu, Wd, Wu_t, Wd_t).view(shp[:-1] + (Wd.shape[0],))


def frozen_mlp(x, Wu, Wd, Wu_t=None, Wd_t=None):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    return FrozenMLPFn.apply(x2 if x2.is_contiguous() else x2.contiguous(), Wu, Wd, Wu_t, Wd_t).view(shp[:-1] + (Wd.shape[0],))


class FrozenLinearOwnFn(torch.autograd.Function):
    """y = x W^T for a frozen bias-free Linear on csrc/gemm.hip; dx = dy W reads W as stored (K-major B operand)."""

    @staticmethod
    def forward(ctx, x2, W, Wt=None):
        ctx.have_t = Wt is not None
        ctx.save_for_backward(Wt if Wt is not None else W)
        return ops.gemm_nt(x2, W)

    @staticmethod
    def backward(ctx, dy):
        (W,) = ctx.saved_tensors
        dy = dy.contiguous() if dy.dtype == W.dtype else dy.to(W.dtype).contiguous()
        if ctx.have_t:   # stored transposed copy (mode "1t")
            return ops.gemm_nt(dy, W), None, None
        if ops.gemm_kmajor_supported(dy.shape[0], W.shape[1], W.shape[0], dy.stride(0), W.stride(0), False, True, dy.dtype):
            return ops.gemm(dy, W, False, True), None, None
        return ops.gemm_nt(dy, ops.transpose(W, W.dtype)), None, None


def frozen_linear_own(x, W, Wt=None):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    return FrozenLinearOwnFn.apply(x2 if x2.is_contiguous() else x2.contiguous(), W, Wt).view(shp[:-1] + (W.shape[0],))


class SwiGLUFn(torch.autograd.Function):
    """h = silu(gate) * up on the fused [.., 2*I] gate|up projection output (xformers_model/llama.py:216-223), bf16."""

    @staticmethod
    def forward(ctx, gu):
        shp = gu.shape
        gu2 = gu.reshape(-1, shp[-1]).contiguous()
        ctx.save_for_backward(gu2)
        ctx.shp = shp
        return ops.swiglu_fwd(gu2).view(shp[:-1] + (shp[-1] // 2,))

    @staticmethod
    def backward(ctx, dh):
        (gu2,) = ctx.saved_tensors
        dh2 = dh.reshape(-1, ctx.shp[-1] // 2)
        dh2 = dh2.contiguous() if dh2.dtype == torch.bfloat16 else dh2.to(torch.bfloat16).contiguous()
        return ops.swiglu_bwd(gu2, dh2).view(ctx.shp)


def swiglu(gu):
    return SwiGLUFn.apply(gu)


class RopeFlashAttentionFn(torch.autograd.Function):
    """LLaMA self-attention core on the fused q|k|v projection output [B,S,3*H*128] (bf16): RoPE on the q and k heads
    (xformers_model/llama.py:158-166) written to a packed [B,S,2,H,128] buffer by one strided pass, then the causal /
    key-padded flash attention of csrc/flash.hip (no ALiBi) with v read in place from the projection buffer.  Backward:
    dq / dk / dv land in the three slices of ONE [B,S,3*H*128] buffer and the inverse rotation runs in place on its q|k part --
    the buffer is then the operand of the (frozen) projection's dgrad GEMM.  Outputs: ctx [B,S,H*128], rotated k and v views
    ([B,S,H,128], for the KV cache; not differentiable)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, key_valid, n_heads, scale):
        B, S, D3 = qkv.shape
        H, d = n_heads, 128
        qkv = qkv.contiguous()
        qk = torch.empty((B, S, 2, H, d), dtype=torch.bfloat16, device=qkv.device)
        ops.rope_strided(qkv, qk, cos, sin, B * S, S, 2 * H, d, 3 * H * d, 2 * H * d)
        v5 = qkv.view(B, S, 3, H, d)
        o, lse = ops.flash_attn_fwd(qk[:, :, 0], qk[:, :, 1], v5[:, :, 2], None, key_valid, scale, True)
        ctx.save_for_backward(qk, qkv, o, lse, cos, sin, key_valid)
        ctx.cfg = (H, scale)
        k_rot, v = qk[:, :, 1], v5[:, :, 2]
        ctx.mark_non_differentiable(k_rot, v)
        return o.view(B, S, H * d), k_rot, v

    @staticmethod
    def backward(ctx, dout, _dk, _dv):
        qk, qkv, o, lse, cos, sin, key_valid = ctx.saved_tensors
        H, scale = ctx.cfg
        B, S, _ = qkv.shape
        d = 128
        v5 = qkv.view(B, S, 3, H, d)
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, S, 3, H, d)
        dout = dout.to(torch.bfloat16).contiguous().view(B, S, H, d)
        ops.flash_attn_bwd(qk[:, :, 0], qk[:, :, 1], v5[:, :, 2], o, lse, dout, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2], None, key_valid,
                           scale, True)
        ops.rope_strided(dqkv, dqkv, cos, sin, B * S, S, 2 * H, d, 3 * H * d, 3 * H * d, inverse=True)
        return dqkv, None, None, None, None, None


def rope_flash_attention(qkv, cos, sin, key_valid, n_heads, scale, want_kv=False):
    ctx, k_rot, v = RopeFlashAttentionFn.apply(qkv, cos.contiguous(), sin.contiguous(), key_valid, n_heads, scale)
    return ctx, (k_rot if want_kv else None), (v if want_kv else None)


def layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    return LayerNormFn.apply(x, weight, bias, eps, out_dtype or x.dtype)


# ----------------------------------------------------------------------------------------------------------------------
# small building blocks (stand-alone OtterMaskedCrossAttention, latent broadcast, dtype casts)
# ----------------------------------------------------------------------------------------------------------------------


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.cast(dy.contiguous(), ctx.src), None


class ExpandLatentsFn(torch.autograd.Function):
    """latents [n, D] -> [G, n, D] (modeling_otter.py:232); backward = column sum over the G copies (HIP colsum)."""

    @staticmethod
    def forward(ctx, latents, G):
        ctx.G = G
        ctx.pdtype = latents.dtype
        return latents.detach().unsqueeze(0).expand(G, -1, -1).contiguous()  # memory replication only

    @staticmethod
    def backward(ctx, dy):
        G, n, D = dy.shape
        g = ops.colsum(dy.contiguous().view(G, n * D), None, G).view(n, D)
        return g.to(ctx.pdtype), None


class LinearFn(torch.autograd.Function):
    """y = x W^T for a bias-free nn.Linear; x [rows, in] in the compute dtype."""

    @staticmethod
    def forward(ctx, x2, W):
        cd = x2.dtype
        ctx.save_for_backward(x2, W)
        return ops.gemm_nt(x2, shadows.w(W, cd))

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        cd = x2.dtype
        dy = dy.contiguous()
        dx = _dgrad(dy, W, cd) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            dW = _wgrad_rows(dy, x2, param=W)
            dW = dW.to(W.dtype) if dW is not None else None
        return dx, dW


class AttnCoreFn(torch.autograd.Function):
    """o = softmax(mask(scale q k^T)) v on [B,Tq,H*64] / [B,M,2*H*64] buffers (attention core only)."""

    @staticmethod
    def forward(ctx, q3, kv3, heads, tt, n_per_media, mask_mode):
        inner = q3.shape[-1]
        scale = HEAD_DIM ** -0.5
        o, lse = ops.attn_fwd(q3, kv3[..., :inner], kv3[..., inner:], heads, tt, n_per_media, mask_mode, scale)
        ctx.save_for_backward(q3, kv3, o, lse, tt)
        ctx.meta = (heads, n_per_media, mask_mode, scale, inner)
        return o

    @staticmethod
    def backward(ctx, do):
        q3, kv3, o, lse, tt = ctx.saved_tensors
        heads, n_per_media, mask_mode, scale, inner = ctx.meta
        dq, dkv = ops.attn_bwd(q3, kv3[..., :inner], kv3[..., inner:], o, do.contiguous(), lse, heads, tt, n_per_media, mask_mode,
                               scale)
        return dq, dkv, None, None, None, None


class FlashSelfAttentionFn(torch.autograd.Function):
    """Causal / ALiBi / key-padding self-attention of the frozen decoder host on the fused Wqkv output
    (mpt/attention.py:22-84; bias :447-464): qkv [B,S,3*H*128] bf16 -> ctx [B,S,H*128] bf16.  q, k, v and the three
    gradient slices are addressed in place inside qkv / dqkv (no chunk or cat copies)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, slopes, key_valid, scale, causal):
        B, S, D3 = qkv.shape
        qkv = qkv.contiguous()
        v5 = qkv.view(B, S, 3, n_heads, 128)
        o, lse = ops.flash_attn_fwd(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], slopes, key_valid, scale, causal)
        ctx.save_for_backward(qkv, o, lse, slopes, key_valid)
        ctx.cfg = (n_heads, scale, causal)
        return o.view(B, S, n_heads * 128)

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse, slopes, key_valid = ctx.saved_tensors
        n_heads, scale, causal = ctx.cfg
        B, S, _ = qkv.shape
        v5 = qkv.view(B, S, 3, n_heads, 128)
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, S, 3, n_heads, 128)
        dout = dout.to(torch.bfloat16).contiguous().view(B, S, n_heads, 128)
        ops.flash_attn_bwd(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, dout, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                           slopes, key_valid, scale, causal)
        return dqkv, None, None, None, None, None


def flash_self_attention(qkv, n_heads, slopes, key_valid, scale, causal=True):
    return FlashSelfAttentionFn.apply(qkv, n_heads, slopes, key_valid, scale, causal)


class CrossEntropyBf16Fn(torch.autograd.Function):
    """Mean token cross-entropy (ignore_index -100) on bf16 logits [rows, V] without materialising an fp32 copy
    (mpt/modeling_mpt.py:428-435 once the labels are rolled).  fp32 arithmetic on the bf16 values = what
    F.cross_entropy(logits.float(), labels) computes; the gradient is produced directly in bf16.  Any negative label is
    ignored (torch: only -100, others assert); a label >= V makes the loss NaN (torch: device assert); an all-ignored batch
    returns NaN like torch."""

    @staticmethod
    def forward(ctx, logits2d, labels):
        K_ = ops.K
        K_.require_cuda(logits2d, labels)
        rows, V = logits2d.shape
        if logits2d.dtype != torch.bfloat16 or logits2d.stride(1) != 1 or labels.dtype != torch.int64 or not labels.is_contiguous():
            raise K_.OtterHipError("cross_entropy: bf16 [rows, V] logits with unit column stride and contiguous int64 labels")
        lse = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
        nll = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
        K_.check(K_.lib().otter_cross_entropy_fwd(logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), lse.data_ptr(), nll.data_ptr(),
                                                  rows, V, K_.stream()), "cross_entropy_fwd")
        n_valid = (labels >= 0).sum().to(torch.float32)
        ctx.save_for_backward(logits2d, labels, lse, n_valid)
        # no clamp: a batch with every label ignored is 0 / 0 = NaN, exactly F.cross_entropy's mean over zero rows
        return nll.sum() / n_valid

    @staticmethod
    def backward(ctx, dloss):
        K_ = ops.K
        logits2d, labels, lse, n_valid = ctx.saved_tensors
        rows, V = logits2d.shape
        dlogits = torch.empty((rows, V), dtype=torch.bfloat16, device=logits2d.device)
        dl = dloss.to(torch.float32).contiguous()
        K_.check(K_.lib().otter_cross_entropy_bwd(logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), lse.data_ptr(), dl.data_ptr(),
                                                  n_valid.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), rows, V, K_.stream()),
                 "cross_entropy_bwd")
        return dlogits, None


def cross_entropy_bf16(logits2d, labels):
    return CrossEntropyBf16Fn.apply(logits2d, labels)


def masked_cross_attention(x, media, tt, mask_mode, heads, eps, norm_w, norm_b, Wq, Wkv, Wo):
    """OtterMaskedCrossAttention.forward (modeling_otter.py:262-340) used stand-alone: returns to_out(attn) in the compute
    dtype (bf16 under autocast, like the reference)."""
    B, T, D = x.shape
    _, T_img, n, Dv = media.shape
    inner = Wq.shape[0]
    if inner != heads * HEAD_DIM:
        raise RuntimeError(f"masked cross-attention: inner dim {inner} != heads*64 (dim_head must be 64)")
    cd = compute_dtype_for(x)
    xn = layer_norm(x, norm_w, norm_b, eps, cd).reshape(B * T, D)
    q = LinearFn.apply(xn, Wq)
    med = CastFn.apply(media.reshape(B * T_img * n, Dv).contiguous(), cd)
    kv = LinearFn.apply(med, Wkv)
    o = AttnCoreFn.apply(q.view(B, T, inner), kv.view(B, T_img * n, 2 * inner), heads, tt, n, mask_mode)
    return LinearFn.apply(o.reshape(B * T, inner), Wo).view(B, T, D)


# ----------------------------------------------------------------------------------------------------------------------
# OtterGatedCrossAttentionBlock  (otter/modeling_otter.py:262-340 + 373-395)
# ----------------------------------------------------------------------------------------------------------------------


class GatedCrossAttentionFn(torch.autograd.Function):
    """y = block(x, media).  Inputs: x [B,T,D]; media [B,T_img,n,Dv]; tt int32 [B,T] or None."""

    @staticmethod
    def forward(ctx, x, media, tt, mask_mode, heads, eps, norm_w, norm_b, Wq, Wkv, Wo, attn_gate, ffn_w, ffn_b, W1, W2,
                ff_gate, delta=None):
        """delta (otter_amd extension, round 4): the previous decoder layer's un-added FFN output (bf16 [B,T,D]); the block then
        starts from x + delta, the add fused into its first LayerNorm pass (mpt.py hands its residual adds to the NEXT norm; before
        round 4 a gated block forced a separate fp32 add + a bf16 cast of its gradient: 45 us per block at C2)."""
        B, T, D = x.shape
        _, T_img, n, Dv = media.shape
        inner = Wq.shape[0]
        if inner != heads * HEAD_DIM:
            raise RuntimeError(f"gated cross-attention: inner dim {inner} != heads*64 (dim_head must be 64)")
        cd = compute_dtype_for(x)
        rd = x.dtype
        N, M = B * T, T_img * n
        x2 = x.reshape(N, D).contiguous()
        scale = HEAD_DIM ** -0.5
        ga, gf = _flat_gate(attn_gate), _flat_gate(ff_gate)
        # --- masked cross attention ---
        if delta is not None:
            x2, xn, mean1, rstd1 = ops.add_layernorm_fwd(x2, delta.reshape(N, D).contiguous(), norm_w.detach(), norm_b.detach(), cd, eps)
        else:
            xn, mean1, rstd1 = ops.layernorm_fwd(x2, norm_w.detach(), norm_b.detach(), cd, eps)
        # the media projection (32 tiles) beside the query projection (128 tiles, half the chip): two streams
        med_in = media.reshape(B * M, Dv).contiguous()
        med = med_in if med_in.dtype == cd else torch.empty((B * M, Dv), dtype=cd, device=x.device)
        kv = torch.empty((B * M, Wkv.shape[0]), dtype=cd, device=x.device)          # [B*M, 2*inner]
        wkv_c, wq_c = shadows.w(Wkv, cd), shadows.w(Wq, cd)
        side = _SideStream(x.device).fork()
        with side.ctx():
            if med is not med_in:
                ops.K.check(ops.K.lib().otter_cast(med_in.data_ptr(), ops.K.dt(med_in), med.data_ptr(), ops.K.dt(med), med_in.numel(), ops.K.stream()), "cast")
            ops.gemm_nt(med, wkv_c, out=kv)
        q = ops.gemm_nt(xn, wq_c)
        side.join()
        kv3 = kv.view(B, M, 2 * inner)
        o, lse = ops.attn_fwd(q.view(B, T, inner), kv3[..., :inner], kv3[..., inner:], heads, tt, n, mask_mode, scale)
        o2 = o.view(N, inner)
        x1 = ops.gemm_nt(o2, shadows.w(Wo, cd), out_dtype=rd, kind=EPI_SCALE_RES, gate=ga, R=x2)   # attn*tanh(g)+x
        # --- gated feed-forward ---
        f, mean2, rstd2 = ops.layernorm_fwd(x1, ffn_w.detach(), ffn_b.detach(), cd, eps)
        need_grad = any(ctx.needs_input_grad)
        u = torch.empty((N, W1.shape[0]), dtype=cd, device=x.device) if need_grad else None
        h = ops.gemm_nt(f, shadows.w(W1, cd), kind=EPI_GELU, C2=u)
        y = ops.gemm_nt(h, shadows.w(W2, cd), out_dtype=rd, kind=EPI_SCALE_RES, gate=gf, R=x1)       # ff*tanh(g)+x1
        if need_grad:
            ctx.save_for_backward(x2, xn, mean1, rstd1, q, kv, o2, lse, x1, mean2, rstd2, f, u, h, med, tt, norm_w, Wq, Wkv,
                                  Wo, attn_gate, ffn_w, W1, W2, ff_gate)
            ctx.meta = (B, T, D, T_img, n, Dv, inner, heads, mask_mode, scale, cd, rd, media.dtype)
            ctx.delta_dtype = delta.dtype if delta is not None else None
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        (x2, xn, mean1, rstd1, q, kv, o2, lse, x1, mean2, rstd2, f, u, h, med, tt, norm_w, Wq, Wkv, Wo, attn_gate, ffn_w, W1,
         W2, ff_gate) = ctx.saved_tensors
        B, T, D, T_img, n, Dv, inner, heads, mask_mode, scale, cd, rd, media_dtype = ctx.meta
        N, M = B * T, T_img * n
        dev = dy.device
        ga, gf = _flat_gate(attn_gate), _flat_gate(ff_gate)
        dy2 = dy.reshape(N, D).contiguous()
        # ---- feed-forward branch:  y = (h W2^T) tanh(gf) + x1 ----
        # every product reads its operands as they lie (K-major GEMM): no transposes of dy / h / dU / f, no W^T shadows of W1 / W2
        dy_cd = dy2 if dy2.dtype == cd else ops.cast(dy2, cd)
        part = torch.empty(ops.gemm_num_partials(N, W1.shape[0], cd), dtype=torch.float32, device=dev)
        dU = _dgrad(dy_cd, W2, cd, kind=EPI_GATE_BWD, gate=gf, aux=u, aux_gelu=True, partial=part)
        d_ff_gate = ops.reduce_partials(part, gate=gf)
        dW2 = _wgrad_rows(dy_cd, h, gate=gf, param=W2)
        dW1 = _wgrad_rows(dU, f, param=W1)
        df = _dgrad(dU, W1, cd)
        dx1, dg2, db2 = ops.layernorm_bwd(df, x1, ffn_w.detach(), mean2, rstd2, rd, dres=dy2)
        # ---- attention branch:  x1 = (o Wo^T) tanh(ga) + x ----
        # Two streams (see _SideStream): the weight gradients of the skinny projections (128 workgroups each = half the chip) run beside
        # the chain that produces dx -- dWo beside dO + the attention backward, dWq / dWkv beside dxn + the LayerNorm backward.
        dx1T, dx1_cd = ops.transpose(dx1, cd, want_same=True)
        sink = grad_sink

        def wgrad_out(param):       # the sink's bucket view, or a fresh fp32 tensor allocated on THIS stream (consumed here after the join)
            out = sink.take(param) if sink is not None else None
            return (out, True) if out is not None else (torch.empty(param.shape, dtype=torch.float32, device=dev), False)

        dWo_out, dWo_sunk = wgrad_out(Wo)
        side = _SideStream(dev).fork()
        with side.ctx():
            o2T = ops.transpose(o2, cd)
            ops.gemm_nt(dx1T, o2T, out=dWo_out, kind=EPI_STORE, gate=ga)
        part2 = torch.empty(ops.gemm_num_partials(N, inner, cd), dtype=torch.float32, device=dev)
        dO = ops.gemm_nt(dx1_cd, shadows.wt(Wo, cd), kind=EPI_GATE_BWD, gate=ga, aux=o2, aux_gelu=False, partial=part2)
        d_attn_gate = ops.reduce_partials(part2, gate=ga)
        kv3 = kv.view(B, M, 2 * inner)
        dq, dkv = ops.attn_bwd(q.view(B, T, inner), kv3[..., :inner], kv3[..., inner:], o2.view(B, T, inner),
                               dO.view(B, T, inner), lse, heads, tt, n, mask_mode, scale)
        dq2, dkv2 = dq.view(N, inner), dkv.view(B * M, 2 * inner)
        side.join()
        if dWo_sunk:
            sink.ready(Wo)          # (on the main stream, after the join: the reducer orders its collective behind THIS stream)
        dWo = None if dWo_sunk else dWo_out
        dWq_out, dWq_sunk = wgrad_out(Wq)
        dWkv_out, dWkv_sunk = wgrad_out(Wkv)
        side.fork()
        with side.ctx():
            dq2T, xnT = ops.transpose(dq2, cd), ops.transpose(xn, cd)
            ops.gemm_nt(dq2T, xnT, out=dWq_out, kind=EPI_STORE)
            dkv2T, medT = ops.transpose(dkv2, cd), ops.transpose(med, cd)
            ops.gemm_nt(dkv2T, medT, out=dWkv_out, kind=EPI_STORE)
        dxn = ops.gemm_nt(dq2, shadows.wt(Wq, cd))
        dmedia = None
        if ctx.needs_input_grad[1]:
            dmedia = ops.gemm_nt(dkv2, shadows.wt(Wkv, cd), out_dtype=media_dtype).view(B, T_img, n, Dv)
        # the gradient of a deferred delta is dx in delta's dtype: written by the same LayerNorm-backward pass when that is bf16
        ddelta = None
        want_dd = ctx.delta_dtype is not None and ctx.needs_input_grad[17]
        fused_dd = want_dd and ctx.delta_dtype == torch.bfloat16 and rd == torch.float32
        if fused_dd:
            ddelta = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
        dx, dg1, db1 = ops.layernorm_bwd(dxn, x2, norm_w.detach(), mean1, rstd1, rd, dres=dx1, dx_bf16=ddelta)
        if want_dd and not fused_dd:
            ddelta = dx if ctx.delta_dtype == dx.dtype else ops.cast(dx, ctx.delta_dtype)
        side.join()
        for prm, sunk in ((Wq, dWq_sunk), (Wkv, dWkv_sunk)):
            if sunk:
                sink.ready(prm)
        dWq = None if dWq_sunk else dWq_out
        dWkv = None if dWkv_sunk else dWkv_out

        def pg(g, p):
            if g is None:  # already delivered through the grad sink
                return None
            return g.to(p.dtype) if g.dtype != p.dtype else g

        return (dx.view(B, T, D), dmedia, None, None, None, None, pg(dg1, norm_w), pg(db1, norm_w), pg(dWq, Wq), pg(dWkv, Wkv),
                pg(dWo, Wo), pg(d_attn_gate, attn_gate), pg(dg2, ffn_w), pg(db2, ffn_w), pg(dW1, W1), pg(dW2, W2),
                pg(d_ff_gate, ff_gate), ddelta.view(B, T, D) if ddelta is not None else None)


# ----------------------------------------------------------------------------------------------------------------------
# OtterPerceiverBlock  (otter/modeling_otter.py:151-184)
# ----------------------------------------------------------------------------------------------------------------------


class PerceiverBlockFn(torch.autograd.Function):
    """latents' = block(x, latents).  x [G,n1,D] media features (G = b*T), latents [G,n2,D]."""

    @staticmethod
    def forward(ctx, x, latents, heads, eps, nm_w, nm_b, nl_w, nl_b, Wq, Wkv, Wo, ff_w, ff_b, W1, W2):
        G, n1, D = x.shape
        n2 = latents.shape[1]
        inner = Wq.shape[0]
        if inner != heads * HEAD_DIM:
            raise RuntimeError(f"perceiver: inner dim {inner} != heads*64 (dim_head must be 64)")
        cd = compute_dtype_for(x)
        rd = latents.dtype
        nk = n1 + n2
        scale = HEAD_DIM ** -0.5
        x2 = x.reshape(G * n1, D).contiguous()
        l2 = latents.reshape(G * n2, D).contiguous()
        # norm_media(x) and norm_latents(latents) land directly in the [x ; latents] buffer `to_kv` reads (no torch.cat)
        kv_in = torch.empty((G * nk, D), dtype=cd, device=x.device)
        _, mean_m, rstd_m = ops.layernorm_fwd(x2, nm_w.detach(), nm_b.detach(), cd, eps, y=kv_in, ymap=RowMap(n1, nk, 0))
        ln = torch.empty((G * n2, D), dtype=cd, device=x.device)
        _, mean_l, rstd_l = ops.layernorm_fwd(l2, nl_w.detach(), nl_b.detach(), cd, eps, y=kv_in, ymap=RowMap(n2, nk, n1), y2=ln)
        # to_q (16 tiles at C2) beside to_kv (160 tiles): two streams, see _SideStream -- every GEMM of the resampler is a small grid
        q = torch.empty((G * n2, inner), dtype=cd, device=x.device)                # [G*n2, inner]
        wq_c, wkv_c = shadows.w(Wq, cd), shadows.w(Wkv, cd)
        side = _SideStream(x.device).fork()
        with side.ctx():
            ops.gemm_nt(ln, wq_c, out=q)
        kv = ops.gemm_nt(kv_in, wkv_c)                                             # [G*nk, 2*inner]
        side.join()
        kv3 = kv.view(G, nk, 2 * inner)
        o, lse = ops.attn_fwd(q.view(G, n2, inner), kv3[..., :inner], kv3[..., inner:], heads, None, 1, MASK_NONE, scale)
        o2 = o.view(G * n2, inner)
        out1 = ops.gemm_nt(o2, shadows.w(Wo, cd), out_dtype=rd, kind=EPI_SCALE_RES, R=l2)            # to_out + residual
        f, mean_f, rstd_f = ops.layernorm_fwd(out1, ff_w.detach(), ff_b.detach(), cd, eps)
        need_grad = any(ctx.needs_input_grad)
        u = torch.empty((G * n2, W1.shape[0]), dtype=cd, device=x.device) if need_grad else None
        h = ops.gemm_nt(f, shadows.w(W1, cd), kind=EPI_GELU, C2=u)
        y = ops.gemm_nt(h, shadows.w(W2, cd), out_dtype=rd, kind=EPI_SCALE_RES, R=out1)              # ff + residual
        if need_grad:
            ctx.save_for_backward(x2, l2, kv_in, ln, mean_m, rstd_m, mean_l, rstd_l, q, kv, o2, lse, out1, mean_f, rstd_f, f, u, h,
                                  nm_w, nl_w, Wq, Wkv, Wo, ff_w, W1, W2)
            ctx.meta = (G, n1, n2, D, inner, heads, scale, cd, rd, x.dtype)
        return y.view(G, n2, D)

    @staticmethod
    def backward(ctx, dy):
        (x2, l2, kv_in, ln, mean_m, rstd_m, mean_l, rstd_l, q, kv, o2, lse, out1, mean_f, rstd_f, f, u, h, nm_w, nl_w, Wq, Wkv, Wo,
         ff_w, W1, W2) = ctx.saved_tensors
        G, n1, n2, D, inner, heads, scale, cd, rd, xdtype = ctx.meta
        nk = n1 + n2
        dy2 = dy.reshape(G * n2, D).contiguous()
        # Round 4: the five weight gradients (and the operand transposes they need at these small-grid shapes) run on a SECOND STREAM beside
        # the chain that produces dx / dlatents -- every GEMM of the resampler is 16-160 workgroups on 256 CUs and 15-30 us of latency, so the
        # two streams share the chip instead of queueing (see _SideStream; the side stream re-synchronises with this one before each
        # product whose operands were produced here, and is joined before the function returns).
        dev = dy.device
        sink = grad_sink
        side = _SideStream(dev)
        sunk = []

        def wgrad_side(dy_rows, x_rows, param, fork=True):
            """dW[out,in] = dy_rows^T . x_rows on the side stream, into the sink's bucket view or a tensor allocated on THIS stream."""
            out = sink.take(param) if sink is not None else None
            if out is not None:
                sunk.append(param)
            else:
                out = torch.empty(param.shape, dtype=torch.float32, device=dev)
            if fork:
                side.fork()
            with side.ctx():
                rows, n_out = dy_rows.shape
                if (os.environ.get("OTTER_NO_KMAJOR") != "1"
                        and ops.gemm_kmajor_supported(n_out, x_rows.shape[1], rows, dy_rows.stride(0), x_rows.stride(0), True, True, dy_rows.dtype)):
                    ops.gemm(dy_rows, x_rows, True, True, out=out, kind=EPI_STORE)
                else:
                    ops.gemm_nt(ops.transpose(dy_rows, cd), ops.transpose(x_rows, cd), out=out, kind=EPI_STORE)
            return None if (sunk and sunk[-1] is param) else out

        # feed-forward: y = gelu(f W1^T) W2^T + out1
        dy_cd = dy2 if dy2.dtype == cd else ops.cast(dy2, cd)
        dW2 = wgrad_side(dy_cd, h, W2)
        dU = _dgrad(dy_cd, W2, cd, kind=EPI_GATE_BWD, aux=u, aux_gelu=True)
        dW1 = wgrad_side(dU, f, W1)
        df = _dgrad(dU, W1, cd)
        dout1, dgf, dbf = ops.layernorm_bwd(df, out1, ff_w.detach(), mean_f, rstd_f, rd, dres=dy2)
        # attention: out1 = o Wo^T + latents
        d1_cd = dout1 if dout1.dtype == cd else ops.cast(dout1, cd)
        dWo = wgrad_side(d1_cd, o2, Wo)
        dO = ops.gemm_nt(d1_cd, shadows.wt(Wo, cd))
        kv3 = kv.view(G, nk, 2 * inner)
        dq, dkv = ops.attn_bwd(q.view(G, n2, inner), kv3[..., :inner], kv3[..., inner:], o2.view(G, n2, inner),
                               dO.view(G, n2, inner), lse, heads, None, 1, MASK_NONE, scale)
        dq2, dkv2 = dq.view(G * n2, inner), dkv.view(G * nk, 2 * inner)
        dWq = wgrad_side(dq2, ln, Wq)
        dWkv = wgrad_side(dkv2, kv_in, Wkv, fork=False)
        dkv_in = ops.gemm_nt(dkv2, shadows.wt(Wkv, cd))                            # [G*nk, D] grads of [xn ; ln]
        # d(ln) = dq Wq (through to_q) + the latent rows of dkv_in (through to_kv)
        dln = ops.gemm_nt(dq2, shadows.wt(Wq, cd))
        ops.add_rows_(dln, dkv_in, RowMap(n2, nk, n1))
        dl, dgl, dbl = ops.layernorm_bwd(dln, l2, nl_w.detach(), mean_l, rstd_l, rd, dres=dout1)
        dx = None
        need_dx = ctx.needs_input_grad[0]
        # norm_media backward reads its dy rows out of dkv_in through the same row map the forward wrote with
        dxm, dgm, dbm = ops.layernorm_bwd(dkv_in, x2, nm_w.detach(), mean_m, rstd_m, xdtype, dymap=RowMap(n1, nk, 0),
                                          need_dx=need_dx)
        if need_dx:
            dx = dxm.view(G, n1, D)
        side.join()
        for prm in sunk:             # on THIS stream, after the join: the reducer orders its collective behind every writer
            sink.ready(prm)

        def pg(g, p):
            if g is None:  # already delivered through the grad sink
                return None
            return g.to(p.dtype) if g.dtype != p.dtype else g

        return (dx, dl.view(G, n2, D), None, None, pg(dgm, nm_w), pg(dbm, nm_w), pg(dgl, nl_w), pg(dbl, nl_w), pg(dWq, Wq),
                pg(dWkv, Wkv), pg(dWo, Wo), pg(dgf, ff_w), pg(dbf, ff_w), pg(dW1, W1), pg(dW2, W2))


# ----------------------------------------------------------------------------------------------------------------------
# OtterHD / Fuyu path (config C5): Persimmon attention, squared ReLU, patch scatter
# ----------------------------------------------------------------------------------------------------------------------


class SqReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.sqrelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous()
        return ops.sqrelu_bwd(x, dy)


def sqrelu(x):
    return SqReLUFn.apply(x)


def _persimmon_pad128(H: int) -> bool:
    """Round 2's layout (heads zero-padded to 128 for the 128-wide flash kernels): odd head counts, or OTTER_FUYU_PAD128=1 (A/B runs)."""
    return H % 2 == 1 or os.environ.get("OTTER_FUYU_PAD128", "0") == "1"


class PersimmonAttentionFn(torch.autograd.Function):
    """Persimmon self-attention core (fuyu/modeling_persimmon.py:262-312) on the per-head interleaved projection output
    qkv [B,S,H*3*64] (bf16): q/k LayerNorm + partial rotary in one pass (otter_qk_norm_rope_fwd) into compact [B,S,H,64] q / k, causal
    flash attention on 64-wide heads (csrc/flash.hip, two heads per workgroup) that reads v IN PLACE from qkv and writes ctx directly
    as [B,S,H*64]; the backward writes dv straight into the v slots of dqkv and the LayerNorm / rotary backward fills the q / k slots.
    (Round 2 zero-padded q / k / v / dO to 128 columns and gathered ctx from a padded output: `_persimmon_pad128`.)"""

    @staticmethod
    def forward(ctx, qkv, gq, bq, gk, bk, cos, sin, H, rot, eps, scale):
        B, S, _ = qkv.shape
        qkv = qkv.contiguous()
        gqf, bqf, gkf, bkf = (t.detach().float().contiguous() for t in (gq, bq, gk, bk))
        pad = _persimmon_pad128(H)
        q, k, v, stats = ops.qk_norm_rope_fwd(qkv, gqf, bqf, gkf, bkf, cos, sin, H, rot, eps, width=128 if pad else 64, copy_v=pad)
        o, lse = ops.flash_attn_fwd(q, k, v, None, None, scale, True)            # [B,S,H,128 | 64]
        ctx.save_for_backward(qkv, stats, q, k, v, o, lse, gqf, gkf, cos, sin)
        ctx.cfg = (H, rot, scale, gq.dtype, pad)
        k_c, v_c = k[..., :64], v[..., :64]          # normalised + rotated keys and the values, for a KV cache (not differentiable)
        ctx.mark_non_differentiable(k_c, v_c)
        return (o[..., :64].reshape(B, S, H * 64) if pad else o.view(B, S, H * 64)), k_c, v_c

    @staticmethod
    def backward(ctx, dctx, _dk, _dv):
        qkv, stats, q, k, v, o, lse, gqf, gkf, cos, sin = ctx.saved_tensors
        H, rot, scale, pdt, pad = ctx.cfg
        B, S, _ = qkv.shape
        if pad:
            do = torch.zeros((B, S, H, 128), dtype=torch.bfloat16, device=qkv.device)
            do[..., :64] = dctx.reshape(B, S, H, 64)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ops.flash_attn_bwd(q, k, v, o, lse, do, dq, dk, dv, None, None, scale, True)
            dqkv, dgq, dbq, dgk, dbk = ops.qk_norm_rope_bwd(dq, dk, dv, qkv, stats, gqf, gkf, cos, sin, H, rot)
        else:
            do = dctx.reshape(B, S, H, 64)
            do = do.contiguous() if do.dtype == torch.bfloat16 else do.to(torch.bfloat16).contiguous()
            dqkv = torch.empty_like(qkv)
            dq, dk = torch.empty_like(q), torch.empty_like(k)
            dv = dqkv.view(B, S, H, 3, 64)[:, :, :, 2]
            ops.flash_attn_bwd(q, k, v, o, lse, do, dq, dk, dv, None, None, scale, True)
            dqkv, dgq, dbq, dgk, dbk = ops.qk_norm_rope_bwd(dq, dk, None, qkv, stats, gqf, gkf, cos, sin, H, rot, dqkv=dqkv)
        return dqkv, dgq.to(pdt), dbq.to(pdt), dgk.to(pdt), dbk.to(pdt), None, None, None, None, None, None


def persimmon_attention(qkv, q_ln, k_ln, cos, sin, n_heads, rot, scale, want_kv=False):
    """ctx [B,S,H*64], or (ctx, k [B,H,S,64], v [B,H,S,64]) with want_kv (the layout of the plain path's cache)."""
    ctx, k_c, v_c = PersimmonAttentionFn.apply(qkv, q_ln.weight, q_ln.bias, k_ln.weight, k_ln.bias, cos.contiguous(), sin.contiguous(), n_heads,
                                               rot, q_ln.eps, scale)
    if want_kv:
        return ctx, k_c.transpose(1, 2), v_c.transpose(1, 2)
    return ctx


class ScatterPatchRowsFn(torch.autograd.Function):
    """FuyuForCausalLM.gather_continuous_embeddings (fuyu/modeling_fuyu.py:44-77) as one HIP pass; the backward routes each
    row's gradient to the word embedding or to its patch embedding."""

    @staticmethod
    def forward(ctx, word, patch, idx):
        idx = idx.contiguous()
        ctx.save_for_backward(idx)
        ctx.pshape, ctx.pdtype = patch.shape, patch.dtype
        return ops.scatter_rows(word.contiguous(), patch.contiguous(), idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        is_patch = idx >= 0
        dword = dy.masked_fill(is_patch[..., None], 0) if ctx.needs_input_grad[0] else None
        dpatch = None
        if ctx.needs_input_grad[1]:
            B, P, D = ctx.pshape
            dpatch = torch.zeros((B * P, D), dtype=torch.float32, device=dy.device)
            b_ix, s_ix = torch.nonzero(is_patch, as_tuple=True)
            dpatch.index_put_((b_ix * P + idx[b_ix, s_ix],), dy[b_ix, s_ix].float(), accumulate=True)
            dpatch = dpatch.view(B, P, D).to(ctx.pdtype)
        return dword, dpatch, None


def scatter_patch_rows(word, patch, idx):
    return ScatterPatchRowsFn.apply(word, patch, idx)


_SPAN_LIMIT = (1 << 32) - (1 << 20)   # the K-major kernel addresses its operands with 32-bit byte offsets


def _kmajor_wgrad_any_size(dy2, x2):
    """dW = dy^T x (fp32) on the K-major kernel, with the token rows cut into chunks when an operand spans 4 GB or more (Fuyu's 262144-row
    vocabulary: dy is 5.9 GB): the first chunk stores, the others accumulate.  None when the shape does not qualify."""
    rows, n_out = dy2.shape
    n_in = x2.shape[1]
    span = max(dy2.stride(0), x2.stride(0)) * 2 * rows
    parts = 1 if span < _SPAN_LIMIT else -(-span // (_SPAN_LIMIT - (1 << 28)))
    step = -(-rows // parts)
    step = -(-step // 128) * 128
    if not ops.gemm_kmajor_supported(n_out, n_in, min(step, rows), dy2.stride(0), x2.stride(0), True, True, dy2.dtype):
        return None
    cuts = [(r0, min(r0 + step, rows)) for r0 in range(0, rows, step)]
    if any(r1 - r0 < 128 for r0, r1 in cuts):      # (a tail shorter than one K-tile pair: leave the whole product to the library)
        return None
    dW = torch.empty((n_out, n_in), dtype=torch.float32, device=dy2.device)
    for i, (r0, r1) in enumerate(cuts):
        ops.gemm(dy2[r0:r1], x2[r0:r1], True, True, out=dW, kind=EPI_STORE, accumulate=(i > 0))
    return dW


def _kmajor_dgrad_any_size(dy2, Wb, sqrelu_of=None):
    """dx = dy W (W stored [out, in] = the K-major B operand), dy cut into row blocks when it spans 4 GB or more.  None when unsupported.
    With `sqrelu_of` = h [rows, n_in] the GEMM's tail multiplies by 2 relu(h): the input gradient of W . relu(h)^2 in one launch."""
    rows, n_out = dy2.shape
    n_in = Wb.shape[1]
    if Wb.stride(0) * 2 * n_out >= _SPAN_LIMIT:
        return None
    span = dy2.stride(0) * 2 * rows
    parts = 1 if span < _SPAN_LIMIT else -(-span // (_SPAN_LIMIT - (1 << 28)))
    step = -(-rows // parts)
    step = -(-step // 256) * 256
    if not ops.gemm_kmajor_supported(min(step, rows), n_in, n_out, dy2.stride(0), Wb.stride(0), False, True, dy2.dtype):
        return None
    epi = (lambda r0, r1: {}) if sqrelu_of is None else (lambda r0, r1: dict(kind=EPI_GATE_BWD, aux=sqrelu_of[r0:r1], aux_gelu="sqrelu"))
    if parts == 1:
        return ops.gemm(dy2, Wb, False, True, **epi(0, rows))
    dx = torch.empty((rows, n_in), dtype=dy2.dtype, device=dy2.device)
    for r0 in range(0, rows, step):
        r1 = min(r0 + step, rows)
        if not ops.gemm_kmajor_supported(r1 - r0, n_in, n_out, dy2.stride(0), Wb.stride(0), False, True, dy2.dtype):
            torch.mm(dy2[r0:r1], Wb, out=dx[r0:r1])      # a short last block (< 192 tiles)
            if sqrelu_of is not None:
                dx[r0:r1] = ops.sqrelu_bwd(sqrelu_of[r0:r1].contiguous(), dx[r0:r1].contiguous())
        else:
            ops.gemm(dy2[r0:r1], Wb, False, True, out=dx[r0:r1], **epi(r0, r1))
    return dx


class TrainableLinearFn(torch.autograd.Function):
    """y = x W^T + b for a TRAINABLE nn.Linear under bf16 compute with fp32 masters (every Linear of the fully fine-tuned Fuyu
    decoder): the bf16 operand copy of W is the cached shadow that FusedAdamW refreshes inside its own update pass (autocast
    re-casts all 9.4 B master weights every forward: 56 GB of traffic per step at Fuyu-8B), the weight gradient comes out of
    the GEMM in fp32, and the bias gradient is the HIP column sum (torch's bf16 reduction: 186 us per layer, 27 ms per step)."""

    @staticmethod
    def forward(ctx, x, W, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = (x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)).contiguous()
        Wb = shadows.w(W, torch.bfloat16)
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.shp = shp
        ctx.bdtype = b.dtype if b is not None else None
        return F.linear(x2, Wb, b.detach().to(torch.bfloat16) if b is not None else None).view(shp[:-1] + (W.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        N = W.shape[0]
        dy2 = dy.reshape(-1, N)
        dy2 = (dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            Wb = shadows.w(W, torch.bfloat16)
            dx = None
            if os.environ.get("OTTER_NO_KMAJOR") != "1" and os.environ.get("OTTER_NO_KMAJOR_DGRAD") != "1" and dy2.is_cuda:
                dx = _kmajor_dgrad_any_size(dy2, Wb)     # dy W with W as stored = the K-major B operand (hipBLASLt's NN form is its slow one)
            dx = (dx if dx is not None else torch.mm(dy2, Wb)).view(ctx.shp)
        dW = None
        if ctx.needs_input_grad[1]:
            # round 3: dy^T x on the K-major kernel of csrc/gemm.hip -- both operands as they lie, fp32 out, any number of token rows
            # (hipBLASLt's TN form with an fp32 output: 543-628 us at the FFN shapes against 386-394, tools/wgrad_paths.py)
            dW = _kmajor_wgrad_any_size(dy2, x2) if (os.environ.get("OTTER_NO_KMAJOR") != "1" and dy2.is_cuda) else None
            if dW is None:
                try:
                    dW = torch.mm(dy2.t(), x2, out_dtype=torch.float32)
                except TypeError:  # older torch: no out_dtype
                    dW = torch.mm(dy2.t(), x2).float()
            dW = dW if dW.dtype == W.dtype else dW.to(W.dtype)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy2, None, dy2.shape[0])
            db = db if db.dtype == ctx.bdtype else db.to(ctx.bdtype)
        return dx, dW, db


class SqReLULinearFn(torch.autograd.Function):
    """y = relu(h)^2 W^T + b, the second half of the Persimmon MLP (fuyu/modeling_persimmon.py:180-194: fused_mlp_func "sqrelu") for a
    TRAINABLE Linear: like TrainableLinearFn, and the activation's backward rides in the tail of the input-gradient GEMM
    (dh = (dy W) . 2 relu(h) in one launch of the K-major kernel: no separate pass over three [tokens, 4 hidden] tensors)."""

    @staticmethod
    def forward(ctx, h, W, b):
        shp = h.shape
        h2 = h.reshape(-1, shp[-1])
        h2 = (h2 if h2.dtype == torch.bfloat16 else h2.to(torch.bfloat16)).contiguous()
        a2 = ops.sqrelu_fwd(h2)
        Wb = shadows.w(W, torch.bfloat16)
        ctx.save_for_backward(h2, a2, W)
        ctx.has_bias = b is not None
        ctx.shp = shp
        ctx.bdtype = b.dtype if b is not None else None
        return F.linear(a2, Wb, b.detach().to(torch.bfloat16) if b is not None else None).view(shp[:-1] + (W.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        h2, a2, W = ctx.saved_tensors
        N = W.shape[0]
        dy2 = dy.reshape(-1, N)
        dy2 = (dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)).contiguous()
        dh = None
        if ctx.needs_input_grad[0]:
            Wb = shadows.w(W, torch.bfloat16)
            if os.environ.get("OTTER_NO_KMAJOR") != "1" and os.environ.get("OTTER_NO_KMAJOR_DGRAD") != "1" and os.environ.get("OTTER_NO_SQRELU_TAIL") != "1":
                dh = _kmajor_dgrad_any_size(dy2, Wb, sqrelu_of=h2)
            if dh is None:
                da = None
                if os.environ.get("OTTER_NO_KMAJOR") != "1" and os.environ.get("OTTER_NO_KMAJOR_DGRAD") != "1":
                    da = _kmajor_dgrad_any_size(dy2, Wb)
                dh = ops.sqrelu_bwd(h2, da if da is not None else torch.mm(dy2, Wb))
            dh = dh.view(ctx.shp)
        dW = None
        if ctx.needs_input_grad[1]:
            dW = _kmajor_wgrad_any_size(dy2, a2) if os.environ.get("OTTER_NO_KMAJOR") != "1" else None
            if dW is None:
                try:
                    dW = torch.mm(dy2.t(), a2, out_dtype=torch.float32)
                except TypeError:  # older torch: no out_dtype
                    dW = torch.mm(dy2.t(), a2).float()
            dW = dW if dW.dtype == W.dtype else dW.to(W.dtype)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy2, None, dy2.shape[0])
            db = db if db.dtype == ctx.bdtype else db.to(ctx.bdtype)
        return dh, dW, db


def sqrelu_linear(mod, h):
    """mod(relu(h)^2) for the Persimmon MLP's second Linear; the fused autograd path under the conditions of `trainable_linear`."""
    W = mod.weight
    if (h.is_cuda and h.dtype == torch.bfloat16 and h.shape[-1] % 8 == 0 and W.requires_grad and W.dtype == torch.float32
            and compute_dtype_for(h) == torch.bfloat16 and torch.is_grad_enabled() and W.shape[0] % 8 == 0 and os.environ.get("OTTER_TORCH_LINEAR") != "1"):
        return SqReLULinearFn.apply(h, W, mod.bias)
    a = sqrelu(h) if (h.is_cuda and h.dtype == torch.bfloat16 and h.shape[-1] % 8 == 0) else torch.square(F.relu(h))
    return trainable_linear(mod, a)


def trainable_linear(mod, x):
    """nn.Linear forward; the HIP-assisted autograd path when the layer trains in fp32 masters under bf16 compute on the GPU."""
    W = mod.weight
    if (x.is_cuda and W.requires_grad and W.dtype == torch.float32 and compute_dtype_for(x) == torch.bfloat16 and torch.is_grad_enabled()
            and W.shape[0] % 8 == 0 and os.environ.get("OTTER_TORCH_LINEAR") != "1"):
        return TrainableLinearFn.apply(x, W, mod.bias)
    return F.linear(x, W, mod.bias)
