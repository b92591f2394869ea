"""TEST INFRASTRUCTURE ONLY -- a CPU stand-in for the libotter_hip.so entry points of the fusion path, for host-logic tests.

The product has no CPU path (every wrapper in otter_amd.ops raises on a CPU tensor).  Some host logic can only be exercised by
running a whole model -- the reference's own training loop driving otter_amd through the drop-in shim, the DP reducer's
gradient-sink protocol on the real autograd graph, trainable-only checkpoints after an optimizer step -- and this container
has no GPU.  `oracle_backend()` therefore swaps the autograd functions of `otter_amd.functional` that the host modules call
for numpy-oracle implementations (oracle/otter_oracle.py: the same checker the GPU parity tests use), inside a context manager,
from tests only.  Nothing here is importable from the product package (tests/test_host_contract.py checks that the product
never imports `oracle`), and no parity claim rests on it: it checks PLUMBING (names, call order, gradient routing), not kernels.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle import otter_oracle as O
from otter_amd import functional as OF
from otter_amd import modeling_otter as MO
from otter_amd import ops
from otter_amd._capi import MASK_EQ, MASK_NONE


def _np(t):
    return t.detach().to(torch.float32).cpu().numpy()


def _route(param, g, like):
    """Deliver a parameter gradient the way functional._wgrad does: into the DP reducer's bucket view when a gradient sink is
    installed (returns None to autograd), else back to autograd."""
    if param is None or not param.requires_grad:
        return None
    gt = torch.from_numpy(np.ascontiguousarray(g)).to(like.dtype).view_as(param)
    sink = OF.grad_sink
    if sink is not None and param.ndim == 2:          # the product routes the GEMM weight gradients (2-D) through the sink
        out = sink.take(param)
        if out is not None:
            out.copy_(gt)
            sink.ready(param)
            return None
    return gt


class GatedCrossAttentionFn(torch.autograd.Function):
    NAMES = ("attn.norm.weight", "attn.norm.bias", "attn.to_q.weight", "attn.to_kv.weight", "attn.to_out.weight", "attn_gate",
             "feed_forward.0.weight", "feed_forward.0.bias", "feed_forward.1.weight", "feed_forward.3.weight", "ff_gate")

    @staticmethod
    def forward(ctx, x, media, tt, mask_mode, heads, eps, *params):
        assert abs(eps - 1e-5) < 1e-12
        ctx.extra = len(params) - len(GatedCrossAttentionFn.NAMES)      # the product passes `deferred` (None on the CPU: mpt._gated_takes_deferred)
        assert ctx.extra in (0, 1) and all(t is None for t in params[len(GatedCrossAttentionFn.NAMES):])
        params = params[:len(GatedCrossAttentionFn.NAMES)]
        p = {"b." + n: _np(t) for n, t in zip(GatedCrossAttentionFn.NAMES, params)}
        y, c = O.gated_xattn_block_fwd(p, "b.", _np(x), _np(media), None, True, mask_mode == MASK_EQ, heads,
                                       tt=None if mask_mode == MASK_NONE else tt.cpu().numpy())
        ctx.p, ctx.c, ctx.params = p, c, params
        ctx.dtypes = (x.dtype, media.dtype)
        return torch.from_numpy(y).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        dx, dmedia, g = O.gated_xattn_block_bwd(ctx.p, "b.", _np(dy), ctx.c)
        grads = [_route(t, g["b." + n], t) for n, t in zip(GatedCrossAttentionFn.NAMES, ctx.params)]
        return (torch.from_numpy(dx).to(ctx.dtypes[0]), torch.from_numpy(dmedia).to(ctx.dtypes[1]), None, None, None, None, *grads, *([None] * ctx.extra))


class PerceiverBlockFn(torch.autograd.Function):
    NAMES = ("norm_media.weight", "norm_media.bias", "norm_latents.weight", "norm_latents.bias", "to_q.weight", "to_kv.weight",
             "to_out.weight", "feed_forward.0.weight", "feed_forward.0.bias", "feed_forward.1.weight", "feed_forward.3.weight")

    @staticmethod
    def forward(ctx, x, latents, heads, eps, *params):
        p = {"b." + n: _np(t) for n, t in zip(PerceiverBlockFn.NAMES, params)}
        y, c = O.perceiver_block_fwd(p, "b.", _np(x)[None], _np(latents)[None], heads)
        ctx.p, ctx.c, ctx.params = p, c, params
        ctx.dtypes = (x.dtype, latents.dtype)
        return torch.from_numpy(y[0]).to(latents.dtype)

    @staticmethod
    def backward(ctx, dy):
        dx, dlat, g = O.perceiver_block_bwd(ctx.p, "b.", _np(dy)[None], ctx.c)
        grads = [_route(t, g["b." + n], t) for n, t in zip(PerceiverBlockFn.NAMES, ctx.params)]
        return (torch.from_numpy(np.ascontiguousarray(dx[0])).to(ctx.dtypes[0]), torch.from_numpy(np.ascontiguousarray(dlat[0])).to(ctx.dtypes[1]),
                None, None, *grads)


class _Apply:
    """`X.apply(...)` facade over a plain differentiable torch function."""

    def __init__(self, fn):
        self.apply = fn


def _layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps).to(out_dtype or x.dtype)


def _add_layer_norm(x, delta, weight, bias, eps=1e-5, out_dtype=None):
    xsum = x + delta.to(x.dtype)
    return xsum, _layer_norm(xsum, weight, bias, eps, out_dtype)


def _text_time(media_locations, attend_previous=True):
    return torch.from_numpy(O.text_time(media_locations.cpu().numpy(), attend_previous).astype(np.int32))


def _broadcast_emb_add(x4, emb):
    return x4 + emb[: x4.shape[1]].reshape(1, x4.shape[1], 1, x4.shape[-1]).to(x4.dtype)


@contextlib.contextmanager
def oracle_backend():
    """Inside the context the fusion modules of otter_amd.modeling_otter (and the LayerNorms of its MPT host) run on CPU tensors."""
    saved = [(OF, "GatedCrossAttentionFn", OF.GatedCrossAttentionFn), (OF, "PerceiverBlockFn", OF.PerceiverBlockFn),
             (OF, "ExpandLatentsFn", OF.ExpandLatentsFn), (OF, "layer_norm", OF.layer_norm), (OF, "add_layer_norm", OF.add_layer_norm),
             (MO, "_BroadcastEmbAddFn", MO._BroadcastEmbAddFn), (ops, "text_time", ops.text_time)]
    OF.GatedCrossAttentionFn = GatedCrossAttentionFn
    OF.PerceiverBlockFn = PerceiverBlockFn
    OF.ExpandLatentsFn = _Apply(lambda latents, G: latents.unsqueeze(0).expand(G, -1, -1))
    OF.layer_norm = _layer_norm
    OF.add_layer_norm = _add_layer_norm
    MO._BroadcastEmbAddFn = _Apply(_broadcast_emb_add)
    ops.text_time = _text_time
    try:
        yield
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
