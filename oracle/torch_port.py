"""Torch-CPU restatement of the fusion path's modules: the SAME ATen operator sequence the reference's own modules run, with autograd for
the backward -- i.e. the reference's arithmetic ENGINE (torch CPU kernels, MKL / oneDNN GEMMs, autograd) on this repository's weight
dictionaries.

TEST INFRASTRUCTURE (lives under oracle/): only tests/ and bench.py's `cpu_baseline` leg import it; the product never does.
Why it exists next to the numpy oracle (oracle/otter_oracle.py): `cpu_baseline` is meant to be the reference's CPU path timed on the GPU
box's host cores, and /root/reference does not travel to that box.  The numpy port measures numpy's BLAS; this file measures what the
reference would actually execute (torch ops + autograd), op for op:

  perceiver_resampler   src/otter_ai/models/otter/modeling_otter.py:151-184 (block), :213-235 (resampler)
  masked_cross_attention                                            :262-340 (non-xformers branch)
  gated_xattn_block                                                 :373-395
  mpt_block             src/otter_ai/models/mpt/blocks.py:68-88, attention.py:22-84 (attn_impl "torch"), norm.py:16-45 (LPLayerNorm == LayerNorm in fp32)
  unembed_loss          src/otter_ai/models/mpt/modeling_mpt.py:419-435 (tied un-embedding, labels rolled by -1, CE ignore_index -100)

Pinned by tests/test_oracle_golden.py::test_torch_port_* against the fixtures the reference's own modules produced (tests/golden/*.npz,
oracle/gen_golden.py): outputs and every gradient, same tolerance as the numpy oracle.
Parameters: dict name -> torch.Tensor with the reference's state-dict names (oracle/synth.py shape tables)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def to_torch(p: dict, requires_grad: bool = True) -> dict:
    """numpy weight dict -> torch leaf tensors (fp32)."""
    return {k: torch.from_numpy(v.copy()).requires_grad_(requires_grad) for k, v in p.items()}


def _heads(t, h):  # "b t n (h d) -> b h t n d" / "b n (h d) -> b h n d" (einops.rearrange in the reference)
    *lead, n, hd = t.shape
    return t.reshape(*lead, n, h, hd // h).movedim(-2, -3)


def _merge(t):  # inverse of _heads
    t = t.movedim(-3, -2)
    *lead, n, h, d = t.shape
    return t.reshape(*lead, n, h * d)


def perceiver_block(p, pre, x, latents, heads=8):
    """modeling_otter.py:151-184: LN(media), LN(latents), q from latents, k/v from cat(media, latents), softmax(q k^T - amax) v, to_out +
    residual (added by the resampler loop, :231), then the feed-forward LN -> Linear -> GELU -> Linear + residual (:232)."""
    D = x.shape[-1]
    xn = F.layer_norm(x, (D,), p[pre + "norm_media.weight"], p[pre + "norm_media.bias"])
    ln = F.layer_norm(latents, (D,), p[pre + "norm_latents.weight"], p[pre + "norm_latents.bias"])
    q = F.linear(ln, p[pre + "to_q.weight"])
    kv = F.linear(torch.cat((xn, ln), dim=-2), p[pre + "to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    q = q * (q.shape[-1] ** -0.5)
    scores = torch.matmul(q, k.transpose(-1, -2))                  # (the reference spells the two products as einsums: the same bmm kernels)
    scores = scores - scores.amax(-1, keepdim=True).detach()       # :176 -- the shift is detached in the reference too
    probs = torch.softmax(scores, dim=-1)
    out1 = F.linear(_merge(torch.matmul(probs, v)), p[pre + "to_out.weight"]) + latents
    f = F.layer_norm(out1, (D,), p[pre + "feed_forward.0.weight"], p[pre + "feed_forward.0.bias"])
    return F.linear(F.gelu(F.linear(f, p[pre + "feed_forward.1.weight"])), p[pre + "feed_forward.3.weight"]) + out1


def perceiver_resampler(p, pre, x, heads=8):
    """x [b,T,F,v,D] -> [b,T,n,D].  modeling_otter.py:213-235."""
    b, T, Fr, v, D = x.shape
    if (pre + "frame_embs") in p:
        x = x + p[pre + "frame_embs"][:Fr][None, None, :, None, :]
    x = x.reshape(b, T, Fr * v, D)
    if (pre + "media_time_embs") in p:
        x = x + p[pre + "media_time_embs"][:T]
    lat = p[pre + "latents"]
    latents = lat[None, None].expand(b, T, *lat.shape)
    i = 0
    while (pre + f"layers.{i}.to_q.weight") in p:
        latents = perceiver_block(p, pre + f"layers.{i}.", x, latents, heads)
        i += 1
    return F.layer_norm(latents, (D,), p[pre + "norm.weight"], p[pre + "norm.bias"])


def masked_cross_attention(p, pre, x, media, media_locations=None, attend_previous=True, only_attend_immediate_media=True, heads=8):
    """x [B,T,D], media [B,T_img,n,Dv] -> [B,T,D].  modeling_otter.py:262-340."""
    B, T_img, n, Dv = media.shape
    xn = F.layer_norm(x, (x.shape[-1],), p[pre + "norm.weight"], p[pre + "norm.bias"])
    q = F.linear(xn, p[pre + "to_q.weight"])
    kv = F.linear(media.reshape(B, T_img * n, Dv), p[pre + "to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    q = q * (q.shape[-1] ** -0.5)
    scores = torch.matmul(q, k.transpose(-1, -2))
    text_time = None
    if media_locations is not None:
        ml = torch.as_tensor(media_locations, dtype=torch.bool)
        text_time = ml.cumsum(dim=-1)                                         # :298
        media_time = torch.arange(T_img) + 1
        if not attend_previous:                                                # :301-311
            text_time = text_time.clone()
            text_time[~ml] += 1
            text_time[text_time > ml.sum(-1, keepdim=True).expand_as(text_time)] = 0
        op = torch.eq if only_attend_immediate_media else torch.ge             # :313-319
        allowed = op(text_time[:, None, :, None], media_time.repeat_interleave(n)[None, None, None, :])
        scores = scores.masked_fill(~allowed, -torch.finfo(scores.dtype).max)  # :321 (finite fill BEFORE the shift)
    scores = scores - scores.amax(-1, keepdim=True).detach()                   # :323
    probs = torch.softmax(scores, dim=-1)
    if media_locations is not None and only_attend_immediate_media:           # :326-330: rows without a preceding image are zeroed
        probs = probs.masked_fill((text_time == 0)[:, None, :, None], 0.0)
    return F.linear(_merge(torch.matmul(probs, v)), p[pre + "to_out.weight"])


def gated_xattn_block(p, pre, x, media, media_locations=None, attend_previous=True, only_attend_immediate_media=True, heads=8):
    """modeling_otter.py:373-395."""
    a = masked_cross_attention(p, pre + "attn.", x, media, media_locations, attend_previous, only_attend_immediate_media, heads)
    x = a * p[pre + "attn_gate"].tanh() + x
    f = F.layer_norm(x, (x.shape[-1],), p[pre + "feed_forward.0.weight"], p[pre + "feed_forward.0.bias"])
    ff = F.linear(F.gelu(F.linear(f, p[pre + "feed_forward.1.weight"])), p[pre + "feed_forward.3.weight"])
    return ff * p[pre + "ff_gate"].tanh() + x


def alibi_bias(n_heads, seq_len, max_len, alibi_bias_max=8):
    """mpt/attention.py:447-464 (gen_slopes + build_alibi_bias), sliced to the last seq_len keys as modeling_mpt.py:135-139 does."""
    _n = 2 ** math.ceil(math.log2(n_heads))
    m = torch.arange(1, _n + 1, dtype=torch.float32) * (alibi_bias_max / _n)
    slopes = 1.0 / torch.pow(2, m)
    if _n != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    pos = torch.arange(1 - max_len, 1, dtype=torch.int32)[-seq_len:].float()
    return pos[None, None, None, :] * slopes[None, :, None, None]


def mpt_block(p, pre, x, n_heads, attn_bias):
    """Pre-LN block with the fused Wqkv and the "torch" attention implementation: blocks.py:68-88, attention.py:22-84 (scores materialised as
    [B, h, S, S], additive ALiBi bias, causal triangle filled with finfo.min, softmax, attn @ v)."""
    B, S, D = x.shape
    a = F.layer_norm(x, (D,), p[pre + "norm_1.weight"], p.get(pre + "norm_1.bias"))
    q, k, v = F.linear(a, p[pre + "attn.Wqkv.weight"]).chunk(3, dim=2)
    q, k, v = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    w = q.matmul(k.transpose(-1, -2)) * (1.0 / math.sqrt(D // n_heads))
    if attn_bias is not None:
        w = w + attn_bias[..., -S:]
    causal = ~torch.tril(torch.ones(S, S, dtype=torch.bool))
    w = w.masked_fill(causal[None, None], torch.finfo(w.dtype).min)
    ctx = _merge(torch.softmax(w, dim=-1).matmul(v))
    x = x + F.linear(ctx, p[pre + "attn.out_proj.weight"])
    m = F.layer_norm(x, (D,), p[pre + "norm_2.weight"], p.get(pre + "norm_2.bias"))
    return x + F.linear(F.gelu(F.linear(m, p[pre + "ffn.up_proj.weight"])), p[pre + "ffn.down_proj.weight"])


def unembed_loss(h, wte, labels):
    """modeling_mpt.py:419-435: logits = h @ wte^T (tied), labels rolled by -1 with the last position ignored, mean CE."""
    logits = F.linear(h, wte)
    lab = torch.roll(torch.as_tensor(labels), shifts=-1, dims=-1).clone()
    lab[..., -1] = -100
    return logits, F.cross_entropy(logits.reshape(-1, logits.shape[-1]), lab.reshape(-1), ignore_index=-100)
