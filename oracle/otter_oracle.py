"""CPU oracle for the Otter vision->language fusion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement (forward AND hand-derived backward) of the reference's algorithm
for the path named in BASELINE.json / SURVEY.md section 8.  It is the checker for the HIP path in
``otter_amd/`` -- it is never imported by the product.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: the reference (Luodian/Otter) ships no golden vectors for this path,
so the oracle is pinned against outputs of the reference's own PyTorch modules run on CPU in the build
container: ``oracle/gen_golden.py`` imports ``/root/reference/src/otter_ai`` and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against them.

Reference lines restated (all under /root/reference/src/otter_ai/models/):
  layer_norm / linear / gelu       torch semantics used at otter/modeling_otter.py:136-147,253-257,365-368
  perceiver_block_*                otter/modeling_otter.py:151-184   (OtterPerceiverBlock.forward)
  perceiver_resampler_*            otter/modeling_otter.py:213-235   (OtterPerceiverResampler.forward)
  text_time                        otter/modeling_otter.py:296-311   (media-time cumsum, attend_previous)
  masked_cross_attention_*         otter/modeling_otter.py:262-340   (OtterMaskedCrossAttention.forward)
  gated_xattn_block_*              otter/modeling_otter.py:373-395   (OtterGatedCrossAttentionBlock.forward)
  alibi_slopes / mpt_block_*       mpt/attention.py:22-84,447-464 ; mpt/blocks.py:68-88 ; mpt/norm.py:16-45
  mpt_lm_*                         mpt/modeling_mpt.py:172-305,383-436 (wte, blocks, norm_f, tied unembed, CE on rolled labels)
  clip_vision_fwd                  third-party transformers==4.35.1 CLIPVisionModel (restated in-repo at
                                   /root/reference/xformers_model/clip.py:50-199,393-446); forward only, frozen
  otter_* / greedy_decode          otter/modeling_otter.py:486-510,917-1042 (conditioning, forward, generate quirk 3.2)
  rms_norm_* / rope_*              /root/reference/xformers_model/llama.py:95-112,115-166 (config C4 only)

All functions compute in the dtype of their inputs (float32 mirrors the reference CPU path; float64 is
used by tests that need a tight error bound).  Parameters are passed as flat ``{state_dict_key: ndarray}``
dicts with exactly the reference's parameter names.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

NEG_MAX32 = -float(np.finfo(np.float32).max)

# --------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------


def layer_norm_fwd(x, w, b, eps=1e-5):
    """torch.nn.functional.layer_norm over the last axis (biased variance, eps inside the sqrt)."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xhat = xc * rstd
    y = xhat * w if w is not None else xhat
    if b is not None:
        y = y + b
    return y, (xhat, rstd, w)


def layer_norm_bwd(dy, cache):
    xhat, rstd, w = cache
    g = dy * w if w is not None else dy
    red = tuple(range(dy.ndim - 1))
    dw = (dy * xhat).sum(red) if w is not None else None
    db = dy.sum(red)
    m1 = g.mean(-1, keepdims=True)
    m2 = (g * xhat).mean(-1, keepdims=True)
    dx = (g - m1 - xhat * m2) * rstd
    return dx, dw, db


def linear_fwd(x, W, bias=None):
    """nn.Linear: y = x @ W^T (+ bias); W is [out, in]."""
    y = x @ W.T
    if bias is not None:
        y = y + bias
    return y


def linear_bwd(dy, x, W, need_bias=False):
    dx = dy @ W
    dW = dy.reshape(-1, dy.shape[-1]).T @ x.reshape(-1, x.shape[-1])
    if need_bias:
        return dx, dW, dy.reshape(-1, dy.shape[-1]).sum(0)
    return dx, dW


def gelu_fwd(u):
    """nn.GELU() default = exact erf form."""
    t = u.dtype.type
    return t(0.5) * u * (t(1.0) + _erf(u * t(1.0 / math.sqrt(2.0))).astype(u.dtype))


def gelu_grad(u):
    t = u.dtype.type
    cdf = t(0.5) * (t(1.0) + _erf(u * t(1.0 / math.sqrt(2.0))).astype(u.dtype))
    pdf = np.exp(t(-0.5) * u * u) * t(1.0 / math.sqrt(2.0 * math.pi))
    return cdf + u * pdf


def quick_gelu(u):
    return u / (1.0 + np.exp(-1.702 * u)).astype(u.dtype)


def _neg_max(dtype):
    return -np.finfo(dtype).max


def softmax_lastdim(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


# --------------------------------------------------------------------------------------------------
# perceiver resampler  (modeling_otter.py:129-235)
# --------------------------------------------------------------------------------------------------


def _split_heads(t, h):  # [..., n, h*d] -> [..., h, n, d]
    *lead, n, hd = t.shape
    t = t.reshape(*lead, n, h, hd // h)
    return np.moveaxis(t, -2, -3)


def _merge_heads(t):  # [..., h, n, d] -> [..., n, h*d]
    t = np.moveaxis(t, -3, -2)
    *lead, n, h, d = t.shape
    return t.reshape(*lead, n, h * d)


def perceiver_block_fwd(p, pre, x, latents, heads=8):
    """x [b,T,n1,D] media features, latents [b,T,n2,D].  modeling_otter.py:151-184."""
    xn, c_nm = layer_norm_fwd(x, p[pre + "norm_media.weight"], p[pre + "norm_media.bias"])
    ln, c_nl = layer_norm_fwd(latents, p[pre + "norm_latents.weight"], p[pre + "norm_latents.bias"])
    Wq, Wkv, Wo = p[pre + "to_q.weight"], p[pre + "to_kv.weight"], p[pre + "to_out.weight"]
    inner = Wq.shape[0]
    d = inner // heads
    scale = x.dtype.type(d ** -0.5)
    q = linear_fwd(ln, Wq)
    kv_in = np.concatenate([xn, ln], axis=-2)
    kv = linear_fwd(kv_in, Wkv)
    k, v = kv[..., :inner], kv[..., inner:]
    qh = _split_heads(q, heads) * scale  # [b,T,h,n2,d]
    kh = _split_heads(k, heads)
    vh = _split_heads(v, heads)
    sim = qh @ np.swapaxes(kh, -1, -2)
    attn = softmax_lastdim(sim)
    oh = attn @ vh
    o = _merge_heads(oh)
    out1 = linear_fwd(o, Wo) + latents
    f, c_ff = layer_norm_fwd(out1, p[pre + "feed_forward.0.weight"], p[pre + "feed_forward.0.bias"])
    u = linear_fwd(f, p[pre + "feed_forward.1.weight"])
    hdn = gelu_fwd(u)
    y = linear_fwd(hdn, p[pre + "feed_forward.3.weight"]) + out1
    cache = dict(c_nm=c_nm, c_nl=c_nl, ln=ln, kv_in=kv_in, qh=qh, kh=kh, vh=vh, attn=attn, o=o, c_ff=c_ff,
                 f=f, u=u, hdn=hdn, n1=x.shape[-2], heads=heads, scale=scale, inner=inner)
    return y, cache


def perceiver_block_bwd(p, pre, dy, c):
    """Returns (dx_media, dlatents, grads dict keyed by full parameter names)."""
    g = {}
    W1, W3 = p[pre + "feed_forward.1.weight"], p[pre + "feed_forward.3.weight"]
    dh, g[pre + "feed_forward.3.weight"] = linear_bwd(dy, c["hdn"], W3)
    du = dh * gelu_grad(c["u"])
    df, g[pre + "feed_forward.1.weight"] = linear_bwd(du, c["f"], W1)
    dout1, g[pre + "feed_forward.0.weight"], g[pre + "feed_forward.0.bias"] = layer_norm_bwd(df, c["c_ff"])
    dout1 = dout1 + dy
    Wq, Wkv, Wo = p[pre + "to_q.weight"], p[pre + "to_kv.weight"], p[pre + "to_out.weight"]
    do, g[pre + "to_out.weight"] = linear_bwd(dout1, c["o"], Wo)
    dlat = dout1.copy()
    doh = _split_heads(do, c["heads"])
    attn = c["attn"]
    dvh = np.swapaxes(attn, -1, -2) @ doh
    dattn = doh @ np.swapaxes(c["vh"], -1, -2)
    dsim = attn * (dattn - (dattn * attn).sum(-1, keepdims=True))
    dqh = dsim @ c["kh"]
    dkh = np.swapaxes(dsim, -1, -2) @ c["qh"]
    dq = _merge_heads(dqh * c["scale"])
    dkv = np.concatenate([_merge_heads(dkh), _merge_heads(dvh)], axis=-1)
    dkv_in, g[pre + "to_kv.weight"] = linear_bwd(dkv, c["kv_in"], Wkv)
    dln, g[pre + "to_q.weight"] = linear_bwd(dq, c["ln"], Wq)
    n1 = c["n1"]
    dxn = dkv_in[..., :n1, :]
    dln = dln + dkv_in[..., n1:, :]
    dl2, g[pre + "norm_latents.weight"], g[pre + "norm_latents.bias"] = layer_norm_bwd(dln, c["c_nl"])
    dlat = dlat + dl2
    dx, g[pre + "norm_media.weight"], g[pre + "norm_media.bias"] = layer_norm_bwd(dxn, c["c_nm"])
    return dx, dlat, g


def perceiver_resampler_fwd(p, pre, x, heads=8):
    """x [b,T,F,v,D] -> [b,T,n,D].  modeling_otter.py:213-235."""
    b, T, F, v, D = x.shape
    if (pre + "frame_embs") in p:
        x = x + p[pre + "frame_embs"][:F][None, None, :, None, :]
    x = x.reshape(b, T, F * v, D)
    if (pre + "media_time_embs") in p:
        x = x + p[pre + "media_time_embs"][:T]
    lat0 = p[pre + "latents"]
    latents = np.broadcast_to(lat0, (b, T) + lat0.shape).copy()
    depth = 0
    while (pre + f"layers.{depth}.to_q.weight") in p:
        depth += 1
    caches = []
    for i in range(depth):
        latents, c = perceiver_block_fwd(p, pre + f"layers.{i}.", x, latents, heads)
        caches.append(c)
    y, c_norm = layer_norm_fwd(latents, p[pre + "norm.weight"], p[pre + "norm.bias"])
    return y, dict(blocks=caches, c_norm=c_norm, shape=(b, T, F, v, D), depth=depth)


def perceiver_resampler_bwd(p, pre, dy, c):
    g = {}
    dlat, g[pre + "norm.weight"], g[pre + "norm.bias"] = layer_norm_bwd(dy, c["c_norm"])
    b, T, F, v, D = c["shape"]
    dx = np.zeros((b, T, F * v, D), dtype=dy.dtype)
    for i in reversed(range(c["depth"])):
        dxi, dlat, gi = perceiver_block_bwd(p, pre + f"layers.{i}.", dlat, c["blocks"][i])
        dx += dxi
        g.update(gi)
    g[pre + "latents"] = dlat.sum((0, 1))
    if (pre + "media_time_embs") in p:
        gm = np.zeros_like(p[pre + "media_time_embs"])
        gm[:T] = dx.sum((0, 2))[:, None, :]
        g[pre + "media_time_embs"] = gm
    dx = dx.reshape(b, T, F, v, D)
    if (pre + "frame_embs") in p:
        gf = np.zeros_like(p[pre + "frame_embs"])
        gf[:F] = dx.sum((0, 1, 3))
        g[pre + "frame_embs"] = gf
    return dx, g


# --------------------------------------------------------------------------------------------------
# masked / gated cross attention  (modeling_otter.py:238-395)
# --------------------------------------------------------------------------------------------------


def text_time(media_locations, attend_previous=True):
    """cumsum of the <image> indicator (+ the attend_previous=False rewrite).  modeling_otter.py:298-311."""
    ml = np.asarray(media_locations).astype(bool)
    tt = np.cumsum(ml.astype(np.int64), axis=-1)
    if not attend_previous:
        tt = tt.copy()
        tt[~ml] += 1
        cnt = ml.sum(-1, keepdims=True)
        tt[tt > cnt] = 0
    return tt


def masked_cross_attention_fwd(p, pre, x, media, media_locations=None, attend_previous=True,
                               only_attend_immediate_media=True, heads=8, tt=None):
    """x [B,T,D], media [B,T_img,n,Dv] -> [B,T,D].  modeling_otter.py:262-340 (non-xformers branch).
    `tt`: a precomputed text_time [B,T] (what `text_time(media_locations, attend_previous)` returns) may be passed instead of
    media_locations -- the product computes the scan once per forward and shares it between its layers."""
    if tt is not None and media_locations is None:
        media_locations = True   # only its presence is used below once tt is given
    B, T_img, n, Dv = media.shape
    xn, c_n = layer_norm_fwd(x, p[pre + "norm.weight"], p[pre + "norm.bias"])
    Wq, Wkv, Wo = p[pre + "to_q.weight"], p[pre + "to_kv.weight"], p[pre + "to_out.weight"]
    inner = Wq.shape[0]
    d = inner // heads
    scale = x.dtype.type(d ** -0.5)
    q = linear_fwd(xn, Wq)
    med = media.reshape(B, T_img * n, Dv)
    kv = linear_fwd(med, Wkv)
    k, v = kv[..., :inner], kv[..., inner:]
    qh = _split_heads(q, heads) * scale  # [B,h,T,d]
    kh = _split_heads(k, heads)
    vh = _split_heads(v, heads)
    sim = qh @ np.swapaxes(kh, -1, -2)  # [B,h,T,M]
    allowed = None
    zero_rows = None
    if media_locations is not None:
        if tt is None:
            tt = text_time(media_locations, attend_previous)  # [B,T]
        tt = np.asarray(tt).astype(np.int64)
        media_time = np.repeat(np.arange(T_img) + 1, n)  # [M]
        if only_attend_immediate_media:
            allowed = tt[:, None, :, None] == media_time[None, None, None, :]
        else:
            allowed = tt[:, None, :, None] >= media_time[None, None, None, :]
        sim = np.where(allowed, sim, _neg_max(sim.dtype))
    attn = softmax_lastdim(sim)
    if media_locations is not None and only_attend_immediate_media:
        zero_rows = (tt == 0)[:, None, :, None]
        attn = np.where(zero_rows, 0, attn).astype(sim.dtype)
    oh = attn @ vh
    o = _merge_heads(oh)
    y = linear_fwd(o, Wo)
    cache = dict(c_n=c_n, xn=xn, med=med, qh=qh, kh=kh, vh=vh, attn=attn, o=o, allowed=allowed,
                 zero_rows=zero_rows, heads=heads, scale=scale, media_shape=media.shape)
    return y, cache


def masked_cross_attention_bwd(p, pre, dy, c):
    g = {}
    Wq, Wkv, Wo = p[pre + "to_q.weight"], p[pre + "to_kv.weight"], p[pre + "to_out.weight"]
    do, g[pre + "to_out.weight"] = linear_bwd(dy, c["o"], Wo)
    doh = _split_heads(do, c["heads"])
    attn = c["attn"]
    dvh = np.swapaxes(attn, -1, -2) @ doh
    dattn = doh @ np.swapaxes(c["vh"], -1, -2)
    if c["zero_rows"] is not None:
        dattn = np.where(c["zero_rows"], 0, dattn).astype(dattn.dtype)
        # softmax output p (pre-zeroing) is needed for the softmax jacobian; rows that were zeroed get no grad
        # at all, rows that are fully masked have constant (uniform) p whatever sim is.
    # softmax backward on the *pre-zeroing* probabilities: for non-zeroed rows attn == p.
    dsim = attn * (dattn - (dattn * attn).sum(-1, keepdims=True))
    if c["allowed"] is not None:
        dsim = np.where(c["allowed"], dsim, 0).astype(dsim.dtype)  # masked_fill kills the gradient to sim
    dqh = dsim @ c["kh"]
    dkh = np.swapaxes(dsim, -1, -2) @ c["qh"]
    dq = _merge_heads(dqh * c["scale"])
    dkv = np.concatenate([_merge_heads(dkh), _merge_heads(dvh)], axis=-1)
    dmed, g[pre + "to_kv.weight"] = linear_bwd(dkv, c["med"], Wkv)
    dxn, g[pre + "to_q.weight"] = linear_bwd(dq, c["xn"], Wq)
    dx, g[pre + "norm.weight"], g[pre + "norm.bias"] = layer_norm_bwd(dxn, c["c_n"])
    return dx, dmed.reshape(c["media_shape"]), g


def gated_xattn_block_fwd(p, pre, x, media, media_locations=None, attend_previous=True,
                          only_attend_immediate_media=True, heads=8, tt=None):
    """modeling_otter.py:373-395."""
    a, c_a = masked_cross_attention_fwd(p, pre + "attn.", x, media, media_locations, attend_previous,
                                        only_attend_immediate_media, heads, tt=tt)
    ga = np.tanh(p[pre + "attn_gate"]).astype(x.dtype)
    x1 = a * ga + x
    f, c_ff = layer_norm_fwd(x1, p[pre + "feed_forward.0.weight"], p[pre + "feed_forward.0.bias"])
    u = linear_fwd(f, p[pre + "feed_forward.1.weight"])
    hdn = gelu_fwd(u)
    ff = linear_fwd(hdn, p[pre + "feed_forward.3.weight"])
    gf = np.tanh(p[pre + "ff_gate"]).astype(x.dtype)
    y = ff * gf + x1
    return y, dict(c_a=c_a, a=a, ga=ga, c_ff=c_ff, f=f, u=u, hdn=hdn, ff=ff, gf=gf)


def gated_xattn_block_bwd(p, pre, dy, c):
    g = {}
    gf, ga = c["gf"], c["ga"]
    g[pre + "ff_gate"] = np.array([(dy * c["ff"]).sum() * (1.0 - float(gf[0]) ** 2)], dtype=dy.dtype)
    dff = dy * gf
    dh, g[pre + "feed_forward.3.weight"] = linear_bwd(dff, c["hdn"], p[pre + "feed_forward.3.weight"])
    du = dh * gelu_grad(c["u"])
    df, g[pre + "feed_forward.1.weight"] = linear_bwd(du, c["f"], p[pre + "feed_forward.1.weight"])
    dx1, g[pre + "feed_forward.0.weight"], g[pre + "feed_forward.0.bias"] = layer_norm_bwd(df, c["c_ff"])
    dx1 = dx1 + dy
    g[pre + "attn_gate"] = np.array([(dx1 * c["a"]).sum() * (1.0 - float(ga[0]) ** 2)], dtype=dy.dtype)
    da = dx1 * ga
    dx, dmedia, ga_ = masked_cross_attention_bwd(p, pre + "attn.", da, c["c_a"])
    g.update(ga_)
    return dx + dx1, dmedia, g


# --------------------------------------------------------------------------------------------------
# frozen host: MPT decoder (mpt/blocks.py, mpt/attention.py, mpt/modeling_mpt.py) -- no biases (no_bias:true)
# --------------------------------------------------------------------------------------------------


def alibi_slopes(n_heads, alibi_bias_max=8):
    """mpt/attention.py:447-455 (gen_slopes)."""
    _n = 2 ** math.ceil(math.log2(n_heads))
    m = np.arange(1, _n + 1, dtype=np.float32) * np.float32(alibi_bias_max / _n)
    slopes = (1.0 / np.power(np.float32(2), m)).astype(np.float32)
    if _n != n_heads:
        slopes = np.concatenate([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes


def mpt_attn_bias(n_heads, s_k, total_len, alibi_bias_max=8, dtype=np.float32):
    """ALiBi bias [1,h,1,s_k] for the LAST s_k key positions of a max_seq_len=total_len table.
    mpt/attention.py:458-464 + the slicing at mpt/modeling_mpt.py:135-139 / attention.py:52-55."""
    pos = np.arange(1 - total_len, 1, dtype=np.int32)[-s_k:].astype(np.float32)
    return (pos[None, None, None, :] * alibi_slopes(n_heads, alibi_bias_max)[None, :, None, None]).astype(dtype)


def mpt_attention_core(qh, kh, vh, scale, slopes=None, key_padding=None, causal=True, dctx=None):
    """The attention core of mpt/attention.py:22-84 on split heads [B,h,S,d] with the ALiBi bias of :447-464 built in its
    key-position form (slope * (j - (Sk-1))), key padding as masked_fill(finfo.min) (modeling_mpt.py:135-144) and the
    causal triangle (attention.py:64-72).  Returns ctx [B,h,Sq,d] and, when dctx is given, (dq, dk, dv)."""
    s_q, s_k = qh.shape[2], kh.shape[2]
    w = (qh @ np.swapaxes(kh, -1, -2)) * scale
    if slopes is not None:
        w = w + (np.asarray(slopes, w.dtype)[None, :, None, None] * np.arange(1 - s_k, 1, dtype=w.dtype)[None, None, None, :])
    minv = np.finfo(w.dtype).min
    if key_padding is not None:
        w = np.where(key_padding[:, None, None, -s_k:].astype(bool), w, minv)
    if causal and s_q != 1:
        s = max(s_q, s_k)
        cm = ~np.tril(np.ones((s, s), dtype=bool))[-s_q:, -s_k:]
        w = np.where(cm[None, None], minv, w)
    pr = softmax_lastdim(w)
    ctx = pr @ vh
    if dctx is None:
        return ctx
    dvh = np.swapaxes(pr, -1, -2) @ dctx
    dpr = dctx @ np.swapaxes(vh, -1, -2)
    dw = pr * (dpr - (dpr * pr).sum(-1, keepdims=True)) * scale
    return ctx, (dw @ kh, np.swapaxes(dw, -1, -2) @ qh, dvh)


def mpt_block_fwd(p, pre, x, n_heads, attn_bias, key_padding=None, past_kv=None):
    """Pre-LN block, fused Wqkv, causal softmax attention with additive bias.  blocks.py:68-88, attention.py:22-84.
    past_kv = (k_past [B,h,d,S0], v_past [B,h,S0,d]) as in the reference's torch cache layout."""
    B, S, D = x.shape
    d = D // n_heads
    a, c1 = layer_norm_fwd(x, p[pre + "norm_1.weight"], p.get(pre + "norm_1.bias"))
    qkv = linear_fwd(a, p[pre + "attn.Wqkv.weight"])
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    qh = _split_heads(q, n_heads)  # [B,h,S,d]
    kh = _split_heads(k, n_heads)
    vh = _split_heads(v, n_heads)
    if past_kv is not None and len(past_kv) != 0:
        kh = np.concatenate([np.swapaxes(past_kv[0], -1, -2), kh], axis=2)
        vh = np.concatenate([past_kv[1], vh], axis=2)
    new_past = (np.swapaxes(kh, -1, -2), vh)
    s_q, s_k = qh.shape[2], kh.shape[2]
    scale = x.dtype.type(1.0 / math.sqrt(d))
    w = (qh @ np.swapaxes(kh, -1, -2)) * scale
    if attn_bias is not None:
        w = w + attn_bias[..., -s_k:].astype(w.dtype)
    minv = np.finfo(w.dtype).min
    if key_padding is not None:
        w = np.where(key_padding[:, None, None, -s_k:].astype(bool), w, minv)
    if s_q != 1:
        s = max(s_q, s_k)
        causal = ~np.tril(np.ones((s, s), dtype=bool))[-s_q:, -s_k:]
        w = np.where(causal[None, None], minv, w)
    pr = softmax_lastdim(w)
    ctx = _merge_heads(pr @ vh)
    bo = linear_fwd(ctx, p[pre + "attn.out_proj.weight"])
    x1 = x + bo
    m, c2 = layer_norm_fwd(x1, p[pre + "norm_2.weight"], p.get(pre + "norm_2.bias"))
    u = linear_fwd(m, p[pre + "ffn.up_proj.weight"])
    hdn = gelu_fwd(u)
    y = x1 + linear_fwd(hdn, p[pre + "ffn.down_proj.weight"])
    cache = dict(c1=c1, a=a, qh=qh, kh=kh, vh=vh, pr=pr, ctx=ctx, c2=c2, m=m, u=u, hdn=hdn, scale=scale,
                 n_heads=n_heads)
    return y, cache, new_past


def mpt_block_bwd_input(p, pre, dy, c):
    """dgrad only (the block is frozen): returns dx."""
    dh = dy @ p[pre + "ffn.down_proj.weight"]
    du = dh * gelu_grad(c["u"])
    dm = du @ p[pre + "ffn.up_proj.weight"]
    dx1, _, _ = layer_norm_bwd(dm, c["c2"])
    dx1 = dx1 + dy
    dctx = dx1 @ p[pre + "attn.out_proj.weight"]
    doh = _split_heads(dctx, c["n_heads"])
    pr = c["pr"]
    dvh = np.swapaxes(pr, -1, -2) @ doh
    dpr = doh @ np.swapaxes(c["vh"], -1, -2)
    dw = pr * (dpr - (dpr * pr).sum(-1, keepdims=True)) * c["scale"]
    dqh = dw @ c["kh"]
    dkh = np.swapaxes(dw, -1, -2) @ c["qh"]
    dqkv = np.concatenate([_merge_heads(dqh), _merge_heads(dkh), _merge_heads(dvh)], axis=-1)
    da = dqkv @ p[pre + "attn.Wqkv.weight"]
    dx, _, _ = layer_norm_bwd(da, c["c1"])
    return dx + dx1


def cross_entropy_rolled(logits, labels):
    """mpt/modeling_mpt.py:428-435: labels rolled by -1 (flat roll, then last column := -100), mean CE, ignore -100."""
    B, S, V = logits.shape
    lab = np.roll(labels.reshape(-1), -1).reshape(B, S).copy()
    lab[:, -1] = -100
    flat = logits.reshape(-1, V)
    lf = lab.reshape(-1)
    valid = lf != -100
    z = flat - flat.max(-1, keepdims=True)
    lse = np.log(np.exp(z).sum(-1, keepdims=True))
    logp = z - lse
    idx = np.where(valid, lf, 0)
    nll = -logp[np.arange(flat.shape[0]), idx]
    n = max(int(valid.sum()), 1)
    loss = (nll * valid).sum() / n
    dlogits = np.exp(logp)
    dlogits[np.arange(flat.shape[0]), idx] -= 1.0
    dlogits = dlogits * (valid[:, None] / n)
    return logits.dtype.type(loss), dlogits.reshape(B, S, V).astype(logits.dtype)


# --------------------------------------------------------------------------------------------------
# frozen host: CLIP vision tower (forward only)
# --------------------------------------------------------------------------------------------------


def clip_vision_fwd(p, pre, pixels, n_heads, patch, eps=1e-5):
    """pixels [N,3,H,W] -> last_hidden_state [N,1+v,D] (before post_layernorm), as CLIPVisionModel(...)[0]."""
    N, C, H, W = pixels.shape
    gh, gw = H // patch, W // patch
    Wp = p[pre + "embeddings.patch_embedding.weight"]  # [D,3,patch,patch]
    D = Wp.shape[0]
    pat = pixels.reshape(N, C, gh, patch, gw, patch).transpose(0, 2, 4, 1, 3, 5).reshape(N, gh * gw, C * patch * patch)
    pe = pat @ Wp.reshape(D, -1).T
    cls = np.broadcast_to(p[pre + "embeddings.class_embedding"], (N, 1, D))
    h = np.concatenate([cls, pe], axis=1) + p[pre + "embeddings.position_embedding.weight"][None]
    h, _ = layer_norm_fwd(h, p[pre + "pre_layrnorm.weight"], p[pre + "pre_layrnorm.bias"], eps)
    i = 0
    d = D // n_heads
    while (pre + f"encoder.layers.{i}.layer_norm1.weight") in p:
        lp = pre + f"encoder.layers.{i}."
        a, _ = layer_norm_fwd(h, p[lp + "layer_norm1.weight"], p[lp + "layer_norm1.bias"], eps)
        q = _split_heads(linear_fwd(a, p[lp + "self_attn.q_proj.weight"], p[lp + "self_attn.q_proj.bias"]), n_heads)
        k = _split_heads(linear_fwd(a, p[lp + "self_attn.k_proj.weight"], p[lp + "self_attn.k_proj.bias"]), n_heads)
        v = _split_heads(linear_fwd(a, p[lp + "self_attn.v_proj.weight"], p[lp + "self_attn.v_proj.bias"]), n_heads)
        pr = softmax_lastdim((q * h.dtype.type(d ** -0.5)) @ np.swapaxes(k, -1, -2))
        ctx = _merge_heads(pr @ v)
        h = h + linear_fwd(ctx, p[lp + "self_attn.out_proj.weight"], p[lp + "self_attn.out_proj.bias"])
        m, _ = layer_norm_fwd(h, p[lp + "layer_norm2.weight"], p[lp + "layer_norm2.bias"], eps)
        m = quick_gelu(linear_fwd(m, p[lp + "mlp.fc1.weight"], p[lp + "mlp.fc1.bias"]))
        h = h + linear_fwd(m, p[lp + "mlp.fc2.weight"], p[lp + "mlp.fc2.bias"])
        i += 1
    return h


# --------------------------------------------------------------------------------------------------
# whole model: OtterForConditionalGeneration over an MPT decoder
# --------------------------------------------------------------------------------------------------


class OtterSpec:
    """Static description of a model instance (what OtterConfig carries in the reference)."""

    def __init__(self, n_layers, d_model, n_heads, max_seq_len, cross_attn_every_n_layers, media_token_id,
                 clip_heads, clip_patch, alibi_bias_max=8, only_attend_immediate_media=True, xattn_heads=8,
                 clip_eps=1e-5):
        self.n_layers, self.d_model, self.n_heads, self.max_seq_len = n_layers, d_model, n_heads, max_seq_len
        self.every = cross_attn_every_n_layers
        self.media_token_id = media_token_id
        self.clip_heads, self.clip_patch, self.clip_eps = clip_heads, clip_patch, clip_eps
        self.alibi_bias_max = alibi_bias_max
        self.immediate = only_attend_immediate_media
        self.xattn_heads = xattn_heads

    def has_xattn(self, layer_idx):
        return (layer_idx + 1) % self.every == 0


def otter_encode_vision(p, spec, vision_x):
    """modeling_otter.py:975-997: (b T F) flatten -> CLIP -> drop CLS -> perceiver."""
    b, T, F = vision_x.shape[:3]
    pix = vision_x.reshape((b * T * F,) + vision_x.shape[3:])
    feats = clip_vision_fwd(p, "vision_encoder.vision_model.", pix, spec.clip_heads, spec.clip_patch, spec.clip_eps)
    feats = feats[:, 1:, :]
    feats = feats.reshape(b, T, F, feats.shape[1], feats.shape[2])
    vis, c = perceiver_resampler_fwd(p, "perceiver.", feats)
    return vis, c


def otter_lm_fwd(p, spec, vis, input_ids, attention_mask=None, past=None, media_locations=None, keep_caches=True):
    """OtterLMMixin.forward + MPTForCausalLM.forward (modeling_otter.py:486-510, modeling_mpt.py:172-305,383-426).
    Returns (logits, caches, new_past).  keep_caches=False drops the per-layer backward caches as it goes (forward-only use at
    the full OTTER-MPT7B size: ~1 GB of fp32 intermediates per layer and sample batch)."""
    LP = "lang_encoder.transformer."
    ids = np.asarray(input_ids)
    if media_locations is None:
        media_locations = ids == spec.media_token_id
    wte = p[LP + "wte.weight"]
    x = wte[ids]
    s_past = 0 if not past or len(past[0]) == 0 else past[0][0].shape[3]
    s_k = ids.shape[1] + s_past
    bias = mpt_attn_bias(spec.n_heads, s_k, spec.max_seq_len, spec.alibi_bias_max, np.float32)
    kp = None
    if attention_mask is not None:
        kp = np.asarray(attention_mask).astype(bool)
    caches = []
    new_past = []
    for i in range(spec.n_layers):
        bp = LP + f"blocks.{i}."
        cx = None
        if spec.has_xattn(i):
            x, cx = gated_xattn_block_fwd(p, bp + "gated_cross_attn_layer.", x, vis, media_locations, True,
                                          spec.immediate, spec.xattn_heads)
        x, cb, npast = mpt_block_fwd(p, bp + "decoder_layer.", x, spec.n_heads, bias, kp,
                                     past[i] if past else None)
        caches.append((cx, cb) if keep_caches else None)
        new_past.append(npast)
    xf, cf = layer_norm_fwd(x, p[LP + "norm_f.weight"], p.get(LP + "norm_f.bias"))
    logits = xf @ wte.T
    return logits, dict(layers=caches, cf=cf, xf=xf, ids=ids), new_past


def otter_forward(p, spec, vision_x, input_ids, attention_mask=None, labels=None, keep_caches=True):
    vis, cv = otter_encode_vision(p, spec, vision_x)
    logits, cl, _ = otter_lm_fwd(p, spec, vis, input_ids, attention_mask, keep_caches=keep_caches)
    out = dict(logits=logits, vis=vis)
    if labels is not None:
        out["loss"], out["dlogits"] = cross_entropy_rolled(logits, np.asarray(labels))
    out["_caches"] = (cv, cl)
    return out


def otter_backward(p, spec, out):
    """Gradients of the loss wrt every *trainable* parameter of the reference recipe
    (perceiver.*, *.gated_cross_attn_layer.*, wte) -- modeling_otter.py:897-905."""
    LP = "lang_encoder.transformer."
    cv, cl = out["_caches"]
    dlogits = out["dlogits"]
    wte = p[LP + "wte.weight"]
    g = {}
    V, D = wte.shape
    gw = dlogits.reshape(-1, V).T @ cl["xf"].reshape(-1, D)
    dxf = dlogits @ wte
    dx, _, _ = layer_norm_bwd(dxf, cl["cf"])
    dvis = np.zeros_like(out["vis"])
    for i in reversed(range(spec.n_layers)):
        bp = LP + f"blocks.{i}."
        cx, cb = cl["layers"][i]
        dx = mpt_block_bwd_input(p, bp + "decoder_layer.", dx, cb)
        if cx is not None:
            dx, dmed, gi = gated_xattn_block_bwd(p, bp + "gated_cross_attn_layer.", dx, cx)
            dvis += dmed
            g.update(gi)
    np.add.at(gw, cl["ids"].reshape(-1), dx.reshape(-1, D))
    g[LP + "wte.weight"] = gw
    _, gp = perceiver_resampler_bwd(p, "perceiver.", dvis, cv)
    g.update(gp)
    return g


def greedy_decode(p, spec, vision_x, input_ids, max_new_tokens, eos_token_id=None, use_cache=False, trace=None):
    """Greedy loop over OtterLMMixin.forward, restating the behaviour of generate() (modeling_otter.py:999-1042)
    for both decode modes of SURVEY.md section 3.2.  With use_cache=True the step input is the last token only, so
    media_locations is recomputed from that single token (modeling_otter.py:491-492) -> text_time==0 -> the
    cross-attention output is exactly zero on cached steps (the reference quirk).
    trace: optional list; per step a dict(top1, top2, margin, absmax) of the last-position logits per sample is appended (how close
    each greedy choice was -- what a reduced-precision run has to resolve to agree)."""
    vis, _ = otter_encode_vision(p, spec, vision_x)
    ids = np.asarray(input_ids).copy()
    B = ids.shape[0]
    done = np.zeros(B, dtype=bool)
    past = None
    for step in range(max_new_tokens):
        if use_cache:
            if past is None:
                logits, _, past = otter_lm_fwd(p, spec, vis, ids, None, [() for _ in range(spec.n_layers)], keep_caches=False)
            else:
                logits, _, past = otter_lm_fwd(p, spec, vis, ids[:, -1:], None, past, keep_caches=False)
        else:
            logits, _, _ = otter_lm_fwd(p, spec, vis, ids, None, keep_caches=False)
        nxt = logits[:, -1, :].argmax(-1)
        if trace is not None:
            last = logits[:, -1, :]
            two = np.sort(np.partition(last, -2, axis=-1)[:, -2:], axis=-1)
            trace.append(dict(top1=two[:, 1].copy(), top2=two[:, 0].copy(), margin=two[:, 1] - two[:, 0], absmax=np.abs(last).max(-1)))
        if eos_token_id is not None:
            nxt = np.where(done, eos_token_id, nxt)
            done |= nxt == eos_token_id
        ids = np.concatenate([ids, nxt[:, None].astype(ids.dtype)], axis=1)
        if eos_token_id is not None and done.all():
            break
    return ids


# --------------------------------------------------------------------------------------------------
# config-C4 extras: LLaMA RMSNorm and RoPE (xformers_model/llama.py:95-112,115-166)
# --------------------------------------------------------------------------------------------------


def rms_norm_fwd(x, w, eps=1e-6):
    var = (x.astype(np.float32) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + np.float32(eps))
    xn = (x.astype(np.float32) * rstd).astype(x.dtype)
    return w * xn, (xn, rstd, w)


def rms_norm_bwd(dy, cache):
    xn, rstd, w = cache
    g = (dy * w).astype(np.float32)
    dw = (dy * xn).reshape(-1, xn.shape[-1]).sum(0)
    m = (g * xn).mean(-1, keepdims=True)
    return ((g - xn * m) * rstd).astype(dy.dtype), dw


def rope_tables(seq_len, dim, base=10000.0, dtype=np.float32):
    inv = 1.0 / (base ** (np.arange(0, dim, 2, dtype=np.float32) / dim))
    fr = np.outer(np.arange(seq_len, dtype=np.float32), inv)
    emb = np.concatenate([fr, fr], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def rope_fwd(x, cos, sin, rot_dim=None):
    """x [B,S,H,d]; cos/sin [S,rot]; half-split (non-interleaved) layout; partial rotary if rot_dim < d."""
    d = x.shape[-1]
    r = d if rot_dim is None else rot_dim
    xr, xp = x[..., :r], x[..., r:]
    c, s = cos[None, :, None, :r], sin[None, :, None, :r]
    out = xr * c + rotate_half(xr) * s
    return np.concatenate([out, xp], axis=-1) if r < d else out


def rope_bwd(dy, cos, sin, rot_dim=None):
    d = dy.shape[-1]
    r = d if rot_dim is None else rot_dim
    dr, dp = dy[..., :r], dy[..., r:]
    c, s = cos[None, :, None, :r], sin[None, :, None, :r]
    t = dr * s
    h = r // 2
    dx = dr * c + np.concatenate([t[..., h:], -t[..., :h]], axis=-1)
    return np.concatenate([dx, dp], axis=-1) if r < d else dx
