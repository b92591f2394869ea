"""Deterministic synthetic weights / inputs shared by the golden generator, the tests and smoke().

TEST INFRASTRUCTURE (lives under oracle/): the product never imports this.  Every tensor is a pure function of
(seed, name, shape), so a fixture only has to store *outputs*: the generator loads these weights into the
reference modules, the tests load the very same weights into the oracle and into the HIP modules.
"""
from __future__ import annotations

import zlib

import numpy as np


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def tensor(seed: int, name: str, shape, scale=1.0, mean=0.0, dtype=np.float32) -> np.ndarray:
    return (mean + scale * _rng(seed, name).standard_normal(tuple(shape))).astype(dtype)


def param_for(seed: int, key: str, shape) -> np.ndarray:
    """Synthetic value for one state-dict entry, chosen by the entry's role (key suffix)."""
    leaf = key.split(".")[-1]
    if leaf in ("attn_gate", "ff_gate"):
        # gates are zero-initialised in the reference (modeling_otter.py:362,371): identity block.  Parity tests
        # MUST use non-zero gates (SURVEY.md headline fact 5).
        v = 0.35 + 0.4 * _rng(seed, key).random(tuple(shape))
        sign = 1.0 if (zlib.crc32(key.encode()) & 1) else -1.0
        return (sign * v).astype(np.float32)
    if leaf == "bias":
        return tensor(seed, key, shape, 0.1)
    if leaf == "weight" and len(shape) == 1:
        return tensor(seed, key, shape, 0.1, 1.0)
    if leaf in ("latents", "frame_embs", "media_time_embs", "class_embedding"):
        return tensor(seed, key, shape, 0.5)
    if "position_embedding" in key:
        return tensor(seed, key, shape, 0.1)
    if "wte" in key or "embed_tokens" in key:
        return tensor(seed, key, shape, 0.3)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return tensor(seed, key, shape, 1.0 / np.sqrt(max(fan_in, 1)))


def state_dict_for(seed: int, shapes: dict) -> dict:
    """shapes: {key: shape}.  Returns {key: float32 ndarray}."""
    return {k: param_for(seed, k, tuple(s)) for k, s in shapes.items()}


# ---- shape tables for the modules on the path (names = reference state-dict keys) ----


def perceiver_shapes(pre, dim, depth, heads=8, dim_head=64, num_latents=64, ff_mult=4, max_num_frames=None,
                     max_num_media=None):
    inner = heads * dim_head
    s = {pre + "latents": (num_latents, dim)}
    if max_num_frames is not None:
        s[pre + "frame_embs"] = (max_num_frames, dim)
    if max_num_media is not None:
        s[pre + "media_time_embs"] = (max_num_media, 1, dim)
    for i in range(depth):
        lp = pre + f"layers.{i}."
        s[lp + "norm_media.weight"] = (dim,)
        s[lp + "norm_media.bias"] = (dim,)
        s[lp + "norm_latents.weight"] = (dim,)
        s[lp + "norm_latents.bias"] = (dim,)
        s[lp + "to_q.weight"] = (inner, dim)
        s[lp + "to_kv.weight"] = (2 * inner, dim)
        s[lp + "to_out.weight"] = (dim, inner)
        s[lp + "feed_forward.0.weight"] = (dim,)
        s[lp + "feed_forward.0.bias"] = (dim,)
        s[lp + "feed_forward.1.weight"] = (ff_mult * dim, dim)
        s[lp + "feed_forward.3.weight"] = (dim, ff_mult * dim)
    s[pre + "norm.weight"] = (dim,)
    s[pre + "norm.bias"] = (dim,)
    return s


def gated_xattn_shapes(pre, dim, dim_visual, heads=8, dim_head=64, ff_mult=4):
    inner = heads * dim_head
    return {
        pre + "attn_gate": (1,),
        pre + "ff_gate": (1,),
        pre + "attn.norm.weight": (dim,),
        pre + "attn.norm.bias": (dim,),
        pre + "attn.to_q.weight": (inner, dim),
        pre + "attn.to_kv.weight": (2 * inner, dim_visual),
        pre + "attn.to_out.weight": (dim, inner),
        pre + "feed_forward.0.weight": (dim,),
        pre + "feed_forward.0.bias": (dim,),
        pre + "feed_forward.1.weight": (ff_mult * dim, dim),
        pre + "feed_forward.3.weight": (dim, ff_mult * dim),
    }


def mpt_block_shapes(pre, d_model, expansion=4):
    return {
        pre + "norm_1.weight": (d_model,),
        pre + "attn.Wqkv.weight": (3 * d_model, d_model),
        pre + "attn.out_proj.weight": (d_model, d_model),
        pre + "norm_2.weight": (d_model,),
        pre + "ffn.up_proj.weight": (expansion * d_model, d_model),
        pre + "ffn.down_proj.weight": (d_model, expansion * d_model),
    }


def clip_shapes(pre, hidden, layers, inter, image, patch):
    npos = (image // patch) ** 2 + 1
    s = {
        pre + "embeddings.class_embedding": (hidden,),
        pre + "embeddings.patch_embedding.weight": (hidden, 3, patch, patch),
        pre + "embeddings.position_embedding.weight": (npos, hidden),
        pre + "pre_layrnorm.weight": (hidden,),
        pre + "pre_layrnorm.bias": (hidden,),
        pre + "post_layernorm.weight": (hidden,),
        pre + "post_layernorm.bias": (hidden,),
    }
    for i in range(layers):
        lp = pre + f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[lp + f"self_attn.{nm}.weight"] = (hidden, hidden)
            s[lp + f"self_attn.{nm}.bias"] = (hidden,)
        s[lp + "layer_norm1.weight"] = (hidden,)
        s[lp + "layer_norm1.bias"] = (hidden,)
        s[lp + "layer_norm2.weight"] = (hidden,)
        s[lp + "layer_norm2.bias"] = (hidden,)
        s[lp + "mlp.fc1.weight"] = (inter, hidden)
        s[lp + "mlp.fc1.bias"] = (inter,)
        s[lp + "mlp.fc2.weight"] = (hidden, inter)
        s[lp + "mlp.fc2.bias"] = (hidden,)
    return s


def otter_mpt_shapes(n_layers, d_model, vocab, every, clip_hidden=1024, clip_layers=1, clip_inter=64, image=28,
                     patch=14, max_num_frames=None, perceiver_depth=6):
    s = {}
    s.update(clip_shapes("vision_encoder.vision_model.", clip_hidden, clip_layers, clip_inter, image, patch))
    s.update(perceiver_shapes("perceiver.", clip_hidden, perceiver_depth, max_num_frames=max_num_frames))
    LP = "lang_encoder.transformer."
    s[LP + "wte.weight"] = (vocab, d_model)
    for i in range(n_layers):
        if (i + 1) % every == 0:
            s.update(gated_xattn_shapes(LP + f"blocks.{i}.gated_cross_attn_layer.", d_model, clip_hidden))
        s.update(mpt_block_shapes(LP + f"blocks.{i}.decoder_layer.", d_model))
    s[LP + "norm_f.weight"] = (d_model,)
    return s


# the tiny full-model configuration used by golden case "otter_tiny" (and by smoke())
TINY = dict(n_layers=4, d_model=64, n_heads=4, vocab=128, every=2, max_seq_len=64, clip_heads=16, image=28,
            patch=14, clip_layers=1, clip_inter=64, media_token_id=125, eoc_token_id=124, answer_token_id=126,
            pad_token_id=127)


def tiny_batch(seed=0, B=2, T=16, T_img=1, F=1, image=28):
    """(vision_x, input_ids, attention_mask, labels) for the tiny model.  Position 0 BOS-ish id, <image> at 1."""
    r = _rng(seed, "tiny_batch")
    vision_x = r.standard_normal((B, T_img, F, 3, image, image)).astype(np.float32)
    ids = r.integers(0, 120, size=(B, T)).astype(np.int64)
    ids[:, 1] = TINY["media_token_id"]
    ids[0, 0] = 3  # a text token before the first image: text_time == 0 row
    ids[:, 5] = TINY["answer_token_id"]
    ids[:, T - 2] = TINY["eoc_token_id"]
    if T_img > 1:
        ids[:, 8] = TINY["media_token_id"]
    mask = np.ones((B, T), dtype=np.int64)
    labels = np.full((B, T), -100, dtype=np.int64)
    labels[:, 6:T - 1] = ids[:, 6:T - 1]
    return vision_x, ids, mask, labels


# ---- the C2-width same-precision comparator case (oracle/gen_golden_bf16ref.py <-> tests/test_gpu_modules.py) ----


def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 (ties to even) -> fp32, in numpy (bit-identical to torch's .to(bfloat16).float() for finite values)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


C2REF = dict(seed=53, T=512, row_step=32, dim=4096, dim_visual=1024)


def c2_bf16ref_case():
    """(state dict, x [1,T,D], media [1,1,64,Dv], R [1,T,D], media_locations [1,T]) of the one-sample gated cross-attention block at the
    benchmark's width: bf16-representable weights / inputs, so that a fp32 run, a bf16-autocast run and the HIP bf16 path see identical
    numbers and only activation rounding differs."""
    c = C2REF
    sd = state_dict_for(c["seed"], gated_xattn_shapes("blk.", c["dim"], c["dim_visual"]))
    sd = {k: (bf16_round(v) if v.ndim == 2 else v) for k, v in sd.items()}
    x = bf16_round(tensor(c["seed"], "c2ref.x", (1, c["T"], c["dim"])))
    media = bf16_round(tensor(c["seed"], "c2ref.m", (1, 1, 64, c["dim_visual"])))
    R = tensor(c["seed"], "c2ref.R", (1, c["T"], c["dim"]))
    ml = np.zeros((1, c["T"]), dtype=bool)
    ml[0, 1] = True          # <image> at position 1 as in the benchmark batch: row 0 precedes it (zeroed attention row)
    return sd, x, media, R, ml
