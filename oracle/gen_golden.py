"""Generate tests/golden/*.npz by running the REFERENCE's own PyTorch modules on CPU.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
Nothing at test/bench/smoke time reads /root/reference -- the committed fixtures travel instead.

What pins what:
  * module cases (perceiver_*, xattn_*) run src/otter_ai/models/otter/modeling_otter.py classes directly;
  * otter_tiny runs a complete OtterForConditionalGeneration (MPT text config, 1-layer CLIP) incl. loss.backward()
    and a hand-written greedy loop over model.lang_encoder (HF generate() of the pinned transformers==4.35.1 is not
    importable here -- SURVEY.md section 8c) in both decode modes;
  * llama_* cases use transformers' LlamaRMSNorm / apply_rotary_pos_emb (third-party arithmetic for config C4);
  * mpt_attn runs src/otter_ai/models/mpt/attention.py::scaled_multihead_dot_product_attention + build_alibi_bias
    (causal / ALiBi / key padding) -- the checker of the decoder-host flash kernels.
Weights and inputs are pure functions of (seed, name, shape) -- see oracle/synth.py -- so fixtures hold outputs only.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402

REF = os.environ.get("OTTER_REF_ROOT", "/root/reference")   # (a staged scratch copy on the GPU box: tools/stage_reference_loop.sh stage-models)
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    peft = types.ModuleType("peft")

    class _Stub:
        def __init__(self, *a, **k):
            pass

    peft.LoraConfig = _Stub
    peft.TaskType = types.SimpleNamespace(CAUSAL_LM="CAUSAL_LM")
    peft.get_peft_model = lambda m, c: m
    sys.modules.setdefault("peft", peft)
    sys.path.insert(0, REF)
    from src.otter_ai.models.otter import modeling_otter as mo  # type: ignore

    return mo


def canonical(key: str) -> str:
    """State-dict key as the reference's pinned transformers==4.35.1 spells it.  transformers 5.x (this container)
    flattened CLIPVisionModel ('vision_encoder.embeddings...'); the pinned version -- and therefore every Otter
    checkpoint -- has 'vision_encoder.vision_model.embeddings...'."""
    if key.startswith("vision_encoder.") and not key.startswith("vision_encoder.vision_model."):
        return "vision_encoder.vision_model." + key[len("vision_encoder."):]
    return key


def load_synth(module: torch.nn.Module, seed: int, prefix: str = ""):
    sd = module.state_dict()
    new = {}
    shapes = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            continue
        ck = canonical(prefix + k)
        new[k] = torch.from_numpy(synth.param_for(seed, ck, tuple(v.shape)))
        shapes[ck] = tuple(v.shape)
    missing = module.load_state_dict(new, strict=False)
    return shapes, missing


def summarize(g: np.ndarray, nsamp=256):
    """Compact fingerprint of a large gradient: [sum, abs-sum, l2] + a strided sample (see tests/_golden.py)."""
    f = g.reshape(-1).astype(np.float64)
    step = max(1, f.size // nsamp)
    return np.concatenate([[f.sum(), np.abs(f).sum(), np.sqrt((f * f).sum())], f[::step][:nsamp]])


def put_grad(out: dict, key: str, g: np.ndarray, full_limit=8192):
    if g.size <= full_limit:
        out["g:" + key] = g.copy()
    else:
        out["gs:" + key] = summarize(g)


def grads_of(module, prefix=""):
    return {prefix + n: p.grad.detach().numpy().copy() for n, p in module.named_parameters() if p.grad is not None}


def case_perceiver(mo, name, seed, dim, depth, num_latents, xshape, max_num_frames=None):
    torch.manual_seed(0)
    m = mo.OtterPerceiverResampler(dim=dim, depth=depth, num_latents=num_latents, max_num_frames=max_num_frames)
    shapes, _ = load_synth(m, seed, "perceiver.")
    x = torch.from_numpy(synth.tensor(seed, name + ".x", xshape)).requires_grad_(True)
    y = m(x)
    R = torch.from_numpy(synth.tensor(seed, name + ".R", tuple(y.shape)))
    (y * R).sum().backward()
    out = {"y": y.detach().numpy(), "dx": x.grad.numpy()}
    for k, v in grads_of(m, "perceiver.").items():
        put_grad(out, k, v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return {"seed": seed, "dim": dim, "depth": depth, "num_latents": num_latents, "xshape": list(xshape),
            "max_num_frames": max_num_frames, "keys": sorted(shapes)}


XATTN_LOCS = {
    # (B=2, T=24); T_img=3
    "base": [[2, 9, 15], [0, 5]],
    "overflow": [[1, 4, 8, 20], [0, 3, 6, 9]],  # 4 <image> tokens but T_img = 3 -> fully masked (uniform) rows
}


def media_locations(kind, B=2, T=24):
    ml = np.zeros((B, T), dtype=bool)
    for b, pos in enumerate(XATTN_LOCS[kind]):
        ml[b, pos] = True
    return ml


def case_xattn(mo, name, seed, loc_kind, attend_previous=True, immediate=True, dim=128, dim_visual=96, T=24, T_img=3,
               n=8):
    torch.manual_seed(0)
    m = mo.OtterGatedCrossAttentionBlock(dim=dim, dim_visual=dim_visual, only_attend_immediate_media=immediate)
    load_synth(m, seed, "blk.")
    x = torch.from_numpy(synth.tensor(seed, "xattn.x", (2, T, dim))).requires_grad_(True)
    media = torch.from_numpy(synth.tensor(seed, "xattn.media", (2, T_img, n, dim_visual))).requires_grad_(True)
    ml = None if loc_kind is None else torch.from_numpy(media_locations(loc_kind, 2, T))
    y = m(x, media, media_locations=ml, attend_previous=attend_previous)
    R = torch.from_numpy(synth.tensor(seed, "xattn.R", tuple(y.shape)))
    (y * R).sum().backward()
    out = {"y": y.detach().numpy(), "dx": x.grad.numpy(), "dmedia": media.grad.numpy()}
    # the attention sub-module alone as well (a4)
    a = m.attn(x.detach(), media.detach(), media_locations=ml, attend_previous=attend_previous)
    out["attn_y"] = a.detach().numpy()
    for k, v in grads_of(m, "blk.").items():
        put_grad(out, k, v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return {"seed": seed, "loc_kind": loc_kind, "attend_previous": attend_previous, "immediate": immediate,
            "dim": dim, "dim_visual": dim_visual, "T": T, "T_img": T_img, "n": n}


class StubTokenizer:
    """Stands in for AutoTokenizer.from_pretrained('mosaicml/mpt-7b-instruct') (no network)."""

    eos_token = "<|endoftext|>"

    def __init__(self):
        t = synth.TINY
        self.map = {"<|endofchunk|>": t["eoc_token_id"], "<image>": t["media_token_id"],
                    "<answer>": t["answer_token_id"], "<PAD>": t["pad_token_id"], "<|endoftext|>": 0}
        self.pad_token = None

    def add_special_tokens(self, d):
        if "pad_token" in d:
            self.pad_token = d["pad_token"]
        return 0

    def encode(self, s):
        return [self.map[s]]

    def __len__(self):
        return synth.TINY["vocab"]


def build_tiny_reference(mo, max_num_frames=None):
    from src.otter_ai.models.otter.configuration_otter import OtterConfig  # type: ignore

    t = synth.TINY
    mo.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: StubTokenizer())
    text_cfg = dict(architectures=["MPTForCausalLM"], d_model=t["d_model"], n_heads=t["n_heads"],
                    n_layers=t["n_layers"], expansion_ratio=4, max_seq_len=t["max_seq_len"], vocab_size=t["vocab"],
                    attn_config=dict(attn_type="multihead_attention", attn_pdrop=0.0, attn_impl="torch", qk_ln=False,
                                     clip_qkv=None, softmax_scale=None, prefix_lm=False, attn_uses_sequence_id=False,
                                     alibi=True, alibi_bias_max=8),
                    no_bias=True, tie_word_embeddings=True, hidden_size=t["d_model"], norm_type="low_precision_layernorm", init_device="cpu", use_cache=False,
                    resid_pdrop=0.0, emb_pdrop=0.0)
    vis_cfg = dict(hidden_size=1024, intermediate_size=t["clip_inter"], num_hidden_layers=t["clip_layers"],
                   num_attention_heads=t["clip_heads"], image_size=t["image"], patch_size=t["patch"],
                   hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=64)
    kw = {}
    if max_num_frames is not None:
        kw["max_num_frames"] = max_num_frames
    cfg = OtterConfig(vision_config=vis_cfg, text_config=text_cfg, cross_attn_every_n_layers=t["every"], **kw)
    model = mo.OtterForConditionalGeneration(cfg)
    return model


def case_otter_tiny(mo, name="otter_tiny", seed=7):
    model = build_tiny_reference(mo)
    model.eval()
    shapes, missing = load_synth(model, seed, "")
    vision_x, ids, mask, labels = synth.tiny_batch(seed)
    for p in model.parameters():
        p.requires_grad_(False)
    model.init_weights()
    out = model(vision_x=torch.from_numpy(vision_x), lang_x=torch.from_numpy(ids),
                attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels))
    out.loss.backward()
    res = {"logits": out.logits.detach().numpy(), "loss": np.array(out.loss.item(), dtype=np.float64)}
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    for n, p in model.named_parameters():
        if p.grad is not None:
            put_grad(res, n, p.grad.numpy())
    # perceiver output for the batch (what is conditioned onto the layers)
    with torch.no_grad():
        model._encode_vision_x(vision_x=torch.from_numpy(vision_x))
        res["vis"] = model.lang_encoder._get_decoder_layers()[0].vis_x.numpy().copy()
        # greedy decode, both modes
        for use_cache in (False, True):
            cur = torch.from_numpy(ids[:, :8].copy())
            past = None
            for _ in range(6):
                if use_cache:
                    if past is None:
                        o = model.lang_encoder(input_ids=cur, use_cache=True)
                    else:
                        o = model.lang_encoder(input_ids=cur[:, -1:], past_key_values=past, use_cache=True)
                    past = o.past_key_values
                else:
                    o = model.lang_encoder(input_ids=cur)
                nxt = o.logits[:, -1, :].argmax(-1, keepdim=True)
                cur = torch.cat([cur, nxt], dim=1)
            res["greedy_cache" if use_cache else "greedy_nocache"] = cur.numpy()
            if use_cache:
                res["greedy_cache_last_logits"] = o.logits[:, -1, :].numpy()
            else:
                res["greedy_nocache_last_logits"] = o.logits[:, -1, :].numpy()
        model.lang_encoder.clear_conditioned_layers()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    keys = {k: list(v) for k, v in shapes.items()}
    return {"seed": seed, "state_dict_shapes": keys, "trainable": trainable,
            "missing": [str(m) for m in missing.missing_keys]}


def tiny_llama_configs():
    t = synth.TINY
    text_cfg = dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=64, intermediate_size=128,
                    num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=4, vocab_size=t["vocab"],
                    max_position_embeddings=64, rms_norm_eps=1e-6, tie_word_embeddings=False, _name_or_path="tiny-llama")
    vis_cfg = dict(hidden_size=1024, intermediate_size=t["clip_inter"], num_hidden_layers=t["clip_layers"],
                   num_attention_heads=t["clip_heads"], image_size=t["image"], patch_size=t["patch"],
                   hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=64)
    return text_cfg, vis_cfg


def case_otter_tiny_llama(mo, name="otter_tiny_llama", seed=13):
    """Config-C4 composition at toy size: LLaMA decoder (HF class, as the reference uses) + video input (F=3 frames,
    max_num_frames=4 -> frame_embs) + gated cross-attention every 2 layers."""
    from src.otter_ai.models.otter.configuration_otter import OtterConfig  # type: ignore

    mo.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: StubTokenizer())
    text_cfg, vis_cfg = tiny_llama_configs()
    cfg = OtterConfig(vision_config=vis_cfg, text_config=dict(text_cfg), cross_attn_every_n_layers=2, max_num_frames=4)
    model = mo.OtterForConditionalGeneration(cfg)
    model.eval()
    shapes, missing = load_synth(model, seed, "")
    vision_x, ids, mask, labels = synth.tiny_batch(seed, F=3)
    for p in model.parameters():
        p.requires_grad_(False)
    model.init_weights()
    out = model(vision_x=torch.from_numpy(vision_x), lang_x=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                labels=torch.from_numpy(labels))
    out.loss.backward()
    res = {"logits": out.logits.detach().numpy(), "loss": np.array(out.loss.item(), dtype=np.float64)}
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    for n, p in model.named_parameters():
        if p.grad is not None:
            put_grad(res, n, p.grad.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    return {"seed": seed, "state_dict_shapes": {k: list(v) for k, v in shapes.items()}, "trainable": trainable}


def case_llama(name="llama_ops", seed=11):
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, apply_rotary_pos_emb

    D, S, H, d = 256, 12, 4, 64
    x = torch.from_numpy(synth.tensor(seed, "rms.x", (2, S, D))).requires_grad_(True)
    n = LlamaRMSNorm(D, eps=1e-6)
    with torch.no_grad():
        n.weight.copy_(torch.from_numpy(synth.tensor(seed, "rms.w", (D,), 0.1, 1.0)))
    y = n(x)
    R = torch.from_numpy(synth.tensor(seed, "rms.R", (2, S, D)))
    (y * R).sum().backward()
    res = {"rms_y": y.detach().numpy(), "rms_dx": x.grad.numpy(), "rms_dw": n.weight.grad.numpy()}
    q = torch.from_numpy(synth.tensor(seed, "rope.q", (2, H, S, d))).requires_grad_(True)
    k = torch.from_numpy(synth.tensor(seed, "rope.k", (2, H, S, d)))
    from oracle.otter_oracle import rope_tables

    cos, sin = rope_tables(S, d)
    cos_t, sin_t = torch.from_numpy(cos)[None], torch.from_numpy(sin)[None]  # [1,S,d]
    qe, ke = apply_rotary_pos_emb(q, k, cos_t, sin_t)
    Rq = torch.from_numpy(synth.tensor(seed, "rope.R", (2, H, S, d)))
    (qe * Rq).sum().backward()
    res.update(rope_q=qe.detach().numpy(), rope_k=ke.detach().numpy(), rope_dq=q.grad.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    return {"seed": seed, "D": D, "S": S, "H": H, "d": d}


def case_mpt_attn(name="mpt_attn", seed=17):
    """The decoder host's attention core exactly as the reference calls it (mpt/attention.py:22-84 with the ALiBi bias of
    :447-464, key padding and the causal mask): pins oracle.mpt_attention_core, the checker of the HIP flash kernels."""
    import_reference()
    from src.otter_ai.models.mpt.attention import build_alibi_bias, scaled_multihead_dot_product_attention  # type: ignore

    B, H, S, d = 2, 4, 40, 32
    lens = [40, 29]
    res = {}
    for tag, causal, alibi, pad in (("a", True, True, True), ("b", False, False, False), ("c", True, True, False)):
        q, k, v = (torch.from_numpy(synth.tensor(seed, f"attn.{tag}.{n}", (B, S, H * d))).requires_grad_(True) for n in "qkv")
        bias = build_alibi_bias(H, S, full=False, alibi_bias_max=8) if alibi else None
        kpm = None
        if pad:
            kpm = torch.zeros(B, S, dtype=torch.bool)
            for b, n in enumerate(lens):
                kpm[b, :n] = True
        out, _, _ = scaled_multihead_dot_product_attention(q, k, v, H, softmax_scale=1.0 / (d ** 0.5), attn_bias=bias,
                                                            key_padding_mask=kpm, is_causal=causal)
        R = torch.from_numpy(synth.tensor(seed, f"attn.{tag}.R", (B, S, H * d)))
        (out * R).sum().backward()
        res.update({f"{tag}_out": out.detach().numpy(), f"{tag}_dq": q.grad.numpy(), f"{tag}_dk": k.grad.numpy(), f"{tag}_dv": v.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    return {"seed": seed, "B": B, "H": H, "S": S, "d": d, "lens": lens,
            "cases": {"a": [True, True, True], "b": [False, False, False], "c": [True, True, False]}}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "mpt_attn":   # add / refresh just this case
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["mpt_attn"] = case_mpt_attn()
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print("wrote mpt_attn")
        return
    mo = import_reference()
    meta = {}
    meta["perceiver_image"] = case_perceiver(mo, "perceiver_image", 1, 128, 2, 16, (2, 2, 1, 10, 128))
    meta["perceiver_video"] = case_perceiver(mo, "perceiver_video", 2, 128, 2, 16, (1, 1, 3, 10, 128), max_num_frames=4)
    meta["xattn_base"] = case_xattn(mo, "xattn_base", 3, "base")
    meta["xattn_overflow"] = case_xattn(mo, "xattn_overflow", 3, "overflow")
    meta["xattn_nomask"] = case_xattn(mo, "xattn_nomask", 3, None)
    meta["xattn_noprev"] = case_xattn(mo, "xattn_noprev", 3, "base", attend_previous=False)
    meta["xattn_ge"] = case_xattn(mo, "xattn_ge", 3, "base", immediate=False)
    meta["otter_tiny"] = case_otter_tiny(mo)
    meta["otter_tiny_llama"] = case_otter_tiny_llama(mo)
    meta["llama_ops"] = case_llama()
    meta["mpt_attn"] = case_mpt_attn()
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
