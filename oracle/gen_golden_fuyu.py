"""Golden fixture for config C5 (OtterHD / Fuyu): runs the REFERENCE's own FuyuForCausalLM
(/root/reference/src/otter_ai/models/fuyu/modeling_fuyu.py; flash_attn is absent here, so its `except ImportError` branch
takes transformers' PersimmonForCausalLM: the decoder arithmetic is third-party, exactly as SURVEY.md section 8c records for
this config) on a tiny configuration with name-seeded weights (oracle/synth.py) and stores inputs' seeds, logits, loss and every
parameter gradient (full for small tensors, fingerprints for large ones) in tests/golden/fuyu_tiny.npz.

TEST INFRASTRUCTURE: runs only in the build container (needs /root/reference); the product never imports it.
Usage: python oracle/gen_golden_fuyu.py"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402
from tests._golden import summarize  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/src/otter_ai/models/fuyu"
SEED = 21


def tiny_fuyu_config():
    from transformers import FuyuConfig

    text = dict(model_type="persimmon", vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                max_position_embeddings=128, qk_layernorm=True, partial_rotary_factor=0.5, hidden_act="relu2", layer_norm_eps=1e-5,
                rope_theta=25000.0, tie_word_embeddings=False)
    return FuyuConfig(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, patch_size=6,
                      num_channels=3, max_position_embeddings=128, image_size=24, text_config=text, tie_word_embeddings=False)


def tiny_fuyu_batch(seed=SEED, B=2, S=20, n_patches=7, patch_dim=108, vocab=120):
    r = np.random.default_rng(seed)
    ids = r.integers(3, vocab, size=(B, S)).astype(np.int64)
    patches = synth.tensor(seed, "fuyu.patches", (B, n_patches, patch_dim))
    idx = np.full((B, S), -1, np.int64)
    idx[0, 1:1 + n_patches] = np.arange(n_patches)
    idx[1, 2:2 + n_patches - 2] = np.arange(n_patches - 2)          # fewer indices than patches (a truncated image)
    mask = np.ones((B, S), np.int64)
    mask[1, 16:] = 0                                                  # right padding
    labels = ids.copy()
    labels[idx >= 0] = -100
    labels[mask == 0] = -100
    labels[:, 0] = -100
    return ids, patches, idx, mask, labels


def import_reference_fuyu():
    pkg = types.ModuleType("ref_fuyu_pkg")
    pkg.__path__ = [REF]
    sys.modules["ref_fuyu_pkg"] = pkg
    spec = importlib.util.spec_from_file_location("ref_fuyu_pkg.modeling_fuyu", os.path.join(REF, "modeling_fuyu.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["ref_fuyu_pkg.modeling_fuyu"] = m
    spec.loader.exec_module(m)
    return m


def main():
    torch.set_num_threads(8)
    mod = import_reference_fuyu()
    cfg = tiny_fuyu_config()
    model = mod.FuyuForCausalLM(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.state_dict_for(SEED, shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.train()
    ids, patches, idx, mask, labels = tiny_fuyu_batch()
    out = model(input_ids=torch.from_numpy(ids), image_patches=torch.from_numpy(patches), image_patches_indices=torch.from_numpy(idx),
                attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels))
    out.loss.backward()
    res = {"logits": out.logits.detach().numpy().astype(np.float32), "loss": np.float32(out.loss.detach())}
    for k, p in model.named_parameters():
        g = p.grad.detach().numpy()
        if g.size <= 4096:
            res["g:" + k] = g.astype(np.float32)
        else:
            res["gs:" + k] = summarize(g)
    np.savez_compressed(os.path.join(OUT, "fuyu_tiny.npz"), **res)
    with open(os.path.join(OUT, "meta.json")) as f:
        meta = json.load(f)
    meta["fuyu_tiny"] = {"seed": SEED, "keys": sorted(shapes), "shapes": {k: list(v) for k, v in shapes.items()},
                         "reference": "src/otter_ai/models/fuyu/modeling_fuyu.py FuyuForCausalLM (transformers PersimmonForCausalLM fallback)",
                         "transformers": __import__("transformers").__version__}
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote fuyu_tiny: loss %.6f, %d tensors" % (float(out.loss), len(res)))


if __name__ == "__main__":
    main()
