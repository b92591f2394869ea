"""Config C4's decoder host (otter_amd/llama.py) against the third-party class the reference uses for it
(transformers.LlamaForCausalLM, modeling_otter.py:54,759-767), same weights, CPU fp32: logits, loss, right padding,
grouped-query heads, and KV-cache decoding == full re-forward.  The bf16 HIP path of the same module is checked on the GPU
(tests/test_gpu_modules.py) against THIS path."""
import numpy as np
import pytest
import torch


def _pair(n_kv=4, seed=0):
    from transformers import LlamaConfig
    from transformers import LlamaForCausalLM as HFLlama

    from otter_amd.llama import LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=n_kv,
                      vocab_size=97, max_position_embeddings=64, rms_norm_eps=1e-6, tie_word_embeddings=False)
    torch.manual_seed(seed)
    ref = HFLlama(cfg).eval()
    mine = LlamaForCausalLM(cfg).eval()
    missing, unexpected = mine.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and all("rotary_emb" in k for k in unexpected), (missing, unexpected)
    assert sorted(mine.state_dict()) == sorted(k for k in ref.state_dict() if "rotary_emb" not in k)
    return cfg, ref, mine


@pytest.mark.parametrize("n_kv", [4, 2])
def test_logits_and_loss_match_transformers(n_kv):
    cfg, ref, mine = _pair(n_kv)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg.vocab_size, (3, 17), generator=g)
    mask = torch.ones_like(ids)
    mask[1, 12:] = 0          # right padding
    labels = ids.clone()
    labels[:, :5] = -100
    labels[mask == 0] = -100
    with torch.no_grad():
        a = ref(input_ids=ids, attention_mask=mask, labels=labels)
        b = mine(input_ids=ids, attention_mask=mask, labels=labels)
    valid = mask.bool()
    err = (a.logits - b.logits)[valid].abs().max() / a.logits[valid].abs().max()
    assert float(err) < 1e-5, float(err)
    assert abs(float(a.loss) - float(b.loss)) < 1e-5 * abs(float(a.loss))


def test_dgrad_matches_transformers():
    cfg, ref, mine = _pair()
    g = torch.Generator().manual_seed(2)
    e = torch.randn(2, 9, cfg.hidden_size, generator=g)
    ea, eb = e.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref(inputs_embeds=ea).logits.square().sum().backward()
    mine(inputs_embeds=eb).logits.square().sum().backward()
    assert float((ea.grad - eb.grad).abs().max() / ea.grad.abs().max()) < 1e-5


def test_cached_decode_equals_full_forward():
    cfg, _, mine = _pair()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg.vocab_size, (2, 11), generator=g)
    with torch.no_grad():
        full = mine(input_ids=ids).logits
        out = mine(input_ids=ids[:, :6], use_cache=True)
        past, steps = out.past_key_values, [out.logits]
        assert len(past) == cfg.num_hidden_layers and past[0][0].shape == (2, cfg.num_key_value_heads, 6, 16)
        for t in range(6, 11):
            out = mine(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
            past = out.past_key_values
            steps.append(out.logits)
    inc = torch.cat(steps, dim=1)
    assert float((inc - full).abs().max() / full.abs().max()) < 1e-5


def test_position_ids_shift_like_transformers():
    """Left-padded decode passes explicit position_ids (HF prepare_inputs_for_generation): same result as transformers."""
    cfg, ref, mine = _pair()
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, cfg.vocab_size, (2, 8), generator=g)
    mask = torch.ones_like(ids)
    mask[0, :3] = 0
    pos = (mask.long().cumsum(-1) - 1).masked_fill(mask == 0, 1)
    with torch.no_grad():
        a = ref(input_ids=ids, attention_mask=mask, position_ids=pos).logits
        b = mine(input_ids=ids, attention_mask=mask, position_ids=pos).logits
    valid = mask.bool()
    assert float((a - b)[valid].abs().max() / a[valid].abs().max()) < 1e-5


def test_host_reference_composition_reproduces_the_reference_fixture():
    """tests/_host_ref.py (transformers' LlamaForCausalLM + the numpy oracle's CLIP / perceiver / gated blocks hooked in front of the
    decoder layers -- the host-side reference of the full-size C4 parity test) against the REFERENCE's own OtterForConditionalGeneration
    over a LLaMA host with 3 video frames (tests/golden/otter_tiny_llama.npz, oracle/gen_golden.py::case_otter_tiny_llama): logits and loss."""
    from oracle import otter_oracle as O
    from oracle import synth
    from oracle.gen_golden import tiny_llama_configs
    from tests import _golden as G
    from tests import _host_ref as H

    m = G.meta()["otter_tiny_llama"]
    gold = G.load("otter_tiny_llama")
    sd = synth.state_dict_for(m["seed"], {k: tuple(v) for k, v in m["state_dict_shapes"].items()})
    text_cfg, _ = tiny_llama_configs()
    hf = H.new_hf_llama(text_cfg)
    H.load_decoder_weights(hf, {k[len("lang_encoder."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("lang_encoder.")})
    t = synth.TINY
    spec = O.OtterSpec(4, 64, 4, 64, 2, t["media_token_id"], t["clip_heads"], t["patch"])
    vision_x, ids, mask, labels = synth.tiny_batch(m["seed"], F=3)
    out = H.otter_llama_forward(hf, sd, spec, vision_x, ids, labels)
    assert G.rel_err(out["logits"], gold["logits"]) < 2e-4
    assert abs(out["loss"] - float(gold["loss"])) < 2e-4 * abs(float(gold["loss"]))


def test_host_reference_backward_reproduces_the_reference_gradients():
    """tests/_host_ref.otter_llama_forward_backward -- the host side of the full-size C4 TRAINING-STEP parity test -- against the gradients of
    the reference's own tiny OTTER-over-LLaMA model (tests/golden/otter_tiny_llama.npz: every trainable tensor, incl. frame embeddings, the
    input embedding and lm_head)."""
    from oracle import otter_oracle as O
    from oracle import synth
    from oracle.gen_golden import tiny_llama_configs
    from tests import _golden as G
    from tests import _host_ref as H

    m = G.meta()["otter_tiny_llama"]
    gold = G.load("otter_tiny_llama")
    sd = synth.state_dict_for(m["seed"], {k: tuple(v) for k, v in m["state_dict_shapes"].items()})
    text_cfg, _ = tiny_llama_configs()
    hf = H.new_hf_llama(text_cfg)
    H.load_decoder_weights(hf, {k[len("lang_encoder."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("lang_encoder.")})
    t = synth.TINY
    spec = O.OtterSpec(4, 64, 4, 64, 2, t["media_token_id"], t["clip_heads"], t["patch"])
    vision_x, ids, mask, labels = synth.tiny_batch(m["seed"], F=3)
    out = H.otter_llama_forward_backward(hf, sd, spec, vision_x, ids, labels)
    assert abs(out["loss"] - float(gold["loss"])) < 2e-4 * abs(float(gold["loss"]))
    assert sorted(out["grads"]) == m["trainable"]
    G.check_grads(gold, out["grads"], 5e-4)
