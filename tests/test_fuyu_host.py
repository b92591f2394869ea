"""Config C5 (OtterHD / Fuyu): otter_amd/fuyu.py's plain path on the CPU against the fixture produced by the reference's own
FuyuForCausalLM (oracle/gen_golden_fuyu.py -> tests/golden/fuyu_tiny.npz): state-dict keys, logits, loss, every gradient;
plus the gather_continuous_embeddings error contract and cached decoding.  The bf16 HIP path is checked on the GPU against the
same fixture (tests/test_gpu_modules.py)."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle.gen_golden_fuyu import SEED, tiny_fuyu_batch, tiny_fuyu_config
from tests import _golden as G


def _model():
    from otter_amd.fuyu import FuyuForCausalLM

    model = FuyuForCausalLM(tiny_fuyu_config())
    m = G.meta()["fuyu_tiny"]
    assert sorted(model.state_dict()) == m["keys"]
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == m["shapes"]
    sd = synth.state_dict_for(SEED, {k: tuple(s) for k, s in m["shapes"].items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return model


def test_forward_backward_match_reference_fixture():
    model = _model().train()
    gold = G.load("fuyu_tiny")
    ids, patches, idx, mask, labels = tiny_fuyu_batch()
    out = model(input_ids=torch.from_numpy(ids), image_patches=torch.from_numpy(patches), image_patches_indices=torch.from_numpy(idx),
                attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels))
    valid = mask.astype(bool)
    assert G.rel_err(out.logits.detach().numpy()[valid], gold["logits"][valid]) < 1e-4
    assert abs(float(out.loss) - float(gold["loss"])) < 1e-5 * abs(float(gold["loss"]))
    out.loss.backward()
    grads = {k: p.grad.detach().numpy() for k, p in model.named_parameters()}
    G.check_grads(gold, grads, 2e-4)


def test_patch_list_input_and_error_contract():
    """The reference's processor hands image_patches over as a list of [1, n_i, patch_dim] tensors (modeling_fuyu.py:125)."""
    model = _model().eval()
    ids, patches, idx, mask, _ = tiny_fuyu_batch()
    t = torch.from_numpy
    with torch.no_grad():
        a = model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(idx), attention_mask=t(mask)).logits
        b = model(input_ids=t(ids), image_patches=[t(patches[i:i + 1]) for i in range(2)], image_patches_indices=t(idx), attention_mask=t(mask)).logits
    assert torch.equal(a, b)
    bad = idx.copy()
    bad[0, 10:18] = np.arange(8)            # more patch positions than patches
    with pytest.raises(ValueError, match="Number of continuous embeddings"):
        model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(bad))
    with pytest.raises(ValueError):
        model()


def test_cached_decode_equals_full_forward():
    model = _model().eval()
    ids, patches, idx, _, _ = tiny_fuyu_batch()
    t = torch.from_numpy
    with torch.no_grad():
        full = model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(idx)).logits
        out = model(input_ids=t(ids[:, :12]), image_patches=t(patches), image_patches_indices=t(idx[:, :12]), use_cache=True)
        past, steps = out.past_key_values, [out.logits]
        for k in range(12, ids.shape[1]):
            out = model(input_ids=t(ids[:, k:k + 1]), past_key_values=past, use_cache=True)
            past = out.past_key_values
            steps.append(out.logits)
    inc = torch.cat(steps, 1)
    assert float((inc - full).abs().max() / full.abs().max()) < 1e-5


def _greedy_reference(model, ids, patches, idx, n_new):
    """Plain argmax loop over full forwards (no cache): what transformers' greedy generate computes for this model."""
    t = torch.from_numpy
    ids_t, idx_t = t(ids), t(idx)
    with torch.no_grad():
        for _ in range(n_new):
            logits = model(input_ids=ids_t, image_patches=t(patches), image_patches_indices=idx_t, use_cache=False).logits
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            ids_t = torch.cat([ids_t, nxt], 1)
            idx_t = torch.cat([idx_t, torch.full_like(nxt, -1)], 1)
    return ids_t


@pytest.mark.parametrize("use_cache", [False, True])
def test_fuyu_generate_greedy(use_cache):
    """FuyuForCausalLM.generate(**processor_outputs, max_new_tokens=) -- the OtterHD inference call
    (pipeline/demos/demo_models.py:171) -- in both decode modes, token-exact against the plain argmax loop; the image patches
    are consumed on the prompt pass only."""
    model = _model().eval()
    ids, patches, idx, mask, _ = tiny_fuyu_batch()
    P = 14
    want = _greedy_reference(model, ids[:, :P], patches, idx[:, :P], 5)
    t = torch.from_numpy
    got = model.generate(input_ids=t(ids[:, :P]), image_patches=t(patches), image_patches_indices=t(idx[:, :P]), max_new_tokens=5,
                         use_cache=use_cache, eos_token_id=-1)
    assert torch.equal(got, want)
    prep = model.prepare_inputs_for_generation(t(ids), past_key_values=((None, None),), image_patches=t(patches), image_patches_indices=t(idx))
    assert prep["input_ids"].shape[1] == 1 and prep["image_patches"] is None and prep["image_patches_indices"] is None


def test_fuyu_generate_beams_and_out_of_range_indices():
    model = _model().eval()
    ids, patches, idx, _, _ = tiny_fuyu_batch()
    t = torch.from_numpy
    P = 14
    a = model.generate(input_ids=t(ids[:, :P]), image_patches=t(patches), image_patches_indices=t(idx[:, :P]), max_new_tokens=4, num_beams=3,
                       use_cache=True, eos_token_id=-1)
    b = model.generate(input_ids=t(ids[:, :P]), image_patches=t(patches), image_patches_indices=t(idx[:, :P]), max_new_tokens=4, num_beams=3,
                       use_cache=False, eos_token_id=-1)
    assert a.shape == (ids.shape[0], P + 4) and torch.equal(a, b)
    # an index >= the number of patches is an IndexError (the reference's advanced indexing raises), never a silent read
    bad = idx.copy()
    bad[0, np.argmax(bad[0] >= 0)] = patches.shape[1]
    with pytest.raises(IndexError, match="continuous embeddings"):
        model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(bad))


def test_persimmon_use_cache_follows_config():
    """past_key_values are returned whenever config.use_cache is true (HF / reference default), in train mode too."""
    model = _model().train()
    ids, patches, idx, _, _ = tiny_fuyu_batch()
    t = torch.from_numpy
    out = model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(idx))
    assert bool(model.config.text_config.use_cache) and out.past_key_values is not None and len(out.past_key_values) == model.config.text_config.num_hidden_layers
    assert model(input_ids=t(ids), image_patches=t(patches), image_patches_indices=t(idx), use_cache=False).past_key_values is None
    txt = model.language_model.generate(t(ids[:, :6]), max_new_tokens=3, eos_token_id=-1)
    assert txt.shape == (ids.shape[0], 9)


def test_host_reference_composition_reproduces_the_reference_fixture():
    """tests/_host_ref.fuyu_forward (transformers' PersimmonForCausalLM + the restated patch-embedding scatter: the host-side reference of the
    full-depth C5 parity test, tests/test_gpu_full_model_c4_c5.py) against the fixture produced by the REFERENCE's own FuyuForCausalLM."""
    from tests import _host_ref as H

    cfg = tiny_fuyu_config()
    m = G.meta()["fuyu_tiny"]
    sd = synth.state_dict_for(SEED, {k: tuple(s) for k, s in m["shapes"].items()})
    gold = G.load("fuyu_tiny")
    ids, patches, idx, mask, labels = tiny_fuyu_batch()
    tc = cfg.text_config.to_dict() if hasattr(cfg.text_config, "to_dict") else dict(cfg.text_config)
    keep = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "max_position_embeddings", "qk_layernorm",
            "partial_rotary_factor", "hidden_act", "layer_norm_eps", "rope_theta", "tie_word_embeddings", "rope_parameters")
    hf = H.new_hf_persimmon({k: tc[k] for k in keep if k in tc})
    out = H.fuyu_forward(hf, {k: torch.from_numpy(v) for k, v in sd.items()}, ids, patches, idx, labels, attention_mask=mask)
    valid = mask.astype(bool)
    assert G.rel_err(out["logits"][valid], gold["logits"][valid]) < 2e-4
    assert abs(out["loss"] - float(gold["loss"])) < 2e-4 * abs(float(gold["loss"]))
