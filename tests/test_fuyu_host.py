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
