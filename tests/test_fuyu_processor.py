"""otter_amd.processing_fuyu.FuyuProcessor: the reference's OtterHD collate (pipeline/mimicit_utils/mimicit_dataset.py:497-505)
calls `fuyu_processor(text=, images=)`, `.get_labels(...)` and `.find_and_remove_tokens(...)`; the library class the shim used to re-export
has neither method and pads on the other side.  Where /root/reference exists (the build container) the reference's own methods are the
expected values; everywhere the hand-worked cases below run."""
import importlib.util
import os
import types

import pytest
import torch

from otter_amd.processing_fuyu import FuyuProcessor

REF = "/root/reference/src/otter_ai/models/fuyu/processing_fuyu.py"
EOS, SPECIAL = 7, 71122


def _proc():
    p = FuyuProcessor.__new__(FuyuProcessor)          # no tokenizer / image-processor files offline: the methods under test need only these
    p.tokenizer = types.SimpleNamespace(eos_token_id=EOS)
    p.pad_token_id, p.dummy_image_index = EOS, -1
    return p


def _ref_proc():
    if not os.path.exists(REF):
        return None
    import sys
    name = "transformers.models.fuyu.image_processing_fuyu"    # needs torchvision (absent here); the reference only takes FuyuBatchFeature from it
    if name not in sys.modules:
        try:
            importlib.import_module(name)
        except ImportError:
            sys.modules[name] = types.SimpleNamespace(FuyuBatchFeature=dict)
    spec = importlib.util.spec_from_file_location("_ref_processing_fuyu", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.FuyuProcessor.__new__(mod.FuyuProcessor)
    r.tokenizer = types.SimpleNamespace(eos_token_id=EOS)
    r.pad_token_id, r.dummy_image_index = EOS, -1
    return r


def _batch(seed, B=6, T=24, n_special=(0, 2, 2, 3, 4, 2)):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(10, 1000, (B, T), generator=g)
    for r, n in enumerate(n_special):
        pos = torch.randperm(T, generator=g)[:n]
        ids[r, pos] = SPECIAL
    return ids


def test_get_labels_hand_case():
    ids = torch.tensor([[1, SPECIAL, 5, 6, SPECIAL, 9, SPECIAL], [1, 2, 3, 4, 5, 6, 7], [SPECIAL, SPECIAL, 3, 4, 5, 6, 7]])
    lab = _proc().get_labels(ids, SPECIAL)
    assert lab.tolist() == [[-100, -100, 5, 6, SPECIAL, -100, -100], [-100] * 7, [-100, SPECIAL, -100, -100, -100, -100, -100]]
    assert _proc().get_labels(ids, SPECIAL, masking_number=-1)[1].tolist() == [-1] * 7


def test_find_and_remove_tokens_hand_case():
    ids = torch.tensor([[1, SPECIAL, 5, SPECIAL, 9], [SPECIAL, 2, 3, 4, 5], [1, 2, 3, 4, 5]])
    lab = ids.clone()
    new_ids, new_lab = _proc().find_and_remove_tokens(ids, lab, SPECIAL)
    assert new_ids.tolist() == [[1, SPECIAL, 5, EOS, 9], [SPECIAL, 2, 3, 4, 5], [1, 2, 3, 4, 5]]   # a single occurrence stays
    assert new_lab.tolist() == new_ids.tolist()
    assert ids.tolist() == new_ids.tolist()          # the reference writes through row views: the caller's tensors change too


@pytest.mark.skipif(not os.path.exists(REF), reason="needs /root/reference (build container)")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_labels_and_token_rewrite_equal_the_reference(seed):
    ref, mine = _ref_proc(), _proc()
    ids = _batch(seed)
    want = ref.get_labels(ids.clone(), SPECIAL)
    got = mine.get_labels(ids.clone(), SPECIAL)
    assert torch.equal(got, want)
    a_ids, a_lab = ids.clone(), want.clone()
    b_ids, b_lab = ids.clone(), want.clone()
    w = ref.find_and_remove_tokens(a_ids, a_lab, SPECIAL)
    g = mine.find_and_remove_tokens(b_ids, b_lab, SPECIAL)
    assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(a_ids, b_ids) and torch.equal(a_lab, b_lab)


def _encodings():
    return [{"input_ids": torch.arange(5)[None] + 100, "image_patches": torch.zeros(1, 4, 6), "image_patches_indices": torch.tensor([[0, 1, 2, 3, -1]])},
            {"input_ids": torch.arange(8)[None] + 200, "image_patches": torch.zeros(1, 6, 6), "image_patches_indices": torch.tensor([[0, 1, 2, 3, 4, 5, -1, -1]])}]


def test_right_padding_with_attention_mask():
    out = _proc()._right_pad_inputs_with_attention_mask(_encodings(), True)
    assert out["input_ids"].tolist() == [[100, 101, 102, 103, 104, EOS, EOS, EOS], list(range(200, 208))]
    assert out["attention_mask"].tolist() == [[1] * 5 + [0] * 3, [1] * 8]
    assert out["image_patches_indices"][0].tolist() == [0, 1, 2, 3, -1, -1, -1, -1]
    assert [tuple(p.shape) for p in out["image_patches"]] == [(1, 4, 6), (1, 6, 6)]
    left = _proc()._left_pad_inputs_with_attention_mask(_encodings(), True)
    assert left["input_ids"][0].tolist() == [EOS, EOS, EOS, 100, 101, 102, 103, 104] and left["attention_mask"][0].tolist() == [0] * 3 + [1] * 5
    ref = _ref_proc()
    if ref is not None:
        w = ref._right_pad_inputs_with_attention_mask(_encodings(), True)
        for k in ("input_ids", "attention_mask", "image_patches_indices"):
            assert torch.equal(out[k], w[k]), k


def test_call_encodes_each_pair_alone_then_right_pads(monkeypatch):
    """`__call__(text=, images=)`: the reference's argument order; one library encoding per (prompt, image), batch right-padded."""
    p = _proc()
    calls = []

    def fake_one(self, text, image):
        calls.append((text, image))
        return _encodings()[len(calls) - 1]

    monkeypatch.setattr(FuyuProcessor, "_encode_one", fake_one)
    out = p(text=["a", "b"], images=["img0", "img1"])
    assert calls == [("a", "img0"), ("b", "img1")]
    assert out["input_ids"].shape == (2, 8) and out["attention_mask"][0].tolist() == [1] * 5 + [0] * 3
    with pytest.raises(ValueError):
        p(text=None, images=None)
    with pytest.raises(ValueError):
        p(text=["a"], images=["i", "j"])
    with pytest.raises(ValueError):
        p(text=["a"], images=["i"], return_attention_mask=False)


def test_shim_exports_the_reference_processor_surface():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = os.path.join(root, "shim")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "src" or k.startswith("src.") or k == "otter_ai" or k.startswith("otter_ai.")}
    sys.path.insert(0, shim)
    try:
        from src.otter_ai.models.fuyu.processing_fuyu import FuyuProcessor as S
        assert S is FuyuProcessor
        for name in ("get_labels", "find_and_remove_tokens", "_right_pad_inputs_with_attention_mask", "_left_pad_inputs_with_attention_mask"):
            assert hasattr(S, name), name
        # the package-level imports of the reference's demos (src/otter_ai/__init__.py)
        from otter_ai import FlamingoForConditionalGeneration, OtterForConditionalGeneration
        from src.otter_ai import OtterForConditionalGeneration as O2
        import otter_amd.modeling_otter as M
        assert OtterForConditionalGeneration is M.OtterForConditionalGeneration is O2
        assert FlamingoForConditionalGeneration is M.FlamingoForConditionalGeneration
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.") or k == "otter_ai" or k.startswith("otter_ai.")]:
            del sys.modules[k]
        sys.modules.update(saved)
