"""Host-side prompt / patch packing of the OtterHD path: the reference's `FuyuProcessor` (src/otter_ai/models/fuyu/processing_fuyu.py:298-640).

The reference class is a fork of transformers' processor with three behavioural differences that its OtterHD collate relies on
(pipeline/mimicit_utils/mimicit_dataset.py:497-505, `prepare_fuyu`):

  * `__call__(text=..., images=...)` encodes every (prompt, image) pair ON ITS OWN and **right**-pads the batch with the eos id
    (`pad_token_id = tokenizer.eos_token_id`, `:321`), attention mask 0 on the pads, patch indices padded with -1
    (`_right_pad_inputs_with_attention_mask`, `:368-408`).  The installed library class left-pads (inference layout);
  * `get_labels(input_ids, special_token_id)` (`:348-366`): targets = the tokens after the FIRST occurrence of the special token up to
    and including the SECOND one, everything else `masking_number`;
  * `find_and_remove_tokens(input_ids, labels, token_id)` (`:324-346`): when a row holds the token more than once, its LAST occurrence
    becomes eos in both the ids and the labels (the name is the reference's; nothing is removed).

Nothing here touches the GPU.  The per-sample encoding itself (image resize / pad / patchify, |SPEAKER| / |NEWLINE| placeholders, box and
point tags) is the library's -- third-party code in the reference as well (it is a copy of transformers 4.35's file)."""
from __future__ import annotations

from typing import List

import torch
from transformers import FuyuProcessor as _HFFuyuProcessor


def _labels_between_first_two(input_ids: torch.Tensor, special_token_id: int, masking_number: int = -100) -> torch.Tensor:
    """Whole-batch form of the reference's per-row loop: position p is a target iff exactly one special token lies strictly before it
    and the row has at least two of them.  (A row with exactly ONE occurrence makes the reference raise -- `len()` of a 0-d tensor
    after `.squeeze()`; here such a row simply has no targets.)"""
    eq = input_ids == special_token_id
    before = torch.cumsum(eq.to(torch.int64), dim=1) - eq.to(torch.int64)      # occurrences strictly before p
    has_two = eq.sum(dim=1, keepdim=True) >= 2
    return torch.where((before == 1) & has_two, input_ids, torch.full_like(input_ids, masking_number))


class FuyuProcessor(_HFFuyuProcessor):
    def __init__(self, image_processor, tokenizer, **kwargs):
        super().__init__(image_processor=image_processor, tokenizer=tokenizer, **kwargs)
        self.pad_token_id = tokenizer.eos_token_id            # processing_fuyu.py:321 (the library pads with 0)
        self.dummy_image_index = -1

    # ---- label construction of the OtterHD collate -------------------------------------------------------------------------------
    def get_labels(self, input_ids: torch.Tensor, special_token_id: int, masking_number: int = -100) -> torch.Tensor:
        return _labels_between_first_two(input_ids, special_token_id, masking_number)

    def find_and_remove_tokens(self, input_ids: torch.Tensor, labels: torch.Tensor, token_id: int):
        eos = self.tokenizer.eos_token_id
        eq = input_ids == token_id
        n = eq.sum(dim=1, keepdim=True)
        T = input_ids.shape[1]
        pos = torch.arange(T, device=input_ids.device).expand_as(input_ids)
        last = torch.where(eq, pos, torch.full_like(pos, -1)).max(dim=1, keepdim=True).values
        hit = (pos == last) & (n > 1)
        # the reference writes through row views, i.e. it also modifies the caller's tensors: kept
        input_ids[hit] = eos
        labels[hit] = eos
        return input_ids.clone(), labels.clone()

    # ---- batching -----------------------------------------------------------------------------------------------------------------
    def _pad_inputs_with_attention_mask(self, model_inputs: List[dict], return_attention_mask: bool, left: bool) -> dict:
        width = max(e["input_ids"].shape[1] for e in model_inputs)
        width_idx = max(e["image_patches_indices"].shape[1] for e in model_inputs)

        def pad(t, w, value):
            fill = torch.full((t.shape[0], w - t.shape[1]), value, dtype=torch.long)
            return torch.cat([fill, t] if left else [t, fill], dim=1)

        out = {"input_ids": torch.cat([pad(e["input_ids"], width, self.pad_token_id) for e in model_inputs], dim=0),
               "image_patches": [e["image_patches"] for e in model_inputs],     # ragged: a list, as in the reference
               "image_patches_indices": torch.cat([pad(e["image_patches_indices"], width_idx, self.dummy_image_index) for e in model_inputs], dim=0)}
        if return_attention_mask:
            out["attention_mask"] = torch.cat([pad(torch.ones_like(e["input_ids"]), width, 0) for e in model_inputs], dim=0)
        return out

    def _right_pad_inputs_with_attention_mask(self, model_inputs: List[dict], return_attention_mask: bool) -> dict:
        return self._pad_inputs_with_attention_mask(model_inputs, return_attention_mask, left=False)

    def _left_pad_inputs_with_attention_mask(self, model_inputs: List[dict], return_attention_mask: bool) -> dict:
        return self._pad_inputs_with_attention_mask(model_inputs, return_attention_mask, left=True)

    def _encode_one(self, text, image) -> dict:
        """One (prompt, image) pair through the library's encoder; a batch of one has no padding, whatever side the library pads on."""
        enc = _HFFuyuProcessor.__call__(self, images=[image], text=[text] if text is not None else None)
        patches = enc["image_patches"]
        if isinstance(patches, (list, tuple)):
            patches = patches[0]
        return {"input_ids": enc["input_ids"], "image_patches": patches, "image_patches_indices": enc["image_patches_indices"]}

    def __call__(self, text=None, images=None, return_attention_mask: bool = True, **kwargs):
        """The reference's argument order (`text` first) and batch layout (right padding).  Text-only input goes to the tokenizer, as in
        the reference (`:569-590`)."""
        from transformers.feature_extraction_utils import BatchFeature

        if not return_attention_mask:
            raise ValueError("`return_attention_mask=False` is not supported for this model.")
        if text is None and images is None:
            raise ValueError("You have to specify either text or images. Both cannot be None.")
        if images is None:
            return self.tokenizer(text=text, return_attention_mask=return_attention_mask, **kwargs)
        if not isinstance(images, (list, tuple)):
            images = [images]
        if text is None:
            texts = [None] * len(images)
        else:
            texts = [text] if isinstance(text, str) else list(text)
        if len(texts) != len(images):
            raise ValueError("FuyuProcessor: %d prompts for %d images" % (len(texts), len(images)))
        encs = [self._encode_one(t, im) for t, im in zip(texts, images)]
        return BatchFeature(data=self._right_pad_inputs_with_attention_mask(encs, return_attention_mask))
