"""Training-step glue for the instruction-following recipe (pipeline/train/instruction_following.py), restated so that
bench.py and the tests can drive one step without the reference's dataset / accelerate / wandb stack:

  masking()            label construction                        instruction_following.py:163-192
  get_grouped_params   weight-decay grouping by parameter name    pipeline/train/train_utils.py:167-183
  TrainStep            forward (bf16 autocast) -> backward -> DP gradient average -> clip_grad_norm_(1.0) -> AdamW
                                                                  instruction_following.py:200-251
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .dp import GradReducer


def masking(input_ids: torch.Tensor, answer_token_id: int, endofchunk_token_id: int, eos_token_id: int,
            masking_number: int = -100) -> torch.Tensor:
    """Labels = input ids on every <answer> ... <|endofchunk|> span (answer token excluded, endofchunk included), eos tokens
    kept, everything else -100, column 0 always -100.  Same result as the reference's per-sample Python loop, computed
    per row with tensor ops (the pairing rule -- each <answer> matched with the first later unused <|endofchunk|> -- is
    kept by walking the (few) answer positions of a row)."""
    B, T = input_ids.shape
    labels = torch.where(input_ids == eos_token_id, input_ids, torch.full_like(input_ids, masking_number))
    for i in range(B):
        row = input_ids[i]
        ans = torch.nonzero(row == answer_token_id, as_tuple=False).flatten().tolist()
        eoc = torch.nonzero(row == endofchunk_token_id, as_tuple=False).flatten().tolist()
        j = 0
        for a in ans:
            while j < len(eoc) and eoc[j] < a:
                j += 1
            if j < len(eoc):
                e = eoc[j]
                labels[i, a + 1:e + 1] = row[a + 1:e + 1]
                j += 1
        for a, e in zip(ans, eoc):  # the reference's second (positional zip) pass, instruction_following.py:184-185
            labels[i, a + 1:e + 1] = row[a + 1:e + 1]
    labels[:, 0] = masking_number
    return labels


def get_grouped_params(model: torch.nn.Module, wd: float):
    def apply_decay(x):
        return ("gated_cross_attn_layer" in x and "ff_gate" not in x and "attn_gate" not in x and "norm" not in x
                and "bias" not in x)

    with_wd, without_wd = [], []
    for n, p in model.named_parameters():
        (with_wd if apply_decay(n) else without_wd).append(p)
    return [{"params": with_wd, "weight_decay": wd}, {"params": without_wd, "weight_decay": 0.0}]


class TrainStep:
    """One optimizer step of the Otter instruction-following recipe on this rank's micro-batch."""

    def __init__(self, model, lr: float = 1e-5, weight_decay: float = 0.1, max_grad_norm: float = 1.0,
                 autocast_dtype: Optional[torch.dtype] = torch.bfloat16, process_group=None, bucket_bytes: int = 640 << 20,
                 fused_optimizer: bool = True, force_reducer: bool = False, hip_optimizer: Optional[bool] = None):
        self.model = model
        self.max_grad_norm = max_grad_norm
        self.autocast_dtype = autocast_dtype
        groups = get_grouped_params(model, weight_decay)
        groups = [{"params": [p for p in g["params"] if p.requires_grad], "weight_decay": g["weight_decay"]} for g in groups]
        self.params = [p for g in groups for p in g["params"]]
        dev = self.params[0].device
        # GPU: clip + AdamW as two HIP sweeps (otter_amd/optim.py); CPU (gloo tests) or OTTER_TORCH_ADAMW=1: torch.optim.AdamW
        if hip_optimizer is None:
            hip_optimizer = dev.type == "cuda" and os.environ.get("OTTER_TORCH_ADAMW") != "1"
        self.hip_optimizer = bool(hip_optimizer)
        if self.hip_optimizer:
            from .optim import FusedAdamW

            self.optimizer = FusedAdamW(groups, lr=lr, max_grad_norm=max_grad_norm)
        else:
            self.optimizer = torch.optim.AdamW(groups, lr=lr, fused=bool(fused_optimizer and dev.type == "cuda"))
        self.world = torch.distributed.get_world_size(process_group) if torch.distributed.is_initialized() else 1
        # single rank: gradients stay ordinary .grad tensors (no bucket indirection, nothing to reduce)
        self.reducer = GradReducer(self.params, bucket_bytes, process_group, force=force_reducer) if (self.world > 1 or force_reducer) else None

    def close(self):
        """Detach the DP reducer's autograd hooks and gradient sink (idempotent).  Call before building another TrainStep /
        GradReducer over the same model; also runs when the object is collected."""
        red, self.reducer = getattr(self, "reducer", None), None
        if red is not None:
            red.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def no_sync(self):
        """Micro-batch accumulation context (accelerate.no_sync / DDP.no_sync): gradients accumulate locally, no collective."""
        import contextlib

        return self.reducer.no_sync() if self.reducer is not None else contextlib.nullcontext()

    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def __call__(self, vision_x, input_ids, attention_mask, labels):
        self.zero_grad()
        dev_type = input_ids.device.type
        if self.autocast_dtype is not None:
            with torch.autocast(device_type=dev_type, dtype=self.autocast_dtype):
                loss = self.model(vision_x=vision_x.to(self.autocast_dtype), lang_x=input_ids, attention_mask=attention_mask,
                                  labels=labels)[0]
        else:
            loss = self.model(vision_x=vision_x, lang_x=input_ids, attention_mask=attention_mask, labels=labels)[0]
        loss.backward()
        if self.reducer is not None:
            self.reducer.wait()
        if self.max_grad_norm is not None and not self.hip_optimizer:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.optimizer.step()   # the HIP optimizer clips inside its step
        return loss.detach()
