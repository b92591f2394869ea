"""Optimizer step of the instruction-following recipe on HIP: `clip_grad_norm_(params, 1.0)` + `AdamW.step()`
(pipeline/train/instruction_following.py:246-251) as two sweeps of libotter_hip.so (csrc/optim.hip) instead of torch's four.

`FusedAdamW` IS a `torch.optim.Optimizer` (LambdaLR / get_cosine_schedule_with_warmup, `accelerate.prepare(optimizer)` and
`isinstance` checks accept it) with torch.optim.AdamW's state layout -- per parameter `step` (fp32 scalar tensor), `exp_avg`,
`exp_avg_sq` -- so optimizer checkpoints written by either load into the other.  Learning rate and weight decay may differ
per param group and every parameter carries its own step count (bias correction per tensor, exactly torch's rule for
parameters that first receive a gradient late or skip steps); betas and eps must be the same in every group.  The update
arithmetic is torch's fused AdamW kernel's.  One deliberate difference: the clip coefficient is applied to the gradients
on the fly, `.grad` itself is left unscaled (nothing reads it before the next zero_grad)."""
from __future__ import annotations

import math
from typing import Iterable, Optional

import numpy as np
import torch

from . import _capi as K
from .functional import shadows

_META = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("shadow", "<u8"), ("numel", "<i8"),
                  ("weight_decay", "<f4"), ("lr", "<f4"), ("bc1", "<f4"), ("bc2_sqrt", "<f4")])
assert _META.itemsize == 64  # include/otter_hip.h: otter_adamw_tensor


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdamW: invalid hyper-parameter (lr, eps, weight_decay >= 0; 0 <= beta < 1)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.refresh_shadows = True
        self.norm_sink = None  # functional.GradNormSink (train.TrainStep, single rank): gradients whose sum of squares came out of their own GEMM
        self.fused_norm_tensors = 0   # how many gradients of the last step took their sum of squares from the producing launch
        self._tables = None
        self.last_norm = None  # device tensor [2]: total gradient norm, clip coefficient

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch only casts (`.to()` returns the SAME tensor when nothing changes): a state dict handed over in memory would
        # leave `step` / moments shared with the optimizer it came from, and two optimizers stepping one counter.  Own copies.
        for st in self.state.values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().to("cpu", torch.float32).clone() if k == "step" else v.detach().clone(memory_format=torch.contiguous_format)
        self._tables = None

    # ---- tables: block -> (tensor, chunk) maps once per set of live tensors; every pointer column every step ----
    def _build(self, live):
        dev = live[0][0].device
        chunk = K.lib().otter_adamw_chunk()
        bt, bc = [], []
        for i, (p, _, _) in enumerate(live):
            n = (p.numel() + chunk - 1) // chunk
            bt.append(np.full(n, i, dtype=np.int32))
            bc.append(np.arange(n, dtype=np.int32))
        bt, bc = np.concatenate(bt), np.concatenate(bc)
        meta = np.zeros(len(live), dtype=_META)
        self._tables = dict(key=[(id(p), p.numel()) for p, _, _ in live], meta=meta, nblocks=int(bt.size),
                            bt=torch.from_numpy(bt).to(dev), bc=torch.from_numpy(bc).to(dev), bt_host=bt, bc_host=bc, sweeps={},
                            partials=torch.empty(int(bt.size), dtype=torch.float32, device=dev),
                            dmeta=torch.empty(meta.nbytes, dtype=torch.uint8, device=dev),
                            hmeta=torch.empty(meta.nbytes, dtype=torch.uint8).pin_memory(),
                            norm=torch.zeros(2, dtype=torch.float32, device=dev))

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live = []
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (tuple(g["betas"]), g["eps"]) != (tuple(g0["betas"]), g0["eps"]):
                raise K.OtterHipError("FusedAdamW: betas / eps must be the same in every param group (lr and weight decay may differ)")
            if g.get("amsgrad") or g.get("maximize"):
                raise K.OtterHipError("FusedAdamW: amsgrad / maximize are not implemented")
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise K.OtterHipError("FusedAdamW: sparse gradients are not supported")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise K.OtterHipError("FusedAdamW: fp32 contiguous parameters and gradients only")
                K.require_cuda(p, p.grad)
                live.append((p, float(g["weight_decay"]), float(g["lr"])))
        if not live:
            return loss
        if self._tables is None or self._tables["key"] != [(id(p), p.numel()) for p, _, _ in live]:
            self._build(live)
        t = self._tables
        b1, b2 = (float(x) for x in g0["betas"])
        # every column is rewritten every step: a Parameter object may get new storage (model.to(), `p.data = ...`, a loaded
        # optimizer state) or a new .grad tensor at any time -- a cached address would be a silent use-after-free
        meta = t["meta"]
        states = [self._init_state(p) for p, _, _ in live]
        for st in states:
            st["step"] += 1
        steps = [float(st["step"]) for st in states]
        sh = [shadows.stale_w(p, torch.bfloat16) if self.refresh_shadows else None for p, _, _ in live]
        meta["p"] = [p.data_ptr() for p, _, _ in live]
        meta["g"] = [p.grad.data_ptr() for p, _, _ in live]
        meta["m"] = [st["exp_avg"].data_ptr() for st in states]
        meta["v"] = [st["exp_avg_sq"].data_ptr() for st in states]
        meta["shadow"] = [0 if x is None else x.data_ptr() for x in sh]
        meta["numel"] = [p.numel() for p, _, _ in live]
        meta["weight_decay"] = [wd for _, wd, _ in live]
        meta["lr"] = [lr for _, _, lr in live]
        meta["bc1"] = [1.0 - math.pow(b1, s) for s in steps]
        meta["bc2_sqrt"] = [math.sqrt(1.0 - math.pow(b2, s)) for s in steps]
        for p, st in zip((p for p, _, _ in live), states):
            if st["exp_avg"].shape != p.shape or st["exp_avg"].device != p.device or st["exp_avg"].dtype != torch.float32 \
                    or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                raise K.OtterHipError("FusedAdamW: optimizer state does not match its parameter (shape / device / fp32 contiguous)")
        if t.get("copied") is not None:
            t["copied"].synchronize()   # the previous step's upload has left the pinned staging buffer
        t["hmeta"].numpy()[:] = meta.view(np.uint8)
        t["dmeta"].copy_(t["hmeta"], non_blocking=True)
        t["copied"] = torch.cuda.Event()
        t["copied"].record()
        lib, stm = K.lib(), K.stream()
        scale = None
        if self.max_grad_norm is not None:
            # Gradients whose sum of squares was written by the GEMM that produced them (functional.GradNormSink; verified against the
            # tensor .grad holds NOW) leave the sweep: their per-tile partials are appended behind the sweep's per-block partials and
            # otter_clip_coef adds up both.  At OTTER-MPT7B that is the sixteen FFN matrices of the gated blocks, 88 % of the 4.9 GB sweep.
            sink, fused = self.norm_sink, {}
            if sink is not None and sink.buf is not None:
                for i, (p, _, _) in enumerate(live):
                    f = sink.fused(p)
                    if f is not None:
                        fused[i] = f
            self.fused_norm_tensors = len(fused)
            bt_d, bc_d, n_sweep, parts, n_parts = t["bt"], t["bc"], t["nblocks"], t["partials"], t["nblocks"]
            if fused:
                key = tuple(sorted(fused))
                sw = t["sweeps"].get(key)
                if sw is None:
                    keep = ~np.isin(t["bt_host"], np.asarray(key, dtype=np.int32))
                    dev = t["bt"].device
                    sw = t["sweeps"][key] = (torch.from_numpy(t["bt_host"][keep]).to(dev), torch.from_numpy(t["bc_host"][keep]).to(dev), int(keep.sum()))
                bt_d, bc_d, n_sweep = sw
                n_fused = sum(n for _, n in fused.values())
                if t.get("parts2") is None or t["parts2"].numel() < n_sweep + n_fused:
                    t["parts2"] = torch.empty(n_sweep + n_fused, dtype=torch.float32, device=t["bt"].device)
                parts, n_parts = t["parts2"], n_sweep + n_fused
                torch.cat([sink.buf[o:o + n] for o, n in (fused[i] for i in key)], out=parts[n_sweep:n_parts])
            if n_sweep > 0:
                K.check(lib.otter_grad_sumsq(t["dmeta"].data_ptr(), bt_d.data_ptr(), bc_d.data_ptr(), n_sweep, parts.data_ptr(), stm), "grad_sumsq")
            K.check(lib.otter_clip_coef(parts.data_ptr(), n_parts, float(self.max_grad_norm), t["norm"].data_ptr(), stm), "clip_coef")
            scale = t["norm"].data_ptr() + 4
            self.last_norm = t["norm"]
        # launch-wide lr / bias corrections are placeholders: every table row carries its own (bc1 != 0)
        K.check(lib.otter_adamw_step(t["dmeta"].data_ptr(), t["bt"].data_ptr(), t["bc"].data_ptr(), t["nblocks"], float(g0["lr"]), b1, b2,
                                     float(g0["eps"]), 1.0, 1.0, scale, stm), "adamw_step")
        # the parameters changed in place behind autograd's back: bump the version counters the bf16 shadow cache keys on
        # (functional._Shadows) -- host-side bookkeeping only, no kernel
        for (p, _, _), x in zip(live, sh):
            torch.autograd.graph.increment_version(p)
            if x is not None:
                shadows.mark_w(p, torch.bfloat16, x)
        return loss
