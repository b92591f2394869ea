"""Optimizer step of the instruction-following recipe on HIP: `clip_grad_norm_(params, 1.0)` + `AdamW.step()`
(pipeline/train/instruction_following.py:246-251) as two sweeps of libotter_hip.so (csrc/optim.hip) instead of torch's four.

Same hyper-parameter surface as torch.optim.AdamW (param groups with lr / betas / eps / weight_decay; one lr, betas and eps
per step call, weight decay per tensor), same update arithmetic as torch's fused kernel.  One deliberate difference:
the clip coefficient is applied to the gradients on the fly, `.grad` itself is left unscaled."""
from __future__ import annotations

import math
from typing import Iterable, Optional

import numpy as np
import torch

from . import _capi as K
from .functional import shadows

_META = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("shadow", "<u8"), ("numel", "<i8"),
                  ("weight_decay", "<f4"), ("reserved", "<i4")])
assert _META.itemsize == 56


class FusedAdamW:
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.param_groups = []
        for g in groups:
            pg = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}
            pg.update(g)
            pg["params"] = [p for p in pg["params"]]
            self.param_groups.append(pg)
        self.max_grad_norm = max_grad_norm
        self.refresh_shadows = True
        self.state = {}
        self._step = 0
        self._tables = None
        self.last_norm = None  # device tensor [2]: total gradient norm, clip coefficient

    # ---- torch.optim.Optimizer surface used by the recipe ----
    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def state_dict(self):
        return {"step": self._step, "state": {i: {k: v for k, v in self.state[p].items()} for i, p in enumerate(self._all()) if p in self.state},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        ps = self._all()
        for i, st in sd["state"].items():
            self.state[ps[int(i)]] = {k: v.to(ps[int(i)].device) for k, v in st.items()}
        self._tables = None

    def _all(self):
        return [p for g in self.param_groups for p in g["params"]]

    # ---- tables: static part once, gradient pointers every step (autograd may hand out new .grad tensors) ----
    def _build(self, live):
        dev = live[0][0].device
        chunk = K.lib().otter_adamw_chunk()
        meta = np.zeros(len(live), dtype=_META)
        bt, bc = [], []
        for i, (p, wd) in enumerate(live):
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                                      "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}
            meta[i] = (p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), 0, p.numel(), wd, 0)
            n = (p.numel() + chunk - 1) // chunk
            bt.append(np.full(n, i, dtype=np.int32))
            bc.append(np.arange(n, dtype=np.int32))
        bt, bc = np.concatenate(bt), np.concatenate(bc)
        self._tables = dict(ids=[id(p) for p, _ in live], meta=meta, nblocks=int(bt.size),
                            bt=torch.from_numpy(bt).to(dev), bc=torch.from_numpy(bc).to(dev),
                            partials=torch.empty(int(bt.size), dtype=torch.float32, device=dev),
                            dmeta=torch.empty(meta.nbytes, dtype=torch.uint8, device=dev),
                            hmeta=torch.empty(meta.nbytes, dtype=torch.uint8).pin_memory(),
                            norm=torch.zeros(2, dtype=torch.float32, device=dev))

    @torch.no_grad()
    def step(self):
        live = []
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (g["lr"], g["betas"], g["eps"]) != (g0["lr"], g0["betas"], g0["eps"]):
                raise K.OtterHipError("FusedAdamW: lr / betas / eps must be the same in every param group (weight decay may differ)")
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise K.OtterHipError("FusedAdamW: fp32 contiguous parameters and gradients only")
                K.require_cuda(p, p.grad)
                live.append((p, float(g["weight_decay"])))
        if not live:
            return
        if self._tables is None or self._tables["ids"] != [id(p) for p, _ in live]:
            self._build(live)
        t = self._tables
        t["meta"]["g"] = [p.grad.data_ptr() for p, _ in live]
        # parameters that have a bf16 copy in the GEMM operand cache get it refreshed by the update kernel itself
        sh = [shadows.stale_w(p, torch.bfloat16) if self.refresh_shadows else None for p, _ in live]
        t["meta"]["shadow"] = [0 if x is None else x.data_ptr() for x in sh]
        if t.get("copied") is not None:
            t["copied"].synchronize()   # the previous step's upload has left the pinned staging buffer
        t["hmeta"].numpy()[:] = t["meta"].view(np.uint8)
        t["dmeta"].copy_(t["hmeta"], non_blocking=True)
        t["copied"] = torch.cuda.Event()
        t["copied"].record()
        self._step += 1
        b1, b2 = g0["betas"]
        lib, st = K.lib(), K.stream()
        scale = None
        if self.max_grad_norm is not None:
            K.check(lib.otter_grad_sumsq(t["dmeta"].data_ptr(), t["bt"].data_ptr(), t["bc"].data_ptr(), t["nblocks"], t["partials"].data_ptr(), st),
                    "grad_sumsq")
            K.check(lib.otter_clip_coef(t["partials"].data_ptr(), t["nblocks"], float(self.max_grad_norm), t["norm"].data_ptr(), st), "clip_coef")
            scale = t["norm"].data_ptr() + 4
            self.last_norm = t["norm"]
        K.check(lib.otter_adamw_step(t["dmeta"].data_ptr(), t["bt"].data_ptr(), t["bc"].data_ptr(), t["nblocks"], float(g0["lr"]), float(b1),
                                     float(b2), float(g0["eps"]), float(1.0 - math.pow(b1, self._step)), float(1.0 - math.pow(b2, self._step)),
                                     scale, st), "adamw_step")
        # the parameters changed in place behind autograd's back: bump the version counters the bf16 shadow cache keys on
        # (functional._Shadows) -- host-side bookkeeping only, no kernel
        for (p, _), x in zip(live, sh):
            torch.autograd.graph.increment_version(p)
            if x is not None:
                shadows.mark_w(p, torch.bfloat16, x)
