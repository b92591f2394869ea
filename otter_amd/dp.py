"""Data-parallel gradient reduction for the Otter training step: one process per GPU, RCCL over xGMI.

The reference gets its all-reduce implicitly from accelerate -> torch DDP (instruction_following.py:311-314,491-494:
25 MB buckets of every requires_grad parameter).  Here the reducer is explicit and shaped for MI355X:

  * the trainable set is tiny in tensor count but large in bytes (perceiver 63 M + 8 x 139.5 M gated blocks + 206.6 M
    tied embedding = 5.54 GB fp32), and xGMI is a point-to-point mesh (7 links x ~153 GB/s) where a ring all-reduce is
    per-link bound -- so buckets are BIG (default 640 MB = one gated cross-attention block): 10 large collectives
    instead of ~220 small ones;
  * parameters' .grad tensors are *views into the flat bucket*: the weight-gradient GEMMs of the hand-written path write
    their result straight into that view (functional.grad_sink: no zero-fill, no accumulate pass, no copy), gradients
    produced by stock autograd (the tied embedding) are copied in once, and the optimizer reads the reduced values in place;
  * a bucket's all-reduce is launched (async, on RCCL's stream) from the post-accumulate-grad hook of its last-arriving
    parameter, i.e. while the frozen decoder layers *below* that cross-attention block are still running their dgrad --
    the overlap window of SURVEY.md section 5.

Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], dtype, device, world: int = 1):
        self.params = params
        # every slice starts on a 256-byte boundary: the weight-gradient GEMMs store into the views with 16-byte vectors
        al = 256 // torch.empty((), dtype=dtype).element_size()
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + al - 1) // al * al
        # (the reduce-scatter + all-gather form splits the flat buffer into `world` equal shards: length a multiple of world x 256 bytes)
        off = (off + al * world - 1) // (al * world) * (al * world)
        self.shard = None                                          # this rank's reduced shard (reduce-scatter + all-gather form only)
        self.flat = torch.zeros(off, dtype=dtype, device=device)   # padding stays zero (zeros reduce to zeros)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
        self.ready = set()   # id(param) whose gradient of the CURRENT backward pass is in its view
        self.work = None


class GradReducer:
    """Bucketed, overlapped gradient averaging across the data-parallel group."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 640 << 20,
                 process_group: Optional[dist.ProcessGroup] = None, grad_dtype: torch.dtype = torch.float32, force: bool = False,
                 row_only: Optional[dict] = None, overlap: Optional[bool] = None, collective: Optional[str] = None):
        """row_only: {parameter: row} -- parameters whose gradient is known to be zero outside ONE row when wait() is called
        (the reference's `--mask_lm_head`, train.mask_embedding): they stay out of the buckets, keep an ordinary .grad, and only
        that row is averaged across ranks."""
        self.group = process_group
        self.row_only = dict(row_only or {})
        self.force = force  # run the collectives even with one rank (exercises the RCCL path on a single GPU)
        # overlap=False (or OTTER_DP_OVERLAP=0): no collective is launched from the gradient hooks; every bucket is reduced in wait(),
        # after backward -- the A/B leg for a multi-GPU run (is the resident RCCL kernel costing the backward GEMMs more than the
        # overlap hides?  DESIGN.md section 7).  Same collectives, same order, same results.
        import os as _os

        self.overlap = (_os.environ.get("OTTER_DP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        # collective="rs_ag" (or OTTER_DP_COLLECTIVE=rs_ag; bench.py --dp-collective): each bucket is averaged as a reduce-scatter into one
        # shard per rank followed by an all-gather of the shards -- the two halves of a ring all-reduce as separate RCCL calls, which on the
        # point-to-point xGMI mesh of an MI355X node RCCL may schedule differently from its fused all-reduce ("direct
        # RS + AG on the 7-link mesh ... compare against stock DDP").  Same averages up to the summation order.  A/B switch for a multi-GPU
        # session; the default stays one all-reduce per bucket.
        self.collective = (collective or _os.environ.get("OTTER_DP_COLLECTIVE", "all_reduce")).lower()
        if self.collective not in ("all_reduce", "rs_ag"):
            raise ValueError("GradReducer: collective must be 'all_reduce' or 'rs_ag', not %r" % (self.collective,))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        ps = [p for p in params if p.requires_grad and p not in self.row_only]
        if not ps:
            raise ValueError("GradReducer: no trainable parameters")
        # reverse registration order ~ the order gradients become ready in backward
        ps = list(reversed(ps))
        self.buckets: List[_Bucket] = []
        cur, cur_bytes = [], 0
        for p in ps:
            nb = p.numel() * torch.empty((), dtype=grad_dtype).element_size()
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(_Bucket(cur, grad_dtype, cur[0].device, self.world))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(_Bucket(cur, grad_dtype, cur[0].device, self.world))
        self._owner = {}
        self._view = {}
        self._hooks = []
        self._written = set()   # id(param) whose bucket view holds this step's gradient
        self.sync = True
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                if p.dtype != grad_dtype:
                    raise ValueError("GradReducer: parameter dtype must equal grad_dtype (fp32 master weights)")
                p.grad = None
                self._owner[p] = b
                self._view[p] = v
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        from . import functional as _F

        # one live reducer per parameter set: an older one over any of these parameters would keep its hooks and rebind
        # p.grad to ITS buckets -- detach it (close() removes its hooks) instead of double-counting
        old = _F.grad_sink
        if old is not None and old is not self and isinstance(old, GradReducer) and any(p in old._owner for p in ps):
            old.close()
        _F.grad_sink = self  # weight-gradient GEMMs now target the bucket views directly

    # ---- step protocol:  zero_grad() ... backward() ... wait() ... optimizer.step() ----
    def zero_grad(self):
        """No memset: every view is either overwritten by its first gradient of the step or zeroed in wait()."""
        self._written.clear()
        for b in self.buckets:
            b.ready.clear()
            b.work = None
            for p in b.params:
                p.grad = None
        for p in self.row_only:
            p.grad = None

    # ---- grad sink protocol (otter_amd.functional._wgrad) ----
    def take(self, p: torch.nn.Parameter):
        """The bucket view to write p's gradient into (overwrite semantics), or None if this step already put one there
        (micro-batch accumulation: the caller then returns its gradient to autograd, which adds it in place)."""
        if p not in self._owner or id(p) in self._written:
            return None
        self._written.add(id(p))
        return self._view[p]

    def ready(self, p: torch.nn.Parameter):
        p.grad = self._view[p]
        self._count(p)

    def _on_grad(self, p: torch.nn.Parameter):
        v = self._view[p]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            # autograd installed a fresh tensor (grad was None): move it into the bucket view
            if id(p) in self._written:
                v.add_(p.grad)
            else:
                v.copy_(p.grad)
                self._written.add(id(p))
            p.grad = v
        else:
            self._written.add(id(p))  # accumulated in place into the view
        self._count(p)

    def _count(self, p: torch.nn.Parameter):
        # per-parameter ready flags, not a counter: under no_sync() accumulation a parameter reports once per micro-batch,
        # and the flags are reset when the synchronising backward begins (no_sync().__exit__), so the hooks of THAT
        # backward launch each bucket as its last gradient arrives (the overlap) instead of everything serialising in wait()
        b = self._owner[p]
        b.ready.add(id(p))
        if len(b.ready) == len(b.params) and b.work is None and self.sync and self.overlap and (self.world > 1 or self.force):
            self._launch(b)

    def _launch(self, b: _Bucket):
        backend = dist.get_backend(self.group)
        if self.collective == "rs_ag":
            n = b.flat.numel() // self.world
            if b.shard is None:
                b.shard = torch.empty(n, dtype=b.flat.dtype, device=b.flat.device)
            if backend == "nccl":     # both calls are enqueued now: RCCL runs them in order on its stream, wait() waits for the second
                dist.reduce_scatter_tensor(b.shard, b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                b.work = dist.all_gather_into_tensor(b.flat, b.shard, group=self.group, async_op=True)
            else:                     # gloo (CPU tests): no stream order between asynchronous calls
                dist.reduce_scatter_tensor(b.shard, b.flat, op=dist.ReduceOp.SUM, group=self.group)
                b.shard.div_(self.world)
                b.work = dist.all_gather_into_tensor(b.flat, b.shard, group=self.group, async_op=True)
            return
        if backend == "nccl":
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            b._needs_div = True

    def wait(self):
        """Block the current stream until every bucket is reduced.  Buckets whose hooks never fired (unused parameters)
        are reduced here so that all ranks issue the same collectives."""
        for b in self.buckets:  # parameters that produced no gradient this step contribute zeros
            for p, v in zip(b.params, b.views):
                if id(p) not in self._written:
                    v.zero_()
                    self._written.add(id(p))
                    p.grad = v
        if (self.world > 1 or self.force) and self.sync:
            for b in self.buckets:
                if b.work is None:
                    self._launch(b)
            rows = self._launch_rows()
            for b in self.buckets:
                b.work.wait()
                if getattr(b, "_needs_div", False):
                    b.flat.div_(self.world)
                    b._needs_div = False
            self._finish_rows(rows)
        for b in self.buckets:
            b.ready.clear()

    def _launch_rows(self):
        """One small collective for all row-only parameters (every rank holds the same parameter set in the same order)."""
        if not self.row_only:
            return None
        items = list(self.row_only.items())
        parts = []
        for p, r in items:
            g = p.grad
            parts.append(g[r].reshape(-1).to(torch.float32) if g is not None else torch.zeros(p.shape[1:].numel(), dtype=torch.float32, device=p.device))
        flat = torch.cat(parts)
        avg = dist.get_backend(self.group) == "nccl"
        work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
        return items, flat, work, avg

    def _finish_rows(self, rows):
        if rows is None:
            return
        items, flat, work, avg = rows
        work.wait()
        if not avg:
            flat.div_(self.world)
        off = 0
        for p, r in items:
            n = p.shape[1:].numel()
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad[r] = flat[off:off + n].view(p.shape[1:]).to(p.grad.dtype)
            off += n

    def close(self):
        """Detach from autograd and from the weight-gradient GEMMs (the parameters keep their current .grad views)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import functional as _F

        if _F.grad_sink is self:
            _F.grad_sink = None

    def no_sync(self):
        red = self

        class _Ctx:
            def __enter__(self_inner):
                red.sync = False

            def __exit__(self_inner, *a):
                red.sync = True
                for b in red.buckets:   # the next backward is the synchronising one: count its gradients from zero
                    b.ready.clear()
                return False

        return _Ctx()

    def bucket_summary(self):
        return [(len(b.params), b.flat.numel() * b.flat.element_size()) for b in self.buckets]
