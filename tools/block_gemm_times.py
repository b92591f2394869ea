"""Every GEMM launch of one OtterGatedCrossAttentionBlock forward + backward at the C2 shapes, timed one by one (events around each call,
a synchronisation after each: durations, not a step time).  Usage: block_gemm_times.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd.modeling_otter import OtterGatedCrossAttentionBlock

dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = OtterGatedCrossAttentionBlock(dim=4096, dim_visual=1024).to(dev)
with torch.no_grad():
    for n, p in blk.named_parameters():
        if p.ndim >= 2:
            p.normal_(0.0, 0.02)
    blk.attn_gate.fill_(0.5); blk.ff_gate.fill_(0.5)
B, T = 8, 512
x = torch.randn(B, T, 4096, device=dev, requires_grad=True)
media = torch.randn(B, 1, 64, 1024, device=dev)
dy = torch.randn(B, T, 4096, device=dev)
ml = torch.zeros(B, T, dtype=torch.bool, device=dev); ml[:, 1] = True
log = []
KIND = {0: "store", 1: "gelu", 2: "scale_res", 3: "gate_bwd"}


def wrap(name, fn):
    def inner(A, Bm, *a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(A, Bm, *a, **kw)
        e.record(); torch.cuda.synchronize()
        if name == "gemm":
            ta, tb = a[0], a[1]
            M = A.shape[1] if ta else A.shape[0]; N = Bm.shape[1] if tb else Bm.shape[0]; Kd = A.shape[0] if ta else A.shape[1]
        else:
            ta = tb = False
            M, Kd = A.shape[-2], A.shape[-1]; N = Bm.shape[-2]
        log.append((name, M, N, Kd, "A^T" if ta else "A", "B^T" if tb else "B", KIND.get(kw.get("kind", 0), "?"), str(out.dtype).split(".")[-1],
                    "acc" if kw.get("accumulate") else "", s.elapsed_time(e) * 1e3))
        return out
    return inner


ops_gemm, ops_gemm_nt = ops.gemm, ops.gemm_nt
for it in range(4):
    if it == 3:
        ops.gemm_nt = wrap("gemm_nt", ops_gemm_nt)
        ops.gemm = wrap("gemm", lambda A, Bm, *a, **kw: ops_gemm(A, Bm, *a, **kw))
    for p in blk.parameters():
        p.grad = None
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x, media, media_locations=ml, attend_previous=True)
    y.backward(dy)
torch.cuda.synchronize()
tot = 0.0
for r in log:
    print("%-8s M=%-6d N=%-6d K=%-6d %-3s %-3s %-9s %-8s %-3s %8.1f us   %6.0f TF" % (*r, 2.0 * r[1] * r[2] * r[3] / r[9] / 1e6))
    tot += r[9]
print("GEMM launches: %d, %.1f us in total" % (len(log), tot))
