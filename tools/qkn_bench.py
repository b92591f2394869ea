"""otter_qk_norm_rope_fwd / _bwd at the C5 shape (B=8, 1396 tokens, 64 heads): time per launch and effective bandwidth."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

B, S, H, rot = 8, 1396, 64, 32
qkv = torch.randn(B, S, H * 192, device="cuda").to(torch.bfloat16)
g = [torch.randn(64, device="cuda") for _ in range(4)]
cos = torch.randn(S, rot, device="cuda"); sin = torch.randn(S, rot, device="cuda")
q, k, v, stats = ops.qk_norm_rope_fwd(qkv, g[0], g[1], g[2], g[3], cos, sin, H, rot, 1e-5, width=64, copy_v=False)
dq, dk = torch.randn_like(q), torch.randn_like(k)
dqkv = torch.empty_like(qkv)


def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

tf = bench(lambda: ops.qk_norm_rope_fwd(qkv, g[0], g[1], g[2], g[3], cos, sin, H, rot, 1e-5, width=64, copy_v=False))
tb = bench(lambda: ops.qk_norm_rope_bwd(dq, dk, None, qkv, stats, g[0], g[2], cos, sin, H, rot, dqkv=dqkv))
n = B * S * H * 64 * 2
print("fwd %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s; includes the partial-sum reduction in torch)   iters=%s cap=%s" % (
    tf, 2 * n * 2 / tf / 1e6, tb, 3 * n * 2 / tb / 1e6, os.environ.get("OTTER_QKN_ITERS", "64"), os.environ.get("OTTER_QKN_CAP", "2048")))
