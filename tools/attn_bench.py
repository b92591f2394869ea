"""Attention cores of the fusion path at C2 shapes: MFMA (variant 0) vs fp32 VALU kernels (variant 1), forward and backward."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

def bench(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / iters * 1e3, 1)

for name, (B, H, Tq, M, n, mode) in {"xattn C2": (8, 8, 512, 64, 64, 1), "perceiver C2": (8, 8, 64, 320, 1, 0), "perceiver video": (8, 8, 64, 2112, 1, 0)}.items():
    inner = H * 64
    q = torch.randn(B, Tq, inner, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B, M, 2 * inner, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, Tq, inner, device="cuda").to(torch.bfloat16)
    tt = None
    if mode:
        tt = torch.ones(B, Tq, dtype=torch.int32, device="cuda"); tt[:, 0] = 0
    row = {"case": name}
    for v in (0, 1):
        ops.set_attn_variant(v)
        o, lse = ops.attn_fwd(q, kv[..., :inner], kv[..., inner:], H, tt, n, mode, 0.125)
        row["fwd_us_v%d" % v] = bench(lambda: ops.attn_fwd(q, kv[..., :inner], kv[..., inner:], H, tt, n, mode, 0.125))
        row["bwd_us_v%d" % v] = bench(lambda: ops.attn_bwd(q, kv[..., :inner], kv[..., inner:], o, do, lse, H, tt, n, mode, 0.125))
    ops.set_attn_variant(0)
    print(json.dumps(row))
