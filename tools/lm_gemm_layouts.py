"""hipBLASLt on the frozen decoder's GEMM shapes: forward (x W^T, "TN"), dgrad as torch does it (dy W, "NN") and dgrad
against a pre-transposed copy of the frozen weight (dy (W^T)^T, "TN" again)."""
import json, os, sys
import torch
import torch.nn.functional as F

def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / iters * 1e3, 1)

M = 4096
for name, (n_out, n_in) in {"Wqkv": (12288, 4096), "out_proj": (4096, 4096), "up_proj": (16384, 4096), "down_proj": (4096, 16384)}.items():
    W = torch.randn(n_out, n_in, device="cuda").to(torch.bfloat16)
    Wt = W.t().contiguous()
    x = torch.randn(M, n_in, device="cuda").to(torch.bfloat16)
    dy = torch.randn(M, n_out, device="cuda").to(torch.bfloat16)
    r = {"layer": name}
    for rep in range(2):
        r["fwd_TN_us_%d" % rep] = bench(lambda: F.linear(x, W))
        r["dgrad_NN_us_%d" % rep] = bench(lambda: torch.matmul(dy, W))
        r["dgrad_TN_us_%d" % rep] = bench(lambda: F.linear(dy, Wt))
    print(json.dumps(r))
