"""Weight-gradient product dW = dy^T x (contraction over the 4096 token rows) two ways: the shipped path (otter_transpose of
both operands + the NT GEMM with fp32 output) and hipBLASLt's native TN form through torch.mm(..., out_dtype=fp32).
Data point for the TN-native GEMM item of DESIGN.md section 8."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / iters * 1e3, 1)

M = 4096
for name, (n_out, n_in) in {"W1 [16384x4096]": (16384, 4096), "W2 [4096x16384]": (4096, 16384), "Wo [4096x512]": (4096, 512)}.items():
    dy = torch.randn(M, n_out, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, n_in, device="cuda").to(torch.bfloat16)
    def ours():
        return ops.gemm_nt(ops.transpose(dy, torch.bfloat16), ops.transpose(x, torch.bfloat16), out_dtype=torch.float32)
    def gemm_only(a=ops.transpose(dy, torch.bfloat16), b=ops.transpose(x, torch.bfloat16)):
        return ops.gemm_nt(a, b, out_dtype=torch.float32)
    def lib():
        return torch.mm(dy.t(), x, out_dtype=torch.float32)
    r = {"weight": name}
    for rep in range(2):
        r["transposes+gemm_us_%d" % rep] = bench(ours)
        r["gemm_only_us_%d" % rep] = bench(gemm_only)
        r["hipblaslt_tn_us_%d" % rep] = bench(lib)
    err = (ours() - lib()).abs().max().item() / lib().abs().max().item()
    r["rel_diff"] = round(err, 6)
    print(json.dumps(r))
