"""Small-grid GEMM shapes of the gated cross-attention block / perceiver (N or M = 512): default config choice vs the 128x128 register-staged
kernel (variant 1), the 128x128 LDS-DMA ring (variant 25) and hipBLASLt.  Usage: gemm_small.py [MxNxK ...]"""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops


def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


SHAPES = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]]
for (M, N, K) in SHAPES or [(4096, 512, 4096), (512, 4096, 4096), (4096, 4096, 512), (512, 4096, 1024), (512, 1024, 4096), (2560, 1024, 1024)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for r in range(3):
        for name, v in (("default", 0), ("t128", 1), ("ring128", 25)):
            ops.set_gemm_variant(v)
            res.setdefault(name, []).append(bench(lambda: ops.gemm_nt(A, B, out=C)))
            err = (C.float() - (A.float() @ B.float().t())).abs().max().item() if r == 0 else 0
            if r == 0: res.setdefault("err_" + name, []).append(round(err, 3))
        res.setdefault("torch", []).append(bench(lambda: torch.matmul(A, B.t(), out=C)))
    ops.set_gemm_variant(0)
    print(json.dumps({"shape": [M, N, K], **{k: round(statistics.median(v), 1) for k, v in res.items()}}), flush=True)
