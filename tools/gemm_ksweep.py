"""Fixed vs per-K cost of the bf16 GEMM: time M=4096, N=16384 at several K for our variants and hipBLASLt (torch.matmul),
fit T(K) = a + b*K.  a = prologue + epilogue + launch, b = main loop per unit K.  Usage: gemm_ksweep.py [variants] [M N]"""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops


def bench(fn, iters=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [13, 18]
M, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 16384)
Ks = [1024, 2048, 4096, 8192]
res = {}
for K in Ks:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    legs = [("v%d" % v, v) for v in variants] + [("torch", None)]
    for r in range(4):   # rotated order + a throw-away run per leg: the slot after a hipBLASLt burst runs slow
        for name, v in legs[r % len(legs):] + legs[:r % len(legs)]:
            if v is None:
                fn = lambda: torch.matmul(A, B.t(), out=C)
            else:
                ops.set_gemm_variant(v)
                fn = lambda: ops.gemm_nt(A, B, out=C)
            bench(fn)
            res.setdefault((name, K), []).append(bench(fn))
ops.set_gemm_variant(0)
for name in ["v%d" % v for v in variants] + ["torch"]:
    t = [statistics.median(res[(name, K)]) for K in Ks]
    # least squares a + b*K
    n = len(Ks); sx = sum(Ks); sy = sum(t); sxx = sum(k * k for k in Ks); sxy = sum(k * y for k, y in zip(Ks, t))
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    print(json.dumps({"kernel": name, "us": [round(x, 1) for x in t], "fixed_us": round(a, 1), "us_per_1024K": round(b * 1024, 1),
                      "mainloop_TF": round(2.0 * M * N * 1024 / (b * 1024) / 1e6, 1)}), flush=True)
