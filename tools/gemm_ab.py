"""Interleaved A/B rounds of GEMM variants at the gated-FFN shapes (random data).  Usage: gemm_ab.py [rounds] [v1,v2,...]"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 5, 6]
for (M, N, K) in [(4096, 16384, 4096), (4096, 4096, 16384), (16384, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    # The slot right after the hipBLASLt leg runs ~15 % slow (the chip comes out of it power-throttled: DESIGN.md 4.3), so the order of
    # the legs rotates from round to round and every leg is preceded by a throw-away run of the SAME kernel.
    legs = [("v%d" % v, v) for v in variants] + [("torch", None)]
    for r in range(rounds):
        for name, v in legs[r % len(legs):] + legs[:r % len(legs)]:
            if v is None:
                fn = lambda: torch.matmul(A, B.t(), out=C)
            else:
                ops.set_gemm_variant(v)
                fn = lambda: ops.gemm_nt(A, B, out=C)
            bench(fn)
            res.setdefault(name, []).append(bench(fn))
    ops.set_gemm_variant(0)
    fl = 2.0 * M * N * K
    print(json.dumps({"shape": [M, N, K], **{k: {"min_us": round(min(v), 1), "med_us": round(statistics.median(v), 1),
                                                "TF_med": round(fl / statistics.median(v) / 1e6, 1)} for k, v in res.items()}}), flush=True)
