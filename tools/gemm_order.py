"""Tile-order sweep of the default large-grid GEMM (otter_gemm_set_debug bits 9-12: super-tile shape 2^lm x 2^(5-lm), M- or N-major walk)
at the three gated-FFN shapes, interleaved rounds, bf16 and fp32 outputs.  Usage: gemm_order.py [rounds] [variant]"""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi

def bench(fn, iters=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ops.set_gemm_variant(variant)
orders = [0] + [lm + 1 + 8 * nm for lm in range(6) for nm in (0, 1)]
for (M, N, K) in [(4096, 16384, 4096), (4096, 4096, 16384), (16384, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for odt in (torch.bfloat16, torch.float32):
        C = torch.empty(M, N, device="cuda", dtype=odt)
        res = {}
        for r in range(rounds):
            for o in orders:
                _capi.lib().otter_gemm_set_debug(o << 9)
                res.setdefault(o, []).append(bench(lambda: ops.gemm_nt(A, B, out=C, out_dtype=odt)))
            _capi.lib().otter_gemm_set_debug(0)
            res.setdefault("torch", []).append(bench(lambda: torch.matmul(A, B.t(), out=C)) if odt == torch.bfloat16 else 0.0)
        print(json.dumps({"shape": [M, N, K], "out": str(odt).split(".")[-1],
                          "med_us": {("%dx%d%s" % (1 << ((o & 7) - 1), 32 >> ((o & 7) - 1), "N" if o & 8 else "M") if o else "default") if o != "torch" else "torch":
                                     round(statistics.median(v), 1) for o, v in res.items()}}), flush=True)
