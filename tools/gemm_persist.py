"""Persistent grid (256 workgroups walking the tiles) vs one workgroup per tile for the default large-grid GEMM, with hipBLASLt beside it;
rotated leg order + throw-away runs, K sweep for the fixed cost.  Usage: gemm_persist.py"""
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

lib = _capi.lib()
for (M, N) in [(4096, 16384), (16384, 4096), (4096, 4096)]:
    out = {"MN": [M, N]}
    fits = {}
    for K in (1024, 4096, 16384) if (M, N) == (4096, 4096) else (1024, 2048, 4096, 8192):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        legs = [("persistent", 0), ("per_tile", 16 << 9), ("torch", None)]
        res = {}
        for r in range(3):
            for name, dbg in legs[r % 3:] + legs[:r % 3]:
                if dbg is None:
                    fn = lambda: torch.matmul(A, B.t(), out=C)
                else:
                    lib.otter_gemm_set_debug(dbg)
                    fn = lambda: ops.gemm_nt(A, B, out=C)
                bench(fn)
                res.setdefault(name, []).append(bench(fn))
            lib.otter_gemm_set_debug(0)
        for k, v in res.items():
            fits.setdefault(k, []).append((K, statistics.median(v)))
    for k, pts in fits.items():
        n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        out[k] = {"us": [round(p[1], 1) for p in pts], "fixed_us": round(a, 1), "mainloop_TF": round(2.0 * M * N / b / 1e6, 1)}
    print(json.dumps(out), flush=True)
