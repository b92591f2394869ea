"""Does the output row stride set the fixed cost of the 4096 x 16384 GEMM?  Same product, C as a view of a wider buffer (ldc = N + pad).
Usage: gemm_ldc.py"""
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

M, N = 4096, 16384
for odt in (torch.bfloat16, torch.float32):
    out = {"out": str(odt).split(".")[-1]}
    for pad in (256, 0, 64, 0, 1024, 0, 256):
        pts = []
        for K in (1024, 4096):
            A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
            C = torch.empty(M, N + pad, device="cuda", dtype=odt)[:, :N]
            fn = lambda: ops.gemm_nt(A, B, out=C, out_dtype=odt)
            bench(fn)
            pts.append(statistics.median([bench(fn) for _ in range(3)]))
        b = (pts[1] - pts[0]) / 3072.0
        out.setdefault("pad%d" % pad, []).append({"us": [round(p, 1) for p in pts], "fixed_us": round(pts[0] - b * 1024, 1)})
    print(json.dumps(out), flush=True)
