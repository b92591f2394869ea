"""Load-path experiment on the FFN-shape GEMM: loads-only time of register staging (v2) vs LDS-DMA (v3)."""
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

M, N, K = 4096, 16384, 4096
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
cfgs = {"v3_full": (3, 16), "v3_loads": (3, 2 | 16), "v3_compute": (3, 1 | 16), "v3_neither": (3, 3 | 16),
        "v2_full": (2, 16), "v2_loads": (2, 2 | 16), "v2_compute": (2, 1 | 16), "v2_neither": (2, 3 | 16)}
res = {k: [] for k in cfgs}
for r in range(4):
    for k, (v, f) in cfgs.items():
        ops.set_gemm_variant(v)
        _capi.lib().otter_gemm_set_debug(f)
        res[k].append(bench(lambda: ops.gemm_nt(A, B, out=C)))
_capi.lib().otter_gemm_set_debug(0)
ops.set_gemm_variant(0)
print(json.dumps({k: [round(min(v), 1), round(statistics.median(v), 1)] for k, v in res.items()}))
