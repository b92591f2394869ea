"""Ablation of the bf16 GEMM at the FFN shape: full vs no-global-loads vs no-MFMA (diagnostic flags; wrong results)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi

def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

M, N, K = 4096, 16384, 4096
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for v in (3,):
    ops.set_gemm_variant(v)
    row = {"variant": v}
    for name, flag in (("full", 0), ("noload", 1), ("nomfma", 2), ("neither", 3), ("noepi", 4), ("nokloop", 8), ("nothing", 12), ("nostores", 16), ("nokloop_nostores", 24)):
        _capi.lib().otter_gemm_set_debug(flag)
        ms = bench(lambda: ops.gemm_nt(A, B, out=C))
        row[name + "_us"] = round(ms * 1e3, 1)
    _capi.lib().otter_gemm_set_debug(0)
    row["full_TF"] = round(2 * M * N * K / row["full_us"] / 1e6, 1)
    row["noload_TF_equiv"] = round(2 * M * N * K / row["noload_us"] / 1e6, 1)
    print(json.dumps(row), flush=True)
ops.set_gemm_variant(0)
