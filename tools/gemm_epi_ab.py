"""FFN-shape GEMM with each fused tail (the in-situ mix of bench.py), for A/B runs of two builds via OTTER_LIB_PATH."""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd._capi import EPI_STORE, EPI_GELU, EPI_SCALE_RES, EPI_GATE_BWD

def bench(fn, iters=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

M, N, K = 4096, 16384, 4096
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
gate = torch.full((1,), 0.5, device="cuda")
C16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
C16b = torch.empty_like(C16)
C32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
part = torch.empty(ops.gemm_num_partials(M, N, torch.bfloat16), dtype=torch.float32, device="cuda")
# down-projection shape for SCALE_RES: [4096 x 4096 x 16384]
A2 = torch.randn(M, N, device="cuda").to(torch.bfloat16)
B2 = torch.randn(K, N, device="cuda").to(torch.bfloat16)
R = torch.randn(M, K, device="cuda")
Y = torch.empty(M, K, device="cuda")
cases = {
    "store_bf16": lambda: ops.gemm_nt(A, B, out=C16),
    "store_f32": lambda: ops.gemm_nt(A, B, out=C32),
    "gelu_2out": lambda: ops.gemm_nt(A, B, out=C16, kind=EPI_GELU, C2=C16b),
    "gate_bwd_gelu": lambda: ops.gemm_nt(A, B, out=C16, kind=EPI_GATE_BWD, gate=gate, aux=aux, aux_gelu=True, partial=part),
    "scale_res_f32": lambda: ops.gemm_nt(A2, B2, out=Y, kind=EPI_SCALE_RES, gate=gate, R=R),
}
from otter_amd import _capi
res = {}
for rnd in range(3):   # interleaved: wide (16-byte) fused tail vs the 4-wide one (debug bit 256)
    for flag, tag in ((0, "wide"), (256, "narrow")):
        _capi.lib().otter_gemm_set_debug(flag)
        for k, f in cases.items():
            res.setdefault(k + ":" + tag, []).append(bench(f))
_capi.lib().otter_gemm_set_debug(0)
print(json.dumps({k: round(statistics.median(v), 1) for k, v in res.items()}))
