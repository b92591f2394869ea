"""Interleaved A/B rounds on the FFN-shape GEMM: K rotation on/off x padded/unpadded leading dimensions."""
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
Cp = torch.empty(M, N + 64, device="cuda", dtype=torch.bfloat16)[:, :N]
Ap = torch.empty(M, K + 64, device="cuda", dtype=torch.bfloat16)[:, :K]; Ap.copy_(A)
Bp = torch.empty(N, K + 64, device="cuda", dtype=torch.bfloat16)[:, :K]; Bp.copy_(B)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ops.set_gemm_variant(variant)
cfgs = {
    "rot": (0, A, B, C), "norot": (32, A, B, C),
    "rot_padC": (0, A, B, Cp), "norot_padC": (32, A, B, Cp),
    "rot_padAB": (0, Ap, Bp, C), "norot_padAB": (32, Ap, Bp, C),
    "rot_padall": (0, Ap, Bp, Cp), "norot_padall": (32, Ap, Bp, Cp),
    "rot_nostores": (16, A, B, C), "norot_nostores": (48, A, B, C),
    "torch": None,
}
res = {k: [] for k in cfgs}
for rnd in range(4):
    for k, c in cfgs.items():
        if c is None:
            res[k].append(bench(lambda: torch.matmul(A, B.t(), out=C)))
        else:
            _capi.lib().otter_gemm_set_debug(c[0])
            res[k].append(bench(lambda: ops.gemm_nt(c[1], c[2], out=c[3])))
_capi.lib().otter_gemm_set_debug(0)
print(json.dumps({k: [round(min(v), 1), round(statistics.median(v), 1)] for k, v in res.items()}))
