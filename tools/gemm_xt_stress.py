"""Round 6: race screen of the cross-tile form of variant 26 (hand-counted vmcnt / raw barriers across tile boundaries): every launch of a mix of
shapes -- several tiles per workgroup, edge tiles between full ones, nk = 4 .. 256, odd CU budgets, all four epilogues -- is compared BIT FOR BIT
with the plain form's result (otter_gemm_set_debug bits 14-15), `reps` times each, while a second stream keeps the memory system busy.
Usage: gemm_xt_stress.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd import _capi as K
from otter_amd._capi import EPI_GELU, EPI_GATE_BWD, EPI_SCALE_RES, EPI_STORE

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bf = torch.bfloat16
torch.manual_seed(0)


def mode(xt):
    K.check(K.lib().otter_gemm_set_debug(((2 if xt else 1) << 14)), "set_debug")


shapes = [(4096, 16384, 4096), (4096, 4096, 16384), (4352, 4096, 384), (4096, 4096, 512), (4100, 12288, 256), (8192, 8192, 1024), (3900, 16384, 768)]
noise_a = torch.randn(64 << 20, device="cuda")
noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
bad = total = 0
for (M, N, Kd) in shapes:
    A = (torch.randn(M, Kd, device="cuda") * 0.05).to(bf)
    B = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    R = torch.randn(M, N, device="cuda")
    aux = torch.randn(M, N, device="cuda").to(bf)
    gate = torch.full((1,), 0.3, device="cuda")
    part = torch.empty(ops.gemm_num_partials(M, N, bf), device="cuda")
    C2 = torch.empty(M, N, device="cuda", dtype=bf)
    legs = {
        "store bf16": lambda: ops.gemm_nt(A, B),
        "store f32": lambda: ops.gemm_nt(A, B, out_dtype=torch.float32),
        "gelu": lambda: (ops.gemm_nt(A, B, kind=EPI_GELU, C2=C2), C2.clone())[0],
        "scale_res f32": lambda: ops.gemm_nt(A, B, out_dtype=torch.float32, kind=EPI_SCALE_RES, gate=gate, R=R),
        "gate_bwd": lambda: ops.gemm_nt(A, B, kind=EPI_GATE_BWD, gate=gate, aux=aux, aux_gelu=True, partial=part),
    }
    for budget in (0, 250, 37):
        ops.set_gemm_cu_budget(budget)
        for name, fn in legs.items():
            mode(False)
            ref = fn().clone()
            mode(True)
            for r in range(reps if budget == 0 else max(4, reps // 8)):
                if r % 3 == 0:
                    with torch.cuda.stream(side):      # a concurrent HBM-bound copy: other latencies, other landing order of the DMA pieces
                        noise_b.copy_(noise_a)
                got = fn()
                total += 1
                if not torch.equal(got, ref):
                    bad += 1
                    d = (got.float() - ref.float()).abs()
                    print("MISMATCH %s %s budget %d rep %d: %d elements differ, max %.4g" % ((M, N, Kd), name, budget, r, int((d > 0).sum()), float(d.max())))
    ops.set_gemm_cu_budget(0)
    torch.cuda.synchronize()
    print("shape %s done" % ((M, N, Kd),), flush=True)
K.check(K.lib().otter_gemm_set_debug(0), "set_debug")
print("gemm_xt_stress: %d launches compared bit for bit with the plain form, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
