"""Run the FFN-shape bf16 GEMM a few times (for rocprofv3 --pmc passes).  Usage: gemm_one.py [variant] [iters] [M N K] [kmajor: 0 | b | ab]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M, N, K = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (4096, 16384, 4096)
km = sys.argv[6] if len(sys.argv) > 6 else "0"
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if v >= 0:
    ops.set_gemm_variant(v)
for _ in range(it):
    if km != "0":
        if _ == 0:
            At, Bt = A.t().contiguous(), B.t().contiguous()
        ops.gemm(At if km == "ab" else A, Bt, km == "ab", True, out=C)
    elif v >= 0:
        ops.gemm_nt(A, B, out=C)
    else:  # hipBLASLt through torch, for side-by-side counter passes
        torch.matmul(A, B.t(), out=C)
torch.cuda.synchronize()
print("done", v, it, M, N, K)
