"""Where does the K-major GEMM differ from the NT GEMM?  (debug aid)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 256)
ta, tb = (sys.argv[4] == "1", sys.argv[5] == "1") if len(sys.argv) > 5 else (True, True)
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
base = ops.gemm_nt(A, B, out_dtype=torch.float32)
for trial in range(3):
    C = ops.gemm(A.t().contiguous() if ta else A, B.t().contiguous() if tb else B, ta, tb, out_dtype=torch.float32)
    bad = (C != base)
    nb = int(bad.sum())
    print("trial", trial, "mismatches", nb, "of", M * N)
    if nb:
        r, c = torch.nonzero(bad, as_tuple=True)
        r, c = r.cpu().numpy(), c.cpu().numpy()
        print(" rows%256 hist(16-bins):", np.bincount((r % 256) // 16, minlength=16))
        print(" cols%256 hist(16-bins):", np.bincount((c % 256) // 16, minlength=16))
        print(" tile_m hist:", np.bincount(r // 256, minlength=M // 256)[:16], " tile_n hist:", np.bincount(c // 256, minlength=N // 256)[:16])
        print(" row%16:", np.bincount(r % 16, minlength=16), " col%16:", np.bincount(c % 16, minlength=16))
        d = (C - base).abs()
        print(" max abs diff", float(d.max()), " base max", float(base.abs().max()))
        # is the wrong value a partial sum?  compare with products over K halves
        i, j = int(r[0]), int(c[0])
        a, b = A[i].float(), B[j].float()
        print(" first bad", i, j, "got", float(C[i, j]), "want", float(base[i, j]), "sum k<K/2", float((a[:K // 2] * b[:K // 2]).sum()),
              "sum k>=K/2", float((a[K // 2:] * b[K // 2:]).sum()), " per-64 partials", [round(float((a[k:k + 64] * b[k:k + 64]).sum()), 3) for k in range(0, K, 64)])
