"""Round 6d: launch time of the FFN-shape products on the library named by OTTER_LIB_PATH -- for the ablation builds of variant 26's K loop
(-DOTTER_T4_ABL=mask: 1 no LDS-DMA in the K loop, 2 no fragment reads, 4 no barriers; results wrong by construction, TIMING ONLY).
8 rotating operand sets (cold), random (AMP=0.05) and zero operands, median of `reps` rounds of 8 launches.
Usage: OTTER_LIB_PATH=... gemm_abl_time.py [reps]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M, D, FF, NSET = 4096, 4096, 16384, 8
bf = torch.bfloat16
label = os.path.basename(os.environ.get("OTTER_LIB_PATH", "default"))
for amp in (0.05, 0.0):
    xs = [(torch.randn(M, D, device="cuda") * amp).to(bf) for _ in range(NSET)]
    W1 = [(torch.randn(FF, D, device="cuda") * amp).to(bf) for _ in range(NSET)]
    hs = [(torch.randn(M, FF, device="cuda") * amp).to(bf) for _ in range(NSET)]
    W2 = [(torch.randn(D, FF, device="cuda") * amp).to(bf) for _ in range(NSET)]
    o1 = torch.empty(M, FF, device="cuda", dtype=bf)
    o2 = torch.empty(M, D, device="cuda", dtype=bf)
    legs = {"up   K=4096 ": lambda i: ops.gemm_nt(xs[i], W1[i], out=o1), "down K=16384": lambda i: ops.gemm_nt(hs[i], W2[i], out=o2)}
    for name, fn in legs.items():
        for i in range(NSET):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(NSET):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / NSET * 1e3)
        print("%-34s %s %s  us per launch: min %.1f median %.1f" % (label, "random" if amp else "zeros ", name, min(ts), statistics.median(ts)), flush=True)
