"""Ablation of the phased bf16 GEMM at the FFN shape with the compile-time diagnostics builds (results are wrong by
construction; timing only).  Usage: gemm_ablate.py [variant] -- run once per build, e.g.
    for m in 0 1 2 3 4 8 9 10; do python -m otter_amd.build --diag $m; done            (here, cross-compiled)
    for m in 0 1 2 3 4 8 9 10; do OTTER_LIB_PATH=otter_amd/lib/libotter_hip_diag$m.so python tools/gemm_ablate.py 10; done
mask bits: 1 = no DMA, 2 = no MFMA, 4 = no epilogue traffic, 8 = no LDS fragment reads, 16 = no epilogue, 128 = plain instead of
non-temporal stores in the fused tail (this one computes correct results)."""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M, N, K = 4096, 16384, 4096
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.set_gemm_variant(variant)
us = [bench(lambda: ops.gemm_nt(A, B, out=C)) for _ in range(3)]
ops.set_gemm_variant(0)
print(json.dumps({"lib": os.path.basename(os.environ.get("OTTER_LIB_PATH") or "libotter_hip.so"), "variant": variant,
                  "us_med": round(statistics.median(us), 1), "us_min": round(min(us), 1),
                  "TF_equiv": round(2 * M * N * K / statistics.median(us) / 1e6, 1)}), flush=True)
