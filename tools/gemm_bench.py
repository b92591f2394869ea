"""Micro-benchmark of the hand-written bf16 MFMA GEMM schedules at the shapes of the gated cross-attention block
(interleaved rounds, random data, torch.cuda.Event on the launch stream).  Usage: python tools/gemm_bench.py"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops  # noqa: E402
from otter_amd._capi import EPI_GELU, EPI_STORE  # noqa: E402

SHAPES = [(4096, 16384, 4096), (4096, 4096, 16384), (16384, 4096, 4096), (4096, 512, 4096), (4096, 4096, 512), (512, 1024, 1024)]


def bench(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    res = []
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        row = {"shape": [M, N, K]}
        for rnd in range(2):
            for v in (1, 3, 4, 5):
                ops.set_gemm_variant(v)
                ms = bench(lambda: ops.gemm_nt(A, B, out=C))
                row.setdefault("v%d" % v, []).append(round(2 * M * N * K / ms / 1e9, 1))
            ms = bench(lambda: torch.matmul(A, B.t(), out=C))
            row.setdefault("torch", []).append(round(2 * M * N * K / ms / 1e9, 1))
        ops.set_gemm_variant(2)
        ms = bench(lambda: ops.gemm_nt(A, B, out=C, kind=EPI_GELU))
        row["v2_gelu"] = round(2 * M * N * K / ms / 1e9, 1)
        ops.set_gemm_variant(0)
        res.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
