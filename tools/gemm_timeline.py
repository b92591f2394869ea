"""Tile-phase timeline of the one-wave-per-SIMD GEMM (variants 18-20): shader-clock timestamps at tile start / prologue done /
K loop done / tail done / tile end for two blocks (otter_gemm_set_debug bit 64).  Usage: gemm_timeline.py [variant] [M N K]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi as K

v = int(sys.argv[1]) if len(sys.argv) > 1 else 18
M, N, Kd = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 16384, 4096)
epi = sys.argv[5] if len(sys.argv) > 5 else "store"      # store | gelu | res | gate_bwd | store_f32
A = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
B = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.set_gemm_variant(v)
C2 = torch.empty_like(C)
aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
R = torch.randn(M, N, device="cuda")
Cf = torch.empty(M, N, device="cuda")
gate = torch.full((1,), 0.5, device="cuda")
part = torch.zeros(ops.gemm_num_partials(M, N, torch.bfloat16), device="cuda")


def launch():
    if epi == "gelu":
        ops.gemm_nt(A, B, out=C, kind=K.EPI_GELU, C2=C2)
    elif epi == "res":
        ops.gemm_nt(A, B, out=Cf, kind=K.EPI_SCALE_RES, gate=gate, R=R)
    elif epi == "gate_bwd":
        ops.gemm_nt(A, B, out=C, kind=K.EPI_GATE_BWD, gate=gate, aux=aux, aux_gelu=True, partial=part)
    elif epi == "store_f32":
        ops.gemm_nt(A, B, out=Cf)
    else:
        ops.gemm_nt(A, B, out=C)


for _ in range(3):
    launch()
torch.cuda.synchronize()
K.lib().otter_gemm_set_debug(64)
launch()
torch.cuda.synchronize()
K.lib().otter_gemm_set_debug(0)
buf = np.zeros(512, dtype=np.uint64)
K.check(K.lib().otter_gemm_read_timeline(buf.ctypes.data_as(ctypes.c_void_p), 512), "timeline")
t = buf.reshape(2, 4, 8, 8).astype(np.int64)
ntile = max(1, (M // 256) * (N // 256) // 256)
print("epilogue:", epi)
for b in range(1):
    t0 = t[b, :, 0, 0].min()
    print("block %d (cycles since its first tile start; wave 0 .. 3)" % (0 if b == 0 else 131))
    for tile in range(min(ntile, 2)):
        for w in range(0, 4, 3):
            m = t[b, w, tile, :5] - t0
            print("  tile %d wave %d: start %8d | prologue %6d | kloop %7d | tail %6d | sync %5d" %
                  (tile, w, m[0], m[1] - m[0], m[2] - m[1], m[3] - m[2], m[4] - m[3]))
ops.set_gemm_variant(0)
