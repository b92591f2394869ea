"""The frozen decoder's eight GEMMs per layer at C2 (4096 token rows, MPT-7B widths): hipBLASLt as the default path issues them
(forward F.linear(x, W); input gradient F.linear(dy, W^T copy)) + the separate GELU / GELU' kernels, against csrc/gemm.hip as the
OTTER_OWN_DECODER_GEMM=1 path issues them (forward gemm_nt, GELU fused into the up-projection's tail; input gradients on the K-major
kernel against the weight as stored, GELU' fused).  Legs rotated, one throw-away run in front of every timed burst (whatever
follows a hipBLASLt burst runs ~15 % slower for ~4 ms)."""
import json, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, functional as OF
from otter_amd._capi import EPI_GELU, EPI_GATE_BWD

M, D = 4096, 4096
dev = "cuda"
bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device=dev) * 0.05).to(bf)
x = rnd(M, D)
W = {"Wqkv": rnd(3 * D, D), "out_proj": rnd(D, D), "up_proj": rnd(4 * D, D), "down_proj": rnd(D, 4 * D)}
Wt = {k: v.t().contiguous() for k, v in W.items()}

def timeit(fn, iters=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

legs = {}
for name, w in W.items():
    N, K = w.shape
    xin = rnd(M, K)
    dy = rnd(M, N)
    y = torch.empty(M, N, device=dev, dtype=bf)
    legs[name + " fwd  lib"] = (lambda xin=xin, w=w: F.linear(xin, w))
    legs[name + " fwd  own"] = (lambda xin=xin, w=w: ops.gemm_nt(xin, w))
    legs[name + " dgrad lib"] = (lambda dy=dy, wt=Wt[name]: F.linear(dy, wt))
    legs[name + " dgrad own"] = (lambda dy=dy, w=w: ops.gemm(dy, w, False, True))
u = rnd(M, 4 * D); du = rnd(M, 4 * D); dyd = rnd(M, D)
ubuf = torch.empty_like(u)
legs["up_proj+gelu fwd  lib"] = lambda: ops.gelu_fwd(F.linear(x, W["up_proj"]))
legs["up_proj+gelu fwd  own"] = lambda: ops.gemm_nt(x, W["up_proj"], kind=EPI_GELU, C2=ubuf)

def lib_down_dgrad():
    g = F.linear(dyd, Wt["down_proj"])
    return ops.gelu_bwd(u, g)
legs["down_proj dgrad+gelu' lib"] = lib_down_dgrad
legs["down_proj dgrad+gelu' own"] = lambda: ops.gemm(dyd, W["down_proj"], False, True, kind=EPI_GATE_BWD, aux=u, aux_gelu=True)

res = {k: [] for k in legs}
for rep in range(3):
    for k, fn in legs.items():
        res[k].append(round(timeit(fn), 1))
for k, v in res.items():
    print("%-32s %s us   min %.1f" % (k, v, min(v)))
tot = lambda tag, names: sum(min(res[n + tag]) for n in names)
plain = ["Wqkv fwd ", "out_proj fwd ", "down_proj fwd ", "Wqkv dgrad", "out_proj dgrad", "up_proj dgrad"]
lib = tot(" lib", plain) + min(res["up_proj+gelu fwd  lib"]) + min(res["down_proj dgrad+gelu' lib"])
own = tot(" own", plain) + min(res["up_proj+gelu fwd  own"]) + min(res["down_proj dgrad+gelu' own"])
print(json.dumps({"per_layer_us": {"hipblaslt_plus_elementwise": round(lib, 1), "own_fused": round(own, 1)}, "x32_layers_ms": {"lib": round(lib * 32 / 1e3, 2), "own": round(own * 32 / 1e3, 2)}}))
