"""Own GEMM vs hipBLASLt with WARM operands (the same buffers every launch: 134 MB of weights + activations sit in the 256 MB Infinity Cache)
and with COLD operands (launches rotate over 8 weight / activation sets = 1.6 GB, as the 32 layers of the step do)."""
import json, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd._capi import EPI_GELU

M, D = 4096, 4096
NSET = 8
bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device="cuda") * float(os.environ.get("AMP", "0.05"))).to(bf)
xs = [rnd(M, D) for _ in range(NSET)]
Wu = [rnd(4 * D, D) for _ in range(NSET)]
hs = [rnd(M, 4 * D) for _ in range(NSET)]
Wd = [rnd(D, 4 * D) for _ in range(NSET)]
ubuf = torch.empty(M, 4 * D, device="cuda", dtype=bf)

def timeit(fn, iters=16):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / iters * 1e3, 1)

legs = {}
for mode, sel in (("warm", lambda i: 0), ("cold", lambda i: i % NSET)):
    legs["up   lib  " + mode] = lambda i, sel=sel: F.linear(xs[sel(i)], Wu[sel(i)])
    legs["up   own  " + mode] = lambda i, sel=sel: ops.gemm_nt(xs[sel(i)], Wu[sel(i)])
    legs["up+gelu own " + mode] = lambda i, sel=sel: ops.gemm_nt(xs[sel(i)], Wu[sel(i)], kind=EPI_GELU, C2=ubuf)
    legs["down lib  " + mode] = lambda i, sel=sel: F.linear(hs[sel(i)], Wd[sel(i)])
    legs["down own  " + mode] = lambda i, sel=sel: ops.gemm_nt(hs[sel(i)], Wd[sel(i)])
res = {k: [] for k in legs}
for rep in range(3):
    for k, fn in legs.items():
        res[k].append(timeit(fn))
for k, v in res.items():
    print("%-22s %s us  min %.1f" % (k, v, min(v)))
