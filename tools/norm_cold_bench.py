"""LayerNorm stream kernels at the C2 shape with WARM operands (one buffer set, 235 MB: fits the 256 MB Infinity Cache, which is what
tools/norm_bench.py measures) and COLD ones (8 sets in rotation = 1.9 GB, as the 32 decoder layers of the step present them), beside a plain
fp32 copy of the same bytes and the fused AdamW sweep (the repository's best streaming kernel) for the achievable HBM rate."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

def bench(fn, iters=24):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

R, D, NS = 4096, 4096, 8
xs = [torch.randn(R, D, device="cuda") for _ in range(NS)]
deltas = [torch.randn(R, D, device="cuda").to(torch.bfloat16) for _ in range(NS)]
dys = [torch.randn(R, D, device="cuda").to(torch.bfloat16) for _ in range(NS)]
dress = [torch.randn(R, D, device="cuda") for _ in range(NS)]
outs = [torch.empty(R, D, device="cuda") for _ in range(NS)]
g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
_, mean, rstd = ops.layernorm_fwd(xs[0], g, b, torch.bfloat16)
MB = R * D / 1e6
res = {}
for mode, sel in (("warm", lambda i: 0), ("cold", lambda i: i % NS)):
    t = bench(lambda i: ops.layernorm_fwd(xs[sel(i)], g, b, torch.bfloat16)); res["ln_fwd " + mode] = [round(t, 1), round(MB * 6 / t, 2)]
    t = bench(lambda i: ops.add_layernorm_fwd(xs[sel(i)], deltas[sel(i)], g, b, torch.bfloat16)); res["add_ln_fwd " + mode] = [round(t, 1), round(MB * 12 / t, 2)]
    t = bench(lambda i: ops.layernorm_bwd(dys[sel(i)], xs[sel(i)], g, mean, rstd, torch.float32, dres=dress[sel(i)], need_dw=False)); res["ln_bwd_dres " + mode] = [round(t, 1), round(MB * 14 / t, 2)]
    t = bench(lambda i: outs[sel(i)].copy_(xs[sel(i)])); res["copy_f32 " + mode] = [round(t, 1), round(MB * 8 / t, 2)]
    t = bench(lambda i: torch.add(xs[sel(i)], dress[sel(i)], out=outs[sel(i)])); res["add_f32 (2 in, 1 out) " + mode] = [round(t, 1), round(MB * 12 / t, 2)]
for k, v in res.items():
    print("%-32s %7.1f us  %5.2f TB/s" % (k, v[0], v[1]))
