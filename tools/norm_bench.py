"""LayerNorm row kernels at the C2 residual-stream shape (4096 rows x 4096, fp32 stream, bf16 branches): us and effective TB/s.
Usage: OTTER_NORM_VARIANT=0|1 norm_bench.py   (0 = generic kernels, 1 = coalesced kernels; csrc/norm.hip)"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

R, D = 4096, 4096
x = torch.randn(R, D, device="cuda")
delta = torch.randn(R, D, device="cuda").to(torch.bfloat16)
g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
dy = torch.randn(R, D, device="cuda").to(torch.bfloat16)
dres = torch.randn(R, D, device="cuda")
dxb = torch.empty(R, D, device="cuda", dtype=torch.bfloat16)
y, mean, rstd = ops.layernorm_fwd(x, g, b, torch.bfloat16)
out = {"variant": os.environ.get("OTTER_NORM_VARIANT", "default")}
MB = R * D / 1e6
t = bench(lambda: ops.layernorm_fwd(x, g, b, torch.bfloat16)); out["ln_fwd"] = [round(t, 1), round(MB * 6 / t, 2)]
t = bench(lambda: ops.add_layernorm_fwd(x, delta, g, b, torch.bfloat16)); out["add_ln_fwd"] = [round(t, 1), round(MB * 12 / t, 2)]
t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, torch.float32, dres=dres, need_dw=False)); out["ln_bwd_dres"] = [round(t, 1), round(MB * 14 / t, 2)]
t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, torch.float32, dres=dres, need_dw=False, dx_bf16=dxb)); out["ln_bwd_dres_bf16copy"] = [round(t, 1), round(MB * 16 / t, 2)]
t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, torch.float32, need_dw=False)); out["ln_bwd"] = [round(t, 1), round(MB * 10 / t, 2)]
t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, torch.float32, dres=dres, need_dw=True)); out["ln_bwd_dres_dw"] = [round(t, 1), round(MB * 14 / t, 2)]
t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, torch.float32, need_dw=True, need_dx=False)); out["ln_bwd_dw_only"] = [round(t, 1), round(MB * 6 / t, 2)]
t = bench(lambda: ops.colsum(dy, None, R)); out["colsum_bf16"] = [round(t, 1), round(MB * 2 / t, 2)]
print(json.dumps(out))
