"""clip + AdamW over ~1.4 B fp32 parameters (the C2 trainable set's size) with bf16 shadows: ms per step and effective TB/s.
Usage: OTTER_ADAMW_VARIANT=n adamw_bench.py [n_tensors] [numel_each]   (variants: csrc/optim.hip)"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.optim import FusedAdamW

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 16384
ps = [torch.nn.Parameter(torch.randn(ne, device="cuda") * 0.02) for _ in range(nt)]
for p in ps:
    p.grad = torch.randn(ne, device="cuda") * 1e-3
opt = FusedAdamW(ps, lr=1e-5, weight_decay=0.1, max_grad_norm=1.0)
opt.refresh_shadows = False
for _ in range(2):
    opt.step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    opt.step()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
n = nt * ne
print(json.dumps({"variant": os.environ.get("OTTER_ADAMW_VARIANT", "0"), "params_B": round(n / 1e9, 3), "ms": round(ms, 3),
                  "TBps_32B_per_param": round(n * 32 / ms / 1e9, 2)}))
