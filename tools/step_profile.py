"""Who launches which elementwise kernels in one train step?  torch.profiler with stacks on a reduced-depth model
(--layers N decoder layers; the per-layer pattern is what matters).  Usage: step_profile.py [layers] [kernel-substring]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from otter_amd.train import TrainStep

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
needles = sys.argv[2:] if len(sys.argv) > 2 else ["CUDAFunctor_add"]
dev = torch.device("cuda:0")
model = bench.build_model(dev, 0, debug_layers=layers)
batch = bench.synth_batch(model, 8, 512, dev, 0)[:4]
step = TrainStep(model, lr=1e-5, weight_decay=0.1, max_grad_norm=1.0, autocast_dtype=torch.bfloat16)
for _ in range(2):
    step(*batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(*batch)
    torch.cuda.synchronize()
for needle in needles:
    agg = collections.Counter()
    tot = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"):
            kern = [k for k in e.kernels if needle in k.name]
            if not kern:
                continue
            stack = [f for f in (e.stack or []) if "otter_amd" in f or "bench.py" in f or "torch/autograd" in f][:3]
            key = (e.name, str(e.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in stack) or "(no py frame: autograd engine)")
            agg[key] += len(kern)
            tot[key] += sum(k.duration for k in kern)
    print("=== kernels matching %r: %d launches, %.0f us per step" % (needle, sum(agg.values()), sum(tot.values())))
    for key, n in sorted(agg.items(), key=lambda kv: -tot[kv[0]])[:14]:
        print("  ", n, round(tot[key]), "us", key)
