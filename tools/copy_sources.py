"""Which call sites produce the dtype-copy kernels of a C2 training step?  torch.profiler with stacks, aten::copy_ / aten::to grouped by
input shape and python frame.  Usage: copy_sources.py [layers]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from otter_amd.train import TrainStep

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
model = bench.build_model(dev, debug_layers=layers)
step = TrainStep(model, lr=1e-5, weight_decay=0.1, max_grad_norm=1.0)
batch = bench.synth_batch(model, 8, 512, dev, 1000)[:4]
for _ in range(2):
    step(*batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(*batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy") and ev.device_time_total > 0:
        shp = str(ev.input_shapes[:2])
        frames = [f for f in (ev.stack or []) if "otter_amd" in f or "bench.py" in f]
        key = (ev.name, shp, frames[0][-90:] if frames else "?")
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%5d calls %9.1f us  %s %s  <- %s" % (v[0], v[1], k[0], k[1], k[2]))
