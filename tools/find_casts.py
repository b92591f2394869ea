"""Which host lines issue dtype-converting copies in one training step?  (torch.profiler, aten::_to_copy / aten::copy_ grouped by Python stack)"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from otter_amd.train import TrainStep

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = bench.build_model(dev, debug_layers=layers)
step = TrainStep(model)
batch = bench.synth_batch(model, 8, 512, dev, 0)[:4]
for _ in range(2):
    step(*batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(*batch)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for e in prof.events():
    if e.name in ("aten::_to_copy", "aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::contiguous", "aten::clone"):
        st = [f for f in (e.stack or []) if "otter_amd" in f or "bench.py" in f]
        key = (e.name, st[0].split("/")[-1] if st else "?", str(e.input_shapes)[:60])
        agg[key] += 1
        tim[key] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for k, v in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:40]:
    print("%6d x %8.1f us total  %s" % (v, tim[k], k))
