"""One OtterGatedCrossAttentionBlock forward + backward at the C2 shapes (B=8 x 512 tokens, 64 media latents), N iterations -- run under
`rocprofv3 --kernel-trace --stats` for the per-kernel anatomy of the block (bench.py's roofline.gated_block is the timed version).
Usage: block_profile.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.modeling_otter import OtterGatedCrossAttentionBlock

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = OtterGatedCrossAttentionBlock(dim=4096, dim_visual=1024).to(dev)
with torch.no_grad():
    for n, p in blk.named_parameters():
        if p.ndim >= 2:
            p.normal_(0.0, 0.02)
    blk.attn_gate.fill_(0.5); blk.ff_gate.fill_(0.5)
B, T = 8, 512
x = torch.randn(B, T, 4096, device=dev, requires_grad=True)
media = torch.randn(B, 1, 64, 1024, device=dev)
dy = torch.randn(B, T, 4096, device=dev)
ml = torch.zeros(B, T, dtype=torch.bool, device=dev); ml[:, 1] = True
for i in range(iters + 2):
    if i == 2:
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    for p in blk.parameters():
        p.grad = None            # as after zero_grad(set_to_none=True): weight gradients are written, not accumulated
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x, media, media_locations=ml, attend_previous=True)
    y.backward(dy)
e.record(); torch.cuda.synchronize()
print("ms per fwd+bwd:", s.elapsed_time(e) / iters)
