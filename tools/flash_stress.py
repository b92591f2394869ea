"""Stress of the decoder-host flash backward (persistent dK/dV form at B x H = 256): thousands of launches, every result compared bit for bit
with the first one -- a data race between the DMA ring, the staging area and the LDS transpose would show as a run-to-run difference."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd.mpt import alibi_slopes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
# (the last three shapes run the PER-BLOCK dK/dV kernel: B x H is neither a multiple of the CU count nor >= 8 rounds -- the form whose row-store
#  epilogue stages in the Q / dO ring)
for (B, H, S, padded) in ((8, 32, 512, False), (8, 32, 384, True), (4, 64, 1024, False), (3, 20, 448, False), (1, 24, 640, False), (5, 12, 300, True)):
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16).cuda()
    dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).cuda()
    sl = alibi_slopes(H, 8).float().cuda()
    kv = None
    if padded:
        lens = torch.tensor([S, 300, 129, 128, 127, S, 1, 200][:B])
        kv = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).cuda()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    scale = 1 / math.sqrt(128)
    o, lse = ops.flash_attn_fwd(q, k, v, sl, kv, scale, True)
    ref = torch.full_like(qkv, float("nan"))
    ops.flash_attn_bwd(q, k, v, o, lse, dout, ref[:, :, 0], ref[:, :, 1], ref[:, :, 2], sl, kv, scale, True)
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    d = torch.empty_like(qkv)
    # a second stream keeps the memory system busy with unrelated traffic part of the time
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    for it in range(N):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        d.fill_(float("nan"))
        o2, lse2 = ops.flash_attn_fwd(q, k, v, sl, kv, scale, True)
        ops.flash_attn_bwd(q, k, v, o2, lse2, dout, d[:, :, 0], d[:, :, 1], d[:, :, 2], sl, kv, scale, True)
        if not (torch.equal(d, ref) and torch.equal(o2, o)):
            bad += 1
    torch.cuda.synchronize()
    print("B=%d H=%d S=%d padded=%s: %d launches, %d differ from the first, %.1f s" % (B, H, S, padded, N, bad, time.time() - t0))
