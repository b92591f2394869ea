"""Where does the persistent dK/dV kernel differ from the per-block kernel? (diagnostic)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd.mpt import alibi_slopes
B, H, S = 8, 32, 512
g = torch.Generator().manual_seed(B + S)
qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16).cuda()
dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).cuda()
sl = alibi_slopes(H, 8).float().cuda()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
scale = 1 / math.sqrt(128)
outs = []
for variant in (0, 7, 0, 7):
    ops.set_flash_variant(variant)
    o, lse = ops.flash_attn_fwd(q, k, v, sl, None, scale, True)
    d = torch.full_like(qkv, float("nan"))
    ops.flash_attn_bwd(q, k, v, o, lse, dout, d[:, :, 0], d[:, :, 1], d[:, :, 2], sl, None, scale, True)
    torch.cuda.synchronize()
    outs.append(d.float())
print("pers run-to-run equal:", torch.equal(outs[0], outs[2]), " per-block run-to-run equal:", torch.equal(outs[1], outs[3]))
for i, name in enumerate(["dq", "dk", "dv"]):
    a, b = outs[0][:, :, i], outs[1][:, :, i]
    diff = (a - b).abs()
    print(name, "max abs diff", diff.max().item(), "max ref", b.abs().max().item(), "n differing", int((diff > 0).sum()))
    if diff.max() > 0:
        idx = (diff > 0).nonzero()
        print("  first differing [b, s, h, d]:", idx[:5].tolist())
        per_s = (diff > 0).sum(dim=(0, 2, 3))
        nz = per_s.nonzero().flatten()
        print("  rows (s) with differences: count", len(nz), "min", int(nz.min()), "max", int(nz.max()))
        per_d = (diff > 0).sum(dim=(0, 1, 2))
        print("  d columns with differences:", per_d.nonzero().flatten().tolist()[:40])
        print("  rel of worst:", (diff / (b.abs() + 1e-9)).max().item())

# which one is right?  fp64 reference of one differing head
a, b_ = outs[0][:, :, 1], outs[1][:, :, 1]
idx = ((a - b_).abs() > 0).nonzero()
seen = set()
for t in idx.tolist():
    bb, ss, hh, _ = t
    if (bb, hh) in seen: continue
    seen.add((bb, hh))
    if len(seen) > 4: break
    Q, K, V, DO = (x[bb, :, hh].double() for x in (q, k, v, dout))
    w = Q @ K.T * scale + sl[hh].double() * torch.arange(1 - S, 1, device="cuda", dtype=torch.float64)[None, :]
    w = w.masked_fill(~torch.ones(S, S, dtype=torch.bool, device="cuda").tril(), float("-inf"))
    P = torch.softmax(w, -1)
    dV = P.T @ DO
    dP = DO @ V.T
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    dK = dS.T @ Q * scale
    rows = sorted(set(r[1] for r in idx.tolist() if r[0] == bb and r[2] == hh))
    for ss in rows[:3]:
        ep = (outs[0][bb, ss, 1, hh].double() - dK[ss]).abs().max().item()
        eb = (outs[1][bb, ss, 1, hh].double() - dK[ss]).abs().max().item()
        nd = int(((outs[0][bb, ss, 1, hh] - outs[1][bb, ss, 1, hh]).abs() > 0).sum())
        print(f"b={bb} h={hh} key row {ss}: |pers - ref| {ep:.5f}  |per-block - ref| {eb:.5f}  differing d: {nd}  |dK row| max {dK[ss].abs().max().item():.3f}")
