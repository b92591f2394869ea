"""Whole-launch picture of the decoder-host flash kernels at C2: begin / end of EVERY workgroup (s_memrealtime, 10 ns ticks) and the CU it
ran on.  Needs `python -m otter_amd.build --flash-timing` and OTTER_LIB_PATH=otter_amd/lib/libotter_hip_flashtiming.so.

Prints per kernel: launch span, the distribution of workgroup lifetimes by tile count, how long each CU was occupied, the idle gaps."""
import ctypes, json, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi
from otter_amd.mpt import alibi_slopes

B, S, H = 8, 512, 32
qkv = torch.randn(B, S, 3, H, 128, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, S, H, 128, device="cuda").to(torch.bfloat16)
sl = alibi_slopes(H, 8).float().cuda()
blk = torch.zeros(3 * 4096 * 4, dtype=torch.int64, device="cuda")
lib = _capi.lib()
lib.otter_flash_set_block_stamps.argtypes = [ctypes.c_void_p]
assert lib.otter_flash_set_block_stamps(blk.data_ptr()) == 0
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
scale = 1 / math.sqrt(128)
dqkv = torch.empty_like(qkv)
for _ in range(4):
    o, lse = ops.flash_attn_fwd(q, k, v, sl, None, scale, True)
    ops.flash_attn_bwd(q, k, v, o, lse, dout, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], sl, None, scale, True)
torch.cuda.synchronize()
t = blk.cpu().numpy().reshape(3, 4096, 4)
names = ["forward", "dQ", "dK/dV"]
for kid in range(3):
    r = t[kid]
    n = int((r[:, 1] > 0).sum())
    r = r[:n]
    beg, end, hw, xcc = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    t0 = beg.min()
    span = (end.max() - t0) / 100.0
    life = (end - beg) / 100.0
    cu = ((xcc & 0xF) << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 0x7) << 8)   # (xcc, se, sh|cu)
    ucu = np.unique(cu)
    busy = np.array([life[cu == c].sum() for c in ucu])
    last = np.array([(end[cu == c].max() - t0) / 100.0 for c in ucu])
    first = np.array([(beg[cu == c].min() - t0) / 100.0 for c in ucu])
    nper = np.array([(cu == c).sum() for c in ucu])
    order = np.argsort(beg)
    print(json.dumps({"kernel": names[kid], "workgroups": n, "span_us": round(span, 2), "distinct_cus": len(ucu),
                      "wg_life_us": {"min": round(life.min(), 2), "mean": round(life.mean(), 2), "max": round(life.max(), 2)},
                      "first_wg_start_spread_us": round(float(np.percentile((beg - t0) / 100.0, 25)), 2),
                      "wgs_per_cu": {"min": int(nper.min()), "max": int(nper.max())},
                      "cu_first_start_us": {"min": round(first.min(), 2), "max": round(first.max(), 2)},
                      "cu_last_end_us": {"min": round(last.min(), 2), "mean": round(last.mean(), 2), "max": round(last.max(), 2)},
                      "cu_sum_of_wg_lifetimes_us": {"min": round(busy.min(), 2), "mean": round(busy.mean(), 2), "max": round(busy.max(), 2)}}))
    # by LPT rank (block id // (B*H): rank 0 = the longest blocks, launched first)
    nbh = B * H
    for rk in range(n // nbh):
        idx = np.arange(rk * nbh, (rk + 1) * nbh)
        bs, es = (beg[idx] - t0) / 100.0, (end[idx] - t0) / 100.0
        print(json.dumps({"kernel": names[kid], "rank": rk, "begin_us": [round(float(bs.min()), 2), round(float(np.median(bs)), 2), round(float(bs.max()), 2)],
                          "life_us": [round(float(life[idx].min()), 2), round(float(life[idx].mean()), 2), round(float(life[idx].max()), 2)],
                          "end_us": [round(float(es.min()), 2), round(float(np.median(es)), 2), round(float(es.max()), 2)]}))
    # who shares a CU: for each CU the ranks of its workgroups in begin order and their (begin, end)
    for c in ucu[:3]:
        m = np.where(cu == c)[0]
        m = m[np.argsort(beg[m])]
        print(json.dumps({"kernel": names[kid], "cu": int(c), "wgs": [[int(i // nbh), round(float((beg[i] - t0) / 100.0), 2), round(float((end[i] - t0) / 100.0), 2)] for i in m]}))
