"""Persimmon / Fuyu-8B attention at the C5 shape (B=4, 64 heads x 64, 1396 tokens, causal, no bias): the head-pair flash kernels on the
interleaved [B,S,H,3,64] projection buffer against round 2's route (heads zero-padded to 128 on the 128-wide kernels).
Usage: flash64_bench.py [B] [S] [H]      FLASH_VARIANT=0|2 selects the block order (longest first / plain grid)."""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1396
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


buf = torch.randn(B, S, H, 3, 64, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, S, H, 64, device="cuda").to(torch.bfloat16)
scale = 1 / math.sqrt(64)
q, k, v = (buf[:, :, :, i] for i in range(3))
qc, kc = q.contiguous(), k.contiguous()       # what otter_qk_norm_rope_fwd hands over: compact q / k, v in place
variant = int(os.environ.get("FLASH_VARIANT", "0"))
ops.set_flash_variant(variant)
o, lse = ops.flash_attn_fwd(qc, kc, v, None, None, scale, True)
dbuf = torch.empty_like(buf)
dq, dk = torch.empty_like(qc), torch.empty_like(kc)
res = {"B": B, "S": S, "H": H, "variant": variant}
res["pair_fwd_us"] = bench(lambda: ops.flash_attn_fwd(qc, kc, v, None, None, scale, True))
res["pair_bwd_us"] = bench(lambda: ops.flash_attn_bwd(qc, kc, v, o, lse, dout, dq, dk, dbuf[:, :, :, 2], None, None, scale, True))
pad = lambda t: torch.cat([t, torch.zeros_like(t)], -1).contiguous()
qp, kp, vp, dop = pad(q), pad(k), pad(v), pad(dout)
op, lsep = ops.flash_attn_fwd(qp, kp, vp, None, None, scale, True)
dqp, dkp, dvp = torch.empty_like(qp), torch.empty_like(qp), torch.empty_like(qp)
res["pad128_fwd_us"] = bench(lambda: ops.flash_attn_fwd(qp, kp, vp, None, None, scale, True))
res["pad128_bwd_us"] = bench(lambda: ops.flash_attn_bwd(qp, kp, vp, op, lsep, dop, dqp, dkp, dvp, None, None, scale, True))
fl = 4 * B * H * S * S * 64 * 0.5            # model FLOPs of the 64-wide heads, causal
res["pair_fwd_TF"] = fl / res["pair_fwd_us"] / 1e6
res["pair_bwd_TF"] = 2.5 * fl / res["pair_bwd_us"] / 1e6
res["max_abs_diff_o"] = float((o.float() - op[..., :64].float()).abs().max())
print(json.dumps({k_: (round(v_, 2) if isinstance(v_, float) else v_) for k_, v_ in res.items()}))
