"""Decoder-host attention at the C2 shape (B=8, 32 heads x 128, 512 tokens, causal + ALiBi): HIP flash kernels vs the
additive-mask SDPA path they replace.  Usage: flash_bench.py [B] [S] [causal] [sdpa] [padded]"""
import json, math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd.mpt import alibi_slopes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
CAUSAL = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
SDPA = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
PAD = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False   # right-padded batch: key_valid mask with lengths in [S/2, S]
H = 32

def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

qkv = torch.randn(B, S, 3, H, 128, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, S, H, 128, device="cuda").to(torch.bfloat16)
sl = alibi_slopes(H, 8).float().cuda()
scale = 1 / math.sqrt(128)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
KV = None
if PAD:
    lens = torch.randint(S // 2, S + 1, (B,), device="cuda")
    KV = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).to(torch.uint8).contiguous()
o, lse = ops.flash_attn_fwd(q, k, v, sl, KV, scale, CAUSAL)
dqkv = torch.empty_like(qkv)
ops.set_flash_variant(int(os.environ.get("FLASH_VARIANT", "0")))
res = {"B": B, "S": S, "H": H, "causal": CAUSAL, "variant": int(os.environ.get("FLASH_VARIANT", "0"))}
res["padded"] = PAD
res["hip_fwd_us"] = bench(lambda: ops.flash_attn_fwd(q, k, v, sl, KV, scale, CAUSAL))
res["hip_bwd_us"] = bench(lambda: ops.flash_attn_bwd(q, k, v, o, lse, dout, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], sl, KV, scale, CAUSAL))
fl = 4 * B * H * S * S * 128 * (0.5 if CAUSAL else 1.0)
res["hip_fwd_TF"] = fl / res["hip_fwd_us"] / 1e6
res["hip_bwd_TF"] = 2.5 * fl / res["hip_bwd_us"] / 1e6
if not SDPA:
    print(json.dumps({k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in res.items()}))
    sys.exit(0)
# the path it replaces: chunked views -> SDPA with an additive [1,H,S,S] bf16 mask
bias = (torch.arange(1 - S, 1, dtype=torch.float32, device="cuda").view(1, 1, 1, S) * sl.view(1, H, 1, 1))
causal = torch.ones(S, S, dtype=torch.bool, device="cuda").tril()
mask = bias.expand(-1, -1, S, -1).masked_fill(~causal, torch.finfo(torch.float32).min).to(torch.bfloat16).expand(B, -1, -1, -1)
qt, kt, vt = (t.transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
def sdpa_f():
    return F.scaled_dot_product_attention(qt, kt, vt, attn_mask=mask, scale=scale)
res["sdpa_fwd_us"] = bench(sdpa_f)
out = sdpa_f(); g = dout.transpose(1, 2)
def sdpa_b():
    out.backward(g, retain_graph=True)
res["sdpa_bwd_us"] = bench(sdpa_b)
print(json.dumps({k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in res.items()}))
