"""In-kernel timeline (s_memtime ticks = shader-clock cycles, ~1.9-2.1 per ns against the 10 ns stamps of tools/flash_launch_picture.py) of the first dK/dV workgroup (key block 0: 16 query tiles at S=512) of the
decoder-host flash attention backward.  Needs `python -m otter_amd.build --flash-timing` and
OTTER_LIB_PATH=otter_amd/lib/libotter_hip_flashtiming.so."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi
from otter_amd.mpt import alibi_slopes

B, S, H = 8, 512, 32
qkv = torch.randn(B, S, 3, H, 128, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, S, H, 128, device="cuda").to(torch.bfloat16)
sl = alibi_slopes(H, 8).float().cuda()
st = torch.zeros(192, dtype=torch.int64, device="cuda")
lib = _capi.lib()
lib.otter_flash_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.otter_flash_set_stamps(st.data_ptr()) == 0
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
scale = 1 / math.sqrt(128)
o, lse = ops.flash_attn_fwd(q, k, v, sl, None, scale, True)
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.flash_attn_bwd(q, k, v, o, lse, dout, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], sl, None, scale, True)
torch.cuda.synchronize()
t = st.cpu().tolist()
rel = lambda i: t[i] - t[0]
rows = []
for it in range(16):
    b = 1 + 5 * it
    rows.append({"it": it, "top": rel(b), "S,dP+reads+dma": t[b + 1] - t[b], "softmax A": t[b + 2] - t[b + 1], "dV,dK(0)+softmax B": t[b + 3] - t[b + 2],
                 "barrier+dV,dK(1)+row reads": t[b + 4] - t[b + 3]})
print(json.dumps({"prologue_to_loop": rel(1), "loop_end": rel(90), "stores_done": rel(91)}))
# persistent per-head kernel: per key block (rank) the end of its tile loop, its stores issued, the switch done (slots 80 + 3 rank ..)
print(json.dumps({"persistent_switches": [{"rank": r, "loop_end": rel(80 + 3 * r), "stores_issued": rel(81 + 3 * r) - rel(80 + 3 * r),
                                           "switch_rest": (rel(82 + 3 * r) - rel(81 + 3 * r)) if r < 3 else None} for r in range(4)]}))
for r in rows:
    print(json.dumps(r))
# forward (v2), block 0 = the last query tile of (batch 0, head 0): 8 key tiles
f = t[96:]
print(json.dumps({"fwd_prologue": f[1] - f[0], "fwd_loop_end": f[90] - f[0], "fwd_stores_done": f[91] - f[0]}))
for kt in range(8):
    b = 1 + 5 * kt
    print(json.dumps({"kt": kt, "top": f[b] - f[0], "wait+barrier": f[b + 1] - f[b], "dma_issue": f[b + 2] - f[b + 1], "S+softmax": f[b + 3] - f[b + 2], "PV": f[b + 4] - f[b + 3]}))
