"""In-kernel timeline (shader cycles) of one forward workgroup (q tile 3 = 8 key tiles at S=512) of the decoder-host flash
attention; needs `python -m otter_amd.build --flash-timing` and OTTER_LIB_PATH=otter_amd/lib/libotter_hip_flashtiming.so."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi
from otter_amd.mpt import alibi_slopes

B, S, H = 8, 512, 32
qkv = torch.randn(B, S, 3, H, 128, device="cuda").to(torch.bfloat16)
sl = alibi_slopes(H, 8).float().cuda()
st = torch.zeros(64, dtype=torch.int64, device="cuda")
lib = _capi.lib()
lib.otter_flash_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.otter_flash_set_stamps(st.data_ptr()) == 0
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
ops.set_flash_variant(1)   # the stamps live in the register-staged first version of the forward
for _ in range(3):
    ops.flash_attn_fwd(q, k, v, sl, None, 1 / math.sqrt(128), True)
torch.cuda.synchronize()
t = st.cpu().tolist()
rel = lambda i: t[i] - t[0]
out = {"gload0_issue": rel(1), "tiles": [], "loop_end": rel(60), "after_store": rel(61)}
for kt in range(8):
    b = 2 + 6 * kt
    out["tiles"].append({"kt": kt, "at_barrier1": rel(b), "lds_written": rel(b + 1) - rel(b), "S_mfma+mask": rel(b + 2) - rel(b + 1),
                         "shfl": rel(b + 3) - rel(b + 2), "exp+rescale": rel(b + 4) - rel(b + 3), "PV": rel(b + 5) - rel(b + 4)})
print(json.dumps(out, indent=0))
