"""Is the in-situ slow-down of the FFN-shape GEMM (398 us in a burst on warm operands -> 439 us inside the training step) CYCLES or CLOCK?
The tile-phase timeline of variant 26 (otter_gemm_set_debug bit 64: s_memtime at tile start / prologue done / K loop done / tail done) in two
conditions on one box: (a) a short burst on one operand set (what tools/gemm_timeline.py and the stand-alone tables measure) and (b) in the
middle of a SUSTAINED loop over 8 rotating operand sets (1.3 GB: nothing survives in the 256 MB Infinity Cache; ~150 launches = the power
state of the training step).  Per condition: launch time by events, cycles per tile phase, and the shader clock implied by
cycles-per-launch / microseconds-per-launch.  Usage: gemm_timeline_sustained.py [epi: store|gelu|store_f32] [kmajor: 0|ab]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops, _capi as K

epi = sys.argv[1] if len(sys.argv) > 1 else "store"
km = sys.argv[2] if len(sys.argv) > 2 else "0"
M, N, Kd = 4096, 16384, 4096
NSET = 8
if os.environ.get("CU_BUDGET"):      # round 6: persistent grid on fewer CUs -- do the tails shorten when fewer CUs write at once?
    print("persistent workgroups:", ops.set_gemm_cu_budget(int(os.environ["CU_BUDGET"])))
bf = torch.bfloat16
AMP = float(os.environ.get("AMP", "0.05"))      # AMP=0: zero operands (no data-dependent power: is the K loop bound by the clock or by the memory path?)
As = [(torch.randn(M, Kd, device="cuda") * AMP).to(bf) for _ in range(NSET)]
Bs = [(torch.randn(N, Kd, device="cuda") * AMP).to(bf) for _ in range(NSET)]
if km == "ab":
    As = [a.t().contiguous() for a in As]
    Bs = [b.t().contiguous() for b in Bs]
C = torch.empty(M, N, device="cuda", dtype=bf)
C2 = torch.empty_like(C)
Cf = torch.empty(M, N, device="cuda")


def launch(i):
    a, b = As[i % NSET], Bs[i % NSET]
    if km == "ab":
        ops.gemm(a, b, True, True, out=Cf if epi == "store_f32" else C)
    elif epi == "gelu":
        ops.gemm_nt(a, b, out=C, kind=K.EPI_GELU, C2=None if os.environ.get("NO_C2") else C2)
    elif epi == "store_f32":
        ops.gemm_nt(a, b, out=Cf)
    else:
        ops.gemm_nt(a, b, out=C)


def timeline(i):
    K.lib().otter_gemm_set_debug(64)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    launch(i)
    e.record()
    torch.cuda.synchronize()
    K.lib().otter_gemm_set_debug(0)
    buf = np.zeros(512, dtype=np.uint64)
    K.check(K.lib().otter_gemm_read_timeline(buf.ctypes.data_as(ctypes.c_void_p), 512), "timeline")
    return buf.reshape(2, 4, 8, 8).astype(np.int64), s.elapsed_time(e) * 1e3


def report(tag, t, us_tl, us_avg):
    print("== %s: %.1f us per launch (events, mean of the last 20); the stamped launch itself %.1f us" % (tag, us_avg, us_tl))
    for b in range(2):
        w = 0
        rows = []
        for tile in range(4):
            m = t[b, w, tile, :5]
            rows.append((m[1] - m[0], m[2] - m[1], m[3] - m[2], m[4] - m[3]))
        total = t[b, w, 3, 4] - t[b, w, 0, 0]
        r = np.array(rows, dtype=np.float64)
        print("   block %3d wave 0, 4 tiles: prologue %6.0f  K loop %7.0f (%.0f per K-tile)  tail %6.0f  sync %5.0f cycles per tile; 4 tiles %d cycles -> %.3f GHz at the launch time"
              % (0 if b == 0 else 131, r[:, 0].mean(), r[:, 1].mean(), r[:, 1].mean() / 64, r[:, 2].mean(), r[:, 3].mean(), total, total / us_tl / 1e3))


# (a) burst, warm operands
for _ in range(3):
    launch(0)
torch.cuda.synchronize()
evs = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); launch(0); e.record(); evs.append((s, e))
torch.cuda.synchronize()
ua = sum(s.elapsed_time(e) for s, e in evs) / len(evs) * 1e3
t, us = timeline(0)
report("burst, warm operands (%s, kmajor=%s)" % (epi, km), t, us, ua)
# (b) sustained, cold operands
torch.cuda.synchronize()
evs = []
for i in range(170):
    if i >= 150:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); launch(i); e.record(); evs.append((s, e))
    else:
        launch(i)
t, us = timeline(170)
ub = sum(s.elapsed_time(e) for s, e in evs) / len(evs) * 1e3
report("sustained loop, 8 rotating operand sets (%s, kmajor=%s)" % (epi, km), t, us, ub)
