"""Anatomy of ONE gated cross-attention block forward + backward at C2 that ADDS UP to the measured time (VERDICT r4 item 2's alternative
deliverable): tools/block_profile.py under rocprofv3 --kernel-trace (csv); over the last iterations the wall time per iteration is split
into (a) time during which exactly one kernel runs, attributed to that kernel, (b) time during which two or more kernels run (the two HIP
streams of functional._SideStream), attributed to the kernel that started first, and (c) idle time (no kernel resident: launch gaps).  Per
kernel: launches per iteration, summed duration, ATTRIBUTED (wall-clock) time, and the residual against a per-kernel floor:
  MFMA-bound GEMMs: flops / (the measured MFMA rate of this box: otter_probe_mfma on random operands, bench.py's calibration)
  HBM-bound sweeps: bytes / (the measured 1 GiB copy rate)
so that every microsecond of the block is named.  Usage (GPU box): python tools/block_timeline.py [iters] > gpurun_out/<tag>_block_anatomy.txt"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out = "/tmp/prof_blocktl"
subprocess.run("rm -rf %s; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d %s -o p -- python %s/tools/block_profile.py %d > /tmp/prof_blocktl.log 2>&1"
               % (out, out, ROOT, iters), shell=True)
wall_line = open("/tmp/prof_blocktl.log").read().strip().split("\n")[-2:]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# iteration boundaries: the first LayerNorm forward of each iteration (norm_fwd / add_layernorm on x); use the cast of dy? simplest: text_time_kernel marks a forward
marks = [i for i, e in enumerate(ev) if "text_time_kernel" in e[2]]
use = marks[-(min(16, len(marks) - 2) + 1):]
lo, hi = use[0], use[-1]
n_it = len(use) - 1
win = ev[lo:hi]
t0, t1 = win[0][0], ev[hi][0]
wall = (t1 - t0) / n_it / 1e3
# sweep
points = []
for s, e, n in win:
    points.append((s, 1, n))
    points.append((min(e, t1), -1, n))
points.sort(key=lambda p: (p[0], -p[1]))
active = []          # kernels in start order
attr = collections.Counter()
idle = 0
overlap = 0
prev = t0
for t, kind, n in points:
    dt = t - prev
    if dt > 0:
        if not active:
            idle += dt
        else:
            attr[active[0]] += dt
            if len(active) > 1:
                overlap += dt
    prev = t
    if kind == 1:
        active.append(n)
    else:
        active.remove(n)
dur, cnt = collections.Counter(), collections.Counter()
for s, e, n in win:
    dur[n] += e - s
    cnt[n] += 1
cal = None
try:
    sys.path.insert(0, ROOT)
    import torch
    import bench
    cal = bench.machine_calibration(torch.device("cuda", 0))
except Exception as ex:  # pragma: no cover
    print("# calibration unavailable:", ex)
mf = cal["random_operands"]["tflops"] * 1e12 if cal else 2.0e15
bw = cal["hbm_copy_tbps"] * 1e12 if cal else 4.7e12
M, D, F = 4096, 4096, 16384
gf = 2.0 * M * D * F


def floor_us(name, calls):
    """(floor in us per iteration, what bounds it) for the kernels whose algorithmic work is known at the C2 shape."""
    if "gemm_bf16_t4_kernel<0, 0, true, true" in name: return 2 * gf / mf * 1e6, "2 weight gradients 16384x4096x4096, MFMA"
    if "gemm_bf16_t4_kernel<3, 0, false, true" in name: return gf / mf * 1e6, "dU = dy W2 with GELU' tail, MFMA"
    if "gemm_bf16_t4_kernel<1, 0, false, false" in name: return gf / mf * 1e6, "FF1 forward with GELU tail, MFMA"
    if "gemm_bf16_t4_kernel<0, 0, false, true" in name: return gf / mf * 1e6, "df = dU W1, MFMA"
    if "gemm_bf16_t4_kernel<2, 0, false, false" in name: return (gf + 2.0 * M * D * 512) / mf * 1e6, "FF2 forward + to_out (gate, fp32 residual), MFMA"
    if "norm_bwd_dx" in name: return 2 * (M * D * (2 + 4 + 4 + 4)) / bw * 1e6, "2 LayerNorm backward sweeps, HBM"
    if "norm_fwd" in name: return 2 * (M * D * (4 + 2)) / bw * 1e6, "2 LayerNorm forward sweeps, HBM"
    if "norm_bwd_dw_partial" in name: return 2 * (M * D * (2 + 4)) / bw * 1e6, "2 LayerNorm weight-gradient column sums, HBM"
    if "cast_kernel" in name: return calls / n_it * (M * D * 6) / bw * 1e6, "fp32 -> bf16 casts, HBM"
    if "transpose_vec_kernel<unsigned short" in name: return calls / n_it * (M * 512 * 4) / bw * 1e6, "bf16 operand transposes, HBM"
    if "transpose_vec_kernel<float" in name: return (M * D * (4 + 2 + 2)) / bw * 1e6, "dx1 -> bf16 copy + transpose, HBM"
    if "gemm_bf16_s4h_kernel" in name or "gemm_bf16_s4_kernel" in name or "gemm_bf16_t4_kernel<0, 0, false, false" in name:
        return calls / n_it * (2.0 * M * 512 * D) / mf * 1e6, "skinny projections 4096x512x4096 (or smaller), MFMA"
    return 0.0, ""


print("# tools/block_timeline.py: one OtterGatedCrossAttentionBlock forward + backward at C2 (B = 8 x 512 tokens), %d iterations under rocprofv3 --kernel-trace" % n_it)
print("# %s" % " | ".join(wall_line))
if cal:
    print("# calibration of this box in this run: MFMA %.0f TFLOP/s on random operands at %.2f GHz (%.0f on zeros), 1 GiB copy %.2f TB/s"
          % (cal["random_operands"]["tflops"], cal["random_operands"]["clock_ghz"], cal["zero_operands"]["tflops"], cal["hbm_copy_tbps"]))
print("# wall per iteration %.1f us = attributed kernel time %.1f + idle (no kernel resident) %.1f;  two kernels resident during %.1f us;  sum of kernel durations %.1f us"
      % (wall, sum(attr.values()) / n_it / 1e3, idle / n_it / 1e3, overlap / n_it / 1e3, sum(dur.values()) / n_it / 1e3))
print("%-86s %7s %9s %9s %9s %9s  %s" % ("kernel", "calls", "dur us", "attrib us", "floor us", "residual", "floor model"))
tot_floor = 0.0
for n, t in attr.most_common(40):
    fl, what = floor_us(n, cnt[n])
    tot_floor += fl
    print("%-86s %7.1f %9.1f %9.1f %9.1f %9.1f  %s" % (n[:86], cnt[n] / n_it, dur[n] / n_it / 1e3, t / n_it / 1e3, fl, t / n_it / 1e3 - fl, what))
print("# sum of floors %.1f us = %.3f of the wall time; the block's 3 x 141.94 GF x 8 = 3.41 TF at the measured MFMA rate alone: %.1f us"
      % (tot_floor, tot_floor / wall, 3 * 141.94e9 * 8 / mf * 1e6))
