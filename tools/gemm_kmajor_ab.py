"""K-major operands of otter_gemm (transpose reads inside variant 26) against the round-2 path (otter_transpose of the operand(s) +
otter_gemm_nt) at the gated block's backward shapes.  Interleaved rounds, medians; events on the launch stream.
Usage: gemm_kmajor_ab.py [rounds]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import ops
from otter_amd._capi import EPI_GATE_BWD, EPI_STORE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda"
torch.manual_seed(0)
R, D, F = 4096, 4096, 16384          # token rows, model width, FFN width
g = torch.full((1,), 0.5, device=dev)


def ev(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


dy = torch.randn(R, D, device=dev).to(torch.bfloat16)
dU = torch.randn(R, F, device=dev).to(torch.bfloat16) * 0.1
h = torch.randn(R, F, device=dev).to(torch.bfloat16)
f = torch.randn(R, D, device=dev).to(torch.bfloat16)
u = torch.randn(R, F, device=dev).to(torch.bfloat16)
W1 = (torch.randn(F, D, device=dev) * 0.02).to(torch.bfloat16)      # [16384, 4096]
W2 = (torch.randn(D, F, device=dev) * 0.02).to(torch.bfloat16)      # [4096, 16384]
W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
o32a = torch.empty(F, D, device=dev)
o32b = torch.empty(D, F, device=dev)
part = torch.empty(ops.gemm_num_partials(R, F, torch.bfloat16), device=dev)

cases = {
    "dW1 = dU^T f      [16384,4096,K=4096] f32": (
        lambda: ops.gemm_nt(ops.transpose(dU, torch.bfloat16), ops.transpose(f, torch.bfloat16), out=o32a),
        lambda: ops.gemm(dU, f, True, True, out=o32a),
        lambda: ops.gemm_nt(W1, W1[:4096], out=o32a[:, :4096]) if False else None),
    "dW2 = g dy^T h    [4096,16384,K=4096] f32": (
        lambda: ops.gemm_nt(ops.transpose(dy, torch.bfloat16), ops.transpose(h, torch.bfloat16), out=o32b, kind=EPI_STORE, gate=g),
        lambda: ops.gemm(dy, h, True, True, out=o32b, kind=EPI_STORE, gate=g), None),
    "dU = dy W2 gelu'  [4096,16384,K=4096] bf16": (
        lambda: ops.gemm_nt(dy, W2t, kind=EPI_GATE_BWD, gate=g, aux=u, aux_gelu=True, partial=part),
        lambda: ops.gemm(dy, W2, False, True, kind=EPI_GATE_BWD, gate=g, aux=u, aux_gelu=True, partial=part), None),
    "df = dU W1        [4096,4096,K=16384] bf16": (
        lambda: ops.gemm_nt(dU, W1t),
        lambda: ops.gemm(dU, W1, False, True), None),
}
# GEMM alone (operands already transposed) for the first two, to separate kernel speed from the deleted transposes
dUT, fT, dyT, hT = (ops.transpose(t, torch.bfloat16) for t in (dU, f, dy, h))
alone = {
    "dW1 = dU^T f      [16384,4096,K=4096] f32": lambda: ops.gemm_nt(dUT, fT, out=o32a),
    "dW2 = g dy^T h    [4096,16384,K=4096] f32": lambda: ops.gemm_nt(dyT, hT, out=o32b, kind=EPI_STORE, gate=g),
}
res = {k: ([], [], []) for k in cases}
for _ in range(2):                    # throw-away: clocks settle
    for k, (a, b, _) in cases.items():
        ev(a, 2); ev(b, 2)
for r in range(rounds):
    order = list(cases.items())
    if r % 2:
        order.reverse()
    for k, (a, b, _) in order:
        legs = [(0, a), (1, b)] if r % 2 == 0 else [(1, b), (0, a)]
        for idx, fn in legs:
            res[k][idx].append(ev(fn))
        if k in alone:
            res[k][2].append(ev(alone[k]))
print("%-48s %14s %14s %14s %8s" % ("launch", "transpose+NT us", "K-major us", "NT alone us", "ratio"))
tot_a = tot_b = 0.0
for k, (a, b, c) in res.items():
    ma, mb = statistics.median(a), statistics.median(b)
    tot_a += ma; tot_b += mb
    print("%-48s %14.1f %14.1f %14s %8.3f" % (k, ma, mb, ("%.1f" % statistics.median(c)) if c else "-", mb / ma))
print("sum: %.1f -> %.1f us per gated block backward (%.1f us saved)" % (tot_a, tot_b, tot_a - tot_b))
