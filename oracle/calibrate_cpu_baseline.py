"""Calibrates bench.py's `cpu_baseline` (kind "port" = the numpy oracle) against the REFERENCE's own modules on the same host, same
sample, same thread count.  Needs the reference's model sources: /root/reference in the build container (8 threads), or -- round 4 -- an
uncommitted scratch copy staged under oracle/_ref/ for ONE gpurun call, so that the ratio is measured on the GPU NODE's own host cores
(tools/stage_reference_loop.sh stage-models; OTTER_REF_ROOT points at it):

    python oracle/calibrate_cpu_baseline.py [out.json]     ->  profiles/r03_cpu_baseline_calibration.json (build container)
                                                                profiles/r04_cpu_baseline_calibration_gpu_node.json (GPU node)

Timed (1 pair: 1 x 224^2 image worth of latents + 512 tokens, fp32, forward + backward, min of 3 after a warm-up):
  * OtterGatedCrossAttentionBlock(dim=4096, dim_visual=1024)      reference: src/otter_ai/models/otter/modeling_otter.py:343-395
  * OtterPerceiverResampler(dim=1024, depth=6)                    :187-235
  * one frozen MPT block (forward + input gradient)               src/otter_ai/models/mpt/blocks.py:19-88 through MPTBlock
against oracle.otter_oracle's functions on identical shapes.  bench.py reads the JSON and prints `cpu_baseline.port_vs_reference`
(ratio of times, reference / port: > 1 means the numpy port is the FASTER of the two, i.e. the reported baseline flatters the CPU)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import otter_oracle as O  # noqa: E402
from oracle import synth  # noqa: E402
from oracle.gen_golden import REF, import_reference  # noqa: E402


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    mo = import_reference()
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    D, Dv, T = 4096, 1024, 512
    try:
        from threadpoolctl import threadpool_info

        blas = max([int(i.get("num_threads", 0)) for i in threadpool_info() if i.get("user_api") == "blas"] or [0])
    except Exception:
        blas = 0
    out = {"host_threads": threads, "numpy_blas_threads": blas, "cpu_count": os.cpu_count(), "sample": "1 pair: [1, 512, 4096] tokens, [1, 1, 64, 1024] latents / [1, 1, 1, 256, 1024] CLIP features, fp32"}
    r = np.random.default_rng(0)

    # ---- gated cross-attention block ----
    blk = mo.OtterGatedCrossAttentionBlock(dim=D, dim_visual=Dv)
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(0.5)
    x = torch.randn(1, T, D)
    media = torch.randn(1, 1, 64, Dv)
    ml = torch.zeros(1, T, dtype=torch.bool)
    ml[0, 1] = True

    def ref_block():
        xx = x.clone().requires_grad_(True)
        y = blk(xx, media, media_locations=ml, attend_previous=True)
        y.backward(torch.ones_like(y))
        for p in blk.parameters():
            p.grad = None

    p = {"b." + k: v.detach().numpy() for k, v in blk.state_dict().items()}
    xn, mn, mln = x.numpy(), media.numpy(), ml.numpy()

    def port_block():
        y, c = O.gated_xattn_block_fwd(p, "b.", xn, mn, mln)
        O.gated_xattn_block_bwd(p, "b.", np.ones_like(y), c)

    out["gated_block"] = {"reference_s": best(ref_block), "port_s": best(port_block)}
    del blk, p

    # ---- perceiver resampler ----
    per = mo.OtterPerceiverResampler(dim=Dv, depth=6)
    feats = torch.randn(1, 1, 1, 256, Dv)

    def ref_per():
        f = feats.clone().requires_grad_(True)
        y = per(f)
        y.backward(torch.ones_like(y))
        for q in per.parameters():
            q.grad = None

    pp = {"p." + k: v.detach().numpy() for k, v in per.state_dict().items()}
    fn_ = feats.numpy()

    def port_per():
        y, c = O.perceiver_resampler_fwd(pp, "p.", fn_)
        O.perceiver_resampler_bwd(pp, "p.", np.ones_like(y), c)

    out["perceiver"] = {"reference_s": best(ref_per), "port_s": best(port_per)}
    del per, pp

    # ---- frozen MPT block: forward + input gradient ----
    sys.path.insert(0, REF)
    from src.otter_ai.models.mpt.blocks import MPTBlock  # type: ignore
    from src.otter_ai.models.mpt.attention import build_alibi_bias, build_attn_bias  # type: ignore

    attn_config = dict(attn_type="multihead_attention", attn_pdrop=0.0, attn_impl="torch", qk_ln=False, clip_qkv=None, softmax_scale=None,
                       prefix_lm=False, attn_uses_sequence_id=False, alibi=True, alibi_bias_max=8)
    mb = MPTBlock(d_model=D, n_heads=32, expansion_ratio=4, attn_config=attn_config, resid_pdrop=0.0, norm_type="low_precision_layernorm",
                  verbose=0, no_bias=True)
    for q in mb.parameters():
        q.requires_grad_(False)
    bias = build_alibi_bias(32, T, full=False, alibi_bias_max=8, dtype=torch.float32)

    def ref_mpt():
        xx = x.clone().requires_grad_(True)
        y = mb(xx, attn_bias=bias, is_causal=True)[0]
        y.backward(torch.ones_like(y))

    pm = {"m." + k: v.detach().numpy() for k, v in mb.state_dict().items()}
    nb = O.mpt_attn_bias(32, T, 2048)

    def port_mpt():
        y, c, _ = O.mpt_block_fwd(pm, "m.", xn, 32, nb)
        O.mpt_block_bwd_input(pm, "m.", np.ones_like(y), c)

    out["mpt_block"] = {"reference_s": best(ref_mpt), "port_s": best(port_mpt)}

    # the step mix bench.py extrapolates with: 8 gated + 32 MPT + 1 perceiver (CLIP and the un-embedding are torch / numpy GEMMs on both sides)
    ref_t = 8 * out["gated_block"]["reference_s"] + 32 * out["mpt_block"]["reference_s"] + out["perceiver"]["reference_s"]
    port_t = 8 * out["gated_block"]["port_s"] + 32 * out["mpt_block"]["port_s"] + out["perceiver"]["port_s"]
    out["step_mix"] = {"reference_s": ref_t, "port_s": port_t, "port_vs_reference": ref_t / port_t}
    for k in ("gated_block", "perceiver", "mpt_block"):
        out[k]["port_vs_reference"] = out[k]["reference_s"] / out[k]["port_s"]
    out["reference_root"] = "staged scratch copy (GPU node)" if "OTTER_REF_ROOT" in os.environ else REF
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_cpu_baseline_calibration.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
