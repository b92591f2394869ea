"""GPU parity at FULL SIZE: the 32-layer OTTER-MPT7B that bench.py times (bench.build_model: MPT-7B decoder, 8 gated
cross-attention blocks with gates 0.5, CLIP ViT-L/14, 6-layer perceiver; src/otter_ai/models/otter/utils/Otter-MPT7B-config.json) against the
numpy oracle's restatement of the reference's forward / generate (modeling_otter.py:917-1042) on the host, at BASELINE configs[0] = C1
(batch 1, 1 x 224^2 image, 32-token prompt with <image> / <answer> / <|endofchunk|>).

Both precisions share ONE set of weights: every parameter is rounded to a bf16-representable value once, so the fp32 parity mode, the bf16
production mode and the fp32 oracle see identical numbers and the bf16 figures measure activation rounding through 32 decoder layers + 8 gated
blocks + 24 CLIP layers + 6 perceiver layers, not weight rounding.

  (i)  fp32 parity mode (fp32 weights, no autocast: gemm_f32_kernel / VALU attention cores / rocBLAS fp32 in the frozen host):
       north-star tolerance -- logits rtol <= 1e-3 (relative to max AND per token row), loss 1e-4, greedy token ids BIT-EXACT in both
       decode modes of SURVEY 3.2.
  (ii) bf16 production mode (frozen weights bf16, bf16 autocast: the kernels the benchmark runs): drift REPORTED (per-row error, cosine,
       loss, greedy agreement, and the fp32 top-2 margin of every greedy step -- gpurun_out/parity_metrics.jsonl -> profiles/, DESIGN.md
       section 5) and bounded by the stated tolerances below.
  (iii) C2-shaped batch (B x 512 tokens, B = OTTER_G1_C2_BATCH, default 4 since round 6; 8 = the bench batch): bf16 loss and per-row logits vs the oracle.
  (iv) THE TRAINING STEP (round 5, VERDICT r4 item 1 -- the metric is a training step, not a forward): the composed backward through
       all 32 layers (flash dQ/dK/dV or the fp32 cores, K-major dgrad / wgrad GEMMs, fork-LayerNorm, the deferred residual add,
       SparseEmbedSink, weight gradients) + grad-norm clip + FusedAdamW of otter_amd.train.TrainStep on a 2 x 128-token batch (several
       key blocks per flash launch) and the plain-autograd backward on the C1 prompt, against O.otter_backward (hand-derived numpy
       backward of the reference's trainable set, modeling_otter.py:897-905; step = instruction_following.py:200-250) and a numpy
       AdamW on the oracle's gradients: fp32 parity mode <= 1e-3 per tensor, bf16 production mode reported per tensor and bounded.

The host leg is ~0.7 TFLOP per C1 forward but streams 32 GB of fp32 weights per pass; the whole module takes a few minutes."""
import os
import sys
import time

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import otter_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NEW_TOKENS = 8
# bf16 production mode, stated tolerances.  Measured on MI355X (profiles/r04_run1_parity_metrics.jsonl):
#   C1  fp32 mode: logits 4.4e-6 (max) / 4.3e-6 (per row), loss 1.7e-7, greedy ids bit-exact in both decode modes
#   C1  bf16 mode: logits 8.5e-3 per row, cosine 0.99997, loss 2.2e-4, 8 / 8 greedy tokens agree in both modes
#                  (fp32 top-2 margins of the eight steps: 0.8 % .. 7.6 % of the largest logit)
#   C2 (B = 2) bf16 forward: logits 9.2e-3 per row, cosine 0.99997, loss 2.7e-5
# -> tolerance = about 2x the measured figure.
BF16_ROW_TOL, BF16_COS_MIN, BF16_LOSS_TOL = 2e-2, 0.9999, 2e-3


def _host_state(model):
    out = {}
    for k, v in model.state_dict().items():
        out[k] = v.detach().to(torch.float32).cpu().numpy() if v.is_floating_point() else v.cpu().numpy()
    return out


@pytest.fixture(scope="module")
def full():
    import psutil

    import bench

    if psutil.virtual_memory().available < 60 << 30:
        pytest.skip("the fp32 host copy of OTTER-MPT7B needs ~33 GB (+ activations): not enough free host memory on this box")
    t0 = time.time()
    model = bench.build_model(DEV, seed=0, frozen_dtype=torch.float32)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).to(torch.float32))       # one weight set for fp32 mode, bf16 mode and the oracle
    model.eval()
    n_par = sum(p.numel() for p in model.parameters())
    assert 8.0e9 < n_par < 8.3e9 and len(model.lang_encoder._get_decoder_layers()) == 32
    assert sum(1 for l in model.lang_encoder._get_decoder_layers() if l.gated_cross_attn_layer is not None) == 8
    p_host = _host_state(model)
    spec = O.OtterSpec(n_layers=32, d_model=4096, n_heads=32, max_seq_len=2048, cross_attn_every_n_layers=4,
                       media_token_id=model.media_token_id, clip_heads=16, clip_patch=14)
    vision_x, ids, mask, labels, _ = bench.synth_batch(model, 1, 32, DEV, seed=4242)
    assert int((ids == model.media_token_id).sum()) == 1 and int((labels != -100).sum()) > 0
    print("[g1] model built + %.1f GB copied to the host in %.0f s" % (sum(v.nbytes for v in p_host.values()) / 2**30, time.time() - t0), flush=True)
    return dict(model=model, p=p_host, spec=spec, batch=(vision_x, ids, mask, labels), bench=bench, memo={})


def _oracle_c1(full):
    """fp32 oracle on the host: forward with loss + greedy in both decode modes (computed once per module)."""
    m = full["memo"]
    if "c1" not in m:
        vision_x, ids, _, labels = full["batch"]
        vx, idn, lab = vision_x.cpu().numpy(), ids.cpu().numpy(), labels.cpu().numpy()
        t0 = time.time()
        out = O.otter_forward(full["p"], full["spec"], vx, idn, None, lab, keep_caches=False)
        t_fwd = time.time() - t0
        res = dict(logits=out["logits"], loss=float(out["loss"]), t_fwd=t_fwd)
        for use_cache in (False, True):
            trace = []
            t0 = time.time()
            res["greedy_cache" if use_cache else "greedy_nocache"] = O.greedy_decode(full["p"], full["spec"], vx, idn, NEW_TOKENS, None, use_cache, trace=trace)
            res["trace_cache" if use_cache else "trace_nocache"] = trace
            res["t_greedy_cache" if use_cache else "t_greedy_nocache"] = time.time() - t0
        print("[g1] oracle: forward %.1f s, greedy %.1f / %.1f s (%d host threads)" % (t_fwd, res["t_greedy_nocache"], res["t_greedy_cache"], os.cpu_count()), flush=True)
        m["c1"] = res
    return m["c1"]


def _run_hip(full, bf16: bool):
    model = full["model"]
    vision_x, ids, mask, labels = full["batch"]
    res = {}
    with torch.no_grad():
        if bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(vision_x=vision_x.to(torch.bfloat16), lang_x=ids, attention_mask=mask, labels=labels)
        else:
            out = model(vision_x=vision_x, lang_x=ids, attention_mask=mask, labels=labels)
        res["logits"] = out.logits.float().cpu().numpy()
        res["loss"] = float(out.loss)
        for use_cache in (False, True):
            if bf16:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    toks = model.generate(vision_x=vision_x.to(torch.bfloat16), lang_x=ids, max_new_tokens=NEW_TOKENS, use_cache=use_cache, eos_token_id=-1)
            else:
                toks = model.generate(vision_x=vision_x, lang_x=ids, max_new_tokens=NEW_TOKENS, use_cache=use_cache, eos_token_id=-1)
            res["greedy_cache" if use_cache else "greedy_nocache"] = toks.cpu().numpy()
    return res


def _agreement(got, want, prompt_len):
    """Fraction of generated tokens that agree up to and including the first disagreement of each row (later ones are conditioned on
    different prefixes), and the index of that first disagreement (-1: none)."""
    agree = total = 0
    first = -1
    for r in range(got.shape[0]):
        for t in range(prompt_len, got.shape[1]):
            total += 1
            if got[r, t] == want[r, t]:
                agree += 1
            else:
                first = t - prompt_len if first < 0 else min(first, t - prompt_len)
                break
    return agree / max(total, 1), first


def test_c1_fp32_parity_mode_logits_loss_and_greedy_bit_exact(full):
    ref = _oracle_c1(full)
    got = _run_hip(full, bf16=False)
    e_max = G.rel_err(got["logits"], ref["logits"])
    e_row = G.row_rel_err(got["logits"][0], ref["logits"][0])
    cos = G.cosine(got["logits"], ref["logits"])
    e_loss = abs(got["loss"] - ref["loss"]) / abs(ref["loss"])
    margins = [float(t["margin"].min() / t["absmax"].max()) for t in ref["trace_nocache"]]
    G.record("full_model_c1_fp32", logits_rel_max=e_max, logits_row_rel=e_row, cosine=cos, loss=got["loss"], loss_ref=ref["loss"], loss_rel=e_loss,
             greedy_nocache_exact=float(np.array_equal(got["greedy_nocache"], ref["greedy_nocache"])),
             greedy_cache_exact=float(np.array_equal(got["greedy_cache"], ref["greedy_cache"])),
             min_top2_margin_rel=min(margins), oracle_forward_s=ref["t_fwd"], host_threads=float(os.cpu_count()))
    assert e_max < 1e-3 and e_row < 1e-3, (e_max, e_row)          # north_star: logits rtol <= 1e-3
    assert e_loss < 1e-4, (got["loss"], ref["loss"])
    assert np.array_equal(got["greedy_nocache"], ref["greedy_nocache"])   # north_star: bit-exact token ids at greedy T=0
    assert np.array_equal(got["greedy_cache"], ref["greedy_cache"])
    assert got["greedy_nocache"].shape == (1, 32 + NEW_TOKENS)


# ----------------------------------------------------------------------------------------------------------------------
# (iv) the training step
# ----------------------------------------------------------------------------------------------------------------------
LP = "lang_encoder.transformer."
LR, WD, MAX_NORM, B1, B2, EPS = 1e-5, 0.1, 1.0, 0.9, 0.999, 1e-8        # the recipe's optimizer (instruction_following.py:385-400,246-251)
FULL_BLOCKS = (3, 31)                                                    # gated blocks compared element by element (first and last)


def _decays(name):
    """train_utils.py:170-171 (get_grouped_params): weight decay only on the gated blocks' matrices."""
    return "gated_cross_attn_layer" in name and "ff_gate" not in name and "attn_gate" not in name and "norm" not in name and "bias" not in name


def _train_batch(full):
    if "train_batch" not in full["memo"]:
        full["memo"]["train_batch"] = full["bench"].synth_batch(full["model"], 2, 128, DEV, seed=31337)[:4]
    return full["memo"]["train_batch"]


def _oracle_grads(full, key, batch):
    """O.otter_forward (with caches) + O.otter_backward on the host, once per batch: {name: gradient} of every trainable parameter."""
    m = full["memo"]
    if key not in m:
        vision_x, ids, _, labels = batch
        t0 = time.time()
        out = O.otter_forward(full["p"], full["spec"], vision_x.cpu().numpy(), ids.cpu().numpy(), None, labels.cpu().numpy(), keep_caches=True)
        t1 = time.time()
        g = O.otter_backward(full["p"], full["spec"], out)
        loss = float(out["loss"])
        del out
        sq = sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values())
        print("[g1] oracle %s: forward %.1f s, backward %.1f s, %d gradient tensors, |g| = %.4e" % (key, t1 - t0, time.time() - t1, len(g), np.sqrt(sq)), flush=True)
        m[key] = dict(g=g, loss=loss, norm=float(np.sqrt(sq)))
    return m[key]


def _tensor_err(got, ref):
    """(relative l2 error of the whole tensor, max error relative to the largest entry, cosine)."""
    a, b = np.asarray(got, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    nb = float(np.sqrt(b @ b)) + 1e-300
    return float(np.sqrt(((a - b) ** 2).sum())) / nb, float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300)), float(a @ b / (np.sqrt(a @ a) * nb + 1e-300))


def _compared_names(model):
    """Element-wise comparison set: the resampler (all of it), gated blocks 3 and 31 (every tensor), the tied embedding; the six other
    gated blocks are compared on norms / gates in full and on their matrices through ||g|| and a 4096-entry strided sample."""
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    full_set = [n for n in names if n.startswith("perceiver.") or n == LP + "wte.weight"
                or any(n.startswith(LP + "blocks.%d.gated_cross_attn_layer." % i) for i in FULL_BLOCKS)]
    rest = [n for n in names if n not in set(full_set)]
    return names, full_set, rest


def _grad_report(model, ref_g, tag):
    """Per-tensor errors of model's .grad against the oracle's gradients; returns (record dict, worst rel-l2, worst tensor name)."""
    names, full_set, rest = _compared_names(model)
    assert sorted(names) == sorted(ref_g), (set(names) ^ set(ref_g))
    prm = dict(model.named_parameters())
    rec, worst, worst_name = {}, 0.0, ""
    for n in full_set:
        assert prm[n].grad is not None, n
        l2, mx, cs = _tensor_err(prm[n].grad.float().cpu().numpy(), ref_g[n])
        rec[n] = [l2, mx, cs]
        if l2 > worst:
            worst, worst_name = l2, n
    for n in rest:
        g = prm[n].grad
        assert g is not None, n
        r = ref_g[n]
        if r.size <= 8192:
            l2, mx, cs = _tensor_err(g.float().cpu().numpy(), r)
        else:     # norm on the device, a strided sample on the host
            step = r.size // 4096
            smp, rs = g.reshape(-1)[::step].float().cpu().numpy(), r.reshape(-1)[::step]
            l2s, mx, cs = _tensor_err(smp, rs)
            nrm = float(g.float().norm())
            rn = float(np.sqrt((r.astype(np.float64) ** 2).sum()))
            l2 = max(l2s, abs(nrm - rn) / rn)
        rec[n] = [l2, mx, cs]
        if l2 > worst:
            worst, worst_name = l2, n
    # The 16 scalar gate gradients (attn_gate / ff_gate: ONE number = a signed sum over every element of a [tokens, 4096] branch output)
    # are judged against the scale of the gate gradients as a group: a gate whose true gradient happens to be near zero has no meaningful
    # relative error of its own in bf16 (round 5: blocks.23 attn_gate 19 % of ITS value = 0.4 % of the largest gate gradient).  The fp32
    # mode is unaffected (every gate within 1e-5 either way).
    gates = [n for n in names if ref_g[n].size == 1]
    scale = max(abs(float(ref_g[n].reshape(-1)[0])) for n in gates)
    for n in gates:
        d = abs(float(prm[n].grad.reshape(-1)[0]) - float(ref_g[n].reshape(-1)[0]))
        rec[n] = [d / scale, d / scale, 1.0]
    worst, worst_name = max((v[0], k) for k, v in rec.items())
    return rec, worst, worst_name


def _numpy_adamw_first_step(p0, g, coef, decay):
    """torch.optim.AdamW's first step on gradient coef * g (state zero before): returns (exp_avg, exp_avg_sq, new parameter)."""
    g = (g.astype(np.float64) * coef)
    m = (1.0 - B1) * g
    v = (1.0 - B2) * g * g
    p = p0.astype(np.float64) * (1.0 - LR * (WD if decay else 0.0))
    p = p - LR * (m / (1.0 - B1)) / (np.sqrt(v) / np.sqrt(1.0 - B2) + EPS)
    return m, v, p


def _bench_batch(full):
    """The benchmark's own batch shape: 8 pairs x 512 tokens (BASELINE configs[1])."""
    if "bench_batch" not in full["memo"]:
        full["memo"]["bench_batch"] = full["bench"].synth_batch(full["model"], 8, 512, DEV, seed=20260930)[:4]
    return full["memo"]["bench_batch"]


def _one_train_step(full, bf16, batch=None):
    """otter_amd.train.TrainStep (the benchmark's step) on the 2 x 128 batch (or `batch`).  Returns the TrainStep (gradients in .grad, AdamW
    state in .optimizer.state) and the parameter snapshot taken before it; the caller restores the parameters (the other legs of this module
    compare against the host copy of the ORIGINAL weights)."""
    from otter_amd.train import TrainStep

    model = full["model"]
    vision_x, ids, mask, labels = batch if batch is not None else _train_batch(full)
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    snap = {n: p.detach().clone() for n, p in trainable}
    model.train()
    step = TrainStep(model, lr=LR, weight_decay=WD, max_grad_norm=MAX_NORM, autocast_dtype=torch.bfloat16 if bf16 else None)
    assert step.hip_optimizer and step.reducer is None and step.embed_sink is not None        # the default single-rank HIP path
    loss = float(step(vision_x, ids, mask, labels))
    torch.cuda.synchronize()
    model.eval()
    return step, snap, loss


def _restore(full, snap):
    with torch.no_grad():
        for n, p in full["model"].named_parameters():
            if n in snap:
                p.copy_(snap[n])
    for p in full["model"].parameters():
        p.grad = None
    torch.cuda.empty_cache()


def _check_adamw(full, step, snap, ref, rec, tol_state, tol_delta):
    """clip coefficient, AdamW moments and the parameter update of the compared tensors against numpy AdamW on the ORACLE's gradients."""
    model = full["model"]
    norm_hip, coef_hip = (float(x) for x in step.optimizer.last_norm.cpu())
    coef_ref = min(1.0, MAX_NORM / (ref["norm"] + 1e-6))                     # torch.nn.utils.clip_grad_norm_
    rec["grad_norm"], rec["grad_norm_ref"], rec["clip_coef"], rec["clip_coef_ref"] = norm_hip, ref["norm"], coef_hip, coef_ref
    assert abs(norm_hip - ref["norm"]) <= tol_state * ref["norm"], (norm_hip, ref["norm"])
    assert abs(coef_hip - coef_ref) <= tol_state * coef_ref, (coef_hip, coef_ref)
    prm = dict(model.named_parameters())
    names = ["perceiver.latents", "perceiver.layers.0.to_kv.weight", "perceiver.layers.5.feed_forward.3.weight", "perceiver.norm.weight"]
    for i in FULL_BLOCKS:
        bp = LP + "blocks.%d.gated_cross_attn_layer." % i
        names += [bp + "attn_gate", bp + "ff_gate", bp + "attn.norm.weight", bp + "attn.to_q.weight", bp + "attn.to_out.weight", bp + "feed_forward.1.weight", bp + "feed_forward.3.weight"]
    names.append(LP + "wte.weight")
    worst_state = worst_delta = 0.0
    for n in names:
        st = step.optimizer.state[prm[n]]
        assert float(st["step"]) == 1.0
        m_ref, v_ref, p_ref = _numpy_adamw_first_step(full["p"][n], ref["g"][n], coef_ref, _decays(n))
        em = _tensor_err(st["exp_avg"].cpu().numpy(), m_ref)[0]
        ev = _tensor_err(st["exp_avg_sq"].cpu().numpy(), v_ref)[0]
        p0 = full["p"][n].astype(np.float64)
        ed = _tensor_err(prm[n].detach().cpu().numpy().astype(np.float64) - p0, p_ref - p0)[0]
        rec["adamw:" + n] = [em, ev, ed]
        worst_state, worst_delta = max(worst_state, em, ev), max(worst_delta, ed)
        assert em <= tol_state and ev <= 2 * tol_state, (n, em, ev)
        if tol_delta is not None:
            assert ed <= tol_delta, (n, ed)
    rec["worst_adamw_state"], rec["worst_adamw_delta"] = worst_state, worst_delta


def test_fp32_parity_mode_training_step_backward_and_adamw_vs_oracle(full):
    """fp32 parity mode, north-star tolerance: every compared gradient within 1e-3 (relative l2 AND max error relative to the largest
    entry) of O.otter_backward, the total gradient norm / clip coefficient / AdamW moments within 1e-3, the parameter update within 2e-2
    (its entries are ~ -lr * sign(g): an entry whose gradient is smaller than the gradient error flips -- ~1e-5 of the entries at this
    accuracy).  Two legs: (a) TrainStep on 2 x 128 tokens, (b) plain autograd (`loss.backward()`, the reference loop's own call through
    the shim: dense tied-embedding gradient, no sink) on the C1 prompt."""
    model = full["model"]
    assert next(p for p in model.parameters() if not p.requires_grad).dtype == torch.float32, "runs before the bf16 legs"
    # (a) TrainStep
    ref = _oracle_grads(full, "train_2x128", _train_batch(full))
    step, snap, loss = _one_train_step(full, bf16=False)
    try:
        rec, worst, worst_name = _grad_report(model, ref["g"], "fp32")
        out = {"loss": loss, "loss_ref": ref["loss"], "worst_grad_rel_l2": worst, "worst_grad": worst_name,
               "worst_grad_rel_max": max(v[1] for v in rec.values()), "min_cosine": min(v[2] for v in rec.values())}
        # rows of the tied embedding the batch looks up: lookup rows (SparseEmbedSink) + un-embedding gradient
        ids = np.unique(_train_batch(full)[1].cpu().numpy())
        wg = dict(model.named_parameters())[LP + "wte.weight"].grad
        out["wte_batch_rows_row_rel"] = G.row_rel_err(wg[torch.from_numpy(ids).to(DEV)].cpu().numpy(), ref["g"][LP + "wte.weight"][ids])
        _check_adamw(full, step, snap, ref, out, tol_state=1e-3, tol_delta=2e-2)
        G.record("full_model_train_step_fp32", **out, per_tensor={k: [float(x) for x in v] for k, v in rec.items() if any(k.startswith(LP + "blocks.%d." % i) for i in FULL_BLOCKS) or k.startswith("perceiver.la") or k.endswith("wte.weight")})
        assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (loss, ref["loss"])
        for n, (l2, mx, cs) in rec.items():
            assert l2 < 1e-3 and mx < 1e-3, (n, l2, mx)
        assert out["wte_batch_rows_row_rel"] < 1e-3
    finally:
        del step
        _restore(full, snap)
    # (b) plain autograd on the C1 prompt
    vision_x, ids_t, mask, labels = full["batch"]
    ref1 = _oracle_grads(full, "c1_prompt", full["batch"])
    model.train()
    out1 = model(vision_x=vision_x, lang_x=ids_t, attention_mask=mask, labels=labels)
    out1.loss.backward()
    torch.cuda.synchronize()
    model.eval()
    try:
        rec1, worst1, worst_name1 = _grad_report(model, ref1["g"], "fp32_c1")
        G.record("full_model_c1_plain_backward_fp32", loss=float(out1.loss), loss_ref=ref1["loss"], worst_grad_rel_l2=worst1, worst_grad=worst_name1,
                 worst_grad_rel_max=max(v[1] for v in rec1.values()), min_cosine=min(v[2] for v in rec1.values()))
        for n, (l2, mx, cs) in rec1.items():
            assert l2 < 1e-3 and mx < 1e-3, (n, l2, mx)
    finally:
        for p in model.parameters():
            p.grad = None
        del out1
        torch.cuda.empty_cache()


def test_bench_shape_training_step_fp32_parity_mode_is_stashed_as_the_same_gpu_reference(full):
    """VERDICT r5 item 5a, first half.  The 2 x 128 legs above run B*H = 64 (batch, head) pairs: the per-block dK/dV kernel, small GEMM grids.
    The benchmark runs 8 x 512: the PERSISTENT dK/dV form (B*H = 256), the 1024-tile persistent / cross-tile GEMM grids, 4096-row LayerNorm
    maps, the side stream.  Here `TrainStep` runs once at exactly that shape in the fp32 parity mode -- itself pinned against the host oracle
    at 2 x 128 (test above) and per kernel at every launch shape (tests/test_gpu_kernels.py) -- and its loss / gradients / norm are kept ON
    THE DEVICE as the reference for the bf16 production step at the same shape (test at the end of this module): no host-oracle time at
    8 x 512 (the oracle's backward would take minutes), and the whole composed step is exercised at the size the benchmark times."""
    model = full["model"]
    assert next(p for p in model.parameters() if not p.requires_grad).dtype == torch.float32, "runs before the bf16 legs"
    step, snap, loss = _one_train_step(full, bf16=False, batch=_bench_batch(full))
    try:
        g = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad}
        assert all(torch.isfinite(v).all() for v in g.values())
        norm, coef = (float(x) for x in step.optimizer.last_norm.cpu())
        full["memo"]["bench_fp32"] = dict(g=g, loss=loss, norm=norm, coef=coef)
        # sanity of the reference itself: random-init model, loss ~ ln(vocab); the clip engages (norm > max_grad_norm)
        assert abs(loss - np.log(50432.0)) < 1.0 and norm > 0 and 0 < coef <= 1.0
        G.record("full_model_bench_shape_fp32_reference", loss=loss, grad_norm=norm, clip_coef=coef, tensors=float(len(g)))
    finally:
        del step
        _restore(full, snap)


def test_c1_bf16_production_mode_drift_reported_and_bounded(full):
    """Runs after the fp32 leg (file order): the frozen weights are cast to bf16 in place -- exactly bench.build_model's layout -- and stay so."""
    ref = _oracle_c1(full)
    model = full["model"]
    for p in model.parameters():
        if not p.requires_grad:
            p.data = p.data.to(torch.bfloat16)            # exact: the values are bf16-representable
    torch.cuda.empty_cache()
    got = _run_hip(full, bf16=True)
    e_row = G.row_rel_err(got["logits"][0], ref["logits"][0])
    e_max = G.rel_err(got["logits"], ref["logits"])
    cos = G.cosine(got["logits"], ref["logits"])
    e_loss = abs(got["loss"] - ref["loss"]) / abs(ref["loss"])
    rec = dict(logits_row_rel=e_row, logits_rel_max=e_max, cosine=cos, loss=got["loss"], loss_ref=ref["loss"], loss_rel=e_loss)
    for mode in ("nocache", "cache"):
        a, first = _agreement(got["greedy_" + mode], ref["greedy_" + mode], 32)
        tr = ref["trace_" + mode]
        rec["agree_" + mode] = a
        rec["first_disagreement_" + mode] = float(first)
        rec["margins_rel_" + mode] = [float(t["margin"][0] / t["absmax"][0]) for t in tr]
        # kernel-level assertion: a step may only disagree where the fp32 top-2 margin is inside the bf16 error bar of the logits
        if first >= 0:
            bar = 4.0 * e_row * float(np.linalg.norm(ref["logits"][0, -1]) / np.sqrt(ref["logits"].shape[-1]))
            assert float(tr[first]["margin"][0]) < max(bar, 2e-2 * float(tr[first]["absmax"][0])), (mode, first, float(tr[first]["margin"][0]), bar)
    G.record("full_model_c1_bf16", **rec)
    assert e_row < BF16_ROW_TOL and cos > BF16_COS_MIN, (e_row, cos)
    assert e_loss < BF16_LOSS_TOL, (got["loss"], ref["loss"])
    # argmax of every prompt position (not only the generated ones): agreement wherever the fp32 margin is clear
    am_g, am_r = got["logits"][0].argmax(-1), ref["logits"][0].argmax(-1)
    srt = np.sort(ref["logits"][0], axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 0.05 * np.abs(ref["logits"][0]).max(-1)
    assert np.array_equal(am_g[clear], am_r[clear])


def test_c2_batch_bf16_loss_and_logits_vs_oracle(full):
    """The bench shape (512 tokens per pair), bf16 production mode, forward: loss and logits of B pairs vs the fp32 oracle on the host."""
    model, bench = full["model"], full["bench"]
    assert next(p for p in model.parameters() if not p.requires_grad).dtype == torch.bfloat16, "runs after the bf16 C1 leg"
    B = int(os.environ.get("OTTER_G1_C2_BATCH", "4"))        # r6: 4 (about a minute of host oracle on 256 threads; the GPU suite has a wall-clock limit); 8 = the benchmark's own batch, whose training step the bench-shape legs below cover
    vision_x, ids, mask, labels, _ = bench.synth_batch(model, B, 512, DEV, seed=977)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(vision_x=vision_x.to(torch.bfloat16), lang_x=ids, attention_mask=mask, labels=labels)
    t0 = time.time()
    ref = O.otter_forward(full["p"], full["spec"], vision_x.cpu().numpy(), ids.cpu().numpy(), None, labels.cpu().numpy(), keep_caches=False)
    t_ref = time.time() - t0
    got = out.logits.float().cpu().numpy()
    e_row = max(G.row_rel_err(got[b], ref["logits"][b]) for b in range(B))
    cos = G.cosine(got, ref["logits"])
    e_loss = abs(float(out.loss) - float(ref["loss"])) / abs(float(ref["loss"]))
    G.record("full_model_c2_bf16_forward", batch=float(B), logits_row_rel=e_row, cosine=cos, loss=float(out.loss), loss_ref=float(ref["loss"]), loss_rel=e_loss,
             oracle_forward_s=t_ref, host_threads=float(os.cpu_count()))
    assert e_row < BF16_ROW_TOL and cos > BF16_COS_MIN, (e_row, cos)
    assert e_loss < BF16_LOSS_TOL, (float(out.loss), float(ref["loss"]))


# bf16 production mode, training step.  Stated tolerances (= about 2x the figures measured on MI355X in round 5, profiles/r05_*parity_metrics.jsonl):
BF16_GRAD_L2_TOL, BF16_GRAD_COS_MIN, BF16_STATE_TOL = 6e-2, 0.998, 6e-2


def test_bf16_production_mode_training_step_backward_and_adamw_vs_oracle(full):
    """The step the benchmark times (frozen weights bf16, bf16 autocast, fp32 masters, TrainStep with SparseEmbedSink + FusedAdamW) on the
    2 x 128 batch against the fp32 oracle's backward: per-tensor relative l2 error and cosine REPORTED (gpurun_out/parity_metrics.jsonl ->
    profiles/) and bounded; gradient norm / clip coefficient / AdamW first moments bounded.  The parameter update itself is reported only:
    its entries are -lr * g / (|g| + eps) ~ -lr * sign(g), and with gradients accurate to ~1e-2 about 1 % of the entries sit inside the
    error bar of zero and flip, which is a property of Adam's first step, not of the kernels (the moments, which are linear / quadratic in
    g, carry the comparison)."""
    model = full["model"]
    assert next(p for p in model.parameters() if not p.requires_grad).dtype == torch.bfloat16, "runs after the bf16 C1 leg"
    ref = _oracle_grads(full, "train_2x128", _train_batch(full))
    step, snap, loss = _one_train_step(full, bf16=True)
    try:
        rec, worst, worst_name = _grad_report(model, ref["g"], "bf16")
        out = {"loss": loss, "loss_ref": ref["loss"], "worst_grad_rel_l2": worst, "worst_grad": worst_name, "min_cosine": min(v[2] for v in rec.values()),
               "median_grad_rel_l2": float(np.median([v[0] for v in rec.values()]))}
        ids = np.unique(_train_batch(full)[1].cpu().numpy())
        wg = dict(model.named_parameters())[LP + "wte.weight"].grad
        out["wte_batch_rows_row_rel"] = G.row_rel_err(wg[torch.from_numpy(ids).to(DEV)].cpu().numpy(), ref["g"][LP + "wte.weight"][ids])
        top = sorted(rec.items(), key=lambda kv: -kv[1][0])[:8]
        print("[g1] bf16 train step: worst gradients (rel l2, rel max, cosine):", [(k, ["%.2e" % x for x in v]) for k, v in top], flush=True)
        G.record("full_model_train_step_bf16_gradients", **out, per_tensor={k: [float(x) for x in v] for k, v in rec.items() if any(k.startswith(LP + "blocks.%d." % i) for i in FULL_BLOCKS) or k.startswith("perceiver.la") or k.endswith("wte.weight")})
        try:
            _check_adamw(full, step, snap, ref, out, tol_state=BF16_STATE_TOL, tol_delta=None)
        finally:
            G.record("full_model_train_step_bf16", **out)
        assert abs(loss - ref["loss"]) <= BF16_LOSS_TOL * abs(ref["loss"]), (loss, ref["loss"])
        for n, (l2, mx, cs) in rec.items():
            assert l2 < BF16_GRAD_L2_TOL and cs > BF16_GRAD_COS_MIN, (n, l2, cs)
    finally:
        del step
        _restore(full, snap)



def test_bench_shape_training_step_bf16_production_mode_vs_fp32_parity_mode(full):
    """VERDICT r5 item 5a, second half: the step `bench.py` times -- bf16 frozen weights, bf16 autocast, fp32 masters, SparseEmbedSink, FusedAdamW --
    at the benchmark's own 8 x 512 shape against the fp32 parity mode of the same step on the same GPU (stashed above).  Every one of the 158
    trainable tensors: relative l2 error and cosine, with the tolerances of the 2 x 128 oracle leg (6e-2 / 0.998; measured there 1.8e-2 worst);
    loss, total gradient norm and the clip coefficient bounded; AdamW's first moments = (1 - beta1) * coef * g checked against the fp32 mode's
    gradients on the compared tensors."""
    model = full["model"]
    assert next(p for p in model.parameters() if not p.requires_grad).dtype == torch.bfloat16, "runs after the bf16 C1 leg"
    ref = full["memo"].get("bench_fp32")
    assert ref is not None, "the fp32 half (test_bench_shape_training_step_fp32_parity_mode_...) must have run in this session"
    step, snap, loss = _one_train_step(full, bf16=True, batch=_bench_batch(full))
    try:
        prm = dict(model.named_parameters())
        rec = {}
        gates = [n for n in ref["g"] if ref["g"][n].numel() == 1]
        gscale = max(abs(float(ref["g"][n].reshape(-1)[0])) for n in gates)
        for n, r in ref["g"].items():
            g = prm[n].grad.detach().float()
            if r.numel() == 1:      # scalar gates: judged against the scale of the gate gradients as a group (see _grad_report)
                d = abs(float(g.reshape(-1)[0]) - float(r.reshape(-1)[0])) / gscale
                rec[n] = [d, 1.0]
                continue
            if n.endswith("wte.weight"):      # the rows the batch looks up + the dense un-embedding part: whole tensor
                pass
            l2 = float((g - r).norm() / (r.norm() + 1e-30))
            cs = float((g.reshape(-1).double() @ r.reshape(-1).double()) / (g.double().norm() * r.double().norm() + 1e-300))
            rec[n] = [l2, cs]
        worst = max((v[0], k) for k, v in rec.items())
        med = float(np.median([v[0] for v in rec.values()]))
        norm, coef = (float(x) for x in step.optimizer.last_norm.cpu())
        out = dict(loss=loss, loss_ref=ref["loss"], worst_grad_rel_l2=worst[0], worst_grad=worst[1], median_grad_rel_l2=med,
                   min_cosine=min(v[1] for v in rec.values()), grad_norm=norm, grad_norm_ref=ref["norm"], clip_coef=coef, clip_coef_ref=ref["coef"])
        print("[g1] bench-shape bf16 vs fp32 mode: loss %.5f / %.5f, worst gradient %.2e (%s), median %.2e, |g| %.4e / %.4e" %
              (loss, ref["loss"], worst[0], worst[1], med, norm, ref["norm"]), flush=True)
        # AdamW first moments of a few tensors against the fp32 mode's gradients: exp_avg = (1 - beta1) * coef * g after the first step
        worst_m = 0.0
        for n in ("perceiver.latents", "perceiver.layers.5.feed_forward.3.weight", LP + "blocks.3.gated_cross_attn_layer.feed_forward.1.weight",
                  LP + "blocks.31.gated_cross_attn_layer.attn.to_q.weight"):
            st = step.optimizer.state[prm[n]]
            want = (1.0 - B1) * ref["coef"] * ref["g"][n]
            worst_m = max(worst_m, float((st["exp_avg"].float() - want).norm() / (want.norm() + 1e-30)))
        out["worst_adamw_exp_avg"] = worst_m
        G.record("full_model_bench_shape_bf16_vs_fp32_mode", **out)
        assert abs(loss - ref["loss"]) <= BF16_LOSS_TOL * abs(ref["loss"]), (loss, ref["loss"])
        assert abs(norm - ref["norm"]) <= BF16_STATE_TOL * ref["norm"] and abs(coef - ref["coef"]) <= BF16_STATE_TOL * ref["coef"]
        for n, (l2, cs) in rec.items():
            assert l2 < BF16_GRAD_L2_TOL and cs > BF16_GRAD_COS_MIN, (n, l2, cs)
        assert worst_m <= BF16_STATE_TOL, worst_m
    finally:
        del step
        _restore(full, snap)
        full["memo"].pop("bench_fp32", None)
