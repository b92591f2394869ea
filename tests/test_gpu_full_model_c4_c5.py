"""GPU parity at FULL SIZE for BASELINE configs[3] (C4) and configs[4] (C5) -- VERDICT r4 item 7, in the style of tests/test_gpu_full_model.py (g1).

C4  OTTER-Video-LLaMA7B: the 32-layer model bench.py --config c4 times (LLaMA-7B host, 8 gated cross-attention blocks, CLIP ViT-L/14, 6-layer
    perceiver with frame embeddings), ONE sample of 8 x 224^2 frames (2048 patch features + 64 latents = 2112 perceiver keys) + a 64-token
    prompt, against tests/_host_ref.py on the host: transformers' LlamaForCausalLM in fp32 (the class the reference instantiates,
    modeling_otter.py:54,759-767; xformers_model/llama.py:286-327 is its in-repo restatement) with the numpy oracle's gated blocks hooked in
    front of the decoder layers, fed by the oracle's CLIP + perceiver (that composition is pinned against the reference's own tiny C4 model,
    tests/test_llama_host.py).  Forward AND the training step (TrainStep's composed backward vs transformers' autograd + the oracle's
    hand-derived backward of the fusion modules).  fp32 parity mode: logits rtol <= 1e-3 (north star), loss 1e-4, gradients 1e-3; bf16 production mode: reported, and bounded by 1.5 x the
    drift of the reference's own class under CPU bf16 autocast on the same model and batch (a random-init LLaMA amplifies bf16 rounding).
C5  OtterHD / Fuyu-8B at full depth (36 Persimmon layers, 9.4 B parameters), one 1080 x 1080 image as 36 x (36 patches + newline) = 1332
    positions + a text tail (bench.py --config c5's sequence), logits + loss against tests/_host_ref.fuyu_forward on the host: transformers'
    PersimmonForCausalLM in fp32 (fuyu/modeling_persimmon.py:286-310 restates it) behind the reference's patch-embedding scatter
    (fuyu/modeling_fuyu.py:44-77,126), a composition pinned against the reference's own FuyuForCausalLM fixture (tests/test_fuyu_host.py).
    bf16 production mode (the only mode of the C5 kernels that the benchmark runs): reported and bounded.

Weights are rounded to bf16-representable values once, so fp32 mode, bf16 mode and the host reference see identical numbers.
Each case skips cleanly when the box has not enough free host memory.  Measured figures go to gpurun_out/parity_metrics.jsonl -> profiles/."""
import os
import sys
import time

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import otter_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests import _host_ref as H  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_ROW_TOL, BF16_COS_MIN, BF16_LOSS_TOL = 2e-2, 0.9999, 2e-3      # as tests/test_gpu_full_model.py (about 2x the measured figures)


def _free_host_gb():
    import psutil

    return psutil.virtual_memory().available / 2**30


@pytest.fixture(scope="module")
def c4():
    import bench

    if _free_host_gb() < 80:
        pytest.skip("LLaMA-7B in fp32 on the host (27 GB) + the fusion modules' numpy copy + activations: not enough free host memory")
    t0 = time.time()
    model = bench.build_model(DEV, seed=0, config="c4", frozen_dtype=torch.float32)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).to(torch.float32))
    model.eval()
    layers = model.lang_encoder._get_decoder_layers()
    assert len(layers) == 32 and sum(1 for l in layers if l.gated_cross_attn_layer is not None) == 8
    assert model.lang_encoder.__class__.__name__ == "LlamaForCausalLM" and model.perceiver.frame_embs is not None
    # host side: the decoder into transformers' class, everything else (CLIP, perceiver, gated blocks) as numpy for the oracle
    hf = H.new_hf_llama(bench.LLAMA7B_TEXT)
    H.load_decoder_weights(hf, model.lang_encoder.state_dict())
    p = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()
         if v.is_floating_point() and (k.startswith("vision_encoder.") or k.startswith("perceiver.") or ".gated_cross_attn_layer." in k)}
    spec = O.OtterSpec(n_layers=32, d_model=4096, n_heads=32, max_seq_len=2048, cross_attn_every_n_layers=4, media_token_id=model.media_token_id,
                       clip_heads=16, clip_patch=14)
    print("[c4] model + host copies in %.0f s" % (time.time() - t0), flush=True)
    yield dict(model=model, hf=hf, p=p, spec=spec, bench=bench, memo={})
    del model, hf, p
    torch.cuda.empty_cache()


def test_c4_video_llama7b_full_size_logits_and_loss_vs_host_reference(c4):
    model, hf, p, spec, bench = c4["model"], c4["hf"], c4["p"], c4["spec"], c4["bench"]
    assert next(q for q in model.parameters() if not q.requires_grad).dtype == torch.float32, "the fp32 legs run first"
    vision_x, ids, mask, labels, _ = bench.synth_batch(model, 1, 64, DEV, seed=777, frames=8)
    assert vision_x.shape[:3] == (1, 1, 8) and int((ids == model.media_token_id).sum()) == 1
    t0 = time.time()
    ref = H.otter_llama_forward(hf, p, spec, vision_x.cpu().numpy(), ids.cpu().numpy(), labels.cpu().numpy())
    t_ref = time.time() - t0
    print("[c4] host reference forward %.1f s (%d host threads)" % (t_ref, os.cpu_count()), flush=True)
    c4["memo"]["fwd"] = (vision_x, ids, mask, labels, ref)
    # ---- fp32 parity mode
    with torch.no_grad():
        out = model(vision_x=vision_x, lang_x=ids, attention_mask=mask, labels=labels)
    got = out.logits.float().cpu().numpy()
    rec = dict(logits_rel_max=G.rel_err(got, ref["logits"]), logits_row_rel=G.row_rel_err(got[0], ref["logits"][0]), cosine=G.cosine(got, ref["logits"]),
               loss=float(out.loss), loss_ref=ref["loss"], loss_rel=abs(float(out.loss) - ref["loss"]) / abs(ref["loss"]), host_forward_s=t_ref)
    G.record("full_model_c4_fp32", **rec)
    assert rec["logits_rel_max"] < 1e-3 and rec["logits_row_rel"] < 1e-3, rec          # north_star: logits rtol <= 1e-3
    assert rec["loss_rel"] < 1e-4, rec
    assert np.array_equal(got[0].argmax(-1), ref["logits"][0].argmax(-1)) or rec["logits_row_rel"] < 1e-5   # greedy choice of every position


def _c4_train_step(c4, bf16):
    """otter_amd.train.TrainStep (bench.py --config c4's step) on a 2-sample x 64-token x 8-frame batch; returns (step, snapshot, loss)."""
    from otter_amd.train import TrainStep

    model, bench = c4["model"], c4["bench"]
    if "train_batch" not in c4["memo"]:
        c4["memo"]["train_batch"] = bench.synth_batch(model, 2, 64, DEV, seed=2024, frames=8)[:4]
    vision_x, ids, mask, labels = c4["memo"]["train_batch"]
    snap = {n: q.detach().clone() for n, q in model.named_parameters() if q.requires_grad}
    model.train()
    step = TrainStep(model, lr=1e-5, weight_decay=0.1, max_grad_norm=1.0, autocast_dtype=torch.bfloat16 if bf16 else None)
    assert step.hip_optimizer and step.reducer is None
    loss = float(step(vision_x, ids, mask, labels))
    torch.cuda.synchronize()
    model.eval()
    return step, snap, loss


def _c4_host_grads(c4, autocast_bf16=False):
    key = "grads_bf16" if autocast_bf16 else "grads_fp32"
    if key not in c4["memo"]:
        vision_x, ids, mask, labels = c4["memo"]["train_batch"]
        t0 = time.time()
        c4["memo"][key] = H.otter_llama_forward_backward(c4["hf"], c4["p"], c4["spec"], vision_x.cpu().numpy(), ids.cpu().numpy(), labels.cpu().numpy(),
                                                         autocast_bf16=autocast_bf16)
        print("[c4] host forward + backward (%s) %.1f s" % ("decoder under CPU bf16 autocast" if autocast_bf16 else "fp32", time.time() - t0), flush=True)
    return c4["memo"][key]


def _c4_grad_errors(model, ref_grads):
    """{name: relative l2 error} of the model's .grad against the host gradients, every trainable tensor; gate scalars against their group scale."""
    from tests.test_gpu_full_model import _tensor_err

    prm = {n: q for n, q in model.named_parameters() if q.requires_grad}
    assert sorted(prm) == sorted(ref_grads), set(prm) ^ set(ref_grads)
    gates = [n for n in prm if ref_grads[n].size == 1]
    scale = max(abs(float(ref_grads[n].reshape(-1)[0])) for n in gates)
    err = {}
    for n, q in prm.items():
        assert q.grad is not None, n
        if n in gates:
            err[n] = abs(float(q.grad.reshape(-1)[0]) - float(ref_grads[n].reshape(-1)[0])) / scale
        else:
            err[n] = _tensor_err(q.grad.float().cpu().numpy(), ref_grads[n])[0]
    return err


def _restore(model, snap):
    with torch.no_grad():
        for n, q in model.named_parameters():
            if n in snap:
                q.copy_(snap[n])
    for q in model.parameters():
        q.grad = None
    torch.cuda.empty_cache()


def test_c4_full_size_training_step_fp32_vs_host_backward(c4):
    """The C4 TRAINING step at full size (round 5): TrainStep (composed backward through the LLaMA host's RMSNorm / RoPE / SwiGLU / flash or fp32
    attention dgrad, the 8 gated blocks, the resampler on 2 112 keys with frame embeddings, the un-tied input embedding and lm_head) against
    tests/_host_ref.otter_llama_forward_backward -- transformers' own autograd through LlamaForCausalLM + the oracle's hand-derived backward of
    the fusion modules, pinned on the reference's tiny C4 gradients.  fp32 parity mode, every trainable tensor (96 of them) within 1e-3."""
    model = c4["model"]
    assert next(q for q in model.parameters() if not q.requires_grad).dtype == torch.float32, "the fp32 legs run first"
    step, snap, loss = _c4_train_step(c4, bf16=False)
    try:
        ref = _c4_host_grads(c4)
        err = _c4_grad_errors(model, ref["grads"])
        worst = max(err.items(), key=lambda kv: kv[1])
        norm_hip = float(step.optimizer.last_norm.cpu()[0])
        norm_ref = float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in ref["grads"].values())))
        G.record("full_model_c4_train_step_fp32", loss=loss, loss_ref=ref["loss"], worst_grad_rel_l2=worst[1], worst_grad=worst[0], tensors=float(len(err)),
                 grad_norm=norm_hip, grad_norm_ref=norm_ref)
        assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (loss, ref["loss"])
        assert worst[1] < 1e-3, worst
        assert abs(norm_hip - norm_ref) <= 1e-3 * norm_ref, (norm_hip, norm_ref)
    finally:
        del step
        _restore(model, snap)


def test_c4_video_llama7b_full_size_bf16_within_the_reference_class_drift(c4):
    """bf16 production mode (the kernels bench.py --config c4 runs), forward: reported, and bounded by 1.5 x the drift of the reference's own
    class under CPU bf16 autocast on the same model and batch."""
    model, hf, p, spec = c4["model"], c4["hf"], c4["p"], c4["spec"]
    vision_x, ids, mask, labels, ref = c4["memo"]["fwd"]
    for q in model.parameters():
        if not q.requires_grad:
            q.data = q.data.to(torch.bfloat16)
    torch.cuda.empty_cache()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out16 = model(vision_x=vision_x.to(torch.bfloat16), lang_x=ids, attention_mask=mask, labels=labels)
    got16 = out16.logits.float().cpu().numpy()
    rec16 = dict(logits_row_rel=G.row_rel_err(got16[0], ref["logits"][0]), cosine=G.cosine(got16, ref["logits"]), loss=float(out16.loss), loss_ref=ref["loss"],
                 loss_rel=abs(float(out16.loss) - ref["loss"]) / abs(ref["loss"]))
    srt = np.sort(ref["logits"][0], axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 0.05 * np.abs(ref["logits"][0]).max(-1)
    rec16["argmax_agree_clear_margin"] = float((got16[0].argmax(-1)[clear] == ref["logits"][0].argmax(-1)[clear]).mean()) if clear.any() else 1.0
    # Same-precision comparator.  A random-init LLaMA-7B amplifies bf16 rounding layer by layer (tools/c4_drift_by_layer.py,
    # profiles/r05_c4_drift_by_layer.txt: the residual stream's per-row error grows from 7e-3 after layer 0 to 5.7e-2 after layer 31 --
    # the MPT host of C2 stays flat at 8e-3), and so does the reference's own class: transformers' LlamaForCausalLM under
    # torch.autocast("cpu", bfloat16), the mode the reference trains in, drifts 1.1e-2 / 3.0e-2 after 1 / 7 layers on the build container.
    # So the bound is the reference class's own bf16 drift on THIS model and batch (decoder under CPU bf16 autocast, fusion modules fp32):
    t0 = time.time()
    ref16 = H.otter_llama_forward(hf, p, spec, vision_x.cpu().numpy(), ids.cpu().numpy(), labels.cpu().numpy(), autocast_bf16=True)
    rec16["ref_bf16_logits_row_rel"] = G.row_rel_err(ref16["logits"][0], ref["logits"][0])
    rec16["ref_bf16_cosine"] = G.cosine(ref16["logits"], ref["logits"])
    rec16["ref_bf16_loss_rel"] = abs(ref16["loss"] - ref["loss"]) / abs(ref["loss"])
    rec16["ref_bf16_forward_s"] = time.time() - t0
    G.record("full_model_c4_bf16", **rec16)
    assert rec16["logits_row_rel"] <= 1.5 * rec16["ref_bf16_logits_row_rel"], rec16
    assert 1.0 - rec16["cosine"] <= 1.5 * 1.5 * (1.0 - rec16["ref_bf16_cosine"]) + 1e-6, rec16         # (1 - cos ~ error^2 / 2)
    assert rec16["loss_rel"] <= max(1.5 * rec16["ref_bf16_loss_rel"], BF16_LOSS_TOL), rec16
    assert rec16["argmax_agree_clear_margin"] == 1.0, rec16


def test_c4_full_size_training_step_bf16_within_the_reference_class_drift(c4):
    """bf16 production mode, the training step: gradients of TrainStep vs the fp32 host backward, reported per tensor; bounded, as a
    root-mean-square over the 96 trainable tensors, by 1.5 x the same figure of the reference's own class (transformers' LlamaForCausalLM under
    CPU bf16 autocast + the fp32 oracle blocks) -- a random-init LLaMA-7B amplifies bf16 rounding, identically in the reference."""
    model = c4["model"]
    assert next(q for q in model.parameters() if not q.requires_grad).dtype == torch.bfloat16, "runs after the bf16 forward leg"
    from tests.test_gpu_full_model import _tensor_err

    step, snap, loss = _c4_train_step(c4, bf16=True)
    try:
        ref = _c4_host_grads(c4)
        ref16 = _c4_host_grads(c4, autocast_bf16=True)
        err = _c4_grad_errors(model, ref["grads"])
        gates = [n for n in err if ref["grads"][n].size == 1]
        scale = max(abs(float(ref["grads"][n].reshape(-1)[0])) for n in gates)
        err_ref = {n: (abs(float(ref16["grads"][n].reshape(-1)[0]) - float(ref["grads"][n].reshape(-1)[0])) / scale if n in gates
                       else _tensor_err(ref16["grads"][n], ref["grads"][n])[0]) for n in err}
        rms = float(np.sqrt(np.mean([v ** 2 for v in err.values()])))
        rms_ref = float(np.sqrt(np.mean([v ** 2 for v in err_ref.values()])))
        worst = max(err.items(), key=lambda kv: kv[1])
        # (per-tensor ratio over the matrices / vectors; the 16 scalar gate gradients -- signed sums over a whole branch whose error is a
        #  random number up to the ~5 % error of the incoming gradient -- count in the RMS only: round 5, layers.3 ff_gate 0.10 vs 0.005)
        worst_ratio = max((err[n] / max(err_ref[n], 5e-3), n) for n in err if n not in gates)
        G.record("full_model_c4_train_step_bf16", loss=loss, loss_ref=ref["loss"], loss_ref_bf16=ref16["loss"], grad_rms_hip=rms, grad_rms_ref_bf16=rms_ref,
                 worst_grad_rel_l2=worst[1], worst_grad=worst[0], worst_ratio=worst_ratio[0], worst_ratio_tensor=worst_ratio[1],
                 tensors_below_reference=float(sum(1 for n in err if err[n] <= err_ref[n])), tensors=float(len(err)))
        assert rms <= 1.5 * rms_ref, (rms, rms_ref)
        assert worst_ratio[0] <= 3.0, worst_ratio
        assert abs(loss - ref["loss"]) <= max(1.5 * abs(ref16["loss"] - ref["loss"]), BF16_LOSS_TOL * abs(ref["loss"])), (loss, ref["loss"], ref16["loss"])
    finally:
        del step
        _restore(model, snap)


def test_c5_fuyu8b_full_depth_logits_and_loss_vs_transformers_fp32():
    from transformers import FuyuConfig

    import bench
    from otter_amd.fuyu import FuyuForCausalLM

    if _free_host_gb() < 100:
        pytest.skip("Fuyu-8B in fp32 on the host (38 GB) + logits over a 262144-entry vocabulary: not enough free host memory")
    text = dict(bench.FUYU8B_TEXT)
    cfg = FuyuConfig(text_config=text, patch_size=30, num_channels=3, **{k: text[k] for k in ("vocab_size", "hidden_size", "intermediate_size",
                     "num_hidden_layers", "num_attention_heads", "max_position_embeddings")})
    t0 = time.time()
    torch.manual_seed(0)
    with torch.device(DEV):
        model = FuyuForCausalLM(cfg)
    g = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 0.02, generator=g)
            p.copy_(p.to(torch.bfloat16).to(p.dtype))        # bf16-representable: the host reference sees the very same numbers
    model.eval()
    n_par = sum(p.numel() for p in model.parameters())
    assert 9.3e9 < n_par < 9.5e9 and cfg.text_config.num_hidden_layers == 36
    hf = H.new_hf_persimmon({k: v for k, v in text.items()})
    state = {k: v for k, v in model.state_dict().items() if v.is_floating_point()}      # (device tensors: copied parameter by parameter)
    print("[c5] model + host copy in %.0f s" % (time.time() - t0), flush=True)
    # bench.py --config c5's sequence: 36 rows x (36 patches + newline) + 64 text tokens = 1396 positions, one sample
    grid, text_len = 36, 64
    S = grid * (grid + 1) + text_len
    gen = torch.Generator(device="cpu").manual_seed(4321)
    idx = torch.full((1, S), -1, dtype=torch.long)
    for r in range(grid):
        idx[:, r * (grid + 1): r * (grid + 1) + grid] = torch.arange(r * grid, (r + 1) * grid)
    patches = torch.randn(1, grid * grid, 2700, generator=gen).to(torch.bfloat16).float()
    ids = torch.randint(10, 262000, (1, S), generator=gen)
    labels = ids.clone()
    labels[:, : grid * (grid + 1) + 8] = -100
    t0 = time.time()
    o_ref = H.fuyu_forward(hf, state, ids.numpy(), patches.numpy(), idx.numpy(), labels.numpy())
    t_ref = time.time() - t0
    ref_logits = o_ref["logits"]
    print("[c5] host reference forward %.1f s (%d host threads)" % (t_ref, os.cpu_count()), flush=True)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(input_ids=ids.to(DEV), image_patches=patches.to(DEV).to(torch.bfloat16), image_patches_indices=idx.to(DEV), labels=labels.to(DEV))
    got = o.logits.float().cpu().numpy()
    # the text tail and the last image rows are what the loss reads; every position is compared
    rec = dict(logits_row_rel=G.row_rel_err(got[0], ref_logits[0]), cosine=G.cosine(got, ref_logits), loss=float(o.loss), loss_ref=o_ref["loss"],
               loss_rel=abs(float(o.loss) - o_ref["loss"]) / abs(o_ref["loss"]), host_forward_s=t_ref, positions=float(S))
    srt = np.sort(ref_logits[0], axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 0.05 * np.abs(ref_logits[0]).max(-1)
    rec["argmax_agree_clear_margin"] = float((got[0].argmax(-1)[clear] == ref_logits[0].argmax(-1)[clear]).mean()) if clear.any() else 1.0
    G.record("full_model_c5_bf16", **rec)
    assert rec["logits_row_rel"] < 3e-2 and rec["cosine"] > 0.9998, rec        # 36 layers at 1396 positions (C5-width single layer: 9.0e-3)
    assert rec["loss_rel"] < BF16_LOSS_TOL, rec
    assert rec["argmax_agree_clear_margin"] == 1.0, rec


def test_c5_fuyu_four_layers_full_width_training_step_gradients_vs_transformers_autograd():
    """VERDICT r5 item 5b.  `bench.py --config c5` times a FULL fine-tune step, and the only full-depth check above is a forward.  Here: four
    Persimmon layers at the full width of Fuyu-8B (hidden 4096, 64 heads of 64, MLP 16384, the 262144-row vocabulary), the benchmark's
    sequence (36 x (36 patches + newline) + 64 text positions = 1396), forward + backward of the product under bf16 autocast -- flash
    attention on the head-pair kernels, qk-LayerNorm, partial RoPE, squared-ReLU tails, K-major weight gradients, the patch projection and
    its scatter -- against transformers' own autograd on the host in fp32 (tests/_host_ref.fuyu_forward_backward), EVERY parameter's
    gradient.  bf16 production mode (the only mode the C5 benchmark runs): per-tensor relative l2 error reported and bounded."""
    from transformers import FuyuConfig

    import bench
    from otter_amd.fuyu import FuyuForCausalLM

    if _free_host_gb() < 60:
        pytest.skip("4 Persimmon layers + two 262144 x 4096 matrices in fp32 on the host with their gradients (~30 GB): not enough free host memory")
    text = dict(bench.FUYU8B_TEXT)
    text["num_hidden_layers"] = 4
    cfg = FuyuConfig(text_config=text, patch_size=30, num_channels=3, **{k: text[k] for k in ("vocab_size", "hidden_size", "intermediate_size",
                     "num_hidden_layers", "num_attention_heads", "max_position_embeddings")})
    torch.manual_seed(0)
    with torch.device(DEV):
        model = FuyuForCausalLM(cfg)
    g = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 0.02, generator=g)
            p.copy_(p.to(torch.bfloat16).to(p.dtype))        # bf16-representable: the host reference sees the very same numbers
    model.train()
    hf = H.new_hf_persimmon({k: v for k, v in text.items()})
    state = {k: v for k, v in model.state_dict().items() if v.is_floating_point()}
    grid, text_len = 36, 64
    S = grid * (grid + 1) + text_len
    gen = torch.Generator(device="cpu").manual_seed(977)
    idx = torch.full((1, S), -1, dtype=torch.long)
    for r in range(grid):
        idx[:, r * (grid + 1): r * (grid + 1) + grid] = torch.arange(r * grid, (r + 1) * grid)
    patches = torch.randn(1, grid * grid, 2700, generator=gen).to(torch.bfloat16).float()
    ids = torch.randint(10, 262000, (1, S), generator=gen)
    labels = ids.clone()
    labels[:, : grid * (grid + 1) + 8] = -100
    t0 = time.time()
    ref = H.fuyu_forward_backward(hf, state, ids.numpy(), patches.numpy(), idx.numpy(), labels.numpy())
    t_ref = time.time() - t0
    print("[c5] host forward + backward of 4 full-width layers: %.1f s (%d host threads), %d gradient tensors" % (t_ref, os.cpu_count(), len(ref["grads"])), flush=True)
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(input_ids=ids.to(DEV), image_patches=patches.to(DEV).to(torch.bfloat16), image_patches_indices=idx.to(DEV), labels=labels.to(DEV))
    o.loss.backward()
    torch.cuda.synchronize()
    names = [n for n, p in model.named_parameters()]
    assert sorted(names) == sorted(ref["grads"]), set(names) ^ set(ref["grads"])
    rec, prm = {}, dict(model.named_parameters())
    for n in names:
        got = prm[n].grad.float().cpu().numpy().astype(np.float64).reshape(-1)
        want = ref["grads"][n].astype(np.float64).reshape(-1)
        nb = float(np.sqrt(want @ want)) + 1e-300
        rec[n] = [float(np.sqrt(((got - want) ** 2).sum())) / nb, float(got @ want / (np.sqrt(got @ got) * nb + 1e-300))]
    worst = max((v[0], k) for k, v in rec.items())
    out = dict(loss=float(o.loss), loss_ref=ref["loss"], loss_rel=abs(float(o.loss) - ref["loss"]) / abs(ref["loss"]), worst_grad_rel_l2=worst[0], worst_grad=worst[1],
               median_grad_rel_l2=float(np.median([v[0] for v in rec.values()])), min_cosine=min(v[1] for v in rec.values()), tensors=float(len(rec)),
               host_fwd_bwd_s=t_ref, positions=float(S))
    top = sorted(rec.items(), key=lambda kv: -kv[1][0])[:6]
    print("[c5] 4-layer train step, bf16: worst gradients (rel l2, cosine):", [(k, ["%.2e" % x for x in v]) for k, v in top], flush=True)
    G.record("c5_four_layer_full_width_train_step_bf16", **out, per_tensor={k: [float(x) for x in v] for k, v in rec.items()})
    assert out["loss_rel"] < BF16_LOSS_TOL, out
    # tolerance: the one-layer / one-tensor figure at this width is 2.5e-2 (test_persimmon_c5_width_bf16_vs_transformers_fp32, dWqkv); the error
    # of a gradient grows with the depth it is propagated through, so 4 layers get 6e-2 and a cosine floor of 0.998 (as the MPT7B step)
    for n, (l2, cs) in rec.items():
        assert l2 < 6e-2 and cs > 0.998, (n, l2, cs)
