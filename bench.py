#!/usr/bin/env python
"""bench.py -- image-text pairs/s of one OTTER-Image-MPT7B instruction-following TRAINING step on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run with
one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env; backend "nccl" = RCCL over xGMI).  Rank 0 prints
ONE JSON line.  A "step" = one full optimizer step on a per-GPU micro-batch of 8 (image 224x224 + 512-token prompt) pairs
(BASELINE.json configs[1]; global batch 8*N -> configs[2] at N=8, weak scaling): CLIP ViT-L/14 forward (frozen) ->
perceiver resampler -> 32 MPT-7B blocks with 8 gated cross-attention blocks (hand-written HIP: LayerNorm, MFMA GEMMs with
fused GELU / tanh-gate / residual epilogues, masked cross-attention) -> tied un-embedding + CE loss -> backward (dgrad
through the frozen decoder, full backward of perceiver + gated blocks + embeddings) -> DP gradient average -> grad-norm
clip -> AdamW.  Nothing is skipped or cached inside the timed region.  Synthetic data, random-init weights (no network).

Extra objects on the JSON line (tier brief section 4):
  roofline     -- the dominant hand-written kernel = the bf16 MFMA GEMM at the gated-FFN shape M=B*T, N=16384, K=4096
                  (FF1 forward, and the two backward GEMMs of the same shape); achieved = 2*M*N*K / mean launch duration
                  measured with hipEvents on the launch stream during the timed steps (otter_prof_* in the C ABI).
  cpu_baseline -- the same step on this host's cores, fp32, timed in this run on a bounded per-component sample (2 pairs) and extrapolated
                  by component counts (see `sample`): `value` = oracle/torch_port.py, the reference modules' own ATen operator sequence
                  + autograd on torch-CPU (the engine the reference runs on a CPU; kind "port" -- /root/reference is not on this box),
                  `numpy_port` = the parity oracle on the host BLAS; rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this host driver needs dmabuf IPC; the HSA runtime reads the switch when it starts, i.e. before any torch.cuda call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md; vendor 5 PF figure is 2:1 sparse)



def machine_calibration(device):
    """Measured ceilings of THIS box at THIS moment (SURVEY 8d "confirm with a microbenchmark and report measured peaks"),
    outside the timed region, < 1 s: back-to-back v_mfma_f32_16x16x32_bf16 on every SIMD (otter_probe_mfma: the product GEMM's instruction,
    no memory traffic) on zero and on random bf16 operands -- the matrix pipe is power-limited on real data --, the shader clock during
    each (s_memtime / s_memrealtime inside the kernel), and a 1 GiB device-to-device copy."""
    from otter_amd import _capi as K_

    lib = K_.lib()
    n_wg = torch.cuda.get_device_properties(device).multi_processor_count
    g = torch.Generator(device=device).manual_seed(11)
    rnd = (torch.randn(1 << 19, device=device, generator=g) * 0.05).to(torch.bfloat16)      # 1 MiB of N(0, 0.05) operands
    zer = torch.zeros(1 << 19, dtype=torch.bfloat16, device=device)
    out = torch.zeros(2 * n_wg + 256 * n_wg, dtype=torch.int64, device=device)
    iters = 150_000                                                                          # x 64 MFMAs x 16 cycles = 154 M cycles ~ 70 ms
    res = {}
    for tag, src in (("zero_operands", zer), ("random_operands", rnd)):
        K_.check(lib.otter_probe_mfma(src.data_ptr(), out.data_ptr(), 20_000, n_wg, K_.stream()), "probe_mfma")     # reach the power state
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K_.check(lib.otter_probe_mfma(src.data_ptr(), out.data_ptr(), iters, n_wg, K_.stream()), "probe_mfma")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = out[: 2 * n_wg].view(n_wg, 2).cpu().double()
        tfl = n_wg * 4 * iters * 1048576.0 / (ms * 1e-3) / 1e12
        clk = float((t[:, 0] / t[:, 1].clamp(min=1)).median()) * 0.1                         # cycles per 100 MHz tick -> GHz
        implied = tfl * 1e12 / (n_wg * 4 * 1024.0) / 1e9                                     # 1024 FLOP per SIMD and cycle when issue is back to back
        res[tag] = {"tflops": round(tfl, 1), "clock_ghz": round(clk if 0.3 < clk < 4.0 else implied, 3), "issue_implied_ghz": round(implied, 3), "ms": round(ms, 1)}
    nbytes = 1 << 30
    a = torch.empty(nbytes, dtype=torch.uint8, device=device).fill_(1)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    res["hbm_copy_tbps"] = round(4 * 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3)   # bytes read + bytes written
    res["compute_units"] = n_wg
    del a, b
    # The memory path as the GEMM sees it: the FFN-shape product on ZERO operands runs at the un-throttled clock, so what it takes beyond
    # 256 K-tiles x 2048 cycles per CU is the part of the LDS-DMA's loaded latency that the two-stage prefetch does not cover plus four
    # prologue + tail pairs (one K-tile = 2048 / f + ~235 ns on the faster boxes).  Boxes whose MFMA probe and copy
    # rate are within 2 % of each other differ by 11 % on the real kernel; this figure is the one that moves with them.
    try:
        from otter_amd import ops as _ops

        M, N, Kd = 4096, 16384, 4096
        za = torch.zeros(M, Kd, dtype=torch.bfloat16, device=device)
        zb = torch.zeros(N, Kd, dtype=torch.bfloat16, device=device)
        zc = torch.empty(M, N, dtype=torch.bfloat16, device=device)
        for _ in range(3):
            _ops.gemm_nt(za, zb, out=zc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            _ops.gemm_nt(za, zb, out=zc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 8 * 1e3
        f = res["zero_operands"]["clock_ghz"]
        res["ffn_gemm_zero_operands_us"] = round(us, 1)
        res["exposed_latency_ns_per_ktile"] = round((us * 1e3 - 4 * 17000.0 / f) / 256.0 - 2048.0 / f, 1)
        del za, zb, zc
    except Exception as ex:      # (a calibration extra must never cost the bench line)
        res["ffn_gemm_zero_operands_us"] = None
        res["calibration_note"] = "zero-operand GEMM probe failed: %r" % (ex,)
    return res


def in_step_gemm_clock(one_step):
    """Shader clock of an FFN-shape GEMM launch INSIDE a training step: one extra (untimed) step with the kernel's tile-phase stamps switched on
    (otter_gemm_set_debug bit 64: s_memtime and the 100 MHz wall clock at the tile boundaries of workgroup 0); the stamps that remain are those
    of the step's last large-grid launch, whose shape and operand layout the kernel stamps beside them (`launch`: in the C2 step a K-major
    backward product of the first gated block, not necessarily the FFN shape).  The probes of machine_calibration run for ~0.1 s after the steps; on some boxes of the pool the
    sustained step runs its GEMMs hundreds of MHz below what those short probes reach -- this is the figure that shows it."""
    import ctypes

    from otter_amd import _capi as K_

    lib = K_.lib()
    prev = K_.gemm_set_debug(K_._gemm_debug_word | 64)     # borrow the stamp bit, keep whatever else the session has set
    try:
        one_step()
        torch.cuda.synchronize()
    finally:
        K_.gemm_set_debug(prev)
    buf = np.zeros(512, dtype=np.uint64)
    K_.check(lib.otter_gemm_read_timeline(buf.ctypes.data_as(ctypes.c_void_p), 512), "gemm_read_timeline")
    t = buf.reshape(2, 4, 8, 8).astype(np.int64)[0, 0]          # workgroup 0, wave 0: [tile][mark]
    words = buf.reshape(2, 4, 8, 8)[0, 0, :, 7]                  # mark 7: shape / operand layout of the launch that stamped the slot
    tiles = [i for i in range(8) if t[i, 0] > 0 and t[i, 4] > t[i, 0] and t[i, 6] > t[i, 5]]
    if not tiles:
        return None
    # a slot keeps the stamps of the LAST launch that reached it: slot 0 is rewritten by every launch, slots 1-3 only by launches with that many
    # tiles per workgroup.  The slots are grouped by launch shape; an FFN-shape launch (N = 16384, K = 4096: the roofline's kernel) is reported when one left stamps, else the largest product
    groups = {}
    for i in tiles:
        groups.setdefault(int(words[i]), []).append(i)

    def flops(w):
        return ((w >> 42) & 0xfffff) * ((w >> 21) & 0x1fffff) * (w & 0x1fffff)

    ffn = [g for g in groups if ((g >> 21) & 0x1fffff, g & 0x1fffff) == (16384, 4096)]      # the roofline's own shape when one of its launches left stamps
    w = max(ffn or groups, key=flops)
    tiles = groups[w]
    cyc = float(sum(t[i, 4] - t[i, 0] for i in tiles))
    ticks = float(sum(t[i, 6] - t[i, 5] for i in tiles))
    launch = {"M": (w >> 42) & 0xfffff, "N": (w >> 21) & 0x1fffff, "K": w & 0x1fffff, "a_kmajor": bool(w >> 63), "b_kmajor": bool((w >> 62) & 1)}   # (M: 20 bits below the two flags)
    return {"clock_ghz": round(cyc / ticks * 0.1, 3), "tiles": len(tiles), "cycles_per_tile": round(cyc / len(tiles)), "us_per_tile": round(ticks / len(tiles) * 0.01, 1),
            "launch": launch}


def apply_calibration(out, roof, cal):
    """Attach the measured ceilings to the JSON line: `frac` stays against the 2.5 PF constant (comparable across rounds and boxes);
    `frac_of_measured` divides by what back-to-back MFMAs on random operands reach on this box in this run."""
    if cal is None:
        return
    out["calibration"] = cal
    out["hbm_copy_tbps"] = cal["hbm_copy_tbps"]
    out["clock_ghz"] = cal["random_operands"]["clock_ghz"]
    if roof is not None:
        mp = cal["random_operands"]["tflops"]
        roof["measured_peak"] = mp
        roof["measured_peak_zero_operands"] = cal["zero_operands"]["tflops"]
        roof["frac_of_measured"] = round(roof["achieved"] / mp, 4)
        gb = roof.get("gated_block")
        if gb:
            gb["frac_of_measured"] = round(gb["achieved"] / mp, 4)


MPT7B_TEXT = dict(architectures=["MPTForCausalLM"], d_model=4096, n_heads=32, n_layers=32, expansion_ratio=4, max_seq_len=2048,
                  vocab_size=50432, no_bias=True, norm_type="low_precision_layernorm", use_cache=False,
                  attn_config=dict(alibi=True, alibi_bias_max=8, attn_impl="torch", attn_type="multihead_attention"))
LLAMA7B_TEXT = dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                    num_attention_heads=32, num_key_value_heads=32, vocab_size=32004, max_position_embeddings=2048, rms_norm_eps=1e-6,
                    tie_word_embeddings=False, hidden_act="silu", _name_or_path="llama-7b")   # 32000 + <|endofchunk|>, <image>, <answer>, <PAD>
FUYU8B_TEXT = dict(model_type="persimmon", vocab_size=262144, hidden_size=4096, intermediate_size=16384, num_hidden_layers=36,
                   num_attention_heads=64, max_position_embeddings=16384, qk_layernorm=True, partial_rotary_factor=0.5, hidden_act="relu2",
                   layer_norm_eps=1e-5, rope_theta=25000.0, tie_word_embeddings=False)
CLIP_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=768)


def build_model(device, seed=0, debug_layers=0, config="c2", frozen_dtype=torch.bfloat16):
    from otter_amd.configuration_otter import OtterConfig
    from otter_amd.modeling_otter import OtterForConditionalGeneration

    text, vis = dict(LLAMA7B_TEXT if config == "c4" else MPT7B_TEXT), dict(CLIP_L14)
    if debug_layers:
        text["num_hidden_layers" if config == "c4" else "n_layers"] = debug_layers
        vis["num_hidden_layers"] = 2
    extra = dict(max_num_frames=8) if config == "c4" else {}   # OTTER-Video: learned frame embeddings (modeling_otter.py:202-211)
    cfg = OtterConfig(vision_config=vis, text_config=text, cross_attn_every_n_layers=4, **extra)
    torch.manual_seed(seed)
    with torch.device(device):
        model = OtterForConditionalGeneration(cfg)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            # (the resampler's learned latents / frame / media-time embeddings keep the reference's own initialisation, torch.randn --
            # modeling_otter.py:202-205: at N(0, 0.02) the 64 latents come out of the random-init resampler nearly identical, every
            # cross-attention row is then a softmax over 64 copies of one key, and the query-side gradients become a difference of
            # near-equal bf16 numbers -- ill-conditioned for the reference under bf16 autocast just the same)
            if p.ndim >= 2 and name.rsplit(".", 1)[-1] not in ("latents", "frame_embs", "media_time_embs"):
                p.normal_(0.0, 0.02, generator=g)
            if name.endswith("attn_gate") or name.endswith("ff_gate"):
                p.fill_(0.5)  # zero-init gates make the block an identity (modeling_otter.py:362,371)
        # frozen weights live in bf16 (288 GB HBM would hold fp32 too, but bf16 halves the weight stream of the frozen
        # GEMMs); trainable parameters keep fp32 masters, exactly like accelerate's bf16 mixed precision.
        # (frozen_dtype=torch.float32: the fp32 parity mode of tests/test_gpu_full_model.py -- same architecture, same initial values)
        for name, p in model.named_parameters():
            if not p.requires_grad:
                p.data = p.data.to(frozen_dtype)
    model.train()
    return model


def run_c5(args, device, rank, world, use_dist):
    """Config C5 (BASELINE configs[4]): OtterHD = Fuyu-8B fully fine-tuned, one 1080x1080 image per sample as 36x36 linear patch tokens
    (30x30x3 values each) + 36 newline tokens + a text tail; no vision tower, no gated cross-attention.  One step = forward + backward
    (every parameter trains) + DP gradient average + clip + AdamW (otter_amd.optim.FusedAdamW)."""
    from transformers import FuyuConfig

    from otter_amd.dp import GradReducer
    from otter_amd.fuyu import FuyuForCausalLM
    from otter_amd.optim import FusedAdamW

    text = dict(FUYU8B_TEXT)
    if args.debug_layers:
        text["num_hidden_layers"] = args.debug_layers
    cfg = FuyuConfig(text_config=text, patch_size=30, num_channels=3, **{k: text[k] for k in ("vocab_size", "hidden_size", "intermediate_size",
                     "num_hidden_layers", "num_attention_heads", "max_position_embeddings")})
    torch.manual_seed(0)
    with torch.device(device):
        model = FuyuForCausalLM(cfg)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 0.02, generator=g)
    model.train()
    params = [p for p in model.parameters()]
    opt = FusedAdamW([{"params": [p for p in params if p.ndim >= 2], "weight_decay": 0.1}, {"params": [p for p in params if p.ndim < 2], "weight_decay": 0.0}],
                     lr=1e-5, max_grad_norm=1.0)
    reducer = GradReducer(params, 1 << 30, overlap=args.dp_overlap == "on", collective=args.dp_collective) if use_dist else None
    B, text_len = args.batch, 64
    grid, P = 36, 36 * 36
    S = grid * (grid + 1) + text_len                      # 1332 image positions (36 rows x (36 patches + newline)) + text
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    idx = torch.full((B, S), -1, dtype=torch.long)
    for r in range(grid):
        idx[:, r * (grid + 1): r * (grid + 1) + grid] = torch.arange(r * grid, (r + 1) * grid)
    idx = idx.to(device)
    # a fresh batch per step, generated up front, resident in HBM when the timed region starts (a repeated batch is memorised in a few steps)
    n_pool = min(args.steps + args.warmup, 16)
    pool = []
    for _ in range(n_pool):
        patches = torch.randn(B, P, 2700, generator=gen).to(device)
        ids = torch.randint(10, 262000, (B, S), generator=gen)
        labels = ids.clone()
        labels[:, : grid * (grid + 1) + 8] = -100             # the image and the instruction are not targets
        pool.append((patches, ids.to(device), labels.to(device)))
    it = [0]

    def step():
        patches, ids, labels = pool[it[0] % n_pool]
        it[0] += 1
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(input_ids=ids, image_patches=patches.to(torch.bfloat16), image_patches_indices=idx, labels=labels).loss
        loss.backward()
        if reducer is not None:
            reducer.wait()
        opt.step()
        return loss.detach()

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = step()
    # roofline object of this config: the dominant hand-written kernel of the C5 step is the K-major GEMM; the launches of
    # the MLP up-projection's weight gradient dW[16384, 4096] = dy^T x over the B*S token rows are timed with events on the launch stream
    from otter_amd import ops as _ops

    rM, rN, rK = text["intermediate_size"], text["hidden_size"], B * S
    _ops.prof_arm_gemm(rM, rN, rK, max_events=max(64, args.steps * text["num_hidden_layers"] + 8))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    n_launch, gemm_ms, _, _ = _ops.prof_collect_split()
    _ops.prof_disarm()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    if rank == 0:
        roof = None
        if n_launch > 0:
            avg_s = gemm_ms / n_launch / 1e3
            ach = 2.0 * rM * rN * rK / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": "gemm_bf16_t4_kernel, both operands K-major (weight gradient dy^T x) M=%d N=%d K=%d" % (rM, rN, rK),
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "launches": n_launch, "avg_us": round(avg_s * 1e6, 1)}
        n_par = sum(p.numel() for p in params)
        flops = 6.0 * n_par * B * S + 12.0 * text["num_hidden_layers"] * B * S * S * 4096 * 0.5   # dense 6ND + causal attention fwd+bwd
        # memory statistics FIRST: the calibration probes below allocate 2 GiB of copy buffers of their own
        mem = {"peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1), "reserved_mem_gb": round(torch.cuda.max_memory_reserved() / 2**30, 1),
               "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0))}
        cal = machine_calibration(device) if world == 1 else None
        cpu_b, cpu_note = None, None
        if world == 1 and not args.no_cpu_baseline:
            try:      # a reported extra must never cost the bench line
                cpu_b = cpu_baseline_c5(B, S, text)
            except Exception as ex:
                cpu_note = "cpu_baseline failed: %r" % (ex,)
        line = {
            "metric": "image-text pairs/s (train step) OtterHD Fuyu-8B, 1080x1080 image as 1296 patch tokens + %d text tokens" % text_len,
            "value": round(B * world * args.steps / elapsed, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("DEBUG-REDUCED (%d layers) " % args.debug_layers if args.debug_layers else "") + ("DEBUG-ONE-GPU-GLOO " if os.environ.get("OTTER_BENCH_DEBUG_ONE_GPU") == "1" else "") +
                       "OtterHD / Fuyu-8B full fine-tune step (BASELINE configs[4]): %d pairs per GPU, sequence %d = 36x(36 patches + newline) + %d text, "
                       "every parameter trainable (%.2f B), bf16 autocast, fp32 masters" % (B, S, text_len, n_par / 1e9),
                       "global_batch": B * world, "seq_len": S, "parallelism": "dp%d" % world},
            "loss": round(float(loss), 4),
            "roofline": roof,
            "cpu_baseline": cpu_b,
            "model_tflops_per_s": round(flops * args.steps / elapsed / 1e12, 1), **mem}
        if cpu_note:
            line["cpu_baseline_note"] = cpu_note
        apply_calibration(line, roof, cal)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def synth_batch(model, B, T, device, seed, frames=1):
    """SURVEY.md section 8d synthetic batch: BOS at 0, <image> at 1, one <answer> ... <|endofchunk|> span, labels by the
    reference's masking() rule."""
    from otter_amd.train import masking

    g = torch.Generator(device="cpu").manual_seed(seed)
    vision_x = torch.randn(B, 1, frames, 3, 224, 224, generator=g).to(device)   # frames of one sample sit on the F axis
    ids = torch.randint(1, min(50277, model.lang_encoder.config.vocab_size - 8), (B, T), generator=g)
    tok = model.text_tokenizer
    answer_id = tok.encode("<answer>")[-1]
    ids[:, 0] = 0
    ids[:, 1] = model.media_token_id
    ids[:, T // 4] = answer_id
    ids[:, T - 1] = model.eoc_token_id
    mask = torch.ones(B, T, dtype=torch.long)
    ids, mask = ids.to(device), mask.to(device)
    tok_ids = (answer_id, model.eoc_token_id, 0)        # <answer>, <|endofchunk|>, eos: what the reference's masking() keys on
    return vision_x, ids, mask, masking(ids, *tok_ids), tok_ids


def _blas_threads():
    """Threads the host BLAS (numpy's matmul) actually uses: the `cores` field of cpu_baseline (not os.cpu_count())."""
    try:
        from threadpoolctl import threadpool_info

        n = [int(i.get("num_threads", 0)) for i in threadpool_info() if i.get("user_api") == "blas"]
        if n:
            return max(n)
    except Exception:
        pass
    return os.cpu_count()


def _hf_decoder_layer_seconds(kind, T, train_all):
    """One decoder layer of the reference's third-party host at full width on the host cores (torch CPU, fp32), forward + backward:
    transformers' LlamaForCausalLM (what the reference instantiates for OTTER-Video-LLaMA7B, modeling_otter.py:54,759-767) or
    PersimmonForCausalLM (the class the reference's in-repo fuyu/modeling_persimmon.py restates) with ONE layer and a 512-entry
    vocabulary; frozen host -> input gradient only, OtterHD -> every parameter trains."""
    if kind == "llama":
        from transformers import LlamaConfig, LlamaForCausalLM

        cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                          vocab_size=512, max_position_embeddings=2048, rms_norm_eps=1e-6)
        m = LlamaForCausalLM(cfg)
    else:
        from transformers import PersimmonConfig, PersimmonForCausalLM

        cfg = PersimmonConfig(hidden_size=4096, intermediate_size=16384, num_hidden_layers=1, num_attention_heads=64, vocab_size=512,
                              max_position_embeddings=16384, qk_layernorm=True, partial_rotary_factor=0.5, hidden_act="relu2")
        m = PersimmonForCausalLM(cfg)
    m = m.float()
    for p_ in m.parameters():
        p_.requires_grad_(bool(train_all))
    x = torch.randn(1, T, 4096).requires_grad_(True)
    best = None
    prev_thr = torch.get_num_threads()
    torch.set_num_threads(_torch_threads())
    try:
        for _ in range(2):     # first pass pays allocation / thread start-up
            t0 = time.perf_counter()
            out = m.model(inputs_embeds=x).last_hidden_state
            out.sum().backward()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    finally:
        torch.set_num_threads(prev_thr)
    return best


def _port_vs_reference(out, port_value):
    """Cross-check only (round 6): the ratio reference-modules / numpy-port measured ONCE on a GPU node's host with the staged reference
    (oracle/calibrate_cpu_baseline.py -> profiles/r05_cpu_baseline_calibration_gpu_node.json).  Until round 5 the bench line's value was the
    numpy port times this committed constant; now the line times the torch-CPU port itself (the reference's arithmetic engine, in this
    run) and the constant only says what the older method would have predicted for it."""
    for name, where in (("r05_cpu_baseline_calibration_gpu_node.json", "a GPU node of this pool (round 5)"), ("r04_cpu_baseline_calibration_gpu_node.json", "a GPU node of this pool"),
                        ("r03_cpu_baseline_calibration.json", "the build container")):
        cal = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(cal):
            continue
        with open(cal) as f:
            c = json.load(f)
        r = c["step_mix"]["port_vs_reference"]
        out["calibration_cross_check"] = {"file": "profiles/" + name, "measured_in_this_run": False, "where": where, "reference_over_numpy_port": round(r, 3),
                                          "numpy_port_value_over_ratio": round(port_value / r, 5), "calibration_host_threads": c.get("host_threads"),
                                          "calibration_numpy_blas_threads": c.get("numpy_blas_threads")}
        break
    return out


def _torch_threads():
    """Threads for the torch-CPU leg: the physical cores of the host (the reference's DDP recipe leaves torch's intra-op default, which is the
    physical core count too)."""
    try:
        import psutil

        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(T=512, config="c2", pairs=2):
    """The hot path on the host cores, fp32, timed IN THIS RUN per component on `pairs` pairs and extrapolated to the whole step by component
    counts.  Two engines side by side:
      * `value` -- oracle/torch_port.py: the operator sequence of the reference's own modules on torch-CPU kernels with autograd for the
        backward = the engine the reference runs on a CPU (pinned on the reference's fixtures, tests/test_oracle_golden.py::test_torch_port_*);
        the frozen decoder layer of C4 / the CLIP layer go through transformers' own classes (what the reference instantiates);
      * `numpy_port` -- oracle/otter_oracle.py (the parity oracle) on the host BLAS.
    /root/reference itself cannot be on this box; the committed reference-vs-port ratio of earlier rounds stays as a cross-check field."""
    from oracle import otter_oracle as O
    from oracle import synth
    from oracle import torch_port as TP

    D, Dv = 4096, 1024
    V = 50432 if config == "c2" else 32004
    frames = 1 if config == "c2" else 8
    r = np.random.default_rng(0)
    nthr = _torch_threads()
    prev_thr = torch.get_num_threads()
    torch.set_num_threads(nthr)

    def rnd(*s, scale=0.02):
        return (r.standard_normal(s, dtype=np.float32) * scale)

    def timed(fn, reps=2):
        best = None
        for _ in range(reps):     # the first pass pays allocation / thread start-up
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    tn, tt = {}, {}       # seconds per component for `pairs` pairs: numpy port / torch port
    # ---- gated cross-attention block, fwd + bwd ----
    p = {k: rnd(*s) if len(s) == 2 else (np.ones(s, np.float32) if k.endswith("weight") else np.full(s, 0.5, np.float32))
         for k, s in synth.gated_xattn_shapes("b.", D, Dv).items()}
    x, media = rnd(pairs, T, D, scale=1.0), rnd(pairs, 1, 64, Dv, scale=1.0)
    ml = np.zeros((pairs, T), bool)
    ml[:, 1] = True

    def gated_np():
        y, c = O.gated_xattn_block_fwd(p, "b.", x, media, ml)
        O.gated_xattn_block_bwd(p, "b.", y, c)

    tn["gated_block"] = timed(gated_np, 1)
    pt = TP.to_torch(p)
    xt, mt = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(media).requires_grad_(True)

    def gated_t():
        for v in pt.values():
            v.grad = None
        y = TP.gated_xattn_block(pt, "b.", xt, mt, ml)
        y.backward(y.detach())

    tt["gated_block"] = timed(gated_t)
    del p, pt
    # ---- frozen decoder block, fwd + dgrad ----
    if config == "c2":
        p = {k: rnd(*s) if len(s) == 2 else np.ones(s, np.float32) for k, s in synth.mpt_block_shapes("m.", D).items()}
        bias = O.mpt_attn_bias(32, T, 2048)

        def lm_np():
            y, c, _ = O.mpt_block_fwd(p, "m.", x, 32, bias)
            O.mpt_block_bwd_input(p, "m.", y, c)

        tn["lm_block"] = timed(lm_np, 1)
        pt = TP.to_torch(p, requires_grad=False)
        bt = TP.alibi_bias(32, T, 2048)

        def lm_t():
            xt.grad = None
            y = TP.mpt_block(pt, "m.", xt, 32, bt)
            y.backward(y.detach())

        tt["lm_block"] = timed(lm_t)
        del p, pt
    else:
        tt["lm_block"] = tn["lm_block"] = pairs * _hf_decoder_layer_seconds("llama", T, train_all=False)   # transformers' own LlamaForCausalLM layer (torch CPU)
    # ---- perceiver resampler (6 layers), fwd + bwd ----
    extra = dict(max_num_frames=8) if frames > 1 else {}
    p = {k: rnd(*s) if len(s) == 2 and min(s) > 64 else np.ones(s, np.float32) * 0.5
         for k, s in synth.perceiver_shapes("p.", Dv, 6, **extra).items()}
    feats = rnd(pairs, 1, frames, 256, Dv, scale=1.0)

    def perc_np():
        y, c = O.perceiver_resampler_fwd(p, "p.", feats)
        O.perceiver_resampler_bwd(p, "p.", y, c)

    tn["perceiver"] = timed(perc_np, 1)
    pt = TP.to_torch(p)
    ft = torch.from_numpy(feats)

    def perc_t():
        for v in pt.values():
            v.grad = None
        y = TP.perceiver_resampler(pt, "p.", ft)
        y.backward(y.detach())

    tt["perceiver"] = timed(perc_t)
    del p, pt
    # ---- one CLIP ViT-L/14 layer on 257 tokens per frame (forward only, frozen) ----
    cp = {k: rnd(*s) if len(s) >= 2 else np.ones(s, np.float32) * 0.1 for k, s in synth.clip_shapes("v.", 1024, 1, 4096, 224, 14).items()}
    pix = rnd(pairs, 3, 224, 224, scale=1.0)
    tn["clip_layer"] = timed(lambda: O.clip_vision_fwd(cp, "v.", pix, 16, 14), 1)
    del cp
    from transformers import CLIPVisionConfig, CLIPVisionModel     # the class the reference instantiates (modeling_otter.py:768)

    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=1, num_attention_heads=16, image_size=224, patch_size=14)).float().eval()
    pixt = torch.from_numpy(pix)

    def clip_t():
        with torch.no_grad():
            clip(pixel_values=pixt)

    tt["clip_layer"] = timed(clip_t)
    del clip
    # ---- un-embedding + CE: logits fwd, dX (and dW where the matrix trains: the tied MPT embedding; LLaMA's lm_head is frozen) ----
    W = rnd(V, D)
    h = rnd(pairs, T, D, scale=1.0)
    lab = r.integers(0, V, size=(pairs, T))

    def un_np():
        logits = h @ W.T
        _, dl = O.cross_entropy_rolled(logits, lab)
        _ = dl @ W
        if config == "c2":
            _ = dl.reshape(-1, V).T @ h.reshape(-1, D)

    tn["unembed_loss"] = timed(un_np, 1)
    Wt = torch.from_numpy(W).requires_grad_(config == "c2")
    ht = torch.from_numpy(h).requires_grad_(True)

    def un_t():
        Wt.grad = None
        ht.grad = None
        _, loss = TP.unembed_loss(ht, Wt, lab)
        loss.backward()

    tt["unembed_loss"] = timed(un_t)
    del W, Wt
    torch.set_num_threads(prev_thr)

    def per_pair(t):
        return (8 * t["gated_block"] + 32 * t["lm_block"] + t["perceiver"] + 24 * frames * t["clip_layer"] + t["unembed_loss"]) / pairs

    lm = ("MPT block" if config == "c2" else "LLaMA-7B layer through transformers' LlamaForCausalLM (the class the reference instantiates; same figure in both engines)")
    comp = "; ".join("%s %.2f / %.2f s" % (k, tt[k], tn[k]) for k in ("gated_block", "lm_block", "perceiver", "clip_layer", "unembed_loss"))
    sample = ("fp32, %d pairs (%dx224^2 image + %d tokens each), timed in this run per component, torch-CPU port / numpy port: %s "
              "[gated-xattn block fwd+bwd, %s fwd+dgrad, 6-layer perceiver fwd+bwd, 1 CLIP layer fwd (torch leg: transformers' CLIPVisionModel), "
              "unembed+CE fwd+bwd]; step time = 8*gated + 32*lm + perceiver + %d*clip + unembed (optimizer/all-reduce not included)"
              % (pairs, frames, T, comp, lm, 24 * frames))
    vt, vn = 1.0 / per_pair(tt), 1.0 / per_pair(tn)
    out = {"value": round(vt, 5), "unit": "pairs/s", "cores": nthr, "host_logical_cpus": os.cpu_count(), "kind": "port",
           "engine": "torch-cpu: oracle/torch_port.py = the reference modules' ATen operator sequence + autograd (fp32), torch.set_num_threads(%d)" % nthr,
           "pairs_timed": pairs, "sample": sample,
           "numpy_port": {"value": round(vn, 5), "unit": "pairs/s", "cores": _blas_threads(), "engine": "oracle/otter_oracle.py on the host BLAS"}}
    return _port_vs_reference(out, vn)


def cpu_baseline_c5(B, S, text):
    """C5 (OtterHD / Fuyu-8B, every parameter trains): one Persimmon layer at full width through transformers' PersimmonForCausalLM on the
    host cores (fwd + full bwd, one sample of S positions), the 2700 -> 4096 patch projection and the 262144-row un-embedding + CE in numpy;
    step = layers * layer + projection + unembed (the optimizer sweep over 9.4 B parameters is not included)."""
    from oracle import otter_oracle as O

    L, V, D = text["num_hidden_layers"], text["vocab_size"], text["hidden_size"]
    r = np.random.default_rng(0)
    t_layer = _hf_decoder_layer_seconds("persimmon", S, train_all=True)
    Wp = r.standard_normal((D, 2700), dtype=np.float32) * 0.02
    patches = r.standard_normal((1296, 2700), dtype=np.float32)
    t0 = time.perf_counter()
    e = patches @ Wp.T
    _ = e.T @ patches
    t_proj = time.perf_counter() - t0
    W = r.standard_normal((V, D), dtype=np.float32) * 0.02
    h = r.standard_normal((S, D), dtype=np.float32)
    t0 = time.perf_counter()
    logits = h @ W.T
    _, dl = O.cross_entropy_rolled(logits[None], r.integers(0, V, size=(1, S)))
    _ = dl[0] @ W
    _ = dl[0].T @ h
    t_un = time.perf_counter() - t0
    per_pair = L * t_layer + t_proj + t_un
    return {"value": round(1.0 / per_pair, 5), "unit": "pairs/s", "cores": _torch_threads(), "host_logical_cpus": os.cpu_count(), "kind": "port",
            "engine": "torch-cpu (transformers' PersimmonForCausalLM layer, %d threads: %.0f %% of the step) + numpy on the host BLAS (%d threads) for the two projections"
                      % (_torch_threads(), 100.0 * L * t_layer / per_pair, _blas_threads()),
            "sample": ("fp32, 1 pair (%d positions): 1 Persimmon layer fwd+bwd through transformers' PersimmonForCausalLM on torch CPU (%.2fs; the class the "
                       "reference's fuyu/modeling_persimmon.py restates), patch projection fwd+wgrad in numpy (%.2fs), un-embedding + CE fwd+bwd in numpy "
                       "(%.2fs); step time = %d*layer + projection + unembed (optimizer/all-reduce not included)" % (S, t_layer, t_proj, t_un, L))}


def gated_block_roofline(model, batch, B, T, device, iters=20):
    """North-star figure: one OtterGatedCrossAttentionBlock (the model's first one, its real weights) forward + backward at the bench
    shapes, timed stand-alone with events (outside the timed steps), against SURVEY.md 8d's 141.94 GF per 512-token sample and forward
    (x 3 for forward + dgrad + wgrad) and the 2.5 PF dense bf16 peak.  Includes everything the block launches: LayerNorms, cross
    attention, the six FFN-shape GEMMs with their fused tails, the small projections, operand transposes, fp32 weight gradients."""
    blk = None
    for layer in model.lang_encoder._get_decoder_layers():
        if getattr(layer, "gated_cross_attn_layer", None) is not None:
            blk = layer.gated_cross_attn_layer
            break
    if blk is None:
        return None
    _, ids, _, _ = batch
    D = blk.feed_forward[1].weight.shape[1]
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(B, T, D, device=device, generator=g).requires_grad_(True)
    media = torch.randn(B, 1, 64, blk.attn.to_kv.weight.shape[1], device=device, generator=g)
    dy = torch.randn(B, T, D, device=device, generator=g)
    ml = ids == model.media_token_id
    saved = [(p, p.grad) for p in blk.parameters()]

    def once():
        for p, _ in saved:
            p.grad = None        # as after zero_grad(set_to_none=True): the weight gradients are written, not accumulated
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x, media, media_locations=ml, attend_previous=True)
        y.backward(dy)

    for _ in range(4):      # the chip comes out of the timed steps (hipBLASLt bursts, a 7 ms HBM-bound optimizer sweep) in another power state
        once()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        once()
        e.record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) for s, e in evs)
    ms = per[len(per) // 2]                       # median of the per-iteration times; mean and min are in the line as well
    ms_mean, ms_min = sum(per) / len(per), per[0]
    for p, gsave in saved:
        p.grad = gsave
    x.grad = None
    Tm = 64
    fwd = 2.0 * T * (D * 512 + 512 * D + 2 * D * 4 * D) + 2.0 * Tm * media.shape[-1] * 1024 + 4.0 * T * Tm * 512   # per sample
    ach = 3.0 * fwd * B / (ms * 1e-3) / 1e12
    return {"what": "one gated cross-attention block, forward + backward, B=%d x %d tokens, stand-alone; median of %d iterations" % (B, T, iters),
            "ms": round(ms, 3), "ms_mean": round(ms_mean, 3), "ms_min": round(ms_min, 3),
            "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2",
                    help="c2 = OTTER-Image-MPT7B (BASELINE metric, the default); c4 = OTTER-Video-LLaMA7B, 8 frames per sample (configs[3]); "
                         "c5 = OtterHD / Fuyu-8B full fine-tune, 1080x1080 patch tokens (configs[4]; batch 8 as in the reference's OtterHD recipe)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-variant", type=int, default=0)
    ap.add_argument("--flash-variant", type=int, default=0, help="A/B hook: otter_flash_set_variant (2 = plain grid order instead of longest-first)")
    ap.add_argument("--debug-layers", type=int, default=0, help="DEBUG ONLY: shrink MPT to this many layers (not a valid bench)")
    ap.add_argument("--dp-overlap", choices=["on", "off"], default="on",
                    help="N > 1: launch each gradient bucket's RCCL all-reduce from its last gradient's hook, overlapped with the frozen decoder's "
                         "backward (on, the default), or reduce every bucket after backward (off) -- the A/B a multi-GPU session needs (echoed in config)")
    ap.add_argument("--rccl-max-channels", type=int, default=0,
                    help="N > 1: cap RCCL's channel count (NCCL_MAX_NCHANNELS, set before the process group is created): fewer channels = fewer CUs held "
                         "by a resident collective while the backward GEMMs run (0 = RCCL's default; echoed in config)")
    ap.add_argument("--dp-collective", choices=["all_reduce", "rs_ag"], default="all_reduce",
                    help="N > 1: average each gradient bucket with one all-reduce (default) or as reduce-scatter + all-gather (GradReducer.collective; "
                         "SURVEY section 5's 'direct RS + AG' variant) -- echoed in config")
    ap.add_argument("--rccl-algo", default="",
                    help="N > 1: NCCL_ALGO for RCCL (e.g. Ring, Tree; empty = RCCL's choice), set before the process group is created -- echoed in config")
    args = ap.parse_args()
    if args.rccl_max_channels > 0:
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_max_channels)     # read by RCCL at communicator creation (inherited by spawned ranks)
    if args.rccl_algo:
        os.environ["NCCL_ALGO"] = args.rccl_algo

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the ranks ourselves (one process per GPU), exactly the launch the driver uses
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # DEBUG ONLY (not a valid bench): every rank on cuda:0, collectives over gloo -- runs the whole N > 1 code path (spawn, reducer hooks,
    # embedding-row all-gather, max-over-ranks timing) on a one-GPU box, where RCCL refuses two ranks on one device
    one_gpu = os.environ.get("OTTER_BENCH_DEBUG_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("OTTER_FORCE_DIST") == "1"  # the env switch exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if os.environ.get("OTTER_BENCH_FORCE_NONPERSISTENT") == "1":   # A/B switch: one workgroup per tile instead of the persistent grids
        from otter_amd import ops as _ops

        _ops.set_gemm_persistent(False)
    if args.config == "c5":
        # 8 pairs per GPU = the reference's OtterHD recipe (docs/OtterHD.md:66, shared_scripts/Demo_OtterHD.sh: --batch_size=8); the fixed
        # cost of the step (clip + AdamW over 9.41 B parameters, 54 ms) is then spread over twice the pairs of round 2's first runs (B=4)
        return run_c5(args, device, rank, world, use_dist)

    from otter_amd import ops
    from otter_amd.train import TrainStep

    if args.gemm_variant:
        ops.set_gemm_variant(args.gemm_variant)
    if args.flash_variant:
        ops.set_flash_variant(args.flash_variant)
    model = build_model(device, seed=0, debug_layers=args.debug_layers, config=args.config)  # identical replica on every rank (same seed)
    step = TrainStep(model, lr=1e-5, weight_decay=0.1, max_grad_norm=1.0, autocast_dtype=torch.bfloat16,
                     force_reducer=os.environ.get("OTTER_FORCE_DIST") == "1", dp_overlap=args.dp_overlap == "on", dp_collective=args.dp_collective)
    B, T = args.batch, args.seq
    from otter_amd.train import masking

    # A FRESH synthetic batch for every step (one repeated batch let the model memorise it and the `loss` field stopped
    # being a sanity signal): the batches are generated up front and are resident in HBM when the timed region starts (tier brief section 4);
    # more than 32 steps cycle through the pool.
    n_pool = min(args.steps + args.warmup, 32)
    pool = [synth_batch(model, B, T, device, seed=1000 + rank + 7919 * i, frames=8 if args.config == "c4" else 1) for i in range(n_pool)]
    vision_x, ids, amask, labels0, tok_ids = pool[0]
    batch = (vision_x, ids, amask, labels0)
    it = [0]

    def one_step():
        vx, tok, am, _, _ = pool[it[0] % n_pool]
        it[0] += 1
        # the reference builds the labels inside its step (instruction_following.py:163-192): same here, on the device, no host sync
        return step(vx, tok, am, masking(tok, *tok_ids))

    # DIAGNOSTIC (not a valid bench): OTTER_BENCH_OCCUPY_CUS=n parks n workgroups that each pin a whole CU (all of its LDS) on a side stream
    # for the timed region -- what the GEMMs see while a collective's kernel is resident
    n_occ = int(os.environ.get("OTTER_BENCH_OCCUPY_CUS", "0"))
    if os.environ.get("OTTER_BENCH_FORCE_NONPERSISTENT") == "1":   # A/B leg without an occupier: the per-tile grids on a free chip
        ops.set_gemm_persistent(False)
    occ_flag = torch.zeros(1, dtype=torch.int32, device=device) if n_occ else None

    def sync():
        if n_occ:
            torch.cuda.current_stream().synchronize()      # (a device-wide wait would wait for the occupier itself)
            return
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = one_step()
    if n_occ:
        from otter_amd import _capi as _K
        if os.environ.get("OTTER_BENCH_NONPERSISTENT") == "1":     # the grids TrainStep selects while a DP reducer is attached
            ops.set_gemm_persistent(False)
        side = torch.cuda.Stream()
        torch.cuda.current_stream().synchronize()
        _K.check(_K.lib().otter_debug_occupy_cus(n_occ, occ_flag.data_ptr(), 60 * 100_000_000, side.cuda_stream), "occupy_cus")
    M, N, Kd = B * T, 16384, 4096
    ops.prof_arm_gemm(M, N, Kd, max_events=max(64, args.steps * 8 * 3 + 8))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    if n_occ:
        occ_flag.fill_(1)
        torch.cuda.synchronize()
    n_launch, gemm_ms, n_km, km_ms = ops.prof_collect_split()
    ops.prof_disarm()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        pairs = B * world * args.steps
        roof = None
        if n_launch > 0:
            avg_s = gemm_ms / n_launch / 1e3
            ach = 2.0 * M * N * Kd / avg_s / 1e12
            # HBM-side traffic per launch comes from the separate rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
            # WRITE_SIZE) committed under profiles/; PMC collection cannot share a run with the timed region.
            traffic = None
            v = args.gemm_variant or 26
            names = {13: "gemm_bf16_ph_kernel (256x256x64 tile, 8 waves, phased)", 18: "gemm_bf16_r4_kernel (256x256x64 tile, 4 waves, register-resident K-tile)",
                     26: "gemm_bf16_t4_kernel (256x256x64 tile, 4 waves, register-resident K-tile, 16x16x32 MFMA; cross-tile ring for K-contiguous operands, K-tiles rotated by N panel)"}
            traffic_src = None
            for cand in ({18: ["r02_pmc_gemm_ffn_v18.json"], 26: ["r06d_pmc_gemm_ffn_traffic.json", "r06_pmc_gemm_ffn_traffic.json", "r05_pmc_gemm_ffn_traffic.json", "r03_pmc_gemm_ffn_traffic.json"]}.get(v, ["r01_pmc_gemm_ffn_v13.json"])):
                pmc = os.path.join(ROOT, "profiles", cand)          # newest committed PMC pass of this kernel first
                if os.path.exists(pmc) and (M, N, Kd) == (4096, 16384, 4096):
                    with open(pmc) as f:
                        traffic = json.load(f).get("traffic_bytes_per_launch")
                    traffic_src = "profiles/" + cand
                    break
            roof = {"bound": "mfma", "kernel": "%s M=%d N=%d K=%d" % (names.get(v, "gemm variant %d" % v), M, N, Kd), "achieved": round(ach, 1),
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "launches": n_launch, "avg_us": round(avg_s * 1e6, 1), "traffic_source": traffic_src,
                    "algorithmic_bytes": 2 * (M * Kd + N * Kd + M * N)}
            # the same kernel runs in two operand layouts: K-contiguous rows (forward products) and K-major (round 3: the backward
            # products read their operands in place through transpose reads); both are in `achieved`, split here
            if n_km and n_launch > n_km:
                roof["by_layout"] = {"k_contiguous": {"launches": n_launch - n_km, "avg_us": round((gemm_ms - km_ms) / (n_launch - n_km) * 1e3, 1)},
                                     "k_major": {"launches": n_km, "avg_us": round(km_ms / n_km * 1e3, 1)}}
            if world == 1 and os.environ.get("OTTER_FORCE_DIST") != "1" and not n_occ:   # (no DP reducer hooks on the parameters: stand-alone backward is safe)
                roof["gated_block"] = gated_block_roofline(model, batch, B, T, device)
        out = {
            "metric": ("image-text pairs/s (train step) OTTER-MPT7B, 1 img+512 tok" if args.config == "c2"
                       else "video-text pairs/s (train step) OTTER-Video-LLaMA7B, 8 frames+512 tok"),
            "value": round(pairs / elapsed, 3),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("DEBUG-REDUCED (%d layers) " % args.debug_layers if args.debug_layers else "") + ("DEBUG-ONE-GPU-GLOO " if os.environ.get("OTTER_BENCH_DEBUG_ONE_GPU") == "1" else "") + ("DIAGNOSTIC-%d-CUS-OCCUPIED " % n_occ if n_occ else "") +
                                   (("OTTER-Image-MPT7B instruction-following train step, 1x224^2 image + %d tokens per pair, "
                                     "batch %d per GPU (BASELINE configs[1]), LM+CLIP frozen, bf16 autocast, fp32 masters" % (T, B)) if args.config == "c2" else
                                    ("OTTER-Video-LLaMA7B-DenseCaption train step, 8x224^2 frames (T_img=1, F=8: 2048 patches) + %d tokens per pair, "
                                     "batch %d per GPU (BASELINE configs[3]), LM+CLIP frozen, bf16 autocast, fp32 masters" % (T, B))),
                       "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world,
                       "dp_overlap": args.dp_overlap if use_dist else None,
                       "dp_collective": args.dp_collective if use_dist else None,
                       "rccl_algo": (args.rccl_algo or os.environ.get("NCCL_ALGO")) if use_dist else None,
                       "rccl_max_channels": (args.rccl_max_channels or os.environ.get("NCCL_MAX_NCHANNELS")) if use_dist else None,
                       "gemm_grid": "per-tile" if step.grid_mode == 2 else "persistent",
                       # which GEMMs of the frozen decoder run on csrc/gemm.hip (otter_amd/mpt.py _own_mode): "1t" = all of them (default since round 6d)
                       "decoder_gemm": os.environ.get("OTTER_OWN_DECODER_GEMM", "1t") if args.config == "c2" else None,
                       # clip_grad_norm_'s reduction: "fused:<n>" = n weight gradients took sum(dW^2) from their own GEMM launch (single rank),
                       # the rest from the sweep; "sweep" = every gradient (OTTER_NO_FUSED_GRAD_NORM=1, or a DP reducer is attached)
                       "grad_norm": ("fused:%d" % step.optimizer.fused_norm_tensors) if getattr(step, "norm_sink", None) is not None else "sweep"},
            "loss": round(float(loss), 4),
            "roofline": roof,
        }
        if world == 1 and not n_occ:
            cal = machine_calibration(device)
            try:
                cal["in_step_gemm"] = in_step_gemm_clock(one_step)      # (one more optimizer step, after everything that is reported)
            except Exception as ex:
                cal["in_step_gemm"] = None
                cal["calibration_note"] = (cal.get("calibration_note", "") + " in-step clock failed: %r" % (ex,)).strip()
            apply_calibration(out, roof, cal)
        if world == 1 and not args.no_cpu_baseline:
            try:      # a reported extra must never cost the bench line (host OOM, a transformers API change ...)
                out["cpu_baseline"] = cpu_baseline(T, args.config)
            except Exception as ex:
                out["cpu_baseline"] = None
                out["cpu_baseline_note"] = "cpu_baseline failed: %r" % (ex,)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
