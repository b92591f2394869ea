"""GPU parity tests of the decoder-host flash attention (otter_flash_attn_fwd / _bwd, head_dim 128 and 64) through the C ABI.

Inputs are rounded to bf16 first and the oracle (oracle/otter_oracle.py: mpt_attention_core, float64) runs on the rounded
values.  Tolerances, relative to the tensor's max: outputs 1e-2 (one bf16 rounding of P and of the output), gradients
2e-2 (bf16 roundings of P, dS and of the result)."""
import math

import numpy as np
import pytest
import torch

from oracle import otter_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from otter_amd import _capi, ops as _ops

    assert _capi.lib().otter_device_check() > 0, _capi.lib().otter_last_error()
    return _ops


def _case(seed, B, H, S, lens):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16)
    valid = None
    if lens is not None:
        valid = torch.zeros(B, S, dtype=torch.uint8)
        for b, n in enumerate(lens):
            valid[b, :n] = 1
    return qkv, dout, valid


CASES = [
    # B, H, S, causal, alibi, right-padded lengths
    (2, 2, 512, True, True, None),
    (2, 3, 200, True, True, [200, 150]),
    (1, 2, 130, False, False, None),
    (2, 2, 64, True, False, [64, 1]),
    (1, 4, 321, True, True, [300]),
    (2, 2, 17, True, True, None),          # a single 32-row query tile (the dK/dV pipeline's prologue-only path)
    (1, 3, 33, True, False, [33]),         # two query tiles, the second with one row
    (1, 2, 96, True, True, None),          # three tiles: the 3-slot ring wraps exactly once
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])   # 8: the persistent dK/dV kernel wherever a launch has >= 2 key blocks
@pytest.mark.parametrize("B,H,S,causal,alibi,lens", CASES)
def test_flash_attention_fwd_bwd(ops, B, H, S, causal, alibi, lens, variant):
    from otter_amd.mpt import alibi_slopes

    ops.set_flash_variant(variant)

    qkv, dout, valid = _case(B * 1000 + S, B, H, S, lens)
    slopes = alibi_slopes(H, 8).float() if alibi else None
    scale = 1.0 / math.sqrt(128)
    dq5 = qkv.to(DEV)
    q, k, v = dq5[:, :, 0], dq5[:, :, 1], dq5[:, :, 2]  # strided slices of the fused buffer
    sl = slopes.to(DEV) if slopes is not None else None
    kvd = valid.to(DEV) if valid is not None else None
    o, lse = ops.flash_attn_fwd(q, k, v, sl, kvd, scale, causal)
    dqkv = torch.full_like(dq5, float("nan"))
    ops.flash_attn_bwd(q, k, v, o, lse, dout.to(DEV), dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], sl, kvd, scale, causal)
    torch.cuda.synchronize()
    ops.set_flash_variant(0)

    f = lambda t: t.double().numpy().transpose(0, 2, 1, 3)  # [B,S,H,d] -> [B,H,S,d]
    qh, kh, vh = f(qkv[:, :, 0].float()), f(qkv[:, :, 1].float()), f(qkv[:, :, 2].float())
    ctx, (rdq, rdk, rdv) = O.mpt_attention_core(qh, kh, vh, scale, slopes.numpy() if alibi else None,
                                                 valid.numpy() if valid is not None else None, causal, f(dout.float()))
    got = lambda t: t.float().cpu().double().numpy().transpose(0, 2, 1, 3)
    # rows of padded QUERIES (beyond the valid length) are unconstrained only in the sense that nobody consumes them, but
    # the kernel computes them like the reference does (their keys 0..i are partly valid), so compare everything
    assert relmax(got(o), ctx) < 1e-2
    assert bool(torch.isfinite(dqkv.float()).all())
    assert relmax(got(dqkv[:, :, 0]), rdq) < 2e-2
    assert relmax(got(dqkv[:, :, 1]), rdk) < 2e-2
    assert relmax(got(dqkv[:, :, 2]), rdv) < 2e-2
    # LSE against the oracle's row normaliser
    w = (qh @ np.swapaxes(kh, -1, -2)) * scale
    if alibi:
        w = w + slopes.numpy().astype(np.float64)[None, :, None, None] * np.arange(1 - S, 1, dtype=np.float64)[None, None, None, :]
    mask = np.ones((B, 1, S, S), bool)
    if valid is not None:
        mask = mask & valid.numpy().astype(bool)[:, None, None, :]
    if causal:
        mask = mask & np.tril(np.ones((S, S), bool))[None, None]
    w = np.where(mask, w, -np.inf)
    ref_lse = np.log(np.exp(w - w.max(-1, keepdims=True)).sum(-1)) + w.max(-1)
    assert np.abs(lse.cpu().double().numpy() - ref_lse).max() < 2e-3


CASES64 = [
    # B, H, S, causal, alibi, right-padded lengths
    (2, 4, 512, True, False, None),
    (1, 2, 200, True, True, [150]),
    (2, 2, 130, False, False, None),
    (2, 6, 64, True, False, [64, 1]),
    (1, 4, 321, True, True, [300]),
    (2, 2, 17, True, False, None),
    (1, 2, 33, True, True, [33]),
    (1, 8, 96, True, False, None),
    (1, 2, 1396, True, False, None),       # C5's sequence length (1296 patch tokens + 36 newlines + 64 text)
    (1, 2, 1, True, False, None),          # a single token
    (2, 4, 5, False, True, [5, 2]),        # less than one fragment row, non-causal, padded
]


def _views64(layout, B, S, H, g, fill=None):
    """q / k / v style [B,S,H,64] bf16 views in one of the layouts the head-pair kernels address in place."""
    mk = (lambda *sh: (torch.randn(*sh, generator=g) * 0.8).to(torch.bfloat16)) if fill is None else (lambda *sh: torch.full(sh, fill, dtype=torch.bfloat16))
    if layout == "compact":            # three contiguous [B,S,H,64] tensors (head stride 64)
        base = [mk(B, S, H, 64).to(DEV) for _ in range(3)]
        return base, base
    if layout == "interleaved":        # Persimmon's projection buffer [B,S,H,3,64] (head stride 192)
        buf = mk(B, S, H, 3, 64).to(DEV)
        return [buf], [buf[:, :, :, i] for i in range(3)]
    buf = mk(B, S, 3, H, 64).to(DEV)   # "fused": [B,S,3,H,64] (head stride 64, token stride 3*H*64)
    return [buf], [buf[:, :, i] for i in range(3)]


@pytest.mark.parametrize("variant", [0, 2, 8])
@pytest.mark.parametrize("layout", ["compact", "interleaved", "fused"])
@pytest.mark.parametrize("B,H,S,causal,alibi,lens", CASES64)
def test_flash_attention_head_dim_64(ops, B, H, S, causal, alibi, lens, layout, variant):
    """head_dim 64 (fuyu/modeling_persimmon.py:310; two heads per workgroup) against the float64 oracle, in every layout the kernels address
    in place, on both block orders."""
    from otter_amd.mpt import alibi_slopes

    if layout != "interleaved" and variant == 2 and S not in (512, 200, 1396):
        pytest.skip("A/B variants: one layout per small case is enough")
    g = torch.Generator().manual_seed(B * 1000 + S + H)
    _, (q, k, v) = _views64(layout, B, S, H, g)
    dout = torch.randn(B, S, H, 64, generator=g).to(torch.bfloat16).to(DEV)
    valid = None
    if lens is not None:
        valid = torch.zeros(B, S, dtype=torch.uint8)
        for b, n in enumerate(lens):
            valid[b, :n] = 1
    slopes = alibi_slopes(H, 8).float() if alibi else None
    scale = 1.0 / math.sqrt(64)
    sl = slopes.to(DEV) if slopes is not None else None
    kvd = valid.to(DEV) if valid is not None else None
    ops.set_flash_variant(variant)
    try:
        o, lse = ops.flash_attn_fwd(q, k, v, sl, kvd, scale, causal)
        _, (dq, dk, dv) = _views64(layout, B, S, H, g, fill=float("nan"))
        ops.flash_attn_bwd(q, k, v, o, lse, dout, dq, dk, dv, sl, kvd, scale, causal)
        torch.cuda.synchronize()
    finally:
        ops.set_flash_variant(0)
    assert o.shape == (B, S, H, 64) and o.is_contiguous()
    f = lambda t: t.float().cpu().double().numpy().transpose(0, 2, 1, 3)
    qh, kh, vh = f(q), f(k), f(v)
    ctx, (rdq, rdk, rdv) = O.mpt_attention_core(qh, kh, vh, scale, slopes.numpy() if alibi else None,
                                                 valid.numpy() if valid is not None else None, causal, f(dout))
    assert relmax(f(o), ctx) < 1e-2
    for t in (dq, dk, dv):
        assert bool(torch.isfinite(t.float()).all())
    assert relmax(f(dq), rdq) < 2e-2
    assert relmax(f(dk), rdk) < 2e-2
    assert relmax(f(dv), rdv) < 2e-2
    w = (qh @ np.swapaxes(kh, -1, -2)) * scale
    if alibi:
        w = w + slopes.numpy().astype(np.float64)[None, :, None, None] * np.arange(1 - S, 1, dtype=np.float64)[None, None, None, :]
    mask = np.ones((B, 1, S, S), bool)
    if valid is not None:
        mask = mask & valid.numpy().astype(bool)[:, None, None, :]
    if causal:
        mask = mask & np.tril(np.ones((S, S), bool))[None, None]
    w = np.where(mask, w, -np.inf)
    ref_lse = np.log(np.exp(w - w.max(-1, keepdims=True)).sum(-1)) + w.max(-1)
    assert np.abs(lse.cpu().double().numpy() - ref_lse).max() < 2e-3


def test_flash_head_pairs_equal_zero_padded_heads(ops):
    """The head-pair kernels against round 2's route (heads zero-padded to 128 columns on the 128-wide kernels) at C5's attention shape
    (B=4 here 2, 64 heads, 1396 tokens): zero columns add exact zeros to every product, so outputs and gradients agree to rounding of the
    differently grouped sums -- and the pair kernels must leave the neighbouring q / k slots of the interleaved buffers untouched."""
    B, H, S = 2, 64, 1396
    g = torch.Generator().manual_seed(11)
    buf = (torch.randn(B, S, H, 3, 64, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, S, H, 64, generator=g).to(torch.bfloat16).to(DEV)
    scale = 1.0 / math.sqrt(64)
    q, k, v = (buf[:, :, :, i] for i in range(3))
    o, lse = ops.flash_attn_fwd(q, k, v, None, None, scale, True)
    dbuf = torch.full_like(buf, 7.0)
    dq, dk = torch.empty_like(o), torch.empty_like(o)
    ops.flash_attn_bwd(q, k, v, o, lse, dout, dq, dk, dbuf[:, :, :, 2], None, None, scale, True)
    assert bool((dbuf[:, :, :, :2] == 7.0).all())           # only the v slots were written
    pad = lambda t: torch.cat([t, torch.zeros_like(t)], -1).contiguous()
    qp, kp, vp, dop = pad(q), pad(k), pad(v), pad(dout)
    op, lsep = ops.flash_attn_fwd(qp, kp, vp, None, None, scale, True)
    dqp, dkp, dvp = torch.empty_like(qp), torch.empty_like(qp), torch.empty_like(qp)
    ops.flash_attn_bwd(qp, kp, vp, op, lsep, dop, dqp, dkp, dvp, None, None, scale, True)
    torch.cuda.synchronize()
    assert float((lse - lsep).abs().max()) < 1e-5
    for a, b in ((o, op), (dq, dqp), (dk, dkp), (dbuf[:, :, :, 2], dvp)):
        a, b = a.float(), b[..., :64].float()
        assert float((a - b).abs().max()) <= 8e-3 * float(b.abs().max())


def test_flash_head_dim_64_rejects_what_it_cannot_address(ops):
    from otter_amd._capi import OtterHipError

    q = torch.zeros(1, 64, 3, 64, dtype=torch.bfloat16, device=DEV)          # odd head count
    with pytest.raises(OtterHipError, match="even"):
        ops.flash_attn_fwd(q, q, q, None, None, 0.125, True)
    t = torch.zeros(1, 64, 2, 64, dtype=torch.bfloat16, device=DEV).transpose(1, 2).contiguous().transpose(1, 2)   # [B,S,H,64] with head stride S*64
    with pytest.raises(OtterHipError, match="head_stride"):
        ops.flash_attn_fwd(t, t, t, None, None, 0.125, True)


def test_flash_batch_rows_independent(ops):
    """Property at the full C2 shape (B=8, 32 heads, 512 tokens): every (batch, head) pair is independent -- computing a
    single batch row alone gives bit-identical output -- and a causal row never depends on later tokens."""
    from otter_amd.mpt import alibi_slopes

    B, H, S = 8, 32, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B, S, 3, H, 128, generator=g, device=DEV).to(torch.bfloat16)
    sl = alibi_slopes(H, 8).float().to(DEV)
    scale = 1.0 / math.sqrt(128)
    o, _ = ops.flash_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], sl, None, scale, True)
    one = qkv[3:4].contiguous()
    o1, _ = ops.flash_attn_fwd(one[:, :, 0], one[:, :, 1], one[:, :, 2], sl, None, scale, True)
    assert torch.equal(o[3:4], o1)
    pert = qkv.clone()
    pert[:, 300:] = torch.randn_like(pert[:, 300:])
    o2, _ = ops.flash_attn_fwd(pert[:, :, 0], pert[:, :, 1], pert[:, :, 2], sl, None, scale, True)
    # ALiBi is anchored at the LAST key (j - (Sk-1)), but a row's softmax is invariant to the per-row constant shift.
    # Rows whose 32-row wave lies entirely before the perturbation are bit-identical; rows 288..299 share a wave with
    # perturbed rows, and the wave-uniform lazy-rescale decision may then fall on a different tile: same value, possibly a
    # different rounding of P and of the result (bf16): equal to rounding error, not bit for bit.
    assert torch.equal(o[:, :288], o2[:, :288])
    a_, b_ = o[:, 288:300].float(), o2[:, 288:300].float()
    assert float((a_ - b_).abs().max()) <= 1e-2 * float(b_.abs().max())


def test_mpt_host_flash_matches_sdpa_path(ops, monkeypatch):
    """The decoder host with the HIP flash kernel vs the same host on the additive-mask SDPA path (and both vs the
    oracle block stack): logits and input-embedding gradients, right-padded batch."""
    from otter_amd.mpt import MPTConfig, MPTForCausalLM

    torch.manual_seed(0)
    cfg = MPTConfig(d_model=256, n_heads=2, n_layers=2, expansion_ratio=2, max_seq_len=128, vocab_size=96, no_bias=True,
                    tie_word_embeddings=True)
    model = MPTForCausalLM(cfg).to(DEV)
    ids = torch.randint(0, 96, (2, 80), device=DEV)
    am = torch.ones(2, 80, dtype=torch.long, device=DEV)
    am[1, 60:] = 0

    def run():
        model.zero_grad(set_to_none=True)
        for p_ in model.parameters():
            p_.requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(input_ids=ids, attention_mask=am)
        logits = out.logits.float()
        (logits[:, :, :7] * am[:, :, None]).sum().backward()
        return logits.detach().cpu().numpy(), model.transformer.wte.weight.grad.detach().float().cpu().numpy()

    monkeypatch.delenv("OTTER_NO_FLASH", raising=False)
    lf, gf = run()
    monkeypatch.setenv("OTTER_NO_FLASH", "1")
    ls, gs = run()
    valid = am.bool().cpu().numpy()
    assert relmax(lf[valid], ls[valid]) < 2e-2
    assert relmax(gf, gs) < 3e-2


@pytest.mark.parametrize("B,H,S,lens", [(8, 32, 512, None), (8, 32, 384, [384, 300, 129, 128, 127, 384, 1, 200]), (1, 256, 256, None)])
def test_flash_persistent_dkv_equals_per_block_kernel(ops, B, H, S, lens):
    """The benchmark's launch (B x H = 256 = one dK/dV workgroup per CU) takes the persistent per-head kernel (flash.hip: PERS): the same
    tiles in the same order per key block as the one-workgroup-per-key-block kernel (variant 7), K / V fragments through an LDS staging
    area instead of global loads, results out through an LDS transpose.  dQ (same kernel) is bit-identical; dK / dV agree to the bf16
    rounding of single P / dS elements (the two instantiations compile the fp32 softmax to different instruction sequences: an fp32 last
    bit now and then flips the bf16 rounding of a P element; tools/flash_pers_diff.py: 2 271 of 16.7 M dK elements, both forms equally
    far from an fp64 reference) -- and dK / dV right against the oracle on two heads."""
    from otter_amd.mpt import alibi_slopes

    g = torch.Generator().manual_seed(B + S)
    qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    sl = alibi_slopes(H, 8).float().to(DEV)
    kvd = None
    if lens is not None:
        kvd = torch.zeros(B, S, dtype=torch.uint8)
        for b, n in enumerate(lens):
            kvd[b, :n] = 1
        kvd = kvd.to(DEV)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    scale = 1.0 / math.sqrt(128)
    outs = []
    for variant in (0, 7, 0):
        ops.set_flash_variant(variant)
        o, lse = ops.flash_attn_fwd(q, k, v, sl, kvd, scale, True)
        d = torch.full_like(qkv, float("nan"))
        ops.flash_attn_bwd(q, k, v, o, lse, dout, d[:, :, 0], d[:, :, 1], d[:, :, 2], sl, kvd, scale, True)
        outs.append(d)
    ops.set_flash_variant(0)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(outs[0].float()).any())
    assert torch.equal(outs[0], outs[2])                       # run to run
    assert torch.equal(outs[0][:, :, 0], outs[1][:, :, 0])     # dQ
    for i in (1, 2):
        a_, b_ = outs[0][:, :, i].float(), outs[1][:, :, i].float()
        diff = (a_ - b_).abs()
        assert float(diff.max()) <= float(b_.abs().max()) * 2.0 ** -7
        assert float((diff > 0).float().mean()) < 1e-3
    hs = [0, H - 1]
    f = lambda t: t.float().cpu().double().numpy().transpose(0, 2, 1, 3)
    bsel = slice(B - 1, B)
    qh, kh, vh = (f(qkv[bsel, :, i][:, :, hs]) for i in range(3))
    _, (rdq, rdk, rdv) = O.mpt_attention_core(qh, kh, vh, scale, sl.cpu().numpy()[hs], kvd.cpu().numpy()[bsel] if kvd is not None else None, True,
                                              f(dout[bsel][:, :, hs]))
    assert relmax(f(outs[0][bsel, :, 1][:, :, hs]), rdk) < 2e-2
    assert relmax(f(outs[0][bsel, :, 2][:, :, hs]), rdv) < 2e-2


@pytest.mark.parametrize("B,H,S,form", [(8, 32, 512, "persistent dK/dV (B x H = 256 workgroups, the benchmark's launch)"),
                                        (3, 20, 448, "per-block dK/dV (B x H = 60: neither a multiple of the CU count nor 8 rounds)"),
                                        (1, 24, 640, "per-block dK/dV, 5 key blocks")])
def test_flash_backward_repeats_bit_for_bit_under_load(ops, B, H, S, form):
    """300 launches of forward + backward, part of them with unrelated traffic on a second stream: every result equals the first one bit for
    bit -- a race between the DMA ring, the K / V staging area and the LDS transpose would show as a run-to-run difference
    (tools/flash_stress.py runs thousands).  Both forms of the dK/dV kernel: the persistent per-head one (the benchmark's launch) and the
    per-block one, whose row-store epilogue stages dK / dV in the Q / dO ring -- ADVICE r4 found that the ring's last LDS-DMA pieces could
    still be in flight there (fixed in round 5: every wave drains its own pieces before the barrier)."""
    from otter_amd.mpt import alibi_slopes

    g = torch.Generator().manual_seed(77)
    qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    sl = alibi_slopes(H, 8).float().to(DEV)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    scale = 1.0 / math.sqrt(128)
    o, lse = ops.flash_attn_fwd(q, k, v, sl, None, scale, True)
    ref = torch.full_like(qkv, float("nan"))
    ops.flash_attn_bwd(q, k, v, o, lse, dout, ref[:, :, 0], ref[:, :, 1], ref[:, :, 2], sl, None, scale, True)
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, device=DEV)
    d = torch.empty_like(qkv)
    bad = 0
    for it in range(300):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        d.fill_(float("nan"))
        o2, lse2 = ops.flash_attn_fwd(q, k, v, sl, None, scale, True)
        ops.flash_attn_bwd(q, k, v, o2, lse2, dout, d[:, :, 0], d[:, :, 1], d[:, :, 2], sl, None, scale, True)
        bad += int(not (torch.equal(d, ref) and torch.equal(o2, o)))
    torch.cuda.synchronize()
    assert bad == 0


def test_flash_block_order_is_only_an_order(ops):
    """Longest-first block order, also in GROUPS of heads (B*H = 160 heads x 1024 tokens = 80 MB of K + V > the 64 MB group budget ->
    groups of 80 heads), against the plain 3-D grid: the same blocks do the same arithmetic, so every output is bit-identical."""
    from otter_amd.mpt import alibi_slopes

    B, H, S = 2, 80, 1024
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B, S, 3, H, 128, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    sl = alibi_slopes(H, 8).float().to(DEV)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    outs = []
    for variant in (0, 2):
        ops.set_flash_variant(variant)
        o, lse = ops.flash_attn_fwd(q, k, v, sl, None, 1.0 / math.sqrt(128), True)
        d = torch.full_like(qkv, float("nan"))
        ops.flash_attn_bwd(q, k, v, o, lse, dout, d[:, :, 0], d[:, :, 1], d[:, :, 2], sl, None, 1.0 / math.sqrt(128), True)
        outs.append((o, lse, d))
    ops.set_flash_variant(0)
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert not bool(torch.isnan(outs[0][2].float()).any())
