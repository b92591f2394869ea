"""Frozen host of the fusion path: the MPT decoder (reference: src/otter_ai/models/mpt/{modeling_mpt,blocks,attention,
norm,custom_embedding}.py) re-stated for PyTorch-ROCm.

Scope note: the decoder is *frozen* in the Otter recipe and is NOT one of the hand-written
rows of round 1 -- its GEMMs and causal attention go through PyTorch-ROCm (hipBLASLt / SDPA), its LayerNorms through
libotter_hip.so.  What this file must get exactly right is the contract the hot path plugs into: parameter names
(`transformer.wte`, `transformer.blocks.{i}.{norm_1,attn.Wqkv,attn.out_proj,norm_2,ffn.up_proj,ffn.down_proj}`,
`transformer.norm_f`), ALiBi + causal masking, the tied un-embedding, the rolled-label loss and the legacy tuple KV cache
(k: [B,H,d,S], v: [B,H,S,d]) that `generate` relies on.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import PretrainedConfig, PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from . import functional as OF

_ATTN_DEFAULTS = dict(attn_type="multihead_attention", attn_pdrop=0.0, attn_impl="torch", qk_ln=False, clip_qkv=None,
                      softmax_scale=None, prefix_lm=False, attn_uses_sequence_id=False, alibi=True, alibi_bias_max=8)


class MPTConfig(PretrainedConfig):
    """Same fields as the reference's mpt/configuration_mpt.py:14-90 (only what OTTER-MPT7B uses is honoured)."""

    model_type = "mpt"

    def __init__(self, d_model=2048, n_heads=16, n_layers=24, expansion_ratio=4, max_seq_len=2048, vocab_size=50368,
                 resid_pdrop=0.0, emb_pdrop=0.0, learned_pos_emb=True, attn_config=None, init_device="cpu", logit_scale=None,
                 no_bias=False, verbose=0, embedding_fraction=1.0, norm_type="low_precision_layernorm", use_cache=False,
                 init_config=None, **kwargs):
        self.d_model, self.n_heads, self.n_layers = d_model, n_heads, n_layers
        self.expansion_ratio, self.max_seq_len, self.vocab_size = expansion_ratio, max_seq_len, vocab_size
        self.resid_pdrop, self.emb_pdrop, self.learned_pos_emb = resid_pdrop, emb_pdrop, learned_pos_emb
        ac = dict(_ATTN_DEFAULTS)
        ac.update(attn_config or {})
        self.attn_config = ac
        self.init_device, self.logit_scale, self.no_bias, self.verbose = init_device, logit_scale, no_bias, verbose
        self.embedding_fraction, self.norm_type, self.use_cache = embedding_fraction, norm_type, use_cache
        self.init_config = init_config or {}
        kwargs.pop("tie_word_embeddings", None)
        kwargs.setdefault("hidden_size", d_model)  # OtterLMMixin.init_otter reads config.hidden_size (modeling_otter.py:473)
        super().__init__(tie_word_embeddings=True, **kwargs)
        self._validate()

    def _validate(self):
        a = self.attn_config
        if self.d_model % self.n_heads:
            raise ValueError("d_model must be divisible by n_heads")
        if a["attn_type"] != "multihead_attention" or a["prefix_lm"] or a["attn_uses_sequence_id"] or a["qk_ln"] or a["clip_qkv"]:
            raise NotImplementedError("otter_amd's MPT host implements the OTTER-MPT7B attention configuration only "
                                      "(multihead, causal, no qk_ln / clip_qkv / prefix_lm / sequence_id)")
        if not a["alibi"]:
            raise NotImplementedError("learned position embeddings (alibi=False) are not implemented")
        if self.resid_pdrop or self.emb_pdrop or a["attn_pdrop"]:
            # the host never applies dropout (OTTER-MPT7B: all three are 0, mpt config :60-62,74); a non-zero value would
            # silently train differently from the reference
            raise NotImplementedError("resid_pdrop / emb_pdrop / attn_pdrop != 0 are not implemented in otter_amd's MPT host")
        if self.norm_type not in ("low_precision_layernorm", "layernorm"):
            raise NotImplementedError(f"norm_type {self.norm_type}")


def alibi_slopes(n_heads: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """mpt/attention.py:447-455."""
    _n = 2 ** math.ceil(math.log2(n_heads))
    m = torch.arange(1, _n + 1, dtype=torch.float32) * (alibi_bias_max / _n)
    slopes = 1.0 / torch.pow(2, m)
    if _n != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes


class SharedEmbedding(nn.Embedding):
    """mpt/custom_embedding.py:7-11."""

    def forward(self, input: torch.Tensor, unembed: bool = False) -> torch.Tensor:
        if unembed:
            if self.weight.requires_grad and self.weight.dtype == torch.float32 and input.is_cuda and input.dtype == torch.bfloat16:
                # trainable fp32 master (the reference unfreezes the input embeddings, modeling_otter.py:905): bf16 operand copy
                # from the shadow cache the AdamW kernel refreshes, fp32 weight gradient straight out of the GEMM -- instead of
                # a 413 MB cast of the table every forward and a 826 MB cast of its gradient every backward (0.8 ms per step)
                return OF.TrainableLinearFn.apply(input, self.weight, None)
            return F.linear(input, self.weight.to(input.dtype))
        return OF.embedding_rows(input, self.weight)    # F.embedding; under otter_amd's TrainStep the row gradients go to the sparse sink


class _Norm(nn.LayerNorm):
    """LPLayerNorm (mpt/norm.py:16-45) on the HIP LayerNorm kernel: statistics in fp32, output in the compute dtype."""

    def forward(self, x):
        return OF.layer_norm(x, self.weight, self.bias, self.eps, OF.compute_dtype_for(x))

    def add_forward(self, x, delta):
        """(x + delta, LN(x + delta)) fused in one HIP pass over the residual stream."""
        return OF.add_layer_norm(x, delta, self.weight, self.bias, self.eps, OF.compute_dtype_for(x))

    def fork_forward(self, x):
        """(x, LN(x)) as ONE autograd node: the gradient of the carried-on stream enters the LayerNorm backward as its residual term."""
        return OF.fork_layer_norm(x, self.weight, self.bias, self.eps, OF.compute_dtype_for(x))


def _own_mode() -> str:
    """Which GEMMs of the frozen decoder run on csrc/gemm.hip, OTTER_OWN_DECODER_GEMM:
      "1t" (THE DEFAULT since round 6d): EVERY decoder GEMM on the own kernels -- Wqkv, out_proj, up_proj + GELU, down_proj forward; the four input
            gradients against stored transposed copies of the frozen weights (K-contiguous operands, the cross-tile form of variant 26), down_proj's with
            GELU' in the tail.  Same-box interleaved A/B on the round-6d kernels: 123.17 / 123.45 ms per step against 123.42 / 123.74 for "mlp" and
            124.5 for "1" on the pool's fastest box, 128.18 against 127.71 (+0.37 %) on a mid box (profiles/r06d_modes_ab_final_box*.txt) -- inside the pool's box-to-box
            spread, and the decoder then needs no vendor GEMM; it was +1.0 % in round 6b, +4.0 % in round 5.
      "mlp" (the default of rounds 6-6c): the two products of the frozen MLP that carry a fusion -- up_proj + GELU, down_proj's input gradient
            + GELU' (functional.FrozenMLPFusedLegsFn) -- so that the decoder's 64 stand-alone GELU / GELU' passes per step (3.4 ms) are gone;
            the plain products stay on hipBLASLt.  Same-box interleaved A/B, round 6 (cross-tile ring + K-tile rotation in variant 26):
            126.33 ms per step against 126.73 with the library for all of them (profiles/r06_own_decoder_ab.txt); it was +2.8 % in round 5.
      "1":  EVERY decoder GEMM on the own kernels (functional.FrozenMLPFn, input gradients against the weights as stored through the K-major
            kernel: no transposed copies, 13 GB less): +1.2 % on the step (round 5: +4.0 %) -- the library's plain products are still ahead in situ.
      "0" / "lib": hipBLASLt for all of them + stand-alone GELU kernels (the round 1-5 default; A/B switch)."""
    v = os.environ.get("OTTER_OWN_DECODER_GEMM", "1t").lower()
    return {"1": "all", "1t": "allt", "mlp": "mlp", "": "allt", "attn": "attn", "attn_qkv": "attn_qkv", "attn_out": "attn_out"}.get(v, "lib")


def _own_gemm() -> bool:
    return _own_mode() in ("all", "allt")


def _own_transposed() -> bool:
    """ "1t" (round 6b): as "1", but the input gradients read stored transposed copies of the frozen weights (K-contiguous operands, the
    cross-tile form of variant 26) instead of the weights as stored through the K-major form -- the copies the library path keeps anyway."""
    return _own_mode() == "allt"


def _own_mlp_fused_legs() -> bool:
    return _own_mode() in ("mlp", "attn", "attn_qkv", "attn_out")


def _own_attn_linear(out_features: int, in_features: int) -> bool:
    """ "attn" / "attn_qkv" / "attn_out" (round 6c, A/B switches): the default "mlp" mode PLUS the attention projections (Wqkv and / or
    out_proj, forward and input gradient) on csrc/gemm.hip against stored transposed copies -- which of the plain products the library wins."""
    m = _own_mode()
    if m == "attn":
        return True
    if m == "attn_qkv":
        return out_features == 3 * in_features
    if m == "attn_out":
        return out_features == in_features
    return False


def _lin(x, w):
    return F.linear(x, w)


class _FrozenLinearFn(torch.autograd.Function):
    """y = x W^T whose input gradient is computed against a stored transposed copy of the (frozen) weight."""

    @staticmethod
    def forward(ctx, x, w, wt):
        ctx.save_for_backward(wt)
        return _lin(x, w)

    @staticmethod
    def backward(ctx, dy):
        (wt,) = ctx.saved_tensors
        return _lin(dy, wt), None, None


class FrozenAwareLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys).  When the weight is frozen -- as the whole decoder is in the Otter
    recipe -- the only backward product is dx = dy W, which hipBLASLt runs 13-17 % faster in its "x W^T" form against a
    K-contiguous operand (tools/lm_gemm_layouts.py: 330 -> 270, 108 -> 95, 412 -> 355, 421 -> 361 us on the four decoder
    shapes at C2).  The weight never changes, so a transposed copy in the compute dtype is built once (13 GB for MPT-7B
    out of 288 GB of HBM) and the dgrad is issued as F.linear(dy, W^T)."""

    def _copies(self, cd):
        w = self.weight
        key = (w._version, w.data_ptr(), cd)
        if getattr(self, "_fz_key", None) != key:
            wc = w.detach() if w.dtype == cd else w.detach().to(cd)
            self._fz_w, self._fz_wt, self._fz_key = wc, wc.t().contiguous(), key
        return self._fz_w, self._fz_wt

    def _copy_w(self, cd):
        """The weight in the compute dtype only (own-kernel path: the input gradient reads it as stored, no transposed copy)."""
        w = self.weight
        if w.dtype == cd:
            return w.detach()
        key = (w._version, w.data_ptr(), cd)
        if getattr(self, "_fzc_key", None) != key:
            self._fzc_w, self._fzc_key = w.detach().to(cd), key
        return self._fzc_w

    def release_copies(self):
        """Drop the cached compute-dtype / transposed copies (rebuilt on the next training forward)."""
        self._fz_w = self._fz_wt = self._fz_key = None
        self._fzc_w = self._fzc_key = None

    def _apply(self, fn, *a, **k):
        # .to() / .cpu() / .half(): the copies would otherwise stay behind on the old device in the old dtype
        self.release_copies()
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        w = self.weight
        if (w.requires_grad or self.bias is not None or not x.is_cuda or not torch.is_grad_enabled() or not x.requires_grad
                or os.environ.get("OTTER_NO_FROZEN_WT") == "1"):
            return F.linear(x, w, self.bias)
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        if _own_gemm() and cd == torch.bfloat16:
            if _own_transposed():
                wc, wt = self._copies(cd)
                return OF.frozen_linear_own(x if x.dtype == cd else x.to(cd), wc, wt)
            return OF.frozen_linear_own(x if x.dtype == cd else x.to(cd), self._copy_w(cd))
        if cd == torch.bfloat16 and _own_attn_linear(self.out_features, self.in_features):
            wc, wt = self._copies(cd)
            return OF.frozen_linear_own(x if x.dtype == cd else x.to(cd), wc, wt)
        wc, wt = self._copies(cd)
        return _FrozenLinearFn.apply(x if x.dtype == cd else x.to(cd), wc, wt)


class MPTMLP(nn.Module):
    def __init__(self, d_model, expansion_ratio, bias):
        super().__init__()
        self.up_proj = FrozenAwareLinear(d_model, expansion_ratio * d_model, bias=bias)
        self.act = nn.GELU()
        self.down_proj = FrozenAwareLinear(expansion_ratio * d_model, d_model, bias=bias)

    def forward(self, x):
        up, dn = self.up_proj, self.down_proj
        if (_own_gemm() and x.is_cuda and up.bias is None and dn.bias is None and not up.weight.requires_grad and not dn.weight.requires_grad
                and OF.compute_dtype_for(x) == torch.bfloat16):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            if _own_transposed():
                (wu, wu_t), (wd, wd_t) = up._copies(torch.bfloat16), dn._copies(torch.bfloat16)
                return OF.frozen_mlp(xb, wu, wd, wu_t, wd_t)
            return OF.frozen_mlp(xb, up._copy_w(torch.bfloat16), dn._copy_w(torch.bfloat16))
        if (_own_mlp_fused_legs() and x.is_cuda and x.requires_grad and torch.is_grad_enabled() and up.bias is None and dn.bias is None
                and not up.weight.requires_grad and not dn.weight.requires_grad and OF.compute_dtype_for(x) == torch.bfloat16):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            wu, wu_t = up._copies(torch.bfloat16)
            # round 6b: the GELU' leg reads a stored transposed copy of the frozen down_proj (both operands K-contiguous: the cross-tile form of
            # variant 26) instead of the weight as stored through the K-major form: 127.90 vs 128.33 ms per step on one box, 3 x 2 interleaved runs
            # (profiles/r06b_mlp_dgrad_kcontig_ab.txt); 4.3 GB of copies for MPT-7B.  OTTER_MLP_DGRAD_KCONTIG=0: the K-major form (A/B switch).
            if os.environ.get("OTTER_MLP_DGRAD_KCONTIG") != "0":
                wd, wd_t = dn._copies(torch.bfloat16)
                return OF.frozen_mlp_fused_legs(xb, wu, wd, wu_t, wd_t)
            return OF.frozen_mlp_fused_legs(xb, wu, dn._copy_w(torch.bfloat16), wu_t)
        u = self.up_proj(x)
        if u.is_cuda and os.environ.get("OTTER_TORCH_GELU") != "1":
            return self.down_proj(OF.gelu(u))      # csrc/elementwise.hip gelu_fwd / gelu_bwd (same exact-erf form as nn.GELU())
        return self.down_proj(self.act(u))


def ops_decode_attn(q, k, v, slopes, key_valid, scale):
    from . import ops

    return ops.decode_attn(q, k, v, slopes, key_valid, scale)


class MultiheadAttention(nn.Module):
    def __init__(self, d_model, n_heads, bias):
        super().__init__()
        self.d_model, self.n_heads = d_model, n_heads
        self.softmax_scale = 1.0 / math.sqrt(d_model / n_heads)
        self.Wqkv = FrozenAwareLinear(d_model, 3 * d_model, bias=bias)
        self.out_proj = FrozenAwareLinear(d_model, d_model, bias=bias)

    def forward(self, x, past_key_value=None, attn_bias=None, is_causal=True, flash=None, decode=None):
        B, S, D = x.shape
        H, d = self.n_heads, D // self.n_heads
        qkv = self.Wqkv(x)
        if decode is not None and past_key_value is not None and len(past_key_value) != 0 and S == 1:
            # cached decode step on HIP (csrc/decode.hip): the new key / value are appended to the cache in the reference's
            # layout (k [B,H,d,S], v [B,H,S,d]) and the single query attends over it in place -- no SDPA, no [B,H,1,S] bias tensor
            slopes, key_valid = decode
            q5 = qkv.view(B, 1, 3, H, d)
            k = torch.cat([past_key_value[0], q5[:, 0, 1].unsqueeze(-1)], dim=3)            # [B,H,d,S+1]
            v = torch.cat([past_key_value[1], q5[:, 0, 2].unsqueeze(2)], dim=2)             # [B,H,S+1,d]
            o = ops_decode_attn(q5[:, 0, 0], k.transpose(2, 3), v, slopes, key_valid, self.softmax_scale)
            return self.out_proj(o.reshape(B, 1, D)), (k, v)
        if flash is not None:
            # HIP flash attention on the fused projection output (otter_amd/csrc/flash.hip): ALiBi, causal and key-padding
            # masks are evaluated inside the kernel, q/k/v and their gradients are slices of one buffer
            slopes, key_valid = flash
            ctx = OF.flash_self_attention(qkv, H, slopes, key_valid, self.softmax_scale, is_causal)
            if past_key_value is not None:
                kv5 = qkv.view(B, S, 3, H, d)
                past_key_value = (kv5[:, :, 1].permute(0, 2, 3, 1), kv5[:, :, 2].transpose(1, 2))  # k [B,H,d,S], v [B,H,S,d]
            return self.out_proj(ctx), past_key_value
        q, k, v = qkv.chunk(3, dim=2)
        q = q.view(B, S, H, d).transpose(1, 2)  # [B,H,S,d]
        k = k.view(B, S, H, d).transpose(1, 2)
        v = v.view(B, S, H, d).transpose(1, 2)
        if past_key_value is not None:
            if len(past_key_value) != 0:
                k = torch.cat([past_key_value[0].transpose(2, 3), k], dim=2)
                v = torch.cat([past_key_value[1], v], dim=2)
            past_key_value = (k.transpose(2, 3), v)  # reference layout: k [B,H,d,S], v [B,H,S,d]
        if attn_bias is not None and attn_bias.size(0) == 1 and B > 1:
            attn_bias = attn_bias.expand(B, -1, -1, -1)
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_bias, dropout_p=0.0, is_causal=False, scale=self.softmax_scale)
        ctx = ctx.transpose(1, 2).reshape(B, S, D)
        return self.out_proj(ctx), past_key_value


class MPTBlock(nn.Module):
    """mpt/blocks.py:23-88."""

    def __init__(self, config: MPTConfig):
        super().__init__()
        bias = not config.no_bias
        self.norm_1 = _Norm(config.d_model, bias=bias)
        self.attn = MultiheadAttention(config.d_model, config.n_heads, bias)
        self.norm_2 = _Norm(config.d_model, bias=bias)
        self.ffn = MPTMLP(config.d_model, config.expansion_ratio, bias)

    def forward(self, x, past_key_value=None, attn_bias=None, attention_mask=None, is_causal=True, deferred=None,
                defer_out=False, flash=None, decode=None, fork_input=False):
        """`deferred` / `defer_out` (otter_amd extension, used by MPTModel.forward): the FFN output of a block is handed to
        the NEXT block un-added, where the residual add is fused into that block's norm_1 pass (one trip over the fp32
        residual stream instead of two).  With the defaults this is exactly mpt/blocks.py:68-88."""
        if deferred is not None:
            x, a = self.norm_1.add_forward(x, deferred)   # x = x + ffn_out(prev) ; a = norm_1(x)
        elif fork_input and x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
            x, a = self.norm_1.fork_forward(x)            # same values; the stream's two gradients meet inside the LayerNorm backward
        else:
            a = self.norm_1(x)
        b, past_key_value = self.attn(a, past_key_value=past_key_value, attn_bias=attn_bias, is_causal=is_causal, flash=flash, decode=decode)
        x, m = self.norm_2.add_forward(x, b)              # x = x + b ; m = norm_2(x)
        n = self.ffn(m)
        if defer_out:
            return x, None, past_key_value, n
        return x + n, None, past_key_value


def _gated_takes_deferred(block, x, delta) -> bool:
    """otter_amd.modeling_otter.OtterLayer forwards `deferred` to its gated cross-attention block, which fuses x + deferred into its first
    LayerNorm pass (HIP path: GPU tensors, fp32 stream + bf16 addend or equal dtypes).  OTTER_NO_DEFER_INTO_GATED=1: A/B switch."""
    g = getattr(block, "gated_cross_attn_layer", None)
    return (g is not None and getattr(g, "accepts_deferred", False) and x.is_cuda and delta.is_cuda
            and os.environ.get("OTTER_NO_DEFER_INTO_GATED") != "1")


class MPTPreTrainedModel(PreTrainedModel):
    config_class = MPTConfig
    base_model_prefix = "model"
    _no_split_modules = ["MPTBlock"]

    def _init_weights(self, module):
        std = self.config.init_config.get("init_std", 0.02) if isinstance(self.config.init_config, dict) else 0.02
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, 0.0, std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, 0.0, std)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            if module.bias is not None:
                nn.init.zeros_(module.bias)


class MPTModel(MPTPreTrainedModel):
    def __init__(self, config: MPTConfig):
        super().__init__(config)
        self.alibi_bias_max = config.attn_config["alibi_bias_max"]
        self.wte = SharedEmbedding(config.vocab_size, config.d_model)
        self.blocks = nn.ModuleList([MPTBlock(config) for _ in range(config.n_layers)])
        self.norm_f = _Norm(config.d_model, bias=not config.no_bias)
        self.is_causal = True
        self._slopes = None

    def get_input_embeddings(self):
        return self.wte

    def set_input_embeddings(self, value):
        self.wte = value

    def _attn_bias(self, s_k: int, device, attention_mask: Optional[torch.Tensor]):
        """[1 or B, H, 1, s_k] fp32 = ALiBi (key-position form, attention.py:458-464) + padding (modeling_mpt.py:135-144)."""
        if self._slopes is None or self._slopes.device != device:
            self._slopes = alibi_slopes(self.config.n_heads, self.alibi_bias_max).to(device)
        pos = torch.arange(1 - s_k, 1, dtype=torch.float32, device=device)
        bias = pos.view(1, 1, 1, s_k) * self._slopes.view(1, -1, 1, 1)
        if attention_mask is not None:
            am = attention_mask.bool()[:, -s_k:]
            if not bool(am.all()):
                bias = bias.masked_fill(~am.view(-1, 1, 1, s_k), torch.finfo(torch.float32).min)
        return bias

    def _flash_spec(self, x, s_past: int, attention_mask):
        """(alibi slopes [H] fp32, key_valid [B,S] uint8 or None) when the HIP flash-attention kernel covers this call:
        bf16 compute on the GPU, head_dim 128, no KV cache to prepend.  Left padding: a fully masked query row (a pad position) comes
        out as the uniform average of V in the reference and as 0 in the kernel.  Those rows only ever feed the keys / values of their own
        positions, which every real query masks -- the hidden states and logits of the real tokens are identical -- so an inference pass
        (generate()'s left-padded prompt, no autograd) takes the kernel too (round 3); under autograd the additive-mask SDPA path stays,
        because the gradient that reaches pad rows differs.  Otherwise None -> SDPA."""
        if not x.is_cuda or s_past != 0 or not self.is_causal or OF.compute_dtype_for(x) != torch.bfloat16:
            return None
        if self.config.d_model // self.config.n_heads != 128 or os.environ.get("OTTER_NO_FLASH") == "1":
            return None
        key_valid = None
        if attention_mask is not None:
            am = attention_mask.bool()
            if not bool(am.all()):
                if not bool(am[:, 0].all()) and torch.is_grad_enabled():
                    return None
                key_valid = am.to(torch.uint8).contiguous()
        if self._slopes is None or self._slopes.device != x.device:
            self._slopes = alibi_slopes(self.config.n_heads, self.alibi_bias_max).to(x.device)
        return self._slopes.float().contiguous(), key_valid

    def _decode_spec(self, x, S, s_past, attention_mask):
        """(alibi slopes, key_valid or None) when the HIP decode kernel covers this call: one new token over a non-empty cache,
        bf16 compute on the GPU, head_dim 128.  Left-padded prompts are fine here (padded keys are masked, the current token is
        always valid)."""
        if not x.is_cuda or S != 1 or s_past == 0 or OF.compute_dtype_for(x) != torch.bfloat16 or os.environ.get("OTTER_NO_FLASH") == "1":
            return None
        if self.config.d_model // self.config.n_heads != 128:
            return None
        key_valid = None
        if attention_mask is not None:
            am = attention_mask.bool()[:, -(s_past + 1):]
            if not bool(am.all()):
                key_valid = am.to(torch.uint8).contiguous()
        if self._slopes is None or self._slopes.device != x.device:
            self._slopes = alibi_slopes(self.config.n_heads, self.alibi_bias_max).to(x.device)
        return self._slopes.float().contiguous(), key_valid

    def forward(self, input_ids, past_key_values=None, attention_mask=None, use_cache=None, return_dict=True, **unused):
        use_cache = use_cache if use_cache is not None else self.config.use_cache
        if attention_mask is not None and self.training and int(attention_mask[:, 0].sum()) != attention_mask.shape[0]:
            raise NotImplementedError("MPT does not support training with left padding.")
        S = input_ids.size(1)
        if S > self.config.max_seq_len:
            raise ValueError(f"Cannot forward input with seq_len={S}, this model only supports seq_len<={self.config.max_seq_len}")
        x = self.wte(input_ids)
        s_past = 0
        if past_key_values is not None and len(past_key_values) and len(past_key_values[0]) != 0:
            s_past = past_key_values[0][0].size(3)
        s_k = S + s_past
        flash, attn_bias = self._flash_spec(x, s_past, attention_mask), None
        decode = self._decode_spec(x, S, s_past, attention_mask) if flash is None else None
        if flash is None and decode is None:
            # one additive mask per forward, shared by every block: ALiBi (+ padding) and, for S > 1, the causal triangle
            attn_bias = self._attn_bias(s_k, x.device, attention_mask)
            if self.is_causal and S != 1:
                causal = torch.ones(S, s_k, dtype=torch.bool, device=x.device).tril(diagonal=s_k - S)
                attn_bias = attn_bias.expand(-1, -1, S, -1).masked_fill(~causal, torch.finfo(torch.float32).min)
            attn_bias = attn_bias.to(OF.compute_dtype_for(x))
        if use_cache and past_key_values is None:
            past_key_values = [() for _ in range(self.config.n_layers)]
        elif past_key_values is not None and not isinstance(past_key_values, list):
            past_key_values = list(past_key_values)   # filled in place below (the reference takes a list, modeling_mpt.py:290-292)
        # Each block hands its FFN output over un-added (`delta`); the add is fused into the next LayerNorm pass.  A wrapper
        # that runs something on the hidden states before the decoder layer (OtterLayer with a gated cross-attention block)
        # needs the materialised sum, so the add is performed here for those layers.
        delta = None
        for i, block in enumerate(self.blocks):
            pkv = past_key_values[i] if past_key_values is not None else None
            if delta is not None and getattr(block, "gated_cross_attn_layer", None) is not None and not _gated_takes_deferred(block, x, delta):
                x = x + delta       # (a wrapper that cannot take the un-added addend gets the materialised sum)
                delta = None
            out = block(x, past_key_value=pkv, attn_bias=attn_bias, attention_mask=None, is_causal=self.is_causal,
                        deferred=delta, defer_out=True, flash=flash, decode=decode)
            x, pkv = out[0], out[2]
            delta = out[3] if len(out) > 3 else None
            if past_key_values is not None:
                past_key_values[i] = pkv
        if delta is not None:
            _, x = self.norm_f.add_forward(x, delta)
        else:
            x = self.norm_f(x)
        return BaseModelOutputWithPast(last_hidden_state=x, past_key_values=past_key_values)


class MPTForCausalLM(MPTPreTrainedModel):
    def __init__(self, config: MPTConfig):
        super().__init__(config)
        self.transformer = MPTModel(config)
        self.logit_scale = None
        if config.logit_scale is not None:
            ls = config.logit_scale
            self.logit_scale = 1 / math.sqrt(config.d_model) if ls == "inv_sqrt_d_model" else ls

    def get_input_embeddings(self):
        return self.transformer.wte

    def set_input_embeddings(self, value):
        self.transformer.wte = value

    def get_output_embeddings(self):
        return self.transformer.wte

    def set_output_embeddings(self, new_embeddings):
        self.transformer.wte = new_embeddings

    def get_decoder(self):
        return self.transformer

    def set_decoder(self, decoder):
        self.transformer = decoder

    def forward(self, input_ids, past_key_values=None, attention_mask=None, labels=None, use_cache=None, return_dict=True,
                **unused):
        out = self.transformer(input_ids=input_ids, past_key_values=past_key_values, attention_mask=attention_mask,
                               use_cache=use_cache)
        logits = self.transformer.wte(out.last_hidden_state, True)
        if self.logit_scale is not None:
            logits = logits * self.logit_scale
        loss = None
        if labels is not None:
            _labels = torch.roll(labels, shifts=-1)      # flat roll, exactly as modeling_mpt.py:428-435
            _labels[:, -1] = -100
            flat, lab = logits.view(-1, logits.size(-1)), _labels.to(logits.device).view(-1)
            if flat.is_cuda and flat.dtype == torch.bfloat16 and flat.size(-1) % 4 == 0 and os.environ.get("OTTER_TORCH_CE") != "1":
                loss = OF.cross_entropy_bf16(flat, lab.contiguous())   # one pass over the bf16 logits (csrc/loss.hip)
            else:
                loss = F.cross_entropy(flat.float(), lab)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=out.past_key_values)
