"""Frozen LLaMA decoder host of config C4 (OTTER-Video-LLaMA7B), MI355X-native.

The reference takes this decoder from the third-party `transformers` package (modeling_otter.py:54,759-767; an in-repo
restatement of the same arithmetic sits in xformers_model/llama.py:95-327).  This module keeps that class surface -- class
names, constructor (a `transformers.LlamaConfig`), state-dict keys (`model.embed_tokens`, `model.layers.{i}.self_attn.
{q,k,v,o}_proj`, `.mlp.{gate,up,down}_proj`, `.input_layernorm`, `.post_attention_layernorm`, `model.norm`, `lm_head`),
forward signature and output type -- and runs the bf16 training path on libotter_hip.so:

  RMSNorm (+ fused residual add)   otter_add_rmsnorm_fwd / otter_rmsnorm_bwd_ex            llama.py:95-112, 311-318
  q|k|v projection                 ONE GEMM against the concatenated frozen weights (hipBLASLt; the weights never change)
  RoPE                             otter_rope_strided on the q and k heads of that buffer   llama.py:115-166
  causal / padded attention        csrc/flash.hip (MFMA, fwd + bwd), head_dim 128           llama.py:169-213
  SwiGLU                           gate|up as one GEMM + otter_swiglu_fwd / _bwd             llama.py:216-223

fp32 (parity mode), CPU construction, KV-cache decode, grouped-query attention, custom position_ids and head dims other than
128 take the plain PyTorch expression of the same arithmetic (the decoder is the frozen HOST of the fusion path, like
otter_amd/mpt.py; the fusion modules themselves have no such path).  Parity: tests/golden/otter_tiny_llama.npz, generated
by the reference with the installed transformers LlamaForCausalLM (oracle/gen_golden.py)."""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import LlamaConfig, PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from . import functional as OF
from . import ops
from .mpt import FrozenAwareLinear


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class LlamaRotaryEmbedding(nn.Module):
    """cos / sin tables in fp32: inv_freq = theta^(-2i/d), emb = cat(freqs, freqs) (half-split layout)."""

    def __init__(self, config: LlamaConfig):
        super().__init__()
        rs = getattr(config, "rope_scaling", None)
        if rs and (rs.get("rope_type", rs.get("type", "default")) not in ("default", None)):
            raise NotImplementedError("otter_amd's LLaMA host implements the default rotary embedding only (LLaMA-7B)")
        rp = getattr(config, "rope_parameters", None) or {}
        self.theta = float(rp.get("rope_theta", None) or getattr(config, "rope_theta", 10000.0))
        self.dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        self._tab = None

    def tables(self, n: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """(cos, sin) fp32 [n', dim] with n' >= n, cached per device."""
        if self._tab is None or self._tab[0].device != device or self._tab[0].shape[0] < n:
            n_alloc = max(n, 512)
            inv = 1.0 / (self.theta ** (torch.arange(0, self.dim, 2, dtype=torch.float32, device=device) / self.dim))
            fr = torch.arange(n_alloc, dtype=torch.float32, device=device)[:, None] * inv[None, :]
            emb = torch.cat((fr, fr), dim=-1)
            self._tab = (emb.cos().contiguous(), emb.sin().contiguous())
        return self._tab


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, delta=None, out_dtype=None):
        """y = RMSNorm(x [+ delta]); returns y, or (x + delta, y) when delta is given."""
        if x.is_cuda:
            return OF.add_rms_norm(x, delta, self.weight, self.variance_epsilon, out_dtype or x.dtype)
        xs = x if delta is None else x + delta.to(x.dtype)
        h = xs.float()
        y = (self.weight.float() * (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)).to(xs.dtype)).to(
            out_dtype or xs.dtype)
        return y if delta is None else (xs, y)


class _FusedFrozenLinear:
    """x -> x [W_1; W_2; ...]^T for several bias-free FROZEN nn.Linear modules sharing their input (q|k|v, gate|up): one GEMM
    instead of two or three, output slices contiguous in one buffer; the concatenated weight and its transpose (for the only
    backward product of a frozen layer, dx = dy W) are built once per weight version, in the compute dtype.

    Memory: the fused copy and its transpose are PERSISTENT duplicates of frozen weights -- for LLaMA-7B in bf16 about
    2 x 8.6 GB on top of the 13 GB model (288 GB of HBM per MI355X is what this layout is sized for).  `release()` drops them (they are
    rebuilt on the next forward); LlamaForCausalLM.release_fused_copies() does it for the whole model and runs automatically when the
    module is moved or cast (`.to()`, `.cpu()`, `.half()`: nn.Module._apply)."""

    def __init__(self, mods: List[nn.Linear]):
        self.mods = mods
        self._key = None
        self._w = self._wt = None

    def release(self):
        self._key = None
        self._w = self._wt = None

    def usable(self, x) -> bool:
        return (x.is_cuda and all(m.bias is None and not m.weight.requires_grad for m in self.mods)
                and os.environ.get("OTTER_NO_FUSED_LLAMA") != "1")

    def __call__(self, x, cd):
        key = tuple((m.weight._version, m.weight.data_ptr()) for m in self.mods) + (cd,)
        if key != self._key:
            w = torch.cat([m.weight.detach().to(cd) for m in self.mods], dim=0).contiguous()
            self._w, self._wt, self._key = w, w.t().contiguous(), key
        from .mpt import _FrozenLinearFn

        x = x if x.dtype == cd else x.to(cd)
        if torch.is_grad_enabled() and x.requires_grad:
            return _FrozenLinearFn.apply(x, self._w, self._wt)
        return F.linear(x, self._w)


class LlamaMLP(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        bias = bool(getattr(config, "mlp_bias", False))
        if getattr(config, "hidden_act", "silu") != "silu":
            raise NotImplementedError("LLaMA host: hidden_act must be silu")
        self.gate_proj = FrozenAwareLinear(config.hidden_size, config.intermediate_size, bias=bias)
        self.up_proj = FrozenAwareLinear(config.hidden_size, config.intermediate_size, bias=bias)
        self.down_proj = FrozenAwareLinear(config.intermediate_size, config.hidden_size, bias=bias)
        self._gu = _FusedFrozenLinear([self.gate_proj, self.up_proj])

    def forward(self, x):
        cd = OF.compute_dtype_for(x)
        if cd == torch.bfloat16 and self._gu.usable(x) and self.gate_proj.out_features % 8 == 0:
            gu = self._gu(x, cd)                         # [..., 2*I] = gate | up
            return self.down_proj(OF.swiglu(gu))
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaAttention(nn.Module):
    def __init__(self, config: LlamaConfig, layer_idx: int):
        super().__init__()
        self.layer_idx = layer_idx
        self.hidden = config.hidden_size
        self.n_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or self.hidden // self.n_heads
        self.n_kv = getattr(config, "num_key_value_heads", None) or self.n_heads
        bias = bool(getattr(config, "attention_bias", False))
        if getattr(config, "attention_dropout", 0.0):
            raise NotImplementedError("LLaMA host: attention_dropout != 0 is not implemented")
        self.q_proj = FrozenAwareLinear(self.hidden, self.n_heads * self.head_dim, bias=bias)
        self.k_proj = FrozenAwareLinear(self.hidden, self.n_kv * self.head_dim, bias=bias)
        self.v_proj = FrozenAwareLinear(self.hidden, self.n_kv * self.head_dim, bias=bias)
        self.o_proj = FrozenAwareLinear(self.n_heads * self.head_dim, self.hidden, bias=bias)
        self.scale = 1.0 / math.sqrt(self.head_dim)
        self._qkv = _FusedFrozenLinear([self.q_proj, self.k_proj, self.v_proj])

    def flash_ok(self, x, s_past: int, default_positions: bool) -> bool:
        if any(hasattr(p, "lora_delta") for p in (self.q_proj, self.k_proj, self.v_proj)):
            return False      # LoRA adapters (otter_amd/lora.py) live in the modules' own forward: the fused q|k|v weight would bypass them
        return (x.is_cuda and OF.compute_dtype_for(x) == torch.bfloat16 and self.head_dim == 128 and self.n_kv == self.n_heads
                and s_past == 0 and default_positions and self._qkv.usable(x) and os.environ.get("OTTER_NO_FLASH") != "1")

    def forward(self, x, cos, sin, attn_mask=None, key_valid=None, past_key_value=None, use_cache=False, flash=False):
        """x [B,S,D] (compute dtype); cos/sin fp32 [S, d] for this call's positions (or [B,S,d] with custom position_ids);
        attn_mask: additive [B or 1,1,S,Sk] for the plain path; key_valid uint8 [B,S] for the flash path."""
        B, S, _ = x.shape
        H, Hk, d = self.n_heads, self.n_kv, self.head_dim
        if flash:
            qkv = self._qkv(x, torch.bfloat16)                                  # [B,S,3*H*d]
            ctx, k_rot, v = OF.rope_flash_attention(qkv, cos, sin, key_valid, H, self.scale, want_kv=use_cache)
            new_past = (k_rot.transpose(1, 2), v.transpose(1, 2)) if use_cache else None   # [B,H,S,d] like HF's legacy cache
            return self.o_proj(ctx), new_past
        q = self.q_proj(x).view(B, S, H, d).transpose(1, 2)                     # [B,H,S,d]
        k = self.k_proj(x).view(B, S, Hk, d).transpose(1, 2)
        v = self.v_proj(x).view(B, S, Hk, d).transpose(1, 2)
        c, s_ = (cos[:, None], sin[:, None]) if cos.dim() == 3 else (cos[None, None], sin[None, None])
        c, s_ = c.to(q.dtype), s_.to(q.dtype)
        q = q * c + _rotate_half(q) * s_
        k = k * c + _rotate_half(k) * s_
        if past_key_value is not None and len(past_key_value) == 2:
            k = torch.cat([past_key_value[0], k], dim=2)
            v = torch.cat([past_key_value[1], v], dim=2)
        new_past = (k, v) if use_cache else None
        if (S == 1 and k.shape[2] > 1 and x.is_cuda and q.dtype == torch.bfloat16 and d == 128 and Hk == H
                and os.environ.get("OTTER_NO_FLASH") != "1"):
            # cached decode step on HIP (csrc/decode.hip): one query over the [B,H,S,d] cache, read in place
            o = ops.decode_attn(q[:, :, 0], k, v, None, key_valid, self.scale)
            return self.o_proj(o.reshape(B, 1, H * d)), new_past
        if Hk != H:
            k = k.repeat_interleave(H // Hk, dim=1)
            v = v.repeat_interleave(H // Hk, dim=1)
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False, scale=self.scale)
        return self.o_proj(ctx.transpose(1, 2).reshape(B, S, H * d)), new_past


class LlamaDecoderLayer(nn.Module):
    """xformers_model/llama.py:286-327 (pre-norm residual block)."""

    def __init__(self, config: LlamaConfig, layer_idx: int):
        super().__init__()
        self.self_attn = LlamaAttention(config, layer_idx)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states, attention_mask=None, cos=None, sin=None, key_valid=None, past_key_value=None, use_cache=False,
                flash=False, deferred=None, defer_out=False, **unused):
        """`deferred` / `defer_out` (otter_amd extension, used by LlamaModel.forward): the MLP output of a layer is handed to
        the NEXT layer un-added, where the residual add is fused into that layer's input RMSNorm pass (one trip over the fp32
        residual stream instead of two).  With the defaults this is exactly the reference block."""
        x = hidden_states
        cd = OF.compute_dtype_for(x)
        if deferred is not None:
            x, a = self.input_layernorm(x, delta=deferred, out_dtype=cd)       # x = x + mlp_out(prev) ; a = norm(x)
        else:
            a = self.input_layernorm(x, out_dtype=cd)
        b, new_past = self.self_attn(a, cos, sin, attn_mask=attention_mask, key_valid=key_valid, past_key_value=past_key_value,
                                     use_cache=use_cache, flash=flash)
        x, m = self.post_attention_layernorm(x, delta=b, out_dtype=cd)          # x = x + b ; m = norm(x)   (one pass)
        d = self.mlp(m)
        if defer_out:
            return x, new_past, d
        return x + d, new_past


class LlamaPreTrainedModel(PreTrainedModel):
    config_class = LlamaConfig
    base_model_prefix = "model"
    _no_split_modules = ["LlamaDecoderLayer"]
    _supports_sdpa = True            # transformers' attention-implementation check (the config default is "sdpa"); the
    _supports_flash_attn = False     # attention here is this module's own (flash.hip / F.scaled_dot_product_attention)
    _supports_flex_attn = False

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, 0.0, std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, 0.0, std)
        elif isinstance(module, LlamaRMSNorm):
            nn.init.ones_(module.weight)


class LlamaModel(LlamaPreTrainedModel):
    def __init__(self, config: LlamaConfig):
        super().__init__(config)
        self.padding_idx = getattr(config, "pad_token_id", None)
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.rotary_emb = LlamaRotaryEmbedding(config)
        self.post_init()

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    @staticmethod
    def _core(layer):
        return getattr(layer, "decoder_layer", layer)   # OtterLayer wraps the decoder layer (modeling_otter.py:398-442)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                use_cache=None, return_dict=True, **unused):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        # HF / reference default: the config's use_cache, in train mode too; the cache entries are views of buffers the
        # attention path produces anyway
        use_cache = bool(use_cache) if use_cache is not None else bool(getattr(self.config, "use_cache", True))
        x = self.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        B, S = x.shape[:2]
        s_past = 0
        if past_key_values is not None and len(past_key_values) and past_key_values[0] is not None and len(past_key_values[0]) == 2:
            s_past = past_key_values[0][0].shape[2]
        s_k = S + s_past
        cos_t, sin_t = self.rotary_emb.tables(s_k, x.device)
        default_pos = position_ids is None
        if default_pos:
            cos, sin = cos_t[s_past:s_k], sin_t[s_past:s_k]
        else:
            cos, sin = cos_t[position_ids], sin_t[position_ids]                  # [B,S,d]
        am = None
        if attention_mask is not None:
            am = attention_mask.bool()
            if bool(am.all()):
                am = None
        core0 = self._core(self.layers[0]).self_attn
        # the flash kernel zeroes fully masked query rows (the reference's softmax makes them uniform): left padding -> plain path
        flash = core0.flash_ok(x, s_past, default_pos) and (am is None or bool(am[:, 0].all()))
        mask = key_valid = None
        if flash:
            key_valid = am.to(torch.uint8).contiguous() if am is not None else None
        else:
            if S == 1 and s_past > 0 and am is not None:
                key_valid = am[:, -s_k:].to(torch.uint8).contiguous()      # for the HIP decode step (padded keys masked)
            cd = OF.compute_dtype_for(x)
            neg = torch.finfo(torch.float32).min
            mask = torch.zeros(1, 1, S, s_k, dtype=torch.float32, device=x.device)
            if S > 1:
                causal = torch.ones(S, s_k, dtype=torch.bool, device=x.device).tril(diagonal=s_k - S)
                mask = mask.masked_fill(~causal, neg)
            if am is not None:
                mask = mask.expand(B, -1, -1, -1).masked_fill(~am[:, None, None, -s_k:], neg)
            mask = mask.to(cd)
        new_pasts = [] if use_cache else None
        # each layer hands its MLP output over un-added (`delta`); a wrapper that runs something on the hidden states before the
        # decoder layer (OtterLayer with a gated cross-attention block) needs the materialised sum
        delta = None
        for i, layer in enumerate(self.layers):
            pkv = past_key_values[i] if (past_key_values is not None and len(past_key_values) > i) else None
            if delta is not None and getattr(layer, "gated_cross_attn_layer", None) is not None:
                x = x + delta
                delta = None
            x, npkv, delta = layer(x, attention_mask=mask, cos=cos, sin=sin, key_valid=key_valid, past_key_value=pkv, use_cache=use_cache,
                                   flash=flash, deferred=delta, defer_out=True)
            if use_cache:
                new_pasts.append(npkv)
        if delta is not None:
            _, x = self.norm(x, delta=delta, out_dtype=OF.compute_dtype_for(x))
        else:
            x = self.norm(x, out_dtype=OF.compute_dtype_for(x))
        return BaseModelOutputWithPast(last_hidden_state=x, past_key_values=tuple(new_pasts) if use_cache else None)


class LlamaForCausalLM(LlamaPreTrainedModel):
    _tied_weights_keys = None

    def __init__(self, config: LlamaConfig):
        super().__init__(config)
        self.model = LlamaModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def get_decoder(self):
        return self.model

    def set_decoder(self, decoder):
        self.model = decoder

    def release_fused_copies(self):
        """Free every cached duplicate of the frozen weights (fused q|k|v and gate|up matrices, transposed dgrad copies: ~17 GB for
        LLaMA-7B in bf16, see _FusedFrozenLinear); they are rebuilt lazily by the next training forward."""
        from .mpt import FrozenAwareLinear

        for m in self.modules():
            if isinstance(m, FrozenAwareLinear):
                m.release_copies()
            for f in (getattr(m, "_gu", None), getattr(m, "_qkv", None)):
                if isinstance(f, _FusedFrozenLinear):
                    f.release()

    def _apply(self, fn, *a, **k):
        self.release_fused_copies()
        return super()._apply(fn, *a, **k)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, return_dict=True, **unused):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                         inputs_embeds=inputs_embeds, use_cache=use_cache)
        logits = self.lm_head(out.last_hidden_state)
        loss = None
        if labels is not None:
            # HF: logits[..., :-1, :] against labels[..., 1:], mean over the valid targets (ignore_index -100)
            lab = torch.full_like(labels, -100)
            lab[:, :-1] = labels[:, 1:]
            flat, lab = logits.view(-1, logits.size(-1)), lab.to(logits.device).view(-1)
            if flat.is_cuda and flat.dtype == torch.bfloat16 and flat.size(-1) % 4 == 0 and os.environ.get("OTTER_TORCH_CE") != "1":
                loss = OF.cross_entropy_bf16(flat, lab)
            else:
                loss = F.cross_entropy(flat.float(), lab, ignore_index=-100)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=out.past_key_values)
