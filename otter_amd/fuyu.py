"""OtterHD / Fuyu-8B path (BASELINE config C5), MI355X-native.

Reference: src/otter_ai/models/fuyu/modeling_fuyu.py (FuyuForCausalLM: word embeddings + `vision_embed_tokens` linear patch
projection scattered into the sequence, `gather_continuous_embeddings` :44-77, forward :79-143) over
fuyu/modeling_persimmon.py (the decoder with flash-attn's fused ops: fused_layer_norm :286-287,360,374, fused rotary
:303-304, flash_attn_func :310, fused sq-relu MLP :180-194).  No vision tower, no gated cross-attention: a decoder whose
first ~1300 positions are linear projections of 30x30x3 image patches, fully fine-tuned.

Same class surface and state-dict keys as the reference / transformers (`language_model.model.layers.{i}.self_attn.
{query_key_value,dense,q_layernorm,k_layernorm}`, `.mlp.{dense_h_to_4h,dense_4h_to_h}`, `.input_layernorm`,
`.post_attention_layernorm`, `language_model.model.{embed_tokens,final_layernorm}`, `language_model.lm_head`,
`vision_embed_tokens`), taking transformers' `FuyuConfig` / `PersimmonConfig`.

bf16 GPU path (every parameter is trainable here, so the GEMMs are torch / hipBLASLt with their wgrad):
  LayerNorm (+ fused residual add)       csrc/norm.hip
  q / k LayerNorm over head_dim + partial RoPE   csrc/fuyu.hip: otter_qk_norm_rope_fwd / _bwd, reading the per-head interleaved
                                         [H,3,d] projection buffer in place, writing compact [B,S,H,64] q / k
  causal attention                       csrc/flash.hip, head_dim 64 (round 3): two heads per workgroup on the 128-wide tile layouts,
                                         v read in place from the projection buffer, ctx written as [B,S,H*64], dv written into the
                                         v slots of dqkv.  (Round 2 zero-padded the heads to 128 columns: twice the attention FLOPs
                                         and padded q / k / v / dO / o copies; still used for an odd head count, OTTER_FUYU_PAD128=1.)
  squared-ReLU                           csrc/fuyu.hip: otter_sqrelu_fwd / _bwd
  patch embeddings into the sequence     csrc/fuyu.hip: otter_scatter_rows (+ gather for the backward)
fp32 / CPU: the plain PyTorch expression of the same arithmetic (parity mode; pinned by tests/golden/fuyu_tiny.npz, generated
by the reference's own FuyuForCausalLM)."""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import FuyuConfig, PersimmonConfig, PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from . import functional as OF


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _cfg_get(cfg, name, default=None):
    v = getattr(cfg, name, None)
    if v is None:
        rp = getattr(cfg, "rope_parameters", None) or {}
        v = rp.get(name, None)
    return default if v is None else v


class PersimmonMLP(nn.Module):
    def __init__(self, config: PersimmonConfig):
        super().__init__()
        if getattr(config, "hidden_act", "relu2") != "relu2":
            raise NotImplementedError("Persimmon host: hidden_act must be relu2 (squared ReLU)")
        self.dense_h_to_4h = nn.Linear(config.hidden_size, config.intermediate_size)
        self.dense_4h_to_h = nn.Linear(config.intermediate_size, config.hidden_size)

    def forward(self, x):
        # relu(h)^2 and the second Linear in one autograd node: the activation's backward rides in the input-gradient GEMM's tail
        return OF.sqrelu_linear(self.dense_4h_to_h, OF.trainable_linear(self.dense_h_to_4h, x))


class PersimmonAttention(nn.Module):
    def __init__(self, config: PersimmonConfig):
        super().__init__()
        self.hidden = config.hidden_size
        self.n_heads = config.num_attention_heads
        self.head_dim = self.hidden // self.n_heads
        if self.head_dim * self.n_heads != self.hidden:
            raise ValueError("hidden_size must be divisible by num_heads")
        if getattr(config, "attention_dropout", 0.0):
            raise NotImplementedError("Persimmon host: attention_dropout != 0 is not implemented")
        rs = getattr(config, "rope_scaling", None)
        if rs and rs.get("rope_type", rs.get("type", "default")) not in ("default", None):
            raise NotImplementedError("Persimmon host: rope scaling is not implemented (Fuyu-8B uses none)")
        self.rot = int(float(_cfg_get(config, "partial_rotary_factor", 0.5)) * self.head_dim)
        self.theta = float(_cfg_get(config, "rope_theta", 25000.0))
        self.query_key_value = nn.Linear(self.hidden, 3 * self.hidden, bias=True)
        self.dense = nn.Linear(self.hidden, self.hidden, bias=True)
        self.qk_layernorm = bool(getattr(config, "qk_layernorm", True))
        if self.qk_layernorm:
            self.q_layernorm = nn.LayerNorm(self.head_dim, eps=config.layer_norm_eps, elementwise_affine=True)
            self.k_layernorm = nn.LayerNorm(self.head_dim, eps=config.layer_norm_eps, elementwise_affine=True)
        self.scale = 1.0 / math.sqrt(self.head_dim)

    def hip_ok(self, x, s_past, default_pos) -> bool:
        return (x.is_cuda and OF.compute_dtype_for(x) == torch.bfloat16 and self.head_dim == 64 and self.qk_layernorm and self.rot % 16 == 0
                and 0 < self.rot <= 64 and s_past == 0 and default_pos and os.environ.get("OTTER_NO_FLASH") != "1")

    def forward(self, x, cos, sin, attn_mask=None, past_key_value=None, use_cache=False, hip=False):
        B, S, _ = x.shape
        H, d = self.n_heads, self.head_dim
        qkv = OF.trainable_linear(self.query_key_value, x)                      # [B,S,H*3*d], per head (q | k | v)
        if hip:
            if use_cache:
                ctx, k, v = OF.persimmon_attention(qkv, self.q_layernorm, self.k_layernorm, cos, sin, H, self.rot, self.scale, want_kv=True)
                return OF.trainable_linear(self.dense, ctx), (k, v)
            ctx = OF.persimmon_attention(qkv, self.q_layernorm, self.k_layernorm, cos, sin, H, self.rot, self.scale)
            return OF.trainable_linear(self.dense, ctx), None
        q5 = qkv.view(B, S, H, 3, d)
        q, k, v = q5[..., 0, :], q5[..., 1, :], q5[..., 2, :]
        if self.qk_layernorm:
            q = self.q_layernorm(q)
            k = self.k_layernorm(k)
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)       # [B,H,S,d]
        c, s_ = (cos[:, None], sin[:, None]) if cos.dim() == 3 else (cos[None, None], sin[None, None])
        c, s_ = c.to(q.dtype), s_.to(q.dtype)
        r = self.rot
        q = torch.cat((q[..., :r] * c + _rotate_half(q[..., :r]) * s_, q[..., r:]), dim=-1)
        k = torch.cat((k[..., :r] * c + _rotate_half(k[..., :r]) * s_, k[..., r:]), dim=-1)
        if past_key_value is not None and len(past_key_value) == 2:
            k = torch.cat([past_key_value[0], k], dim=2)
            v = torch.cat([past_key_value[1], v], dim=2)
        new_past = (k, v) if use_cache else None
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False, scale=self.scale)
        return self.dense(ctx.transpose(1, 2).reshape(B, S, H * d)), new_past


class PersimmonDecoderLayer(nn.Module):
    """modeling_persimmon.py:322-392 (pre-LN block, dropout 0)."""

    def __init__(self, config: PersimmonConfig):
        super().__init__()
        if getattr(config, "hidden_dropout", 0.0):
            raise NotImplementedError("Persimmon host: hidden_dropout != 0 is not implemented")
        self.self_attn = PersimmonAttention(config)
        self.mlp = PersimmonMLP(config)
        self.input_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.post_attention_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, x, cos, sin, attn_mask=None, past_key_value=None, use_cache=False, hip=False, pending=None, defer=False):
        """hip path: `pending` is the previous layer's MLP output, not yet added to the residual stream -- the add rides in this layer's
        first LayerNorm pass; with `defer` the layer hands its own MLP output back the same way ((x, past, mlp_out) instead of (x, past))."""
        if hip:
            n1, n2 = self.input_layernorm, self.post_attention_layernorm
            if pending is not None:
                x, a = OF.add_layer_norm(x, pending, n1.weight, n1.bias, n1.eps, torch.bfloat16)   # x = x + pending ; a = LN(x)  (one pass)
            else:
                a = OF.layer_norm(x, n1.weight, n1.bias, n1.eps, torch.bfloat16)
            b, new_past = self.self_attn(a, cos, sin, use_cache=use_cache, hip=True)
            x, m = OF.add_layer_norm(x, b, n2.weight, n2.bias, n2.eps, torch.bfloat16)   # x = x + b ; m = LN(x)  (one pass)
            if defer:
                return x, new_past, self.mlp(m)
            return x + self.mlp(m), new_past
        b, new_past = self.self_attn(self.input_layernorm(x), cos, sin, attn_mask=attn_mask, past_key_value=past_key_value, use_cache=use_cache)
        x = x + b
        return x + self.mlp(self.post_attention_layernorm(x)), new_past


class PersimmonPreTrainedModel(PreTrainedModel):
    config_class = PersimmonConfig
    base_model_prefix = "model"
    _no_split_modules = ["PersimmonDecoderLayer"]
    _supports_sdpa = True
    _supports_flash_attn = False
    _supports_flex_attn = False

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, 0.0, std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, 0.0, std)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            nn.init.zeros_(module.bias)


class PersimmonModel(PersimmonPreTrainedModel):
    def __init__(self, config: PersimmonConfig):
        super().__init__(config)
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, getattr(config, "pad_token_id", None))
        self.layers = nn.ModuleList([PersimmonDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.final_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self._tab = None
        self.post_init()

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    def _tables(self, n, device):
        att = self.layers[0].self_attn
        if self._tab is None or self._tab[0].device != device or self._tab[0].shape[0] < n:
            na = max(n, 512)
            inv = 1.0 / (att.theta ** (torch.arange(0, att.rot, 2, dtype=torch.float32, device=device) / att.rot))
            fr = torch.arange(na, dtype=torch.float32, device=device)[:, None] * inv[None, :]
            emb = torch.cat((fr, fr), dim=-1)
            self._tab = (emb.cos().contiguous(), emb.sin().contiguous())
        return self._tab

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, use_cache=None,
                return_dict=True, **unused):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        # HF / reference default: the config's use_cache, in train mode too; the cache entries are views, not copies
        use_cache = bool(use_cache) if use_cache is not None else bool(getattr(self.config, "use_cache", True))
        x = self.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        B, S = x.shape[:2]
        s_past = 0
        if past_key_values is not None and len(past_key_values) and past_key_values[0] is not None and len(past_key_values[0]) == 2:
            s_past = past_key_values[0][0].shape[2]
        s_k = S + s_past
        cos_t, sin_t = self._tables(s_k, x.device)
        default_pos = position_ids is None
        if not default_pos and position_ids.shape[0] == 1 and bool((position_ids[0] == torch.arange(s_past, s_k, device=position_ids.device)).all()):
            default_pos = True            # the reference's forward always passes arange(past, past + S) (modeling_fuyu.py:116-119)
        if default_pos:
            cos, sin = cos_t[s_past:s_k], sin_t[s_past:s_k]
        else:
            cos, sin = cos_t[position_ids], sin_t[position_ids]
        am = None
        if attention_mask is not None:
            am = attention_mask.bool()
            if bool(am.all()):
                am = None
        # HIP path: causal only, like the reference's flash_attn_func(causal=True) which never sees the padding mask
        # (modeling_persimmon.py:310): with right padding the real positions are identical; left padding takes the plain path
        hip = self.layers[0].self_attn.hip_ok(x, s_past, default_pos) and (am is None or bool(am[:, 0].all()))
        mask = None
        if not hip:
            neg = torch.finfo(torch.float32).min
            mask = torch.zeros(1, 1, S, s_k, dtype=torch.float32, device=x.device)
            if S > 1:
                causal = torch.ones(S, s_k, dtype=torch.bool, device=x.device).tril(diagonal=s_k - S)
                mask = mask.masked_fill(~causal, neg)
            if am is not None:
                mask = mask.expand(B, -1, -1, -1).masked_fill(~am[:, None, None, -s_k:], neg)
            mask = mask.to(OF.compute_dtype_for(x))
        new_pasts = [] if use_cache else None
        pending = None      # hip path: a layer's MLP output joins the residual stream inside the NEXT LayerNorm pass (one kernel less per layer)
        for i, layer in enumerate(self.layers):
            pkv = past_key_values[i] if (past_key_values is not None and len(past_key_values) > i) else None
            if hip:
                x, npkv, pending = layer(x, cos, sin, use_cache=use_cache, hip=True, pending=pending, defer=True)
            else:
                x, npkv = layer(x, cos, sin, attn_mask=mask, past_key_value=pkv, use_cache=use_cache)
            if use_cache:
                new_pasts.append(npkv)
        n = self.final_layernorm
        if pending is not None:
            x = OF.add_layer_norm(x, pending, n.weight, n.bias, n.eps, OF.compute_dtype_for(x))[1]
        else:
            x = OF.layer_norm(x, n.weight, n.bias, n.eps, OF.compute_dtype_for(x)) if x.is_cuda else n(x)
        return BaseModelOutputWithPast(last_hidden_state=x, past_key_values=tuple(new_pasts) if use_cache else None)


class PersimmonForCausalLM(PersimmonPreTrainedModel):
    def __init__(self, config: PersimmonConfig):
        super().__init__(config)
        self.model = PersimmonModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def get_decoder(self):
        return self.model

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, return_dict=True, **unused):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                         inputs_embeds=inputs_embeds, use_cache=use_cache)
        logits = OF.trainable_linear(self.lm_head, out.last_hidden_state)
        loss = None
        if labels is not None:
            lab = torch.full_like(labels, -100)
            lab[:, :-1] = labels[:, 1:]
            flat, lab = logits.view(-1, logits.size(-1)), lab.to(logits.device).view(-1)
            if flat.is_cuda and flat.dtype == torch.bfloat16 and flat.size(-1) % 4 == 0 and os.environ.get("OTTER_TORCH_CE") != "1":
                loss = OF.cross_entropy_bf16(flat, lab)
            else:
                loss = F.cross_entropy(flat.float(), lab, ignore_index=-100)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=out.past_key_values)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **generate_kwargs):
        """Text-only decoding (transformers 5.x no longer gives PreTrainedModel a generate()): otter_amd/generation.py over this model."""
        from .generation import generate_tokens

        use_cache = bool(generate_kwargs.pop("use_cache", getattr(self.config, "use_cache", True)))

        def step(ids, mask, past, beam_idx):
            if use_cache and past is not None:
                if beam_idx is not None:
                    past = tuple(tuple(t.index_select(0, beam_idx) for t in layer) for layer in past)
                out = self(input_ids=ids[:, -1:], attention_mask=mask, past_key_values=past, use_cache=True)
            else:
                out = self(input_ids=ids, attention_mask=mask, use_cache=use_cache)
            return out.logits[:, -1, :], (out.past_key_values if use_cache else None)

        return generate_tokens(step, input_ids, attention_mask, **generate_kwargs)


class FuyuPreTrainedModel(PreTrainedModel):
    config_class = FuyuConfig
    base_model_prefix = "fuyu"
    _no_split_modules = ["PersimmonDecoderLayer"]
    _supports_sdpa = True
    _supports_flash_attn = False
    _supports_flex_attn = False

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, 0.0, std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, 0.0, std)


class FuyuForCausalLM(FuyuPreTrainedModel):
    """modeling_fuyu.py:19-143.  `image_patches`: [B, n_patches, patch*patch*channels] (or a list of [1, n_i, ...] tensors as the
    reference's processor emits); `image_patches_indices` [B, S]: -1 for text positions, else the index of the patch whose
    embedding replaces the word embedding at that position."""

    def __init__(self, config: FuyuConfig):
        super().__init__(config)
        self.padding_idx = getattr(config, "pad_token_id", None)
        self.vocab_size = config.text_config.vocab_size
        self.language_model = PersimmonForCausalLM(config.text_config)
        self.vision_embed_tokens = nn.Linear(config.patch_size * config.patch_size * config.num_channels, config.text_config.hidden_size)
        self.post_init()

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.language_model.set_input_embeddings(value)

    def gather_continuous_embeddings(self, word_embeddings: torch.Tensor, continuous_embeddings: List[torch.Tensor],
                                     image_patch_input_indices: torch.Tensor) -> torch.Tensor:
        """modeling_fuyu.py:44-77: out[b, dst] = continuous[b][idx[b, dst]] wherever idx >= 0 (one HIP scatter on the GPU)."""
        if word_embeddings.shape[0] != len(continuous_embeddings):
            raise ValueError(f"Batch sizes must match! Got {len(continuous_embeddings)=} and {word_embeddings.shape[0]=}")
        B = word_embeddings.shape[0]
        idx = image_patch_input_indices
        n_rows = torch.tensor([c.shape[0] for c in continuous_embeddings], device=idx.device)
        n_idx = (idx >= 0).sum(dim=1)
        # one host synchronisation for both checks (the reference's per-sample loop has one per sample)
        bad_count, bad_range = (torch.stack([(n_idx > n_rows).any(), (idx.max(dim=1).values >= n_rows).any()]).tolist() if B else (False, False))
        if bad_count:
            b = int(torch.nonzero(n_idx > n_rows)[0])
            raise ValueError(f"Number of continuous embeddings {continuous_embeddings[b].shape=} does not match number of continuous "
                             f"token ids {int(n_idx[b])} in batch element {b}.")
        if bad_range:   # the reference's `continuous_embeddings[b][src]` raises IndexError for these
            b = int(torch.nonzero(idx.max(dim=1).values >= n_rows)[0])
            raise IndexError(f"image_patch_input_indices of batch element {b} reach {int(idx[b].max())}, but it has only "
                             f"{continuous_embeddings[b].shape[0]} continuous embeddings")
        if word_embeddings.is_cuda and all(c.shape[0] == continuous_embeddings[0].shape[0] for c in continuous_embeddings):
            return OF.scatter_patch_rows(word_embeddings, torch.stack(list(continuous_embeddings), 0), image_patch_input_indices)
        out = word_embeddings.clone()
        for b in range(B):
            dst = torch.nonzero(image_patch_input_indices[b] >= 0, as_tuple=True)[0]
            src = image_patch_input_indices[b][dst]
            out[b, dst] = continuous_embeddings[b][src].to(out.dtype)
        return out

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, image_patches=None,
                                      image_patches_indices=None, **kwargs):
        """modeling_fuyu.py:145-175 (the pinned transformers' hook): with a cache only the last token is fed and the patches are dropped."""
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"), "attention_mask": attention_mask,
                "image_patches": image_patches if past_key_values is None else None,
                "image_patches_indices": image_patches_indices if past_key_values is None else None}

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, image_patches=None, image_patches_indices: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, **generate_kwargs):
        """`model.generate(**processor_outputs, max_new_tokens=...)` as the reference's OtterHD inference calls it
        (pipeline/demos/demo_models.py:171; the pinned transformers mixes GenerationMixin into PreTrainedModel, 5.x no longer does).
        Decoding = otter_amd/generation.py (greedy / beam / sampling, the same restatement OtterForConditionalGeneration.generate uses).
        The image patches enter on the prompt pass only; with `use_cache` (default: the config's) later steps feed one token against
        the KV cache, without it the whole sequence is re-run with the patch positions of the prompt and -1 for the new tokens."""
        from .generation import generate_tokens

        use_cache = bool(generate_kwargs.pop("use_cache", getattr(self.config.text_config, "use_cache", True)))
        L0 = input_ids.shape[1]
        nb = int(generate_kwargs.get("num_beams", 1) or 1)
        patches, idx0 = image_patches, image_patches_indices
        if nb > 1 and patches is not None:          # every beam of a sample sees that sample's image
            patches = (patches.repeat_interleave(nb, dim=0) if torch.is_tensor(patches) else [p for p in patches for _ in range(nb)])
            idx0 = idx0.repeat_interleave(nb, dim=0)

        def step(ids, mask, past, beam_idx):
            if use_cache and past is not None:
                if beam_idx is not None:
                    past = tuple(tuple(t.index_select(0, beam_idx) for t in layer) for layer in past)
                out = self(input_ids=ids[:, -1:], attention_mask=mask, past_key_values=past, use_cache=True)
            else:
                idx = idx0
                if idx is not None and ids.shape[1] > L0:
                    idx = torch.cat([idx, idx.new_full((idx.shape[0], ids.shape[1] - L0), -1)], dim=1)
                out = self(input_ids=ids, image_patches=patches, image_patches_indices=idx, attention_mask=mask, use_cache=use_cache)
            return out.logits[:, -1, :], (out.past_key_values if use_cache else None)

        return generate_tokens(step, input_ids, attention_mask, **generate_kwargs)

    def forward(self, input_ids=None, labels=None, image_patches=None, image_patches_indices=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, use_cache=None, return_dict=True, **unused):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both decoder_input_ids and decoder_inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.language_model.get_input_embeddings()(input_ids)
            if image_patches is not None and past_key_values is None:
                w = self.vision_embed_tokens
                if torch.is_tensor(image_patches):
                    pe = list(w(image_patches.to(w.weight.dtype)))                  # one GEMM for the whole batch
                else:
                    pe = [w(p.to(w.weight.dtype)).squeeze(0) for p in image_patches]
                inputs_embeds = self.gather_continuous_embeddings(inputs_embeds, pe, image_patches_indices)
        return self.language_model(inputs_embeds=inputs_embeds, labels=labels, attention_mask=attention_mask, position_ids=position_ids,
                                   past_key_values=past_key_values, use_cache=use_cache)
