"""Drop-in mirror of the reference's `src/otter_ai/models/otter/modeling_otter.py` for the fusion hot path.

Same class names, constructor arguments, parameter names (state-dict keys) and forward()/generate() signatures as
the reference, so `pipeline/train/instruction_following.py` can import this module in place of
the original.  The arithmetic of every class below runs in libotter_hip.so through the autograd
functions of `otter_amd.functional`; the nn.LayerNorm / nn.Linear children are parameter containers only (their own
forward() is never called), kept so that checkpoints load/save with the reference's key names.

Reference lines: OtterPerceiverBlock :129-184, OtterPerceiverResampler :187-235, OtterMaskedCrossAttention :238-340,
OtterGatedCrossAttentionBlock :343-395, OtterLayer :398-442, OtterLMMixin :445-520, OtterForConditionalGeneration :739-1042.
"""
from __future__ import annotations

import os
import random
import warnings
from typing import List, Optional

import torch
import torch.nn as nn
from transformers import PreTrainedModel
from transformers.modeling_outputs import CausalLMOutputWithPast

from . import functional as OF
from . import ops
from ._capi import MASK_EQ, MASK_GE, MASK_NONE, RowMap
from .clip import CLIPVisionModel
from .configuration_otter import OtterConfig
from .mpt import MPTForCausalLM

__KNOWN_DECODER_LAYERS_ATTR_NAMES = {"llama": "model.layers", "MPTForCausalLM": "transformer.blocks"}


def master_print(*args, **kwargs):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        if dist.get_rank() == 0:
            print(*args, **kwargs)
    else:
        print(*args, **kwargs)


def _infer_decoder_layers_attr_name(model: nn.Module):
    for k, v in __KNOWN_DECODER_LAYERS_ATTR_NAMES.items():
        if k.lower() in model.__class__.__name__.lower():
            return v
    raise ValueError("We require the attribute name for the nn.ModuleList in the decoder storing the transformer block layers.")


def extend_instance(obj, mixin):
    """Apply mixins to a class instance after creation (modeling_otter.py:94-98)."""
    base_cls = obj.__class__
    obj.__class__ = type(base_cls.__name__, (mixin, base_cls), {})


def getattr_recursive(obj, att):
    if att == "":
        return obj
    i = att.find(".")
    return getattr(obj, att) if i < 0 else getattr_recursive(getattr(obj, att[:i]), att[i + 1:])


def setattr_recursive(obj, att, val):
    if "." in att:
        obj = getattr_recursive(obj, ".".join(att.split(".")[:-1]))
    setattr(obj, att.split(".")[-1], val)


def exists(val):
    return val is not None


# ======================================================================================================================
# perceiver resampler
# ======================================================================================================================


class OtterPerceiverBlock(nn.Module):
    def __init__(self, *, dim: int, dim_head: int = 64, heads: int = 8, mult: int = 4):
        super().__init__()
        if dim_head != OF.HEAD_DIM:
            raise NotImplementedError("otter_amd attention kernels are built for dim_head=64 (the reference's only value)")
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner_dim = dim_head * heads
        ff_dim = dim * mult
        self.norm_media = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.feed_forward = nn.ModuleList([nn.LayerNorm(dim), nn.Linear(dim, ff_dim, bias=False), nn.GELU(),
                                           nn.Linear(ff_dim, dim, bias=False)])

    def forward(self, x: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
        """x (b, T, n1, D) image features; latents (b, T, n2, D)."""
        b, T, n1, D = x.shape
        n2 = latents.shape[2]
        ff = self.feed_forward
        y = OF.PerceiverBlockFn.apply(x.reshape(b * T, n1, D), latents.reshape(b * T, n2, D), self.heads, self.norm_media.eps,
                                      self.norm_media.weight, self.norm_media.bias, self.norm_latents.weight,
                                      self.norm_latents.bias, self.to_q.weight, self.to_kv.weight, self.to_out.weight,
                                      ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight)
        return y.view(b, T, n2, D)


class _BroadcastEmbAddFn(torch.autograd.Function):
    """x [outer, F, inner, D] + emb[:F] (broadcast over outer and inner): frame_embs / media_time_embs
    (modeling_otter.py:224-229).  Forward = one in-place row add on a copy; backward = column sums per embedding row."""

    @staticmethod
    def forward(ctx, x4, emb):
        outer, F, inner, D = x4.shape
        y = x4.contiguous().clone()
        e = emb.detach()[:F].contiguous().float()
        ops.K.check(ops.K.lib().otter_add_frame_embs(y.data_ptr(), ops.K.dt(y), e.data_ptr(), outer, F, inner, D, ops.K.stream()),
                    "add_frame_embs")
        ctx.shape = (outer, F, inner, D)
        ctx.emb_shape = emb.shape
        ctx.emb_dtype = emb.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        outer, F, inner, D = ctx.shape
        demb = None
        if ctx.needs_input_grad[1]:
            dy2 = dy.contiguous().view(outer * F * inner, D)
            demb = torch.zeros(ctx.emb_shape, dtype=torch.float32, device=dy.device)
            flat = demb.view(-1, D)
            for f in range(F):
                ops.colsum(dy2, RowMap(inner, F * inner, f * inner), outer * inner, out=flat[f])
            demb = demb.to(ctx.emb_dtype)
        return (dy if ctx.needs_input_grad[0] else None), demb


class OtterPerceiverResampler(nn.Module):
    def __init__(self, *, dim: int, depth: int = 6, dim_head: int = 64, heads: int = 8, num_latents: int = 64,
                 max_num_media: Optional[int] = None, max_num_frames: Optional[int] = None, ff_mult: int = 4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.frame_embs = nn.Parameter(torch.randn(max_num_frames, dim)) if exists(max_num_frames) else None
        self.media_time_embs = nn.Parameter(torch.randn(max_num_media, 1, dim)) if exists(max_num_media) else None
        self.layers = nn.ModuleList([OtterPerceiverBlock(dim=dim, dim_head=dim_head, heads=heads, mult=ff_mult)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (b, T, F, v, D) -> (b, T, n, D)."""
        b, T, F, v, D = x.shape
        if exists(self.frame_embs):
            x = _BroadcastEmbAddFn.apply(x.reshape(b * T, F, v, D), self.frame_embs).view(b, T, F, v, D)
        x = x.reshape(b, T, F * v, D)
        if exists(self.media_time_embs):
            x = _BroadcastEmbAddFn.apply(x.reshape(b, T, F * v, D), self.media_time_embs).view(b, T, F * v, D)
        latents = OF.ExpandLatentsFn.apply(self.latents, b * T).view(b, T, *self.latents.shape)
        for block in self.layers:
            latents = block(x, latents)
        return OF.layer_norm(latents, self.norm.weight, self.norm.bias, self.norm.eps, latents.dtype)


# ======================================================================================================================
# masked / gated cross attention
# ======================================================================================================================


def _mask_mode(media_locations, only_attend_immediate_media):
    if media_locations is None:
        return MASK_NONE
    return MASK_EQ if only_attend_immediate_media else MASK_GE


class OtterMaskedCrossAttention(nn.Module):
    def __init__(self, *, dim: int, dim_visual: int, dim_head: int = 64, heads: int = 8, only_attend_immediate_media: bool = True):
        super().__init__()
        if dim_head != OF.HEAD_DIM:
            raise NotImplementedError("otter_amd attention kernels are built for dim_head=64 (the reference's only value)")
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner_dim = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim_visual, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.only_attend_immediate_media = only_attend_immediate_media

    def forward(self, x, media, media_locations=None, attend_previous: bool = True, text_time=None):
        """x (B, T_txt, D); media (B, T_img, n, D_img); media_locations bool (B, T_txt).  Stand-alone use of the attention
        sub-module (inside the gated block the fused function is used instead)."""
        mode = _mask_mode(media_locations, self.only_attend_immediate_media)
        tt = text_time
        if mode != MASK_NONE and tt is None:
            tt = ops.text_time(media_locations, attend_previous)
        return OF.masked_cross_attention(x, media, tt, mode, self.heads, self.norm.eps, self.norm.weight, self.norm.bias,
                                         self.to_q.weight, self.to_kv.weight, self.to_out.weight)


class OtterGatedCrossAttentionBlock(nn.Module):
    accepts_deferred = True      # forward(..., deferred=): see OtterLayer.forward / otter_amd.mpt._gated_takes_deferred

    def __init__(self, *, dim: int, dim_visual: int, dim_head: int = 64, heads: int = 8, ff_mult: int = 4,
                 only_attend_immediate_media: bool = True):
        super().__init__()
        self.attn = OtterMaskedCrossAttention(dim=dim, dim_visual=dim_visual, dim_head=dim_head, heads=heads,
                                              only_attend_immediate_media=only_attend_immediate_media)
        self.attn_gate = nn.Parameter(torch.tensor([0.0]))
        self.feed_forward = nn.ModuleList([nn.LayerNorm(dim), nn.Linear(dim, dim * ff_mult, bias=False), nn.GELU(),
                                           nn.Linear(dim * ff_mult, dim, bias=False)])
        self.ff_gate = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, media, media_locations=None, attend_previous: bool = True, text_time=None, deferred=None):
        """`deferred` (otter_amd extension): an addend of the residual stream that has not been added yet (the MPT host hands each
        block's FFN output to the NEXT LayerNorm pass); the block computes on x + deferred, the add fused into its first LayerNorm."""
        a, ff = self.attn, self.feed_forward
        mode = _mask_mode(media_locations, a.only_attend_immediate_media)
        tt = text_time
        if mode != MASK_NONE and tt is None:
            tt = ops.text_time(media_locations, attend_previous)
        return OF.GatedCrossAttentionFn.apply(x, media, tt, mode, a.heads, a.norm.eps, a.norm.weight, a.norm.bias, a.to_q.weight,
                                              a.to_kv.weight, a.to_out.weight, self.attn_gate, ff[0].weight, ff[0].bias,
                                              ff[1].weight, ff[3].weight, self.ff_gate, deferred)


class OtterLayer(nn.Module):
    def __init__(self, gated_cross_attn_layer: nn.Module, decoder_layer: nn.Module):
        super().__init__()
        self.gated_cross_attn_layer = gated_cross_attn_layer
        self.decoder_layer = decoder_layer
        self.vis_x = None
        self.media_locations = None
        self.attend_previous = None
        self.text_time = None

    def is_conditioned(self) -> bool:
        return self.vis_x is not None

    def condition_vis_x(self, vis_x) -> None:
        self.vis_x = vis_x

    def condition_media_locations(self, media_locations) -> None:
        self.media_locations = media_locations
        self.text_time = None

    def condition_attend_previous(self, attend_previous) -> None:
        self.attend_previous = attend_previous

    def condition_text_time(self, text_time) -> None:
        """otter_amd extension: the media-time scan is shared by all layers instead of being recomputed per layer."""
        self.text_time = text_time

    def forward(self, lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **decoder_layer_kwargs):
        if self.gated_cross_attn_layer is None:
            return self.decoder_layer(lang_x, attention_mask=attention_mask, **decoder_layer_kwargs)
        if self.vis_x is None:
            raise ValueError("vis_x must be conditioned before forward pass")
        if self.media_locations is None:
            raise ValueError("media_locations must be conditioned before forward pass")
        # otter_amd's MPT host may hand over the previous layer's FFN output un-added (`deferred`, mpt.py): the gated block fuses the add
        # into its first LayerNorm; the decoder layer then starts from the block's (materialised) output, which it both normalises and
        # carries on (`fork_input`: one LayerNorm-backward pass instead of an extra gradient add)
        deferred = decoder_layer_kwargs.pop("deferred", None)
        gkw = {"deferred": deferred} if deferred is not None else {}
        lang_x = self.gated_cross_attn_layer(lang_x, self.vis_x, media_locations=self.media_locations,
                                             attend_previous=self.attend_previous, text_time=self.text_time, **gkw)
        if "defer_out" in decoder_layer_kwargs:       # (only otter_amd's MPTBlock takes the extension keywords)
            decoder_layer_kwargs["fork_input"] = True
        return self.decoder_layer(lang_x, attention_mask=attention_mask, **decoder_layer_kwargs)


class OtterLMMixin(nn.Module):
    """Mixin to add cross-attention layers to a language model (modeling_otter.py:445-520)."""

    def set_decoder_layers_attr_name(self, decoder_layers_attr_name):
        self.decoder_layers_attr_name = decoder_layers_attr_name

    def _get_decoder_layers(self):
        return getattr_recursive(self, self.decoder_layers_attr_name)

    def _set_decoder_layers(self, value):
        setattr_recursive(self, self.decoder_layers_attr_name, value)

    def init_otter(self, media_token_id: int, vis_hidden_size: int, cross_attn_every_n_layers: int,
                   use_media_placement_augmentation: bool):
        gated = nn.ModuleList([
            OtterGatedCrossAttentionBlock(dim=self.config.hidden_size, dim_visual=vis_hidden_size)
            if (layer_idx + 1) % cross_attn_every_n_layers == 0 else None
            for layer_idx, _ in enumerate(self._get_decoder_layers())
        ])
        self._set_decoder_layers(nn.ModuleList([OtterLayer(g, d) for g, d in zip(gated, self._get_decoder_layers())]))
        self.media_token_id = media_token_id
        self.use_media_placement_augmentation = use_media_placement_augmentation
        self.initialized_otter = True

    def forward(self, *input, **kwargs):
        """Condition the Otter layers on the media locations before forward()."""
        if not getattr(self, "initialized_otter", False):
            raise ValueError("Otter layers are not initialized. Please call `init_otter` first.")
        input_ids = kwargs["input_ids"] if "input_ids" in kwargs else input[0]
        media_locations = input_ids == self.media_token_id
        attend_previous = (random.random() < 0.5) if self.use_media_placement_augmentation else True
        tt = ops.text_time(media_locations, attend_previous)
        for layer in self._get_decoder_layers():
            layer.condition_media_locations(media_locations)
            layer.condition_attend_previous(attend_previous)
            layer.condition_text_time(tt)
        return super().forward(*input, **kwargs)

    def is_conditioned(self) -> bool:
        return all(l.is_conditioned() for l in self._get_decoder_layers())

    def clear_conditioned_layers(self) -> None:
        for layer in self._get_decoder_layers():
            layer.condition_vis_x(None)
            layer.condition_media_locations(None)
            layer.condition_attend_previous(None)


# ======================================================================================================================
# model
# ======================================================================================================================


class OtterStubTokenizer:
    """Used only when the MPT tokenizer files are not available locally (no network): carries the four special-token ids
    the model needs.  For the 50432-row MPT-7B vocabulary they are the ids the real tokenizer assigns after
    `add_special_tokens` (50277..50280); for toy vocabularies they are the last four rows."""

    eos_token = "<|endoftext|>"
    pad_token = "<PAD>"

    def __init__(self, vocab_size: int):
        base = 50277 if vocab_size >= 50281 else vocab_size - 4
        self.vocab_size = vocab_size
        self.special = {"<|endofchunk|>": base, "<image>": base + 1, "<answer>": base + 2, "<PAD>": base + 3,
                        "<|endoftext|>": 0}

    def add_special_tokens(self, d):
        return 0

    def encode(self, s):
        return [self.special[s]]

    def __call__(self, s, add_special_tokens=False, **kw):
        return {"input_ids": self.encode(s)}

    def __len__(self):
        return self.vocab_size


def _load_tokenizer(name: str, vocab_size: int):
    """modeling_otter.py:750-757.  The stub is used ONLY when the tokenizer files are not on this machine (OSError from
    `local_files_only=True`: benches and tests run without network) or on explicit request (OTTER_STUB_TOKENIZER=1) -- and
    says so; any other failure (a broken tokenizers install, a corrupt file) propagates, because the stub's hard-coded
    special-token ids would silently change <image> / <|endofchunk|> (and alias real ids of a LLaMA vocabulary)."""
    import os
    import warnings

    if os.environ.get("OTTER_STUB_TOKENIZER") != "1":
        from transformers import AutoTokenizer

        try:
            tok = AutoTokenizer.from_pretrained(name, local_files_only=True)
        except OSError as e:
            warnings.warn("otter_amd: tokenizer files for %r not found locally (%s); using OtterStubTokenizer "
                          "(special-token ids only)" % (name, type(e).__name__), stacklevel=2)
        else:
            tok.add_special_tokens({"additional_special_tokens": ["<|endofchunk|>", "<image>", "<answer>"]})
            if tok.pad_token is None:
                tok.add_special_tokens({"pad_token": "<PAD>"})
            return tok
    return OtterStubTokenizer(vocab_size)


def _use_hip_rmsnorm(lang_encoder: nn.Module) -> None:
    """Config C4 (LLaMA host, third-party transformers class as in the reference): route every LlamaRMSNorm through the
    HIP RMSNorm kernel (same parameters, same state-dict keys; only `forward` is rebound)."""
    import types

    def _fwd(self, hidden_states):
        eps = getattr(self, "variance_epsilon", getattr(self, "eps", 1e-6))
        return OF.rms_norm(hidden_states, self.weight, eps)

    for mod in lang_encoder.modules():
        if mod.__class__.__name__ == "LlamaRMSNorm":
            mod.forward = types.MethodType(_fwd, mod)


class OtterPreTrainedModel(PreTrainedModel):
    config_class = OtterConfig
    base_model_prefix = "otter"
    supports_gradient_checkpointing = False
    _no_split_modules = ["OtterPerceiverBlock", "OtterLayer", "CLIPVisionModel"]

    def _init_weights(self, module):
        """Otter requires no specific initialization."""
        return


class OtterForConditionalGeneration(OtterPreTrainedModel):
    config_class = OtterConfig

    def __init__(self, config: OtterConfig):
        super().__init__(config)
        tc = config.text_config
        arch = (getattr(tc, "architectures", None) or ["MPTForCausalLM"])[0]
        if arch == "MPTForCausalLM":
            text_tokenizer = _load_tokenizer("mosaicml/mpt-7b-instruct", tc.vocab_size)
            lang_encoder = MPTForCausalLM(tc)
        elif arch == "LlamaForCausalLM":
            if os.environ.get("OTTER_HF_LLAMA") == "1":   # A/B switch: the third-party class the reference uses, HIP RMSNorm only
                from transformers import LlamaForCausalLM

                lang_encoder = LlamaForCausalLM(tc)
                _use_hip_rmsnorm(lang_encoder)
            else:
                from .llama import LlamaForCausalLM       # MI355X-native host with the same class surface and state-dict keys

                lang_encoder = LlamaForCausalLM(tc)
            text_tokenizer = _load_tokenizer(getattr(tc, "_name_or_path", "") or "llama", tc.vocab_size)
        else:
            raise NotImplementedError(arch)
        vision_encoder = CLIPVisionModel(config.vision_config)
        self.text_tokenizer = text_tokenizer
        self.eoc_token_id = text_tokenizer.encode("<|endofchunk|>")[-1]
        self.media_token_id = text_tokenizer.encode("<image>")[-1]

        extend_instance(lang_encoder, OtterLMMixin)
        lang_encoder.set_decoder_layers_attr_name(_infer_decoder_layers_attr_name(lang_encoder))
        self.lang_encoder = lang_encoder
        self.cross_attn_every_n_layers = config.cross_attn_every_n_layers
        self.use_media_placement_augmentation = False  # strictly false for Otter (modeling_otter.py:786)
        self.max_num_frames = config.max_num_frames if hasattr(config, "max_num_frames") else None
        vision_encoder.output_tokens = True
        self.vision_encoder = vision_encoder
        self.vis_dim = 1024
        self.perceiver = OtterPerceiverResampler(dim=self.vis_dim, max_num_frames=self.max_num_frames)
        self.lang_encoder.init_otter(media_token_id=self.media_token_id, vis_hidden_size=self.vis_dim,
                                     cross_attn_every_n_layers=self.cross_attn_every_n_layers,
                                     use_media_placement_augmentation=self.use_media_placement_augmentation)
        if "lora_config" in config.__dict__:
            # modeling_otter.py:808-829: low-rank adapters on the frozen decoder's Wqkv (MPT) / q_proj + v_proj (LLaMA); the wrapper
            # nesting, parameter names and class rename of peft are restated in otter_amd/lora.py (peft itself is not installed here)
            from .lora import get_lora_model

            master_print(f"Using LoRA with config:{config.lora_config}")
            self.lang_encoder = get_lora_model(self.lang_encoder, config.lora_config, arch)
            self.lang_encoder.master_print_trainable_parameters()
        self.post_init()

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        model = super().from_pretrained(*args, **kwargs)
        # transformers >= 5 builds the model on the meta device and materialises NEW Parameter objects while loading, which
        # drops the requires_grad flags set by init_weights() in __init__ (the pinned 4.35.1 loaded in place): re-apply.
        model.init_weights()
        return model

    # ---- accessors used by pipeline/train/instruction_following.py ----
    def get_input_embeddings(self) -> nn.Module:
        return self.lang_encoder.get_input_embeddings()

    def set_input_embeddings(self, new_embeddings):
        self.lang_encoder.set_input_embeddings(new_embeddings)

    def get_output_embeddings(self) -> nn.Module:
        return self.lang_encoder.get_output_embeddings()

    def set_output_embeddings(self, new_embeddings):
        self.lang_encoder.set_output_embeddings(new_embeddings)

    def get_image_encoder(self) -> nn.Module:
        return self.vision_encoder

    def get_lang_encoder(self) -> nn.Module:
        return self.lang_encoder

    def init_weights(self):
        """Freeze everything except gated cross-attention layers, the perceiver and the input embeddings
        (modeling_otter.py:851-915)."""
        cfg = self.config.__dict__
        if not cfg.get("train_full_model", False):
            for p in self.parameters():
                p.requires_grad = False
        if cfg.get("train_vision_encoder", False):
            for p in self.vision_encoder.parameters():
                p.requires_grad = True
        if cfg.get("train_lang_encoder", False):
            for p in self.lang_encoder.parameters():
                p.requires_grad = True
        if "lora_config" in cfg:                      # modeling_otter.py:889-894
            for name, p in self.lang_encoder.named_parameters():
                if "lora" in name:
                    p.requires_grad = True
        for name, p in self.lang_encoder.named_parameters():
            if "gated_cross_attn_layer" in name:
                p.requires_grad = True
        for name, p in self.named_parameters():
            if "perceiver" in name:
                p.requires_grad = True
        self.lang_encoder.get_input_embeddings().requires_grad_(True)
        if "LlamaForCausalLM" in self.lang_encoder.__class__.__name__:
            self.lang_encoder.lm_head.requires_grad_(True)

    def forward(self, vision_x: torch.Tensor, lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, use_cached_vision_x: bool = False, clear_conditioned_layers: bool = True,
                past_key_values: Optional[List[torch.FloatTensor]] = None, use_cache: bool = False,
                **kwargs) -> CausalLMOutputWithPast:
        assert (vision_x is not None) or use_cached_vision_x, "Must provide either vision_x or use_cached_vision_x to True."
        if use_cached_vision_x:
            assert vision_x is None, "Expect vision_x to be None when use_cached_vision_x is True."
            assert self.lang_encoder.is_conditioned()
        else:
            self._encode_vision_x(vision_x=vision_x)
        output = self.lang_encoder(input_ids=lang_x, attention_mask=attention_mask, labels=labels,
                                   past_key_values=past_key_values, use_cache=use_cache, **kwargs)
        if clear_conditioned_layers:
            self.lang_encoder.clear_conditioned_layers()
        return output

    def _encode_vision_x(self, vision_x: torch.Tensor):
        """(b, T_img, F, C, H, W) -> CLIP tokens (CLS dropped) -> perceiver -> condition every decoder layer."""
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        b, T, F = vision_x.shape[:3]
        flat = vision_x.reshape((b * T * F,) + tuple(vision_x.shape[3:]))
        with torch.no_grad() if not any(p.requires_grad for p in self.vision_encoder.parameters()) else _nullctx():
            feats = self.vision_encoder(flat)[0][:, 1:, :]
        feats = feats.reshape(b, T, F, feats.shape[1], feats.shape[2])
        vis = self.perceiver(feats)
        for layer in self.lang_encoder._get_decoder_layers():
            layer.condition_vis_x(vis)

    @torch.no_grad()
    def generate(self, vision_x: torch.Tensor, lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                 **generate_kwargs):
        """The reference's call contract (modeling_otter.py:999-1042): encode the vision input once (every beam of a sample is
        conditioned on that sample's media, the reference's `vision_x.repeat_interleave(num_beams)`), decode with
        eos_token_id = <|endofchunk|> unless the caller overrides it, clear the conditioning, return lang_x with the new tokens
        appended.  The decoding itself -- greedy, beam search with `no_repeat_ngram_size` / `bad_words_ids` /
        `length_penalty` / `min_new_tokens`, temperature / top-k / top-p sampling: what the reference's demos, benchmark
        wrappers and serving code pass -- is otter_amd/generation.py (the pinned transformers' algorithm, restated).
        `use_cache` selects between the two decode modes of SURVEY.md section 3.2 (default: the LM config's use_cache, False
        for OTTER-MPT7B)."""
        from .generation import generate_tokens

        num_beams = int(generate_kwargs.get("num_beams", 1) or 1)
        use_cache = bool(generate_kwargs.pop("use_cache", getattr(self.lang_encoder.config, "use_cache", False)))
        generate_kwargs.setdefault("eos_token_id", self.eoc_token_id)
        self._encode_vision_x(vision_x=vision_x)
        if num_beams > 1:   # perceiver output repeated per beam: same conditioning as encoding the repeated frames, 1/num_beams the work
            for layer in self.lang_encoder._get_decoder_layers():
                if layer.vis_x is not None:
                    layer.condition_vis_x(layer.vis_x.repeat_interleave(num_beams, dim=0))
        lm = self.lang_encoder

        def step(ids, mask, past, beam_idx):
            if use_cache and past is not None:
                if beam_idx is not None:
                    past = [tuple(t.index_select(0, beam_idx) for t in layer) for layer in past]   # a list: the MPT host fills it in place
                out = lm(input_ids=ids[:, -1:], attention_mask=mask, past_key_values=past, use_cache=True)
            else:
                out = lm(input_ids=ids, attention_mask=mask, use_cache=use_cache)
            return out.logits[:, -1, :], (out.past_key_values if use_cache else None)

        try:
            return generate_tokens(step, lang_x, attention_mask, **generate_kwargs)
        finally:
            self.lang_encoder.clear_conditioned_layers()


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# the reference exposes the same architecture under these names as well (flamingo/modeling_flamingo.py:696; otter/:539)
OtterModel = OtterForConditionalGeneration
FlamingoForConditionalGeneration = OtterForConditionalGeneration
