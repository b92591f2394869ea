"""Frozen producer of the fusion path: CLIP ViT vision tower -> last_hidden_state [N, 1+v, D].

The reference instantiates transformers==4.35.1 `CLIPVisionModel` (modeling_otter.py:768,991; in-repo restatement at
/root/reference/xformers_model/clip.py:50-199,393-446).  The tower is frozen and inference-only in the Otter recipe
("next" tier).  Round 2: its no_grad inference path runs q|k|v as one GEMM, the attention core on
the head_dim-64 MFMA kernels of csrc/attn_mfma.hip, quick-GELU and the residual-add + LayerNorm pairs as single HIP passes
(`_forward_fused`); the GEMMs stay on hipBLASLt.  The class also pins the *checkpoint contract*: state-dict keys are `vision_model.embeddings.*`, `vision_model.pre_layrnorm.*`,
`vision_model.encoder.layers.{i}.*`, `vision_model.post_layernorm.*` exactly as the pinned transformers version spells
them (transformers 5.x renamed them, which would break every published Otter checkpoint).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.hidden_size
        self.class_embedding = nn.Parameter(torch.randn(D))
        self.patch_embedding = nn.Conv2d(cfg.num_channels, D, kernel_size=cfg.patch_size, stride=cfg.patch_size, bias=False)
        self.num_positions = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.position_embedding = nn.Embedding(self.num_positions, D)

    def forward(self, pixel_values):
        N = pixel_values.shape[0]
        w = self.patch_embedding.weight
        # stride == kernel: the patch "convolution" is a plain GEMM over unfolded patches (MIOpen picked a naive conv kernel
        # for this shape: 1.2 ms per call in the round-1 profile)
        P = w.shape[-1]
        C, Hh, Ww = pixel_values.shape[1:]
        pat = pixel_values.to(w.dtype).reshape(N, C, Hh // P, P, Ww // P, P).permute(0, 2, 4, 1, 3, 5).reshape(N, -1, C * P * P)
        pe = F.linear(pat, w.reshape(w.shape[0], -1))
        cls = self.class_embedding.to(pe.dtype).expand(N, 1, -1)
        return torch.cat([cls, pe], dim=1) + self.position_embedding.weight.to(pe.dtype)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.hidden_size
        self.num_heads = cfg.num_attention_heads
        self.q_proj, self.k_proj = nn.Linear(D, D), nn.Linear(D, D)
        self.v_proj, self.out_proj = nn.Linear(D, D), nn.Linear(D, D)

    def forward(self, x):
        N, S, D = x.shape
        H = self.num_heads
        q = self.q_proj(x).view(N, S, H, D // H).transpose(1, 2)
        k = self.k_proj(x).view(N, S, H, D // H).transpose(1, 2)
        v = self.v_proj(x).view(N, S, H, D // H).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.out_proj(o.transpose(1, 2).reshape(N, S, D))


class _MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.hidden_size, cfg.intermediate_size)
        self.fc2 = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.act = cfg.hidden_act

    def forward(self, x):
        x = self.fc1(x)
        if self.act == "quick_gelu":
            x = x * torch.sigmoid(1.702 * x)
        elif self.act == "gelu":
            x = F.gelu(x)
        else:
            raise NotImplementedError(self.act)
        return self.fc2(x)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = _Attention(cfg)
        self.layer_norm1 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(cfg)
        self.layer_norm2 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _VisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)  # (sic) the HF attribute name
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, pixel_values):
        x = self.pre_layrnorm(self.embeddings(pixel_values))
        if x.is_cuda and not torch.is_grad_enabled():
            return self._forward_fused(x)
        for layer in self.encoder.layers:
            x = layer(x)
        return x  # last_hidden_state: post_layernorm applies to the pooled CLS only (clip.py:434-436)

    def _forward_fused(self, x):
        """Inference path of the frozen tower (the recipe runs it under no_grad, modeling_otter.py:990-991): every residual add
        is folded into the following LayerNorm pass (otter_add_layernorm_fwd), 2 launches per layer instead of 4.  Same
        values: the sum keeps the residual stream's dtype, the normalised rows are rounded once to the GEMM operand dtype."""
        from . import functional as OF
        from . import ops

        shp = x.shape
        cd = OF.compute_dtype_for(x)
        x2 = x.reshape(-1, shp[-1]).contiguous()
        delta = None
        for layer in self.encoder.layers:
            n1, n2 = layer.layer_norm1, layer.layer_norm2
            if delta is None:
                a = ops.layernorm_fwd(x2, n1.weight, n1.bias, cd, n1.eps, need_stats=False)[0]
            else:
                x2, a, _, _ = ops.add_layernorm_fwd(x2, delta, n1.weight, n1.bias, cd, n1.eps, need_stats=False)
            b = self._attn_fused(layer.self_attn, a, shp, cd)
            x2, m, _, _ = ops.add_layernorm_fwd(x2, b.contiguous(), n2.weight, n2.bias, cd, n2.eps, need_stats=False)
            delta = self._mlp_fused(layer.mlp, m, cd)
        return (x2 + delta).view(shp)

    @staticmethod
    def _attn_fused(att, a2, shp, cd):
        """q|k|v as ONE GEMM against the concatenated frozen weights and biases, then the HIP attention core (head_dim 64 MFMA
        kernels of csrc/attn_mfma.hip, unmasked, 257 keys) reading q, k, v as strided views of that buffer: no chunk / transpose
        copies, no torch SDPA."""
        from . import ops
        from ._capi import MASK_NONE

        N, S, D = shp
        H = att.num_heads
        if D // H != 64 or cd != torch.bfloat16 or os.environ.get("OTTER_CLIP_SDPA") == "1":   # env: A/B switch back to torch SDPA
            return att(a2.view(shp)).reshape(-1, D)
        key = tuple((l.weight._version, l.weight.data_ptr(), l.bias._version) for l in (att.q_proj, att.k_proj, att.v_proj)) + (cd,)
        if getattr(att, "_qkv_key", None) != key:
            att._qkv_w = torch.cat([l.weight.detach().to(cd) for l in (att.q_proj, att.k_proj, att.v_proj)], 0).contiguous()
            att._qkv_b = torch.cat([l.bias.detach().to(cd) for l in (att.q_proj, att.k_proj, att.v_proj)], 0).contiguous()
            att._qkv_key = key
        qkv = F.linear(a2.view(N, S, D), att._qkv_w, att._qkv_b)                   # [N, S, 3*D]
        o, _ = ops.attn_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, None, S, MASK_NONE, (D // H) ** -0.5, need_lse=False)
        return F.linear(o, att.out_proj.weight.to(cd), att.out_proj.bias.to(cd)).reshape(-1, D)

    @staticmethod
    def _mlp_fused(mlp, m2, cd):
        from . import ops

        h = F.linear(m2, mlp.fc1.weight.to(cd), mlp.fc1.bias.to(cd))
        if mlp.act == "quick_gelu" and h.is_contiguous() and h.numel() % 8 == 0:
            ops.quick_gelu_(h)                                                       # one in-place HIP pass
        elif mlp.act == "quick_gelu":
            h = h * torch.sigmoid(1.702 * h)
        elif mlp.act == "gelu":
            h = F.gelu(h)
        else:
            raise NotImplementedError(mlp.act)
        return F.linear(h, mlp.fc2.weight.to(cd), mlp.fc2.bias.to(cd)).contiguous()


class CLIPVisionModel(nn.Module):
    """`vision_encoder(x)[0]` is last_hidden_state, as the reference uses it (modeling_otter.py:991)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.vision_model = _VisionTransformer(config)
        self.output_tokens = False

    def forward(self, pixel_values):
        h = self.vision_model(pixel_values)
        pooled = self.vision_model.post_layernorm(h[:, 0, :])
        return (h, pooled)
