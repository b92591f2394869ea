"""OtterConfig: composition of a CLIP vision config and a text (MPT / LLaMA) config.

Mirror of the reference's src/otter_ai/models/otter/configuration_otter.py:15-97 (same JSON schema, so the reference's
`Otter-MPT7B-config.json` and published checkpoints' config.json load unchanged)."""
from __future__ import annotations

import copy

from transformers import PretrainedConfig
from transformers.models.auto import CONFIG_MAPPING
from transformers.models.clip import CLIPVisionConfig

from .mpt import MPTConfig


class OtterConfig(PretrainedConfig):
    model_type = "otter"
    is_composition = True
    has_no_defaults_at_init = True

    def __init__(self, vision_config=None, text_config=None, cross_attn_every_n_layers: int = 4,
                 use_media_placement_augmentation: bool = True, **kwargs):
        super().__init__(**kwargs)
        vision_config = {} if vision_config is None else vision_config
        text_config = {} if text_config is None else text_config
        if isinstance(vision_config, PretrainedConfig):
            vision_config = vision_config.to_dict()
        if isinstance(text_config, PretrainedConfig):
            text_config = text_config.to_dict()
        vision_config = dict(vision_config)
        text_config = dict(text_config)
        vision_config.pop("model_type", None)
        self.vision_config = CLIPVisionConfig(**vision_config)
        arch = (text_config.get("architectures") or [None])[0]
        if arch == "MPTForCausalLM" or (arch is None and text_config.get("model_type", "mpt") == "mpt"):
            text_config.pop("model_type", None)
            text_config.setdefault("architectures", ["MPTForCausalLM"])
            self.text_config = MPTConfig(**text_config)
        elif arch == "LlamaForCausalLM" or text_config.get("model_type") == "llama":
            text_config.pop("model_type", None)
            self.text_config = CONFIG_MAPPING["llama"](**text_config)
        else:
            raise NotImplementedError(
                f"text architecture {arch!r}: otter_amd hosts the fusion path on MPTForCausalLM and LlamaForCausalLM only "
                "(MosaicGPT / RWForCausalLM backbones are outside BASELINE.json's configs)")
        self.cross_attn_every_n_layers = cross_attn_every_n_layers
        self.use_media_placement_augmentation = use_media_placement_augmentation

    def to_dict(self):
        output = copy.deepcopy({k: v for k, v in self.__dict__.items() if k not in ("vision_config", "text_config")})
        output["vision_config"] = self.vision_config.to_dict()
        output["text_config"] = self.text_config.to_dict()
        output["model_type"] = self.__class__.model_type
        output["cross_attn_every_n_layers"] = self.cross_attn_every_n_layers
        output["use_media_placement_augmentation"] = self.use_media_placement_augmentation
        return output
