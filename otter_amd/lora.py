"""LoRA adapters on the frozen decoder: the `lora_config` branch of OtterForConditionalGeneration (modeling_otter.py:808-829).

The reference hands the language model to `peft.get_peft_model(lang_encoder, LoraConfig(r, lora_alpha, lora_dropout, task_type=CAUSAL_LM,
target_modules=...))` with target modules `["Wqkv"]` for MPT and `["q_proj", "v_proj"]` for LLaMA / OPT / GPT-J (`:811-819`), then renames
the wrapper's class to `<Architecture>LoRA` (`:829`) and unfreezes every parameter whose name contains "lora" (`:889-894`).  `peft` is a
third-party dependency (`requirements.txt:28`, `peft>=0.4.0`, unpinned) that is not installed here; what the training loop, the
checkpoint code and the optimizer grouping depend on is restated natively:

  * arithmetic (peft `lora.Linear.forward`):  y = x W^T (+ b) + (alpha / r) * B(A(dropout(x))),  A [r, in] Kaiming-uniform(a = sqrt 5),
    B [out, r] zeros -- the adapter is the identity until trained;
  * parameter / state-dict names (names must contain `lora`): the wrapper nesting `lang_encoder.base_model.model.<...>` and,
    per target module, `<name>.weight` (frozen base, peft 0.4 layout; `<name>.base_layer.weight` of peft >= 0.6 is accepted on load),
    `<name>.lora_A.default.weight`, `<name>.lora_B.default.weight`;
  * attribute forwarding of the two wrapper levels (peft's `__getattr__` chain), so `lang_encoder._get_decoder_layers()`,
    `.transformer.wte`, `.config`, `.get_input_embeddings()` keep working on the wrapped model.

The base projection keeps its own forward (otter_amd.mpt.FrozenAwareLinear: hipBLASLt / own kernels, transposed copy for the input
gradient); the rank-r products are two skinny GEMMs (r = 16: far below any tile of csrc/gemm.hip) and stay on torch."""
from __future__ import annotations

import math
from typing import Iterable, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mpt import FrozenAwareLinear

ADAPTER = "default"

# modeling_otter.py:76-83 and :811-819
MODEL_CLASSES = {"LlamaForCausalLM": "llama", "OPTForCausalLM": "opt", "GPTJForCausalLM": "gptj", "GPTNeoXForCausalLM": "gpt_neox",
                 "MPTForCausalLM": "mpt", "MosaicGPT": "mpt"}
TARGET_MODULES = {"llama": ["q_proj", "v_proj"], "opt": ["q_proj", "v_proj"], "gptj": ["q_proj", "v_proj"], "gpt_neox": ["query_key_value"],
                  "mpt": ["Wqkv"]}


class LoraLinear(FrozenAwareLinear):
    """A target nn.Linear with one low-rank adapter.  Shares the base layer's Parameter objects (no copy)."""

    def __init__(self, base: nn.Linear, r: int, lora_alpha: float, lora_dropout: float = 0.0):
        if r <= 0:
            raise ValueError("LoRA rank must be positive, got %r" % (r,))
        nn.Module.__init__(self)                      # (not nn.Linear.__init__: no second [out, in] allocation)
        self.in_features, self.out_features = base.in_features, base.out_features
        self.weight = base.weight
        self.register_parameter("bias", base.bias)
        dev, dt = base.weight.device, torch.float32   # adapters are fp32 masters like every trainable parameter of the recipe
        self.r = {ADAPTER: int(r)}
        self.lora_alpha = {ADAPTER: lora_alpha}
        self.scaling = {ADAPTER: lora_alpha / r}
        self.lora_dropout = nn.ModuleDict({ADAPTER: nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()})
        self.lora_A = nn.ModuleDict({ADAPTER: nn.Linear(self.in_features, r, bias=False, device=dev, dtype=dt)})
        self.lora_B = nn.ModuleDict({ADAPTER: nn.Linear(r, self.out_features, bias=False, device=dev, dtype=dt)})
        nn.init.kaiming_uniform_(self.lora_A[ADAPTER].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B[ADAPTER].weight)
        self.weight.requires_grad = False             # peft freezes the base layer it wraps
        self._register_load_state_dict_pre_hook(self._accept_base_layer_keys)

    @staticmethod
    def _accept_base_layer_keys(state_dict, prefix, *unused):
        """Checkpoints written with peft >= 0.6 keep the frozen base under `<name>.base_layer.{weight,bias}`."""
        for leaf in ("weight", "bias"):
            k = prefix + "base_layer." + leaf
            if k in state_dict:
                state_dict[prefix + leaf] = state_dict.pop(k)

    def lora_delta(self, x: torch.Tensor) -> torch.Tensor:
        A, B = self.lora_A[ADAPTER], self.lora_B[ADAPTER]
        h = self.lora_dropout[ADAPTER](x)
        if not torch.is_autocast_enabled() and h.dtype != A.weight.dtype:
            h = h.to(A.weight.dtype)
        return B(A(h)) * self.scaling[ADAPTER]

    def forward(self, x):
        y = FrozenAwareLinear.forward(self, x)
        return y + self.lora_delta(x).to(y.dtype)

    def merged_weight(self) -> torch.Tensor:
        """W + (alpha / r) B A -- what peft's merge_adapter() would write into the base layer (used by the tests as the reference)."""
        return self.weight.detach().float() + self.scaling[ADAPTER] * (self.lora_B[ADAPTER].weight.detach().float() @ self.lora_A[ADAPTER].weight.detach().float())

    def extra_repr(self):
        return "in_features=%d, out_features=%d, r=%d, lora_alpha=%s" % (self.in_features, self.out_features, self.r[ADAPTER], self.lora_alpha[ADAPTER])


def _fallthrough_getattr(self, name, inner):
    try:
        return nn.Module.__getattr__(self, name)
    except AttributeError:
        return getattr(nn.Module.__getattr__(self, inner), name)


class LoraModel(nn.Module):
    """peft.tuners.lora.LoraModel: holds the adapted network as `.model`."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def __getattr__(self, name):
        return _fallthrough_getattr(self, name, "model")


class PeftModelForCausalLM(nn.Module):
    """peft.PeftModelForCausalLM as far as Otter uses it: `.base_model` (a LoraModel), forward delegation, attribute fall-through."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.base_model = LoraModel(model)

    def forward(self, *args, **kwargs):
        return self.base_model(*args, **kwargs)

    def __getattr__(self, name):
        return _fallthrough_getattr(self, name, "base_model")

    def get_base_model(self) -> nn.Module:
        return self.base_model.model

    def trainable_parameter_counts(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return tr, sum(p.numel() for p in self.parameters())

    def master_print_trainable_parameters(self):      # modeling_otter.py:828
        tr, al = self.trainable_parameter_counts()
        print("trainable params: %d || all params: %d || trainable%%: %.4f" % (tr, al, 100.0 * tr / max(al, 1)))

    print_trainable_parameters = master_print_trainable_parameters


def _replace_targets(model: nn.Module, targets: Iterable[str], r: int, alpha: float, dropout: float) -> List[str]:
    done = []
    targets = tuple(targets)
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            if child_name in targets and isinstance(child, nn.Linear) and not isinstance(child, LoraLinear):
                setattr(parent, child_name, LoraLinear(child, r, alpha, dropout))
                done.append((parent_name + "." if parent_name else "") + child_name)
    return done


def get_lora_model(lang_encoder: nn.Module, lora_config: dict, architecture: str) -> PeftModelForCausalLM:
    """The reference's `get_peft_model(self.lang_encoder, LoraConfig(...))` + class rename (modeling_otter.py:808-829)."""
    short = MODEL_CLASSES.get(architecture)
    if short is None:
        raise KeyError("LoRA: unknown language model architecture %r (modeling_otter.py:76-83)" % (architecture,))
    for k in ("r", "lora_alpha", "lora_dropout"):
        if k not in lora_config:
            raise KeyError("lora_config needs %r (modeling_otter.py:820-823)" % k)
    hit = _replace_targets(lang_encoder, TARGET_MODULES[short], int(lora_config["r"]), lora_config["lora_alpha"], float(lora_config["lora_dropout"]))
    if not hit:
        raise ValueError("LoRA: none of the target modules %s found in %s" % (TARGET_MODULES[short], lang_encoder.__class__.__name__))
    original = lang_encoder.__class__.__name__
    # the reference renames the (shared) wrapper class in place; a per-architecture subclass gives the same `__class__.__name__`
    cls = type(original + "LoRA", (PeftModelForCausalLM,), {})
    return cls(lang_encoder)
