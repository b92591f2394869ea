"""`src.otter_ai.models.otter.modeling_otter` -> otter_amd (reference: src/otter_ai/models/otter/modeling_otter.py)."""
from otter_amd.modeling_otter import *  # noqa: F401,F403
from otter_amd.modeling_otter import (OtterConfig, OtterForConditionalGeneration, OtterGatedCrossAttentionBlock, OtterLayer, OtterLMMixin,  # noqa: F401
                                      OtterMaskedCrossAttention, OtterModel, OtterPerceiverBlock, OtterPerceiverResampler,
                                      OtterPreTrainedModel, extend_instance, getattr_recursive, master_print, setattr_recursive)
