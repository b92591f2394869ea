"""`src.otter_ai.models.fuyu.modeling_fuyu` -> otter_amd.fuyu (OtterHD path)."""
from otter_amd.fuyu import FuyuForCausalLM  # noqa: F401
