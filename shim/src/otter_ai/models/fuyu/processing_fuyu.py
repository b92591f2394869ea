"""`src.otter_ai.models.fuyu.processing_fuyu`: host-side prompt / patch packing; the reference's class extends transformers' processor
(fuyu/processing_fuyu.py) and never touches the GPU -- the library class is re-exported."""
from transformers import FuyuProcessor  # noqa: F401
