"""`src.otter_ai.models.fuyu.processing_fuyu` -> otter_amd.processing_fuyu: the reference's processor semantics (text-first `__call__`,
per-sample encoding + RIGHT padding with the eos id, `get_labels`, `find_and_remove_tokens` -- what pipeline/mimicit_utils/
mimicit_dataset.py:497-505 calls) on top of the installed transformers' encoder."""
from otter_amd.processing_fuyu import FuyuProcessor  # noqa: F401
