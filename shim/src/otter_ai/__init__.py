"""Drop-in shim package (see shim/README.md).  Mirrors the reference's package root (src/otter_ai/__init__.py): the two model
classes its demos and benchmark wrappers import as `from otter_ai import OtterForConditionalGeneration`."""
from . import models  # noqa: F401
from .models.flamingo.modeling_flamingo import FlamingoForConditionalGeneration  # noqa: F401
from .models.otter.modeling_otter import OtterForConditionalGeneration  # noqa: F401
