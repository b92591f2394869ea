"""`otter_ai` (the reference's pip-installed package name, used by its demos) as an alias of shim/src/otter_ai: submodules resolve
through `__path__`, the package-level names (`from otter_ai import OtterForConditionalGeneration`, src/otter_ai/__init__.py) are
re-exported here."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "src", "otter_ai")]

from otter_ai import models  # noqa: E402,F401
from otter_ai.models.flamingo.modeling_flamingo import FlamingoForConditionalGeneration  # noqa: E402,F401
from otter_ai.models.otter.modeling_otter import OtterForConditionalGeneration  # noqa: E402,F401
