"""`otter_ai` (the reference's pip-installed package name, used by its demos) as an alias of shim/src/otter_ai."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "src", "otter_ai")]
