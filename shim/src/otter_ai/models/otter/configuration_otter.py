"""`src.otter_ai.models.otter.configuration_otter` -> otter_amd.configuration_otter."""
from otter_amd.configuration_otter import OtterConfig  # noqa: F401
