"""Upstream import path ``core.config.DensePipelineConfig`` (core/config.py:7-26)."""
from .types import DensePipelineConfig, TRIANGULATION_MODES  # noqa: F401
