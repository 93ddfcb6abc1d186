"""Upstream import path ``core.camera_models.CameraRecord`` (core/camera_models.py:10-28)."""
from .types import CameraRecord  # noqa: F401
