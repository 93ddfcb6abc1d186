"""MI355X-native dense initialisation for LichtFeld Studio (hot path of the densification plugin).

Layout
    csrc/        hand-written HIP kernels (gfx950) + the C-ABI (`include/lfd_densify.h`)
    core/        host-side mirror of the upstream plugin's `core/` interface for this path
    densify.py   upstream entry points (`dense_init`, `dense_init_from_lfs`, `build_argparser`)
    synthetic.py analytic scenes for tests / smoke / bench

Nothing in this package imports from ``oracle/``; the HIP library is required (no CPU fallback).
"""
from .core.types import CameraRecord, DensePipelineConfig  # noqa: F401

__version__ = "0.1.0"
