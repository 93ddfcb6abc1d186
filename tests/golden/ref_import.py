"""Import the upstream plugin's hot-path modules from /root/reference (development container only).

The upstream package needs its GUI host (``lichtfeld``), ``pycolmap`` and the RoMa-v2 package at
import time.  None of them exist here, so three stub modules are inserted into ``sys.modules`` and a
synthetic parent package whose ``__path__`` points at the read-only checkout is registered, so that
``/root/reference/__init__.py`` (which registers GUI panels) is never executed.

This file is tooling for *generating* golden vectors (tests/golden/make_golden.py).  It is never
imported by tests, smoke() or bench.py at run time: /root/reference does not exist on the GPU box.
"""
from __future__ import annotations

import importlib
import sys
import types

REFERENCE_ROOT = "/root/reference"
_PKG = "_lfd_upstream"


class _Log:
    def __init__(self):
        self.records = []

    def _add(self, level, msg):
        self.records.append((level, str(msg)))

    def info(self, msg):
        self._add("info", msg)

    def warn(self, msg):
        self._add("warn", msg)

    def error(self, msg):
        self._add("error", msg)

    def debug(self, msg):
        self._add("debug", msg)


def load_reference():
    """Return a namespace with the upstream ``core.*`` modules."""
    if _PKG + ".core.pipeline" in sys.modules:
        return _namespace()

    lf = types.ModuleType("lichtfeld")
    lf.log = _Log()
    sys.modules.setdefault("lichtfeld", lf)

    pc = types.ModuleType("pycolmap")
    for name in ("Camera", "Image", "Reconstruction"):
        setattr(pc, name, type(name, (), {}))
    sys.modules.setdefault("pycolmap", pc)

    rm = types.ModuleType("romav2")

    class RoMaV2:  # placeholder: the real model needs torchvision + downloaded weights
        class Cfg:
            def __init__(self, **kw):
                pass

    rm.RoMaV2 = RoMaV2
    sys.modules.setdefault("romav2", rm)

    parent = types.ModuleType(_PKG)
    parent.__path__ = [REFERENCE_ROOT]
    sys.modules[_PKG] = parent
    for sub in ("core.geometry", "core.sampling", "core.config", "core.camera_models",
                "core.writers", "core.image_utils", "core.selection", "core.debug_viz",
                "core.threaded_dataloader", "core.pipeline"):
        importlib.import_module(f"{_PKG}.{sub}")
    return _namespace()


def _namespace():
    ns = types.SimpleNamespace()
    for sub in ("geometry", "sampling", "config", "camera_models", "writers", "image_utils",
                "selection", "debug_viz", "threaded_dataloader", "pipeline"):
        setattr(ns, sub, sys.modules[f"{_PKG}.core.{sub}"])
    ns.log = sys.modules["lichtfeld"].log
    return ns
