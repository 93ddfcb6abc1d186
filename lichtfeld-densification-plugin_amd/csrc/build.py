"""Build the HIP library in-tree: ``python build.py`` -> ../liblfd_densify.so (gfx950).

hipcc cross-compiles without a GPU.  -ffp-contract=off: every rounding in lfd_geometry.hpp is
explicit (mul+add vs fma) so the host build of the per-cell routine and the device build agree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "liblfd_densify.so")
SOURCES = ["lfd_api.hip", "lfd_kernels.hip", "lfd_select.hip", "lfd_writer.hip", "lfd_host.hip", "lfd_image.hip"]
HEADERS = ["lfd_device.hpp", "lfd_geometry.hpp", "lfd_context.hpp", os.path.join("..", "..", "include", "lfd_densify.h")]
# -amdgpu-sched-strategy=max-ilp: the dense kernel is bound by its vector arithmetic (long dependent f64 chains); the
# ILP-first machine scheduler is worth 3.5 % on it (profiles/history.md (r1/ablation.txt)), instruction semantics are unchanged
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building liblfd_densify.so")
    if verbose:
        sys.stderr.write(res.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
