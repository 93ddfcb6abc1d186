"""The C-ABI proven from C (the drop-in boundary is the header): tests/abi/caller.c - C99, `-Wall -Werror`, nothing but
include/lfd_densify.h - is compiled against liblfd_densify.so and run on a case UPSTREAM made (golden g3: `_triangulate_ref` on captured selections,
core/pipeline.py:602-780): same survivor count, same group sizes in upstream's group order, positions within the stated tolerance; a bad argument comes
back as a status and a message.  CPU tier: the CPU twin (lfd_create_host ... lfd_triangulate_indexed_host); ``-m gpu``: the device entry points.
Also here: the ctypes mirror of the structures against the compiler's layout (lfd_struct_layout / lfd_struct_fields)."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from helpers import g3_case, oracle_cams
from lichtfeld_densification_plugin_amd.core import hip_backend as hb

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "lichtfeld-densification-plugin_amd")


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    hb.load_library()                                  # (builds / checks the library first)
    exe = str(tmp_path_factory.mktemp("abi") / "caller")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    rocm_lib = "/opt/rocm/lib"
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "abi", "caller.c"),
           "-L", PKG, "-llfd_densify", "-L", rocm_lib, "-lamdhip64", "-lm", f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{rocm_lib}", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def _dump_case(g3, name, path):
    """golden g3 `name` as the flat file caller.c reads (layout in its header comment)"""
    case = g3_case(g3, name)
    cams = oracle_cams(g3)
    p = case["params"]
    k, H, W = case["k"], case["H"], case["W"]
    warp = np.ascontiguousarray(case["warp"], np.float32)
    ch = warp.shape[-1]
    seg = np.zeros(k, np.int32)
    seg[:len(case["seg_count"])] = case["seg_count"]
    head = np.zeros(16, np.int32)
    head[:12] = [0x4C464443, H, W, case["w_match"], case["h_match"], k, ch, len(cams), case["ref"], case["sel"].size, case["xyz"].shape[0], int(p.no_filter)]
    with open(path, "wb") as f:
        f.write(head.tobytes())
        f.write(struct.pack("<d", float(p.sampson_thresh)))
        f.write(np.array([p.certainty_thresh, 0.9, p.reproj_thresh, p.min_parallax_deg], np.float32).tobytes())
        f.write(np.asarray(case["nbrs"], np.int32).tobytes())
        for field, n in (("K", 9), ("R", 9), ("t", 3), ("P", 12), ("C", 3)):
            f.write(np.stack([np.asarray(getattr(c, field), np.float32).reshape(n) for c in cams]).tobytes())
        f.write(np.array([[c.width, c.height] for c in cams], np.int32).tobytes())
        f.write(np.ascontiguousarray(case["cert"], np.float32).tobytes())
        f.write(warp.tobytes())
        f.write(np.ascontiguousarray(case["image"], np.uint8).tobytes())
        f.write(np.ascontiguousarray(case["sel"], np.int64).tobytes())
        f.write(seg.tobytes())
        f.write(np.ascontiguousarray(case["xyz"], np.float32).tobytes())
    return case


@pytest.mark.parametrize("name", ["a_filter_k3", "c_rect_k3", "f_nosampson_k4"])
def test_a_c_program_drives_the_cpu_twin_through_the_header_alone(caller, g3, tmp_path, name):
    case = _dump_case(g3, name, str(tmp_path / "case.bin"))
    res = subprocess.run([caller, str(tmp_path / "case.bin"), "host"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr + res.stdout
    assert res.stdout.startswith(f"OK host: {case['xyz'].shape[0]} survivors") and "bad argument -> status 1" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a_filter_k3", "c_rect_k3", "f_nosampson_k4"])
def test_a_c_program_drives_the_device_entry_points_through_the_header_alone(caller, g3, tmp_path, name):
    case = _dump_case(g3, name, str(tmp_path / "case.bin"))
    res = subprocess.run([caller, str(tmp_path / "case.bin"), "device"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr + res.stdout
    assert res.stdout.startswith(f"OK device: {case['xyz'].shape[0]} survivors")
    assert "two references chained on one MT19937 stream from C" in res.stdout           # lfd_rng_seed, lfd_triangulate_sampled, lfd_triangulate_sampled_chain


def test_the_ctypes_mirror_is_the_compilers_layout():
    lib = hb.load_library()                            # (load_library itself runs the check: an import-time failure otherwise)
    hb.check_struct_layout(lib)
    n = lib.lfd_struct_layout(None, 0)
    table = (C.c_int32 * n)()
    assert lib.lfd_struct_layout(table, n) == n and table[0] == C.sizeof(hb.lfd_params) == 32 and table[1] == 7 and list(table[2:6]) == [0, 8, 8, 4]
    assert C.sizeof(hb.lfd_batch) == 32 + 11 * 8 and C.sizeof(hb.lfd_points) == 48 and C.sizeof(hb.lfd_tile_segment) == 8 and C.sizeof(hb.lfd_copy_segment) == 24
    # what the Python side really hands over for the two small structures are NumPy / torch rows: the same bytes
    assert np.dtype(np.int32).itemsize * 2 == C.sizeof(hb.lfd_tile_segment) and np.dtype(np.int64).itemsize * 3 == C.sizeof(hb.lfd_copy_segment)


def test_a_reordered_or_retyped_mirror_fails_the_check():
    lib = hb.load_library()

    class lfd_params(C.Structure):                     # two neighbouring floats swapped: every offset is still right, the names tell
        _fields_ = [("sampson_thresh", C.c_double), ("sample_cap", C.c_float), ("certainty_thresh", C.c_float), ("reproj_thresh", C.c_float),
                    ("min_parallax_deg", C.c_float), ("no_filter", C.c_int32), ("flags", C.c_int32)]
    with pytest.raises(hb.HipBackendError, match="lists the fields"):
        hb.check_struct_layout(lib, (lfd_params,) + hb.ABI_STRUCTS[1:])

    class lfd_points(C.Structure):                     # capacity narrowed to 32 bits: the size tells
        _fields_ = [("xyz", C.c_void_p), ("rgb", C.c_void_p), ("err", C.c_void_p), ("cell", C.c_void_p), ("slot", C.c_void_p), ("capacity", C.c_int32)]
    with pytest.raises(hb.HipBackendError, match="does not match the library's layout"):
        hb.check_struct_layout(lib, hb.ABI_STRUCTS[:2] + (lfd_points,) + hb.ABI_STRUCTS[3:])

    class lfd_batch(C.Structure):                      # a field dropped
        _fields_ = [f for f in hb.lfd_batch._fields_ if f[0] != "reserved"]
    with pytest.raises(hb.HipBackendError):
        hb.check_struct_layout(lib, (hb.lfd_params, lfd_batch) + hb.ABI_STRUCTS[2:])
