"""ctypes binding of the C-ABI in ``include/lfd_densify.h`` (the HIP library is the only backend).

Torch is used for what it is good at here - owning device memory and streams; every tensor is
handed to the library as a raw device pointer.  If ``liblfd_densify.so`` is missing or no GPU is
present the constructors raise: there is deliberately no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
import dataclasses
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .types import CameraRecord, DensePipelineConfig

LFD_MAX_SLOTS = 16
LFD_ABI_VERSION = 9
LFD_FLAG_EXACT_COLOUR = 1     # lfd_params.flags: dense mode blends colours with upstream's f64 arithmetic (bit-identical rgb)
LFD_FLAG_TILE_SEGMENTS = 2    # informational: the caller takes the unordered-retirement route (lfd_triangulate_dense_segments)
_LIB_NAME = "liblfd_densify.so"
_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HipBackendError(RuntimeError):
    """Raised for any non-zero status of the HIP library (message from ``lfd_last_error``)."""


class SelectionInexact(HipBackendError):
    """The device selection met a normalised weight below 2^-29 (LFD_SELECT_INEXACT): its exact parallel cumulative sum
    is not guaranteed for such input, so it refused - WITHOUT consuming the MT19937 stream.  The caller runs the host
    stage (core/sampling.py) on the same map with the device's stream state instead (core/pipeline.py does)."""


class lfd_params(C.Structure):
    _fields_ = [("sampson_thresh", C.c_double), ("certainty_thresh", C.c_float), ("sample_cap", C.c_float),
                ("reproj_thresh", C.c_float), ("min_parallax_deg", C.c_float), ("no_filter", C.c_int32),
                ("flags", C.c_int32)]


class lfd_batch(C.Structure):
    _fields_ = [("n_refs", C.c_int32), ("k", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("w_match", C.c_int32), ("h_match", C.c_int32), ("warp_channels", C.c_int32), ("reserved", C.c_int32),
                ("ref_cam", C.POINTER(C.c_int32)), ("n_slots", C.POINTER(C.c_int32)), ("nbr_cam", C.POINTER(C.c_int32)),
                ("cert", C.POINTER(C.c_void_p)), ("warp", C.POINTER(C.c_void_p)), ("image", C.POINTER(C.c_void_p)),
                ("mask_a", C.POINTER(C.c_void_p)), ("mask_b", C.POINTER(C.c_void_p)),
                ("axis_x", C.c_void_p), ("axis_y", C.c_void_p), ("fundamental", C.POINTER(C.c_float))]


class lfd_points(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("rgb", C.c_void_p), ("err", C.c_void_p), ("cell", C.c_void_p),
                ("slot", C.c_void_p), ("capacity", C.c_int64)]


class lfd_tile_segment(C.Structure):
    """one row of the tile table of lfd_triangulate_dense_segments (handled as an (n, 2) int32 tensor on this side)"""
    _fields_ = [("offset", C.c_int32), ("count", C.c_int32)]


class lfd_copy_segment(C.Structure):
    """one copy of lfd_copy_segments (handled as an (n, 3) int64 array on this side)"""
    _fields_ = [("src_offset", C.c_int64), ("dst_offset", C.c_int64), ("nbytes", C.c_int64)]


ABI_STRUCTS = (lfd_params, lfd_batch, lfd_points, lfd_tile_segment, lfd_copy_segment)     # the order lfd_struct_layout reports them in


def check_struct_layout(lib, structs=ABI_STRUCTS) -> None:
    """The ctypes mirrors above against what the COMPILER made of include/lfd_densify.h (lfd_struct_layout: sizeof, field count and every
    field's offset and size per structure; lfd_struct_fields: their names): a field added, dropped, reordered or retyped on one side only is an import error here, not a corrupted
    launch later."""
    lib.lfd_struct_layout.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    lib.lfd_struct_layout.restype = C.c_int
    n = int(lib.lfd_struct_layout(None, 0))
    table = (C.c_int32 * n)()
    lib.lfd_struct_layout(table, n)
    lib.lfd_struct_fields.restype = C.c_char_p
    names = dict(part.split(":") for part in lib.lfd_struct_fields().decode().split(";"))
    vals, at = list(table), 0
    for cls in structs:
        theirs = names.get(cls.__name__, "").split(",")
        if [name for name, *_ in cls._fields_] != theirs:
            raise HipBackendError(f"ctypes mirror of {cls.__name__} lists the fields {[name for name, *_ in cls._fields_]}, the library's header has "
                                  f"{theirs} (include/lfd_densify.h and core/hip_backend.py disagree)")
        if at + 2 > len(vals):
            raise HipBackendError(f"lfd_struct_layout ends before {cls.__name__}: the library is older than this binding")
        size, n_fields = vals[at], vals[at + 1]
        offsets = list(zip(vals[at + 2:at + 2 + 2 * n_fields:2], vals[at + 3:at + 2 + 2 * n_fields:2]))       # (offset, size) per field
        at += 2 + 2 * n_fields
        mine = [(name, getattr(cls, name).offset, getattr(cls, name).size) for name, *_ in cls._fields_]
        if C.sizeof(cls) != size or [(o, z) for _n, o, z in mine] != offsets:
            raise HipBackendError(f"ctypes mirror of {cls.__name__} does not match the library's layout: here sizeof {C.sizeof(cls)} with offsets "
                                  f"(name, offset, size) {mine}, the library has sizeof {size} with (offset, size) {offsets} (include/lfd_densify.h and core/hip_backend.py disagree)")
    if at != len(vals):
        raise HipBackendError("lfd_struct_layout reports structures this binding does not mirror")


_lib = None


def library_path() -> str:
    """In-tree library; ``LFD_DENSIFY_LIB`` overrides it (used to A/B kernel builds while profiling)."""
    return os.environ.get("LFD_DENSIFY_LIB") or os.path.join(_PKG_DIR, _LIB_NAME)


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library and declare its prototypes (works without a GPU: the host
    helpers and symbol table are usable, every compute entry point then returns LFD_ERR_HIP)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HipBackendError(
            f"{path} not found: build it with `python {os.path.join(_PKG_DIR, 'csrc', 'build.py')}` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback for the dense-initialisation path.")
    lib = C.CDLL(path)
    ctxp = C.c_void_p
    lib.lfd_abi_version.restype = C.c_int
    if int(lib.lfd_abi_version()) != LFD_ABI_VERSION:
        raise HipBackendError(f"{path} implements ABI {int(lib.lfd_abi_version())}, this binding ABI {LFD_ABI_VERSION}: rebuild the library "
                              f"(python {os.path.join(_PKG_DIR, 'csrc', 'build.py')})")
    check_struct_layout(lib)
    lib.lfd_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(ctxp)]
    lib.lfd_destroy.argtypes = [ctxp]
    lib.lfd_destroy.restype = None
    lib.lfd_set_stream.argtypes = [ctxp, C.c_void_p]
    lib.lfd_reload_env.argtypes = [ctxp]
    lib.lfd_kernel_timing.argtypes = [ctxp, C.c_int32]
    lib.lfd_kernel_timing_read.argtypes = [ctxp, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]
    lib.lfd_last_error.argtypes = [ctxp]
    lib.lfd_last_error.restype = C.c_char_p
    fptr = C.POINTER(C.c_float)
    lib.lfd_upload_cameras.argtypes = [ctxp, C.c_int32, fptr, fptr, fptr, fptr, fptr, C.POINTER(C.c_int32)]
    lib.lfd_prepare_batch.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params)]
    lib.lfd_aggregate.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_dense.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.POINTER(lfd_points),
                                          C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_dense_ply.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
    lib.lfd_dense_tiles_per_ref.argtypes = [C.c_int32, C.c_int32]
    lib.lfd_triangulate_dense_segments.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.POINTER(lfd_points),
                                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_dense_ply_segments.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lfd_order_segments.argtypes = [ctxp, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(lfd_points), C.POINTER(lfd_points), C.c_void_p]
    lib.lfd_pack_ply_segments.argtypes = [ctxp, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.lfd_pack_points3d_segments.argtypes = [ctxp, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_uint64, C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_indexed.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p,
                                            C.POINTER(C.c_int64), C.POINTER(lfd_points), C.c_void_p, C.c_void_p,
                                            C.c_void_p]
    lib.lfd_triangulate_sampled.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_int32, C.c_float, C.c_int32,
                                            C.c_int32, C.c_float, C.POINTER(lfd_points), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_sampled_multi.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_int32, C.c_float, C.c_int32,
                                                  C.c_int32, C.POINTER(C.c_uint32), C.POINTER(lfd_points), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_sampled_chain.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_int32, C.c_float, C.c_int32,
                                                  C.c_int32, C.POINTER(C.c_float), C.POINTER(lfd_points), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]
    lib.lfd_select_top_m.argtypes = [ctxp, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lfd_pack_ply.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.lfd_pack_points3d.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
    lib.lfd_quantise_rgb.argtypes = [ctxp, C.c_void_p, C.c_int64, C.c_void_p]
    lib.lfd_copy_segments.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.lfd_launch_status.argtypes = [ctxp, C.POINTER(C.c_int32)]
    lib.lfd_get_pair_fundamental.argtypes = [ctxp, C.c_int32, C.POINTER(C.c_double)]
    lib.lfd_rng_seed.argtypes = [ctxp, C.c_uint32]
    lib.lfd_rng_get_state.argtypes = [ctxp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    lib.lfd_rng_set_state.argtypes = [ctxp, C.POINTER(C.c_uint32), C.c_int32]
    lib.lfd_rng_checkpoint.argtypes = [ctxp, C.c_int32]
    lib.lfd_rng_rollback.argtypes = [ctxp, C.c_int32]
    lib.lfd_select_samples.argtypes = [ctxp, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                       C.c_float, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lfd_create_host.argtypes = [C.c_int32, C.POINTER(ctxp)]
    lib.lfd_host_threads.argtypes = [ctxp]
    lib.lfd_host_threads.restype = C.c_int
    lib.lfd_aggregate_host.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_dense_host.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.POINTER(lfd_points),
                                               C.c_void_p, C.c_void_p]
    lib.lfd_triangulate_indexed_host.argtypes = [ctxp, C.POINTER(lfd_batch), C.POINTER(lfd_params), C.c_void_p,
                                                 C.POINTER(C.c_int64), C.POINTER(lfd_points), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lfd_prepare_image.argtypes = [ctxp, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lfd_prepare_mask.argtypes = [ctxp, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    lib.lfd_host_resize_tables.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
    lib.lfd_host_nearest_indices.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.lfd_identity_axis.argtypes = [C.c_int32, fptr]
    lib.lfd_parallax_dot_threshold.argtypes = [C.c_float]
    lib.lfd_parallax_dot_threshold.restype = C.c_float
    lib.lfd_host_fundamental.argtypes = [fptr] * 7
    lib.lfd_host_null_vector.argtypes = [fptr, C.POINTER(C.c_double)]
    lib.lfd_host_null_vector.restype = C.c_int
    lib.lfd_host_eval_correspondence.argtypes = [fptr, fptr, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                                 C.c_int32, C.POINTER(lfd_params), fptr]
    for name in ("lfd_create", "lfd_set_stream", "lfd_reload_env", "lfd_kernel_timing", "lfd_kernel_timing_read", "lfd_upload_cameras", "lfd_prepare_batch", "lfd_aggregate", "lfd_triangulate_dense",
                 "lfd_triangulate_dense_ply", "lfd_triangulate_dense_ply_segments", "lfd_dense_tiles_per_ref", "lfd_triangulate_dense_segments", "lfd_order_segments", "lfd_pack_ply_segments", "lfd_pack_points3d_segments",
                 "lfd_triangulate_indexed", "lfd_triangulate_sampled", "lfd_triangulate_sampled_multi", "lfd_triangulate_sampled_chain", "lfd_launch_status", "lfd_rng_seed", "lfd_rng_get_state", "lfd_rng_set_state",
                 "lfd_rng_checkpoint", "lfd_rng_rollback",
                 "lfd_select_samples", "lfd_select_top_m", "lfd_pack_ply", "lfd_pack_points3d", "lfd_quantise_rgb", "lfd_copy_segments", "lfd_identity_axis",
                 "lfd_host_fundamental", "lfd_get_pair_fundamental", "lfd_create_host", "lfd_aggregate_host",
                 "lfd_triangulate_dense_host", "lfd_triangulate_indexed_host", "lfd_prepare_image", "lfd_prepare_mask",
                 "lfd_host_resize_tables", "lfd_host_nearest_indices",
                 "lfd_host_eval_correspondence"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def make_params(config: DensePipelineConfig, sample_cap: float = 0.9, exact_colour: Optional[bool] = None) -> lfd_params:
    """``exact_colour``: dense mode blends colours in f64 like upstream (bit-identical rgb) instead of f32 (within 2.5e-7);
    default: the configuration's ``exact_colour`` field (False)."""
    if exact_colour is None:
        exact_colour = bool(config.exact_colour)
    return lfd_params(sampson_thresh=float(config.sampson_thresh), certainty_thresh=float(config.certainty_thresh),
                      sample_cap=float(sample_cap), reproj_thresh=float(config.reproj_thresh),
                      min_parallax_deg=float(config.min_parallax_deg), no_filter=1 if config.no_filter else 0,
                      flags=LFD_FLAG_EXACT_COLOUR if exact_colour else 0)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, np.float32))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- host helpers (usable without a GPU) --------------------------------------------------------
def identity_axis(n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    rc = load_library().lfd_identity_axis(n, _fp(out))
    if rc != 0:
        raise HipBackendError("lfd_identity_axis failed")
    return out


def copy_segments(src: torch.Tensor, dst: torch.Tensor, segments, stream: Optional["torch.cuda.Stream"] = None) -> None:
    """lfd_copy_segments: ``dst.bytes[d : d + n] = src.bytes[s : s + n]`` for every row ``(s, d, n)`` of ``segments`` (int64, byte offsets into the two
    contiguous device tensors' storage views) in ONE launch on ``stream`` (default: torch's current stream of that device).  Asynchronous."""
    segs = np.ascontiguousarray(np.asarray(segments, np.int64).reshape(-1, 3))
    if segs.shape[0] == 0:
        return
    if not (src.is_cuda and dst.is_cuda and src.device == dst.device and src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("copy_segments needs two contiguous tensors on one GPU")
    sb, db = src.numel() * src.element_size(), dst.numel() * dst.element_size()
    if segs.min() < 0 or int((segs[:, 0] + segs[:, 2]).max()) > sb or int((segs[:, 1] + segs[:, 2]).max()) > db:
        raise ValueError("copy_segments: a segment leaves its tensor")
    st = stream if stream is not None else torch.cuda.current_stream(src.device)
    rc = load_library().lfd_copy_segments(C.c_void_p(st.cuda_stream), int(src.device.index or 0), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                          segs.ctypes.data_as(C.c_void_p), int(segs.shape[0]))
    if rc != 0:
        raise HipBackendError(f"lfd_copy_segments failed ({rc})")


def parallax_dot_threshold(min_deg: float) -> float:
    return float(load_library().lfd_parallax_dot_threshold(C.c_float(min_deg)))


def host_fundamental(K1, R1, t1, K2, R2, t2) -> np.ndarray:
    a = [_f32(x).reshape(-1) for x in (K1, R1, t1, K2, R2, t2)]
    out = np.empty(9, np.float32)
    rc = load_library().lfd_host_fundamental(*[_fp(x) for x in a], _fp(out))
    if rc != 0:
        raise HipBackendError("lfd_host_fundamental failed")
    return out.reshape(3, 3)


def host_resize_tables(in_size: int, out_size: int):
    """(bounds (out,2), kk (out,ksize)) of the BILINEAR resampling exactly as the device kernel uses them (Pillow's tables)."""
    lib = load_library()
    ks = C.c_int32(0)
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((1,), np.int32)
    lib.lfd_host_resize_tables(in_size, out_size, bounds.ctypes.data_as(C.POINTER(C.c_int32)), kk.ctypes.data_as(C.POINTER(C.c_int32)), 0, C.byref(ks))
    kk = np.zeros((out_size, int(ks.value)), np.int32)
    rc = lib.lfd_host_resize_tables(in_size, out_size, bounds.ctypes.data_as(C.POINTER(C.c_int32)), kk.ctypes.data_as(C.POINTER(C.c_int32)),
                                    int(kk.size), C.byref(ks))
    if rc != 0:
        raise HipBackendError("lfd_host_resize_tables failed")
    return bounds, kk


def host_nearest_indices(in_size: int, out_size: int) -> np.ndarray:
    idx = np.zeros((out_size,), np.int32)
    if load_library().lfd_host_nearest_indices(in_size, out_size, idx.ctypes.data_as(C.POINTER(C.c_int32))) != 0:
        raise HipBackendError("lfd_host_nearest_indices failed")
    return idx


def fundamental_from_world2cam(K1, R1, t1, K2, R2, t2) -> np.ndarray:
    """Upstream's ``fundamental_from_world2cam`` (core/geometry.py:122-130 with ``skew`` :53-55): the same NumPy
    calls in the same order on the same f32 arrays (``np.linalg.inv`` is LAPACK sgetrf/sgetri), so the result is
    upstream's F bit for bit on the same machine.  Handed to the kernels through ``lfd_batch.fundamental``."""
    K1, R1, K2, R2 = (np.asarray(a, np.float32) for a in (K1, R1, K2, R2))
    t1, t2 = np.asarray(t1, np.float32).reshape(3, 1), np.asarray(t2, np.float32).reshape(3, 1)
    R = R2 @ R1.T
    t = (t2 - R @ t1).reshape(3)
    tx, ty, tz = t
    cross = np.array([[0, -tz, ty], [tz, 0, -tx], [-ty, tx, 0]], dtype=np.float32)
    E = cross @ R
    return _inv_intrinsics(K2).T @ E @ _inv_intrinsics(K1)


_KINV_CACHE: dict = {}


def _inv_intrinsics(K: np.ndarray) -> np.ndarray:
    """``np.linalg.inv(K)`` (upstream's call, LAPACK sgetrf/sgetri), remembered by the matrix's bytes: a run has a few hundred cameras and every
    pair asks for both inverses - the same LAPACK call on the same bytes gives the same bytes, so the cache changes no result, only the count
    of LAPACK calls in the per-reference loop (six per reference of three neighbours otherwise)."""
    key = K.tobytes()
    inv = _KINV_CACHE.get(key)
    if inv is None:
        if len(_KINV_CACHE) > 4096:
            _KINV_CACHE.clear()
        inv = _KINV_CACHE[key] = np.linalg.inv(K)
    return inv


def host_null_vector(A) -> "tuple[np.ndarray, int]":
    """Smallest right singular vector (un-normalised, f64) of a 4x4 f32 matrix through the routine the kernels
    triangulate with (host build of csrc/lfd_geometry.hpp), and the number of solves it made."""
    a = _f32(A).reshape(16)
    out = np.zeros(4, np.float64)
    it = load_library().lfd_host_null_vector(_fp(a), out.ctypes.data_as(C.POINTER(C.c_double)))
    if it < 0:
        raise HipBackendError("lfd_host_null_vector failed")
    return out, int(it)


def pack_camera(cam: CameraRecord) -> np.ndarray:
    return np.concatenate([_f32(cam.K).reshape(-1), _f32(cam.R).reshape(-1), _f32(cam.t).reshape(-1),
                           _f32(cam.P).reshape(-1), _f32(cam.C).reshape(-1),
                           np.array([cam.width, cam.height], np.float32)]).astype(np.float32)


def host_eval_correspondence(cam1: CameraRecord, cam2: CameraRecord, xa, ya, xb, yb, w_match, h_match,
                             params: lfd_params) -> np.ndarray:
    """The per-cell routine (host build of the kernels' source) on one correspondence: a debugging /
    unit-test aid, not a code path of the pipeline.  Returns [x,y,z,_,_,_,err,keep]."""
    out = np.zeros(8, np.float32)
    c1, c2 = pack_camera(cam1), pack_camera(cam2)
    rc = load_library().lfd_host_eval_correspondence(_fp(c1), _fp(c2), C.c_float(xa), C.c_float(ya), C.c_float(xb),
                                                     C.c_float(yb), int(w_match), int(h_match), C.byref(params), _fp(out))
    if rc != 0:
        raise HipBackendError("lfd_host_eval_correspondence failed")
    return out


# ---- device path ---------------------------------------------------------------------------------
@dataclasses.dataclass
class ReferenceInputs:
    """Device-resident inputs of one reference view (what RoMa produced for it)."""
    ref_cam: int                              # index into the uploaded camera table
    nbr_cams: List[int]                       # one per neighbour slot
    cert: List[torch.Tensor]                  # per slot (H,W) f32, raw certainty
    warp: List[torch.Tensor]                  # per slot (H,W,2|4) f32
    image: torch.Tensor                       # (h_match,w_match,3) u8
    mask_a: Optional[torch.Tensor] = None     # (h_match,w_match) u8 {0,1}
    mask_b: Optional[List[Optional[torch.Tensor]]] = None
    fundamental: Optional[Sequence[np.ndarray]] = None   # per slot (3,3) f32: upstream's F for the pair (see PreparedBatch)


@dataclasses.dataclass
class TriangulationOutput:
    xyz: torch.Tensor            # (n,3) f32
    rgb: torch.Tensor            # (n,3) f32 in [0,1]
    err: torch.Tensor            # (n,)  f32
    cell: Optional[torch.Tensor]  # (n,) i32
    slot: Optional[torch.Tensor]  # (n,) u8
    ref_offsets: np.ndarray      # (n_refs+1,) i64 host
    seg_counts: np.ndarray       # (n_refs,k) i32 host
    seg_order: Optional[np.ndarray] = None   # indexed mode: slot of the g-th emitted group, -1 = none
    n_selected: Optional[int] = None         # sampled call: cells the selection stage picked
    launch_status: int = 0                   # sampled call: look-back status of the launch (0 = ok)
    sel_status: Optional[np.ndarray] = None  # sampled calls: selection status per reference (0 = ok, else what upstream would have raised for)
    _packed: Optional[torch.Tensor] = None   # the one float buffer xyz / rgb / err are views of
    _cap: int = 0

    def host_arrays(self):
        """(xyz, rgb, err) as NumPy arrays.  When the buffers are small (sampled mode) the whole packed buffer crosses in
        one copy instead of three."""
        n = int(self.xyz.shape[0])
        if self._packed is None or self._cap * 7 > (1 << 20) or n == 0:
            return self.xyz.cpu().numpy(), self.rgb.cpu().numpy(), self.err.cpu().numpy()
        h = self._packed.cpu().numpy()
        c = self._cap
        return h[:3 * c].reshape(c, 3)[:n].copy(), h[3 * c:6 * c].reshape(c, 3)[:n].copy(), h[6 * c:6 * c + n].copy()

    @property
    def count(self) -> int:
        return int(self.ref_offsets[-1])


@dataclasses.dataclass
class SegmentedOutput:
    """What lfd_triangulate_dense_segments leaves on the device: reference r's survivors in rows [r*H*W, r*H*W + ref_counts[r]) of the
    buffers, tile after tile in the order the tiles retired; ``table[r * tiles_per_ref + t] = (offset, count)`` of every tile.  The ordered
    result / the file payload come out of ``HipDensifier.order_segments`` / ``pack_ply_segments`` / ``pack_points3d_segments``."""
    buffers: "OutputBuffers"
    table: torch.Tensor          # (n_refs * tiles_per_ref, 2) i32
    ref_counts: torch.Tensor     # (n_refs,) i64, device
    n_refs: int
    H: int
    W: int
    k: int


class PreparedBatch:
    """ctypes view of a list of ReferenceInputs; keeps the tensors alive."""

    def __init__(self, refs: Sequence[ReferenceInputs], w_match: int, h_match: int,
                 axes: Optional[Sequence[torch.Tensor]] = None, cameras: Optional[Sequence[CameraRecord]] = None):
        """``cameras``: the run's camera records; when given, every pair's fundamental matrix is computed on the host
        with upstream's own NumPy calls (``fundamental_from_world2cam``) and handed to the library, so the Sampson gate
        sees upstream's F bit for bit.  A reference that carries its own ``fundamental`` list keeps it."""
        if not refs:
            raise ValueError("empty batch")
        self.refs = list(refs)
        n = len(self.refs)
        k = max(len(r.cert) for r in self.refs)
        if k < 1 or k > LFD_MAX_SLOTS:
            raise ValueError(f"neighbour slots per reference must be in [1,{LFD_MAX_SLOTS}]")
        H, W = self.refs[0].cert[0].shape
        ch = int(self.refs[0].warp[0].shape[-1])
        dev = self.refs[0].cert[0].device
        self.n_refs, self.k, self.H, self.W, self.channels, self.device = n, k, int(H), int(W), ch, dev
        self.ref_cam = (C.c_int32 * n)(*[int(r.ref_cam) for r in self.refs])
        self.n_slots = (C.c_int32 * n)(*[len(r.cert) for r in self.refs])
        self.nbr_cam = (C.c_int32 * (n * k))()
        self.cert = (C.c_void_p * (n * k))()
        self.warp = (C.c_void_p * (n * k))()
        self.image = (C.c_void_p * n)()
        any_ma = any(r.mask_a is not None for r in self.refs)
        any_mb = any(r.mask_b is not None and any(m is not None for m in r.mask_b) for r in self.refs)
        self.mask_a = (C.c_void_p * n)() if any_ma else None
        self.mask_b = (C.c_void_p * (n * k))() if any_mb else None
        self._keep = []
        for i, r in enumerate(self.refs):
            if not (len(r.cert) == len(r.warp) == len(r.nbr_cams)):
                raise ValueError("cert / warp / nbr_cams length mismatch")
            img = self._chk(r.image, torch.uint8, (h_match, w_match, 3), "image")
            self.image[i] = img.data_ptr()
            if r.mask_a is not None:
                self.mask_a[i] = self._chk(r.mask_a, torch.uint8, (h_match, w_match), "mask_a").data_ptr()
            for j in range(len(r.cert)):
                s = i * k + j
                self.nbr_cam[s] = int(r.nbr_cams[j])
                self.cert[s] = self._chk(r.cert[j], torch.float32, (H, W), "cert").data_ptr()
                self.warp[s] = self._chk(r.warp[j], torch.float32, (H, W, ch), "warp").data_ptr()
                if r.mask_b is not None and r.mask_b[j] is not None:
                    self.mask_b[s] = self._chk(r.mask_b[j], torch.uint8, (h_match, w_match), "mask_b").data_ptr()
        self.fundamental = None
        if cameras is not None or any(r.fundamental is not None for r in self.refs):
            fund = np.zeros((n * k, 9), np.float32)
            for i, r in enumerate(self.refs):
                for j in range(len(r.cert)):
                    if r.fundamental is not None:
                        F = np.asarray(r.fundamental[j], np.float32)
                    elif cameras is not None:
                        a, b = cameras[int(r.ref_cam)], cameras[int(r.nbr_cams[j])]
                        F = fundamental_from_world2cam(a.K, a.R, a.t, b.K, b.R, b.t)
                    else:
                        raise ValueError("fundamental matrices must be given for every reference of a batch or for none")
                    fund[i * k + j] = np.asarray(F, np.float32).reshape(9)
            self.fundamental = np.ascontiguousarray(fund)
        self.axes = None
        if axes is not None:
            ax = self._chk(axes[0], torch.float32, (W,), "axis_x")
            ay = self._chk(axes[1], torch.float32, (H,), "axis_y")
            self.axes = (ax, ay)
        vp = C.POINTER(C.c_void_p)
        self.c = lfd_batch(
            n_refs=n, k=k, H=int(H), W=int(W), w_match=int(w_match), h_match=int(h_match), warp_channels=ch, reserved=0,
            fundamental=self.fundamental.ctypes.data_as(C.POINTER(C.c_float)) if self.fundamental is not None else None,
            ref_cam=C.cast(self.ref_cam, C.POINTER(C.c_int32)), n_slots=C.cast(self.n_slots, C.POINTER(C.c_int32)),
            nbr_cam=C.cast(self.nbr_cam, C.POINTER(C.c_int32)), cert=C.cast(self.cert, vp), warp=C.cast(self.warp, vp),
            image=C.cast(self.image, vp), mask_a=C.cast(self.mask_a, vp) if self.mask_a is not None else None,
            mask_b=C.cast(self.mask_b, vp) if self.mask_b is not None else None,
            axis_x=self.axes[0].data_ptr() if self.axes else None, axis_y=self.axes[1].data_ptr() if self.axes else None)

    def _chk(self, t: torch.Tensor, dtype, shape, what: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise ValueError(f"{what} must be a tensor on {self.device} (the path consumes RoMa's outputs in place, "
                             "all tensors of a batch on one device)")
        if t.dtype != dtype or tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what}: expected {dtype} {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
        if not t.is_contiguous():
            t = t.contiguous()
        self._keep.append(t)
        return t


_tls = threading.local()
# 1-3: what upstream's np.random.choice raises for (the reference drew nothing, there as here).  4-7: this implementation's own refusals -
# the reference did NOT do what upstream would have done with the stream, so a fused call of several references that reports one of them
# is void from that reference on (SELECT_VOIDS_STREAM; core/strategies.py::SampledLoop rolls the stream back and redoes the references).
_SELECT_ERRORS = {1: "probabilities contain NaN", 2: "probabilities are not non-negative", 3: "Fewer non-zero entries in p than size",
                  4: "selection: a weight is below 2^-29 (exact parallel cumsum not guaranteed)",
                  5: "selection made no progress (a bounded wait between its workgroups expired; nothing was committed to the random stream)",
                  6: "selection: too many coverage bins for the device stage",
                  7: "selection: more cells than the output has room for"}
SELECT_VOIDS_STREAM = frozenset((4, 5, 6, 7))
RNG_CHECKPOINTS = 4            # LFD_RNG_CHECKPOINTS of the header


def selection_error(status: int) -> str:
    """What upstream's sampling stage says (np.random.choice's ValueError texts) for a selection status of the fused calls."""
    return _SELECT_ERRORS.get(int(status), f"selection failed with status {int(status)}")


def _read_back_i32(t: torch.Tensor) -> np.ndarray:
    """Small int32 device tensor -> NumPy copy, staged through a per-thread pinned buffer (a pageable ``.cpu()`` costs
    ~2x as much and pinning a fresh buffer per call far more)."""
    n = int(t.numel())
    if not t.is_cuda:
        return t.numpy().copy()
    buf = getattr(_tls, "pinned_i32", None)
    if buf is None or buf.numel() < n:
        buf = torch.empty((max(n, 4096),), dtype=torch.int32).pin_memory()
        _tls.pinned_i32 = buf
    buf[:n].copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf[:n].numpy().copy()


class OutputBuffers:
    """Caller-owned survivor buffers (device).  Allocate once, reuse across launches."""

    def __init__(self, capacity: int, n_refs: int, k: int, device, with_cell: bool = True, with_segments: bool = True):
        """``with_cell``: also return the grid cell / neighbour slot of every survivor (debug previews);
        ``with_segments``: also count survivors per (reference, neighbour) - upstream's group sizes.  Both are
        optional outputs of the C-ABI; upstream's own result is xyz, rgb, err."""
        self.capacity = int(capacity)
        self.with_segments = with_segments
        cap = max(self.capacity, 1)
        # one allocation for the three float outputs (views below): the sampled mode's few thousand points then reach the
        # host in ONE copy (TriangulationOutput.host_arrays)
        self._f = torch.empty((cap * 7,), dtype=torch.float32, device=device)
        self.xyz = self._f[:cap * 3].view(cap, 3)
        self.rgb = self._f[cap * 3:cap * 6].view(cap, 3)
        self.err = self._f[cap * 6:]
        self.cell = torch.empty((cap,), dtype=torch.int32, device=device) if with_cell else None
        self.slot = torch.empty((cap,), dtype=torch.uint8, device=device) if with_cell else None
        # the small integer outputs share ONE buffer so that collect() needs a single device-to-host copy:
        # [ref_offsets i64 x (R+1)] [seg_counts i32 x R*k] [seg_order i32 x R*k]
        n_off, n_seg = 2 * (n_refs + 1), n_refs * k
        self._meta = torch.zeros((n_off + 2 * n_seg + 2 * n_refs + 1,), dtype=torch.int32, device=device)
        self._meta[n_off + n_seg:n_off + 2 * n_seg].fill_(-1)
        self.ref_offsets = self._meta[:n_off].view(torch.int64)
        self.seg_counts = self._meta[n_off:n_off + n_seg].view(n_refs, k)
        self.seg_order = self._meta[n_off + n_seg:n_off + 2 * n_seg].view(n_refs, k)
        self.sel_info = self._meta[n_off + 2 * n_seg:]           # lfd_triangulate_sampled[_multi]: {cells selected, selection status} per reference, then the launch status
        self._n_refs, self._k = n_refs, k
        self.c = lfd_points(xyz=self.xyz.data_ptr(), rgb=self.rgb.data_ptr(), err=self.err.data_ptr(),
                            cell=self.cell.data_ptr() if with_cell else None,
                            slot=self.slot.data_ptr() if with_cell else None, capacity=self.capacity)

    def begin_collect(self, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Enqueue the read-back of the small integer outputs (counts, selection / launch status) into this buffer's own pinned
        landing area and record an event behind it: ``collect`` then waits for THAT event only, not for whatever has been
        launched on the stream since (the next reference's kernels)."""
        if not self._meta.is_cuda:
            return
        if getattr(self, "_pinned_meta", None) is None:
            self._pinned_meta = torch.empty((int(self._meta.numel()),), dtype=torch.int32).pin_memory()
            self._meta_event = torch.cuda.Event()
        st = stream if stream is not None else torch.cuda.current_stream(self._meta.device)
        with torch.cuda.stream(st):
            self._pinned_meta.copy_(self._meta, non_blocking=True)
            self._meta_event.record(st)
        self._meta_pending = True

    def select_status(self, meta: np.ndarray) -> int:
        """Worst selection status over the references of a fused sampled call (0 = every selection went through)."""
        base = 2 * (self._n_refs + 1) + 2 * self._n_refs * self._k
        st = meta[base + 1:base + 2 * self._n_refs:2]
        bad = st[st != 0]
        return int(bad[0]) if bad.size else 0

    def collect(self, indexed: bool = False, check_selection: bool = False) -> TriangulationOutput:
        """Synchronise and trim to the number of survivors.  ``check_selection``: raise what upstream's sampling stage
        would have raised if the fused call's selection refused its input (any reference's; without it the caller reads
        ``sel_status`` reference by reference)."""
        if getattr(self, "_meta_pending", False):             # begin_collect() was called: wait for that copy alone
            self._meta_event.synchronize()
            self._meta_pending = False
            meta = self._pinned_meta.numpy().copy()
        else:
            meta = _read_back_i32(self._meta)                 # one copy through a cached pinned buffer (synchronises)
        if check_selection:
            st = self.select_status(meta)
            if st in (1, 2, 3):
                raise ValueError(_SELECT_ERRORS[st])
            if st == 4:
                raise SelectionInexact(_SELECT_ERRORS[4])
            if st != 0:
                raise HipBackendError(selection_error(st))
        n_off, n_seg = 2 * (self._n_refs + 1), self._n_refs * self._k
        offs = meta[:n_off].view(np.int64).copy()
        n = int(offs[-1])
        if n > self.capacity:
            raise HipBackendError(f"output capacity {self.capacity} too small for {n} survivors")
        return TriangulationOutput(
            xyz=self.xyz[:n], rgb=self.rgb[:n], err=self.err[:n],
            cell=self.cell[:n] if self.cell is not None else None, slot=self.slot[:n] if self.slot is not None else None,
            ref_offsets=offs, seg_counts=meta[n_off:n_off + n_seg].reshape(self._n_refs, self._k).copy(),
            seg_order=meta[n_off + n_seg:n_off + 2 * n_seg].reshape(self._n_refs, self._k).copy() if indexed else None,
            n_selected=int(meta[n_off + 2 * n_seg:n_off + 2 * n_seg + 2 * self._n_refs:2].sum()),
            sel_status=meta[n_off + 2 * n_seg + 1:n_off + 2 * n_seg + 2 * self._n_refs:2].copy(),
            launch_status=int(meta[n_off + 2 * n_seg + 2 * self._n_refs]), _packed=self._f,
            _cap=max(self.capacity, 1))



class HipDensifier:
    """One context = one GPU + one stream (``torch.cuda.current_stream`` of the device at creation,
    unless a stream is given).  Not thread-safe: use one per thread, as the C-ABI requires."""

    def __init__(self, device: Optional[torch.device] = None, stream: Optional[torch.cuda.Stream] = None):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        if not torch.cuda.is_available():
            raise HipBackendError("no GPU visible: the dense-initialisation hot path has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise HipBackendError(f"HipDensifier needs a cuda (HIP) device, got {self.device}")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        rc = self._lib.lfd_create(index, C.c_void_p(self.stream.cuda_stream), C.byref(self._ctx))
        if rc != 0:
            raise HipBackendError(f"lfd_create failed ({rc}): {self._lib.lfd_last_error(None).decode()}")
        self.n_cams = 0

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.lfd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise HipBackendError(f"{what} failed ({rc}): {self._lib.lfd_last_error(self._ctx).decode()}")

    def _same_device(self, batch: "PreparedBatch", out: Optional["OutputBuffers"] = None, *tensors) -> None:
        """Every pointer handed to the library must live where this context computes: a CPU batch (or one of another GPU) given
        to a device context would be a memory fault inside a kernel, a device batch given to the CPU twin a host segfault."""
        if torch.device(batch.device) != self.device:
            raise ValueError(f"batch lives on {batch.device}, this context computes on {self.device}")
        if out is not None and out.xyz.device != self.device:
            raise ValueError(f"output buffers live on {out.xyz.device}, this context computes on {self.device}")
        for t in tensors:
            if t is not None and t.device != self.device:
                raise ValueError(f"tensor on {t.device} handed to a context that computes on {self.device}")

    def upload_cameras(self, cams: Sequence[CameraRecord]) -> None:
        K = _f32(np.stack([np.asarray(c.K, np.float32).reshape(9) for c in cams]))
        R = _f32(np.stack([np.asarray(c.R, np.float32).reshape(9) for c in cams]))
        t = _f32(np.stack([np.asarray(c.t, np.float32).reshape(3) for c in cams]))
        P = _f32(np.stack([np.asarray(c.P, np.float32).reshape(12) for c in cams]))
        Cc = _f32(np.stack([np.asarray(c.C, np.float32).reshape(3) for c in cams]))
        wh = np.ascontiguousarray(np.array([[c.width, c.height] for c in cams], np.int32))
        self._check(self._lib.lfd_upload_cameras(self._ctx, len(cams), _fp(K), _fp(R), _fp(t), _fp(P), _fp(Cc),
                                                 wh.ctypes.data_as(C.POINTER(C.c_int32))), "lfd_upload_cameras")
        self.n_cams = len(cams)

    def pair_fundamentals(self, n_refs: int, k: int) -> np.ndarray:
        """(n_refs, k, 3, 3) f64: the fundamental matrices the kernels of the last prepared batch used (debug read-back)."""
        out = np.zeros((n_refs * k, 9), np.float64)
        self._check(self._lib.lfd_get_pair_fundamental(self._ctx, n_refs * k, out.ctypes.data_as(C.POINTER(C.c_double))),
                    "lfd_get_pair_fundamental")
        return out.reshape(n_refs, k, 3, 3)

    def reload_env(self) -> None:
        """Re-read the profiling switches of the environment (they are read at creation only, lfd_reload_env)."""
        self._check(self._lib.lfd_reload_env(self._ctx), "lfd_reload_env")

    def time_dense_kernels(self, n_launches: int) -> None:
        """The next ``n_launches`` dense launches carry start / stop events of their own (lfd_kernel_timing): the kernel's device-side
        duration, as a kernel trace reports it.  0 switches the timing off."""
        self._check(self._lib.lfd_kernel_timing(self._ctx, int(n_launches)), "lfd_kernel_timing")

    def dense_kernel_times_ms(self) -> np.ndarray:
        """Durations (ms, launch order) of the dense launches timed since ``time_dense_kernels`` / the last read; waits for the last one."""
        cap = 1 << 20
        n = C.c_int32(0)
        buf = np.zeros(cap, np.float32)
        self._check(self._lib.lfd_kernel_timing_read(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), "lfd_kernel_timing_read")
        return buf[:n.value].astype(np.float64)

    def check_launches(self) -> None:
        """Synchronise and raise if a kernel reported a look-back timeout."""
        st = C.c_int32(0)
        self._check(self._lib.lfd_launch_status(self._ctx, C.byref(st)), "lfd_launch_status")

    # -- N1: file payloads packed on the device ------------------------------------------------------------
    @staticmethod
    def _pts(t: torch.Tensor, cols: int, what: str) -> torch.Tensor:
        if t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError(f"{what} must be a float32 device tensor")
        t = t.contiguous()
        if cols and (t.dim() != 2 or t.shape[1] != cols):
            raise ValueError(f"{what} must have shape (n,{cols})")
        return t

    def pack_ply(self, xyz: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
        """(n*15,) u8 device tensor: the PLY body upstream's write_ply emits after its header."""
        xyz, rgb = self._pts(xyz, 3, "xyz"), self._pts(rgb, 3, "rgb")
        n = int(xyz.shape[0])
        out = torch.empty((max(n * 15, 4),), dtype=torch.uint8, device=xyz.device)
        self._check(self._lib.lfd_pack_ply(self._ctx, xyz.data_ptr(), rgb.data_ptr(), n, out.data_ptr()), "lfd_pack_ply")
        return out[:n * 15]

    def pack_points3d(self, xyz: torch.Tensor, rgb: torch.Tensor, err: Optional[torch.Tensor] = None,
                      id_base: int = 0) -> torch.Tensor:
        """(n*43,) u8 device tensor: upstream's points3D.bin body (without the leading u64 count)."""
        xyz, rgb = self._pts(xyz, 3, "xyz"), self._pts(rgb, 3, "rgb")
        e = self._pts(err, 0, "err") if err is not None else None
        n = int(xyz.shape[0])
        out = torch.empty((max(n * 43, 4),), dtype=torch.uint8, device=xyz.device)
        self._check(self._lib.lfd_pack_points3d(self._ctx, xyz.data_ptr(), rgb.data_ptr(), e.data_ptr() if e is not None else None,
                                                n, int(id_base), out.data_ptr()), "lfd_pack_points3d")
        return out[:n * 43]

    def quantise_rgb(self, rgb: torch.Tensor) -> torch.Tensor:
        rgb = self._pts(rgb, 3, "rgb")
        out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
        self._check(self._lib.lfd_quantise_rgb(self._ctx, rgb.data_ptr(), int(rgb.shape[0]), out.data_ptr()), "lfd_quantise_rgb")
        return out

    # -- N3: image preparation on the device -----------------------------------------------------------------
    def prepare_image(self, rgb: torch.Tensor, size_wh, mask01: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``Image.resize(size_wh, BILINEAR)`` of a decoded (h, w, 3) u8 device image, then masked pixels black
        (upstream core/image_utils.py:69-91), bit for bit like Pillow.  Returns a (h_out, w_out, 3) u8 device tensor."""
        if rgb.dtype != torch.uint8 or not rgb.is_cuda or rgb.dim() != 3 or rgb.shape[2] != 3:
            raise ValueError("rgb must be a (h, w, 3) uint8 device tensor")
        rgb = rgb.contiguous()
        w_out, h_out = int(size_wh[0]), int(size_wh[1])
        if mask01 is not None:
            if mask01.dtype != torch.uint8 or not mask01.is_cuda or tuple(mask01.shape) != (h_out, w_out):
                raise ValueError("mask01 must be a (h_out, w_out) uint8 device tensor")
            mask01 = mask01.contiguous()
        out = torch.empty((h_out, w_out, 3), dtype=torch.uint8, device=rgb.device)
        self._check(self._lib.lfd_prepare_image(self._ctx, rgb.data_ptr(), int(rgb.shape[1]), int(rgb.shape[0]), w_out, h_out,
                                                mask01.data_ptr() if mask01 is not None else None, out.data_ptr()), "lfd_prepare_image")
        return out

    def prepare_mask(self, mask_l: torch.Tensor, size_wh, threshold: float = 0.5, invert: bool = False) -> torch.Tensor:
        """``load_mask_resized_np`` after the "L" conversion: NEAREST resize + threshold (core/image_utils.py:40-66)."""
        if mask_l.dtype != torch.uint8 or not mask_l.is_cuda or mask_l.dim() != 2:
            raise ValueError("mask_l must be a (h, w) uint8 device tensor")
        mask_l = mask_l.contiguous()
        w_out, h_out = int(size_wh[0]), int(size_wh[1])
        out = torch.empty((h_out, w_out), dtype=torch.uint8, device=mask_l.device)
        self._check(self._lib.lfd_prepare_mask(self._ctx, mask_l.data_ptr(), int(mask_l.shape[1]), int(mask_l.shape[0]), w_out, h_out,
                                               C.c_float(threshold), 1 if invert else 0, out.data_ptr()), "lfd_prepare_mask")
        return out

    # -- S: selection stage on the device ----------------------------------------------------------------
    def seed_rng(self, seed: int) -> None:
        """Seed the context's legacy MT19937 stream like ``np.random.seed(seed)``."""
        self._check(self._lib.lfd_rng_seed(self._ctx, C.c_uint32(int(seed) & 0xFFFFFFFF)), "lfd_rng_seed")

    def rng_state(self):
        key = (C.c_uint32 * 624)()
        pos = C.c_int32(0)
        self._check(self._lib.lfd_rng_get_state(self._ctx, key, C.byref(pos)), "lfd_rng_get_state")
        return np.frombuffer(key, dtype=np.uint32).copy(), int(pos.value)

    def set_rng_state(self, key: np.ndarray, pos: int) -> None:
        k = np.ascontiguousarray(key, dtype=np.uint32)
        self._check(self._lib.lfd_rng_set_state(self._ctx, k.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos)),
                    "lfd_rng_set_state")

    def checkpoint_rng(self, place: int) -> None:
        """The stream put aside on the device, in stream order (no host wait): lfd_rng_checkpoint."""
        self._check(self._lib.lfd_rng_checkpoint(self._ctx, int(place)), "lfd_rng_checkpoint")

    def rollback_rng(self, place: int) -> None:
        """... and taken back: the stream continues from where ``checkpoint_rng(place)`` saw it."""
        self._check(self._lib.lfd_rng_rollback(self._ctx, int(place)), "lfd_rng_rollback")

    def select_samples(self, best_cert: torch.Tensor, M: int, cap: float = 0.9, border: int = 2, tiles: int = 24,
                       s_override: float = 0.0) -> torch.Tensor:
        """Coverage sampling (filter mode) of one reference's aggregated certainty map on the device;
        returns the selected cells (int64 device tensor, ascending).  Raises ValueError in the cases
        upstream's ``np.random.choice`` does."""
        if best_cert.dtype != torch.float32 or not best_cert.is_cuda or best_cert.dim() != 2:
            raise ValueError("best_cert must be a 2-D float32 device tensor")
        bc = best_cert.contiguous()
        H, W = bc.shape
        cap_n = int(M) + int(tiles) * int(tiles) + 64
        out = torch.empty((cap_n,), dtype=torch.int64, device=bc.device)
        n = C.c_int32(0)
        st = C.c_int32(0)
        rc = self._lib.lfd_select_samples(self._ctx, bc.data_ptr(), H, W, int(M), C.c_float(cap), int(border), int(tiles),
                                          C.c_float(s_override), out.data_ptr(), cap_n, C.byref(n), C.byref(st))
        if rc != 0 and st.value in (1, 2, 3):
            raise ValueError(self._lib.lfd_last_error(self._ctx).decode().replace("selection: ", ""))
        if rc != 0 and st.value == 4:
            raise SelectionInexact(self._lib.lfd_last_error(self._ctx).decode())
        self._check(rc, "lfd_select_samples")
        return out[:int(n.value)]

    TOP_M_MAX = 16384

    def select_top_m(self, best_cert: torch.Tensor, M: int, cap: float = 0.9) -> torch.Tensor:
        """no_filter selection: the M largest capped certainties, descending (ties by cell index)."""
        if best_cert.dtype != torch.float32 or not best_cert.is_cuda or best_cert.dim() != 2:
            raise ValueError("best_cert must be a 2-D float32 device tensor")
        bc = best_cert.contiguous()
        H, W = bc.shape
        cap_n = max(min(int(M), H * W), 1)
        out = torch.empty((cap_n,), dtype=torch.int64, device=bc.device)
        n, st = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.lfd_select_top_m(self._ctx, bc.data_ptr(), H, W, int(M), C.c_float(cap), out.data_ptr(), cap_n,
                                               C.byref(n), C.byref(st)), "lfd_select_top_m")
        return out[:int(n.value)]

    # -- launches (asynchronous on self.stream) -------------------------------------------------------
    def prepare(self, batch: PreparedBatch, params: lfd_params) -> None:
        """Upload the batch's descriptor tables and derive its per-pair constants now (lfd_prepare_batch); the launch that follows
        for the same batch then finds them in place."""
        self._same_device(batch)
        self._check(self._lib.lfd_prepare_batch(self._ctx, C.byref(batch.c), C.byref(params)), "lfd_prepare_batch")

    def launch_aggregate(self, batch: PreparedBatch, params: lfd_params, best_cert: torch.Tensor,
                         best_slot: Optional[torch.Tensor]) -> None:
        self._same_device(batch, None, best_cert, best_slot)
        self._check(self._lib.lfd_aggregate(self._ctx, C.byref(batch.c), C.byref(params), best_cert.data_ptr(),
                                            best_slot.data_ptr() if best_slot is not None else None), "lfd_aggregate")

    def launch_dense(self, batch: PreparedBatch, params: lfd_params, out: OutputBuffers) -> None:
        self._same_device(batch, out)
        self._check(self._lib.lfd_triangulate_dense(self._ctx, C.byref(batch.c), C.byref(params), C.byref(out.c),
                                                    out.ref_offsets.data_ptr(),
                                                    out.seg_counts.data_ptr() if out.with_segments else None),
                    "lfd_triangulate_dense")

    # -- the file payload straight from the kernel ----------------------------------------------------------------------------------------
    def launch_dense_ply(self, batch: PreparedBatch, params: lfd_params, records: torch.Tensor, ref_offsets: torch.Tensor,
                         seg_counts: Optional[torch.Tensor] = None) -> None:
        """lfd_triangulate_dense_ply: ``records`` (uint8, capacity * 15) receives the 15-byte PLY vertex records of the survivors in raster
        order per reference; ``ref_offsets`` (int64, n_refs + 1) their exclusive prefix.  Asynchronous."""
        self._same_device(batch, None, records, ref_offsets, seg_counts)
        if records.dtype != torch.uint8 or ref_offsets.dtype != torch.int64 or not records.is_contiguous():
            raise ValueError("records must be a contiguous uint8 tensor, ref_offsets int64")
        self._check(self._lib.lfd_triangulate_dense_ply(self._ctx, C.byref(batch.c), C.byref(params), records.data_ptr(), int(records.numel()) // 15,
                                                        ref_offsets.data_ptr(), seg_counts.data_ptr() if seg_counts is not None else None, None, None),
                    "lfd_triangulate_dense_ply")

    def triangulate_dense_ply(self, batch: PreparedBatch, params: lfd_params):
        """(PLY body as a uint8 device tensor, ref_offsets host int64): what pack_ply makes of triangulate_dense's result, from one kernel."""
        cap = batch.n_refs * batch.H * batch.W
        rec = torch.empty((max(cap * 15, 4),), dtype=torch.uint8, device=self.device)
        offs = torch.zeros((batch.n_refs + 1,), dtype=torch.int64, device=self.device)
        self.launch_dense_ply(batch, params, rec, offs)
        self.check_launches()
        h = offs.cpu().numpy()
        return rec[:int(h[-1]) * 15], h

    def launch_dense_ply_segments(self, batch: PreparedBatch, params: lfd_params, records: torch.Tensor, ref_counts: torch.Tensor, table: torch.Tensor,
                                  seg_counts: Optional[torch.Tensor] = None) -> None:
        """lfd_triangulate_dense_ply_segments: the PLY records without a look-back - reference r's ``ref_counts[r]`` records start at byte
        ``15 * r * H * W`` of ``records`` (uint8, n_refs * H * W * 15), tile after tile in retirement order; ``table`` (int32 (n_tiles, 2)) says where
        each tile went.  Asynchronous."""
        self._same_device(batch, None, records, ref_counts, table, seg_counts)
        if records.dtype != torch.uint8 or ref_counts.dtype != torch.int64 or table.dtype != torch.int32 or not records.is_contiguous() or not table.is_contiguous():
            raise ValueError("records must be a contiguous uint8 tensor, ref_counts int64, table a contiguous int32 (n_tiles, 2) tensor")
        self._check(self._lib.lfd_triangulate_dense_ply_segments(self._ctx, C.byref(batch.c), C.byref(params), records.data_ptr(), int(records.numel()) // 15,
                                                                 ref_counts.data_ptr(), seg_counts.data_ptr() if seg_counts is not None else None, table.data_ptr()),
                    "lfd_triangulate_dense_ply_segments")

    def tiles_per_ref(self, H: int, W: int) -> int:
        return int(self._lib.lfd_dense_tiles_per_ref(int(H), int(W)))

    # -- unordered retirement (opt-in): tiles claim room with one atomic, the consumers restore raster order from the tile table ------------
    def launch_dense_segments(self, batch: PreparedBatch, params: lfd_params, out: OutputBuffers, table: torch.Tensor,
                              ref_counts: torch.Tensor) -> None:
        self._same_device(batch, out, table, ref_counts)
        if table.dtype != torch.int32 or ref_counts.dtype != torch.int64 or not table.is_contiguous():
            raise ValueError("table must be a contiguous int32 (n_tiles, 2) tensor, ref_counts int64 (n_refs,)")
        self._check(self._lib.lfd_triangulate_dense_segments(self._ctx, C.byref(batch.c), C.byref(params), C.byref(out.c), ref_counts.data_ptr(),
                                                             out.seg_counts.data_ptr() if out.with_segments else None, table.data_ptr()),
                    "lfd_triangulate_dense_segments")

    def triangulate_dense_segments(self, batch: PreparedBatch, params: lfd_params, with_cell: bool = True) -> SegmentedOutput:
        out = OutputBuffers(batch.n_refs * batch.H * batch.W, batch.n_refs, batch.k, self.device, with_cell)
        tpr = int(self._lib.lfd_dense_tiles_per_ref(batch.H, batch.W))
        table = torch.zeros((batch.n_refs * tpr, 2), dtype=torch.int32, device=self.device)
        counts = torch.zeros((batch.n_refs,), dtype=torch.int64, device=self.device)
        self.launch_dense_segments(batch, params, out, table, counts)
        self.check_launches()
        return SegmentedOutput(out, table, counts, batch.n_refs, batch.H, batch.W, batch.k)

    def order_segments(self, seg: SegmentedOutput, into: Optional[OutputBuffers] = None) -> TriangulationOutput:
        """The ordered structure-of-arrays result of a segmented launch: what lfd_triangulate_dense would have returned, bit for bit."""
        src = seg.buffers
        dst = into if into is not None else OutputBuffers(src.capacity, seg.n_refs, seg.k, self.device, with_cell=src.cell is not None)
        self._check(self._lib.lfd_order_segments(self._ctx, seg.n_refs, seg.H, seg.W, seg.table.data_ptr(), C.byref(src.c), C.byref(dst.c),
                                                 dst.ref_offsets.data_ptr()), "lfd_order_segments")
        dst.seg_counts.copy_(src.seg_counts)
        return dst.collect()

    def pack_ply_segments(self, seg: SegmentedOutput):
        """(PLY body in raster order as a u8 device tensor, ref_offsets (n_refs + 1,) host i64) straight from the unordered buffers."""
        cap = seg.buffers.capacity
        out = torch.empty((max(cap * 15, 4),), dtype=torch.uint8, device=self.device)
        offs = torch.zeros((seg.n_refs + 1,), dtype=torch.int64, device=self.device)
        self._check(self._lib.lfd_pack_ply_segments(self._ctx, seg.n_refs, seg.H, seg.W, seg.table.data_ptr(), seg.buffers.xyz.data_ptr(),
                                                    seg.buffers.rgb.data_ptr(), cap, out.data_ptr(), offs.data_ptr()), "lfd_pack_ply_segments")
        h = offs.cpu().numpy()
        return out[:int(h[-1]) * 15], h

    def pack_points3d_segments(self, seg: SegmentedOutput, id_base: int = 0):
        cap = seg.buffers.capacity
        out = torch.empty((max(cap * 43, 4),), dtype=torch.uint8, device=self.device)
        offs = torch.zeros((seg.n_refs + 1,), dtype=torch.int64, device=self.device)
        self._check(self._lib.lfd_pack_points3d_segments(self._ctx, seg.n_refs, seg.H, seg.W, seg.table.data_ptr(), seg.buffers.xyz.data_ptr(),
                                                         seg.buffers.rgb.data_ptr(), seg.buffers.err.data_ptr(), cap, int(id_base), out.data_ptr(),
                                                         offs.data_ptr()), "lfd_pack_points3d_segments")
        h = offs.cpu().numpy()
        return out[:int(h[-1]) * 43], h

    def launch_indexed(self, batch: PreparedBatch, params: lfd_params, sel_idx: torch.Tensor,
                       sel_offsets: Sequence[int], out: OutputBuffers) -> None:
        if sel_idx.dtype != torch.int64 or not sel_idx.is_cuda or not sel_idx.is_contiguous():
            raise ValueError("sel_idx must be a contiguous int64 device tensor")
        self._same_device(batch, out, sel_idx)
        offs = (C.c_int64 * (batch.n_refs + 1))(*[int(v) for v in sel_offsets])
        self._check(self._lib.lfd_triangulate_indexed(self._ctx, C.byref(batch.c), C.byref(params), sel_idx.data_ptr(),
                                                      offs, C.byref(out.c), out.ref_offsets.data_ptr(),
                                                      out.seg_counts.data_ptr(), out.seg_order.data_ptr()),
                    "lfd_triangulate_indexed")

    def launch_sampled(self, batch: PreparedBatch, params: lfd_params, M: int, out: OutputBuffers, cap: float = 0.9,
                       border: int = 2, tiles: int = 24, s_override: float = 0.0, sel_cells: Optional[torch.Tensor] = None) -> None:
        """One reference view through aggregate -> selection -> indexed triangulation in one asynchronous call
        (lfd_triangulate_sampled): no read-back in between, the selection count stays on the device."""
        self._same_device(batch, out, sel_cells)
        self._check(self._lib.lfd_triangulate_sampled(self._ctx, C.byref(batch.c), C.byref(params), int(M), C.c_float(cap), int(border),
                                                      int(tiles), C.c_float(s_override), C.byref(out.c), out.ref_offsets.data_ptr(),
                                                      out.seg_counts.data_ptr(), out.seg_order.data_ptr(), out.sel_info.data_ptr(),
                                                      sel_cells.data_ptr() if sel_cells is not None else None),
                    "lfd_triangulate_sampled")

    def launch_sampled_multi(self, batch: PreparedBatch, params: lfd_params, M: int, out: OutputBuffers, seeds: Sequence[int],
                             cap: float = 0.9, border: int = 2, tiles: int = 24) -> None:
        """Several reference views through the fused call at once, each on its own MT19937 stream (``seeds[r]``, like
        ``np.random.seed``): lfd_triangulate_sampled_multi.  ``out`` needs capacity n_refs * (M + tiles*tiles + 64)."""
        if len(seeds) != batch.n_refs:
            raise ValueError("one seed per reference")
        self._same_device(batch, out)
        arr = (C.c_uint32 * batch.n_refs)(*[int(v) & 0xFFFFFFFF for v in seeds])
        self._check(self._lib.lfd_triangulate_sampled_multi(self._ctx, C.byref(batch.c), C.byref(params), int(M), C.c_float(cap), int(border),
                                                            int(tiles), arr, C.byref(out.c), out.ref_offsets.data_ptr(),
                                                            out.seg_counts.data_ptr(), out.seg_order.data_ptr(), out.sel_info.data_ptr(), None),
                    "lfd_triangulate_sampled_multi")

    def launch_sampled_chain(self, batch: PreparedBatch, params: lfd_params, M: int, out: OutputBuffers,
                             s_overrides: Optional[Sequence[float]] = None, cap: float = 0.9, border: int = 2, tiles: int = 24,
                             sel_cells: Optional[torch.Tensor] = None) -> None:
        """Several reference views through the fused call at once on the context's ONE MT19937 stream, consumed in batch order
        (lfd_triangulate_sampled_chain): the results and the stream afterwards are those of ``n_refs`` successive ``launch_sampled`` calls.
        ``s_overrides[r]`` > 0: upstream's own normaliser of reference r.  ``out`` needs capacity n_refs * (M + tiles*tiles + 64)."""
        if s_overrides is not None and len(s_overrides) != batch.n_refs:
            raise ValueError("one normaliser per reference")
        self._same_device(batch, out, sel_cells)
        arr = (C.c_float * batch.n_refs)(*[float(v) for v in s_overrides]) if s_overrides is not None else None
        self._check(self._lib.lfd_triangulate_sampled_chain(self._ctx, C.byref(batch.c), C.byref(params), int(M), C.c_float(cap), int(border),
                                                            int(tiles), arr, C.byref(out.c), out.ref_offsets.data_ptr(),
                                                            out.seg_counts.data_ptr(), out.seg_order.data_ptr(), out.sel_info.data_ptr(),
                                                            sel_cells.data_ptr() if sel_cells is not None else None),
                    "lfd_triangulate_sampled_chain")

    def triangulate_sampled(self, batch: PreparedBatch, params: lfd_params, M: int, cap: float = 0.9, border: int = 2,
                            tiles: int = 24, s_override: float = 0.0, with_cell: bool = True) -> TriangulationOutput:
        out = OutputBuffers(int(M) + int(tiles) * int(tiles) + 64, 1, batch.k, self.device, with_cell)
        self.launch_sampled(batch, params, M, out, cap, border, tiles, s_override)
        with torch.cuda.stream(self.stream):                       # the read-back is ordered after the launch on ITS stream
            res = out.collect(indexed=True, check_selection=True)  # one copy: counts, selection status, launch status
        if res.launch_status != 0:
            self.check_launches()                                  # resets the device word and raises
        return res

    # -- convenience wrappers (synchronising) -------------------------------------------------------------
    def aggregate(self, batch: PreparedBatch, params: lfd_params):
        best = torch.empty((batch.n_refs, batch.H, batch.W), dtype=torch.float32, device=self.device)
        slot = torch.empty((batch.n_refs, batch.H, batch.W), dtype=torch.uint8, device=self.device)
        self.launch_aggregate(batch, params, best, slot)
        return best, slot

    def triangulate_dense(self, batch: PreparedBatch, params: lfd_params, capacity: Optional[int] = None,
                          with_cell: bool = True) -> TriangulationOutput:
        cap = batch.n_refs * batch.H * batch.W if capacity is None else int(capacity)
        out = OutputBuffers(cap, batch.n_refs, batch.k, self.device, with_cell)
        self.launch_dense(batch, params, out)
        self.check_launches()
        return out.collect()

    def triangulate_indexed(self, batch: PreparedBatch, params: lfd_params, sel_idx: torch.Tensor,
                            sel_offsets: Sequence[int], with_cell: bool = True) -> TriangulationOutput:
        out = OutputBuffers(int(sel_offsets[-1]), batch.n_refs, batch.k, self.device, with_cell)
        self.launch_indexed(batch, params, sel_idx, sel_offsets, out)
        self.check_launches()
        return out.collect(indexed=True)


class HostDensifier:
    """The CPU twin of :class:`HipDensifier` (``lfd_create_host`` + the ``*_host`` entry points of the C-ABI): the host
    build of the kernels' per-cell source over CPU tensors, on ``n_threads`` threads.  Explicitly chosen - for upstream's
    CPU-only configuration, for the CPU baseline of the benchmark and for parity checks - never a fallback: a
    :class:`HipDensifier` does not turn into one when the GPU is missing."""

    def __init__(self, n_threads: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.lfd_create_host(int(n_threads), C.byref(self._ctx))
        if rc != 0:
            raise HipBackendError(f"lfd_create_host failed ({rc}): {self._lib.lfd_last_error(None).decode()}")
        self.device = torch.device("cpu")
        self.n_threads = int(self._lib.lfd_host_threads(self._ctx))
        self.n_cams = 0

    close = HipDensifier.close
    __del__ = HipDensifier.__del__
    _check = HipDensifier._check
    _same_device = HipDensifier._same_device
    upload_cameras = HipDensifier.upload_cameras

    def check_launches(self) -> None:
        pass

    def aggregate(self, batch: PreparedBatch, params: lfd_params):
        self._same_device(batch)
        best = torch.empty((batch.n_refs, batch.H, batch.W), dtype=torch.float32)
        slot = torch.empty((batch.n_refs, batch.H, batch.W), dtype=torch.uint8)
        self._check(self._lib.lfd_aggregate_host(self._ctx, C.byref(batch.c), C.byref(params), best.data_ptr(), slot.data_ptr()),
                    "lfd_aggregate_host")
        return best, slot

    def triangulate_dense(self, batch: PreparedBatch, params: lfd_params, capacity: Optional[int] = None,
                          with_cell: bool = True) -> TriangulationOutput:
        self._same_device(batch)
        cap = batch.n_refs * batch.H * batch.W if capacity is None else int(capacity)
        out = OutputBuffers(cap, batch.n_refs, batch.k, self.device, with_cell)
        self._check(self._lib.lfd_triangulate_dense_host(self._ctx, C.byref(batch.c), C.byref(params), C.byref(out.c),
                                                         out.ref_offsets.data_ptr(), out.seg_counts.data_ptr()),
                    "lfd_triangulate_dense_host")
        return out.collect()

    def triangulate_indexed(self, batch: PreparedBatch, params: lfd_params, sel_idx: torch.Tensor,
                            sel_offsets: Sequence[int], with_cell: bool = True) -> TriangulationOutput:
        if sel_idx.dtype != torch.int64 or sel_idx.is_cuda or not sel_idx.is_contiguous():
            raise ValueError("sel_idx must be a contiguous int64 CPU tensor")
        self._same_device(batch)
        out = OutputBuffers(int(sel_offsets[-1]), batch.n_refs, batch.k, self.device, with_cell)
        offs = (C.c_int64 * (batch.n_refs + 1))(*[int(v) for v in sel_offsets])
        self._check(self._lib.lfd_triangulate_indexed_host(self._ctx, C.byref(batch.c), C.byref(params), sel_idx.data_ptr(), offs,
                                                           C.byref(out.c), out.ref_offsets.data_ptr(), out.seg_counts.data_ptr(),
                                                           out.seg_order.data_ptr()), "lfd_triangulate_indexed_host")
        return out.collect(indexed=True)
