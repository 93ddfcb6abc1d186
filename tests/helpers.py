"""Shared helpers for the parity tests (fixture decoding, oracle camera blocks, guard bands)."""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import densify_oracle as orc  # noqa: E402  (tests are allowed to use the oracle)


def oracle_cams(g, prefix="cam_"):
    out = []
    for i in range(g[prefix + "K"].shape[0]):
        out.append(orc.OracleCamera(K=g[prefix + "K"][i], R=g[prefix + "R"][i], t=g[prefix + "t"][i].reshape(3, 1),
                                    P=g[prefix + "P"][i], C=g[prefix + "C"][i],
                                    width=int(g[prefix + "wh"][i, 0]), height=int(g[prefix + "wh"][i, 1])))
    return out


def g3_case(g3, name):
    pre = name + "_"
    H, W, wm, hm, ref, k = [int(v) for v in g3[pre + "dims"]]
    cfg = json.loads(str(g3[pre + "cfg"]))
    params = orc.OracleParams(certainty_thresh=cfg["certainty_thresh"], reproj_thresh=cfg["reproj_thresh"],
                              sampson_thresh=cfg["sampson_thresh"], min_parallax_deg=cfg["min_parallax_deg"],
                              no_filter=bool(cfg["no_filter"]), matches_per_ref=int(cfg["matches_per_ref"]))
    nbrs = [int(n) for n in g3[pre + "nbrs"]]
    mask_a = g3[pre + "mask_a"] if (pre + "mask_a") in g3.files else None
    masks_b = None
    if mask_a is not None:
        masks_b = [g3[pre + f"mask_b{j}"] if (pre + f"mask_b{j}") in g3.files else None for j in range(k)]
    case = dict(name=name, H=H, W=W, w_match=wm, h_match=hm, ref=ref, k=k, params=params, nbrs=nbrs,
                warp=g3[pre + "warp"], cert=g3[pre + "cert"], image=g3[pre + "image"], sel=g3[pre + "sel"],
                post_cert=g3[pre + "post_cert"], mask_a=mask_a, masks_b=masks_b, none=(pre + "none") in g3.files)
    if not case["none"]:
        for key in ("xyz", "rgb", "err", "seg_nbr_cam", "seg_count", "dbg_matches", "dbg_cert"):
            case[key] = g3[pre + key]
    return case


def guard_band_ok(diag, params, eps_sampson=1e-6, eps_err=2e-3, eps_par=5e-3, eps_z_rel=1e-4):
    """Cells whose decision variables are all farther than epsilon from their thresholds: for those,
    two correct implementations of the f32 DLT must agree on keep/reject."""
    ok = np.ones(diag["err"].shape, bool)
    if not params.no_filter:
        if params.sampson_thresh > 0:
            ok &= np.abs(diag["sampson"] - params.sampson_thresh) > eps_sampson * max(1.0, params.sampson_thresh)
        passed_s = (diag["sampson"] < params.sampson_thresh) if params.sampson_thresh > 0 else np.ones_like(ok)
        near = np.abs(diag["err"] - np.float32(params.reproj_thresh)) <= eps_err + 4.0 * diag["err_noise"]
        near |= np.abs(diag["z1"]) <= eps_z_rel * np.maximum(1.0, np.abs(diag["z1"]))
        near |= np.abs(diag["z2"]) <= eps_z_rel * np.maximum(1.0, np.abs(diag["z2"]))
        if params.min_parallax_deg > 0:
            near |= np.abs(diag["parallax_deg"] - np.float32(params.min_parallax_deg)) <= eps_par
        near |= ~np.isfinite(diag["err"])
        ok &= ~(near & passed_s)
    return ok


def oracle_cam(c):
    return orc.OracleCamera(K=c.K, R=c.R, t=c.t, P=c.P, C=c.C, width=c.width, height=c.height)


def flip_report(kept_cells, sref, cams, w_match, h_match, params, axes, sample=None, sampson_rel=1e-12, seed=0, masks=(None, None)):
    """Cells of one reference that an implementation (``kept_cells`` = the grid cells it kept, dense mode) decides
    differently from the oracle, each classified by the threshold that explains it (``orc.classify_flips``: the band of
    every reject reason is derived from a stated rounding-noise model, see its docstring).

    ``sample``: compare on that many random cells only (the oracle's LAPACK path costs ~4 us per cell).
    Returns dict(cells=compared, flipped=n, out_of_band=n, by_reason={reason: n}, oob_cells=[...], oracle=...)."""
    k = len(sref.nbr_indices)
    H, W = sref.cert.shape[1:]
    certs = [sref.cert[j].cpu().numpy() for j in range(k)]
    warps = [sref.warp[j].cpu().numpy() for j in range(k)]
    ca, cbs = oracle_cam(cams[sref.ref_index]), [oracle_cam(cams[n]) for n in sref.nbr_indices]
    with np.errstate(all="ignore"):
        best_cert, best_k, agg = orc.prepare_reference(certs, warps, params, masks[0], masks[1])
        if sample is None or sample >= H * W:
            cells = np.arange(H * W, dtype=np.int64)
        else:
            cells = np.sort(np.random.RandomState(seed).choice(H * W, size=int(sample), replace=False)).astype(np.int64)
        cand = cells[~(best_cert.reshape(-1)[cells] <= 0)]      # dense mode's candidates: best certainty not <= 0 (orc.triangulate_dense)
        res = orc.triangulate_selected(cand, best_cert, best_k, agg, sref.image.cpu().numpy(), ca, cbs, w_match, h_match,
                                       params, axes=axes)
    keep_orc = np.zeros(H * W, bool)
    keep_orc[res.cell] = True
    keep_impl = np.zeros(H * W, bool)
    keep_impl[np.asarray(kept_cells, np.int64)] = True
    flipped = cells[keep_orc[cells] != keep_impl[cells]]
    reason, _ = orc.classify_flips(flipped, best_k, agg, ca, cbs, w_match, h_match, params, axes=axes, sampson_rel=sampson_rel)
    by = {name: int((reason == i).sum()) for i, name in enumerate(orc.FLIP_REASONS)}
    return dict(cells=int(cells.size), flipped=int(flipped.size), out_of_band=int((reason < 0).sum()), by_reason=by,
                oob_cells=flipped[reason < 0].tolist(), oracle=res, compared=cells, best_k=best_k, agg=agg, cam_a=ca, cams_b=cbs)
