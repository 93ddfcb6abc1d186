"""RoMa-v2 producer of the hot path's inputs (upstream core/matcher.py:74-211).

RoMa-v2 itself is a third-party model (DINOv3 ViT-L/16 + match transformer + DPT head + conv
refiners) and runs unmodified on PyTorch-ROCm; it is not re-implemented here.  This wrapper keeps
upstream's interface - ``w_resized / h_resized / sample_thresh / match_grids_batch / close`` - with
two MI355X-side differences: the outputs STAY on the GPU (the HIP kernels read them in place, there
is no device-to-host copy), and the warp can be handed over as the 2-channel ``warp_AB`` together
with the A-grid axes instead of being concatenated with a materialised identity grid.
"""
from __future__ import annotations

import gc
import os
import sys
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .hostlog import log

ROMA_WEIGHTS_FILE = "romav2.pt"

# name -> (H_lr, W_lr, H_hr, W_hr, bidirectional); "high" is the plugin's own preset
# (upstream core/matcher.py:82-87), the others are RoMaV2.apply_setting presets
PLUGIN_PRESETS = {"high": (640, 640, 960, 960, True)}


def romav2_cached_weights_paths() -> List[str]:
    roots: List[str] = []
    try:
        roots.append(os.path.join(torch.hub.get_dir(), "checkpoints"))
    except Exception:
        pass
    if os.getenv("TORCH_HOME"):
        roots.append(os.path.join(os.path.expanduser(os.environ["TORCH_HOME"]), "hub", "checkpoints"))
    roots.append(os.path.join(os.path.expanduser(os.getenv("XDG_CACHE_HOME", "~/.cache")), "torch", "hub", "checkpoints"))
    seen, out = set(), []
    for r in roots:
        key = os.path.normcase(os.path.normpath(r))
        if key not in seen:
            seen.add(key)
            out.append(os.path.join(r, ROMA_WEIGHTS_FILE))
    return out


def has_cached_romav2_weights() -> bool:
    return any(os.path.isfile(p) for p in romav2_cached_weights_paths())


def _import_romav2():
    try:
        from romav2 import RoMaV2
        return RoMaV2
    except Exception:
        pass
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cand in (os.path.join(here, "RoMaV2", "src"), os.path.join(os.path.dirname(here), "RoMaV2", "src")):
        if os.path.isdir(cand) and cand not in sys.path:
            sys.path.insert(0, cand)
    try:
        from romav2 import RoMaV2
        return RoMaV2
    except Exception as exc:
        raise RuntimeError(
            "RoMa-v2 (package `romav2`) is not importable. Install it (PyTorch-ROCm build) or vendor it under "
            "RoMaV2/src next to this package; the dense-initialisation kernels only consume its outputs.") from exc


def _model_image(im):
    """What RoMaV2._load_image takes: PIL images / paths / arrays as they are; an (h, w, 3) u8 DEVICE tensor (the output of the
    device image preparation) as the (1, 3, h, w) tensor the model expects - no host round trip."""
    if isinstance(im, torch.Tensor) and im.dim() == 3 and im.shape[-1] == 3:
        return im.permute(2, 0, 1).unsqueeze(0)
    return im


class _KeyedDescriptor(torch.nn.Module):
    """Stands in for RoMaV2's child module ``f`` (the DINOv3 ``Descriptor``, RoMaV2/src/romav2/romav2.py:101,177) while a keyed
    call runs: ``forward`` answers from the run's FeatureCache under (camera key, input height, input width) - the size is
    part of the key so that a matcher re-used across settings never serves features of another resolution - and calls the
    wrapped module on a miss.  ``key`` None: plain call."""

    def __init__(self, inner: torch.nn.Module, cache):
        super().__init__()
        self.inner = inner
        self.cache = cache
        self.key = None

    def __getattr__(self, name):
        # anything the model reads from its backbone besides calling it (parameters, buffers, config) is the wrapped module's
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("inner"), name)

    @staticmethod
    def _frozen(levels):
        """What the cache keeps: an immutable snapshot of the per-level LIST.  The model rebinds entries of the list it is handed - in
        bidirectional settings ``_compute_head_preds`` does ``head_input[-1] = f_list[-1] + f_mv + match_emb`` on the NEIGHBOUR's list
        (RoMaV2/src/romav2/matcher.py:57-59,179-186), i.e. on the very object ``self.f(img_B)`` returned - so a cached list handed out as it
        is would come back with its last level replaced (found by tests/golden/check_matcher_contract.py against the real class, `high`).
        The tensors themselves are never written in place."""
        return levels if isinstance(levels, torch.Tensor) else tuple(levels)

    @staticmethod
    def _fresh(kept):
        """... and what a call receives: a new list over the same tensors."""
        return kept if isinstance(kept, torch.Tensor) else list(kept)

    def forward(self, img):
        if self.key is None or self.cache is None:
            return self.inner(img)
        variant = (int(img.shape[-2]), int(img.shape[-1]))
        if not isinstance(self.key, (list, tuple)):
            return self._fresh(self.cache.get_or_compute(self.key, lambda: self._frozen(self.inner(img)), variant=variant))
        # a batch of images, one key each (several pairs per forward): the cameras seen before come from the cache, the others go
        # through the backbone together - ONE pass - and are kept sample by sample
        keys = list(self.key)
        if len(keys) != int(img.shape[0]):
            raise ValueError("one camera key per image of the batch")
        vals = [self.cache.lookup(k, variant) if k is not None else None for k in keys]
        miss = [i for i, v in enumerate(vals) if v is None]
        if miss:
            out = self.inner(img[miss] if len(miss) < len(keys) else img)
            self._returns_tensor = isinstance(out, torch.Tensor)
            levels = [out] if self._returns_tensor else list(out)
            for j, i in enumerate(miss):
                mine = [t[j:j + 1] for t in levels]
                kept = mine[0] if self._returns_tensor else tuple(mine)          # the same (frozen) form the one-image path keeps
                vals[i] = self.cache.store(keys[i], kept, variant) if keys[i] is not None else kept
        per_image = [[v] if isinstance(v, torch.Tensor) else list(v) for v in vals]
        stacked = [torch.cat([v[lv] for v in per_image], 0) for lv in range(len(per_image[0]))]
        return stacked[0] if isinstance(vals[0], torch.Tensor) else stacked


class RomaMatcher:
    """Dense matcher with the reference image's features cached across its neighbours."""

    accepts_device_images = True      # match_grids_batch takes (h, w, 3) u8 device tensors as well as PIL images
    supports_feature_keys = True      # match_grids_batch(..., keys=(ref_key, [nbr_keys])) shares backbone features between references

    def __init__(self, device: str = "cuda", mode: str = "outdoor", setting: str = "fast", two_channel: bool = True,
                 pairs_per_forward: int = 1):
        """``pairs_per_forward`` > 1: the neighbours of a reference go through the model that many at a time (one batched forward,
        RoMaV2/tests/test_bidirectional.py runs B = 8) instead of one pair per forward as upstream's loop does (core/matcher.py:175-188
        there).  Batched GEMMs may round differently from single ones: the default stays 1, upstream's behaviour."""
        del mode
        RoMaV2 = _import_romav2()
        self.device = torch.device(device)
        torch.set_float32_matmul_precision("highest")
        self.model = RoMaV2(RoMaV2.Cfg(compile=False))
        if setting in PLUGIN_PRESETS:
            h_lr, w_lr, h_hr, w_hr, bidir = PLUGIN_PRESETS[setting]
            self.model.H_lr, self.model.W_lr, self.model.H_hr, self.model.W_hr = h_lr, w_lr, h_hr, w_hr
            self.model.bidirectional = bidir
        else:
            self.model.apply_setting(setting)
        self.model.to(self.device).eval()
        self.sample_thresh = 0.9
        self.w_resized, self.h_resized = int(self.model.W_lr), int(self.model.H_lr)
        self.two_channel = bool(two_channel)
        self.pairs_per_forward = max(1, int(pairs_per_forward))
        self._axes: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        log.info(f"RoMaV2 initialized (setting={setting}, H_lr={self.model.H_lr}, W_lr={self.model.W_lr}, device={device})")

    def reference_axes(self, H: int, W: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """The A-grid exactly as upstream builds it (core/matcher.py:132-135), as two 1-D axes."""
        key = (int(H), int(W))
        if key not in self._axes:
            self._axes[key] = (torch.linspace(-1 + 1 / W, 1 - 1 / W, W, device=self.device),
                               torch.linspace(-1 + 1 / H, 1 - 1 / H, H, device=self.device))
        return self._axes[key]

    def set_feature_cache(self, cache) -> None:
        """A core.scheduler.FeatureCache (or None to switch sharing off): backbone features (``model.f`` of the low-resolution
        image) are then computed once per camera key instead of once per (reference, neighbour) pair.  The vendored model is
        not modified: for the duration of a keyed call its child module ``f`` (RoMaV2/src/romav2/romav2.py:101) is replaced by
        a ``_KeyedDescriptor`` - an nn.Module, which is what ``nn.Module.__setattr__`` accepts for a registered child - and
        restored afterwards."""
        self._feature_cache = cache

    @torch.inference_mode()
    def match_grids_batch(self, imA, imB_list: Sequence, keys=None) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """``keys`` = (reference key, [neighbour keys]): camera identities for the feature cache (core/scheduler.py)."""
        if self.model is None:
            raise RuntimeError("RoMaV2 model has been released; create a new matcher before matching.")
        if not imB_list:
            return []
        torch.set_float32_matmul_precision("highest")
        model = self.model
        cache = getattr(self, "_feature_cache", None)
        ref_key, nbr_keys = (None, [None] * len(imB_list))
        if keys is not None and cache is not None:
            ref_key, nbr_keys = keys[0], list(keys[1])
            if len(nbr_keys) != len(imB_list):
                raise ValueError("keys: one neighbour key per neighbour image")
        img_a = model._load_image(_model_image(imA))
        kw = dict(mode="bicubic", align_corners=False, antialias=True)
        a_lr = F.interpolate(img_a, size=(int(model.H_lr), int(model.W_lr)), **kw)
        a_hr = None
        if model.H_hr is not None and model.W_hr is not None:
            a_hr = F.interpolate(img_a, size=(int(model.H_hr), int(model.W_hr)), **kw)
        plain_f = model.f
        keyed = _KeyedDescriptor(plain_f, cache) if cache is not None and keys is not None else None
        out: List[Tuple[torch.Tensor, torch.Tensor]] = []
        if keyed is not None:
            model.f = keyed            # the model's own calls of self.f (romav2.py:177) go through the cache while this call lasts
        try:
            if keyed is not None:
                keyed.key = ref_key
            feats_a = model.f(a_lr)    # DINOv3 features of the reference: once per reference (once per RUN with keys)
            P = self.pairs_per_forward
            for c0 in range(0, len(imB_list), P):
                chunk, chunk_keys = list(imB_list[c0:c0 + P]), list(nbr_keys[c0:c0 + P])
                if len(chunk) == 1:
                    if keyed is not None:
                        keyed.key = chunk_keys[0]  # the neighbour's features: looked up instead of recomputed when seen before
                    pred = model.match_from_features(f_list_A=feats_a, img_A_lr=a_lr, imB=_model_image(chunk[0]), img_A_hr=a_hr)
                else:
                    # several pairs in ONE forward: the neighbours stacked along the batch axis, the reference's features and images
                    # broadcast to them (RoMaV2.match_from_features' own steps - _load_image, _resize_match_image, _match_core -
                    # on a batch, romav2.py:404-428)
                    # (every neighbour is resized on its own first - they may come in different sizes - then stacked)
                    n = len(chunk)
                    sized = [model._resize_match_image(model._load_image(_model_image(b))) for b in chunk]
                    b_lr = torch.cat([s_[0] for s_ in sized], dim=0)
                    b_hr = torch.cat([s_[1] for s_ in sized], dim=0) if sized[0][1] is not None else None
                    if keyed is not None:
                        keyed.key = chunk_keys
                    fa = [t.expand(n, *t.shape[1:]) if t.shape[0] == 1 else t for t in feats_a]
                    pred = model._match_core(f_list_A=fa, img_A_lr=a_lr.expand(n, -1, -1, -1), img_B_lr=b_lr,
                                             img_A_hr=a_hr.expand(n, -1, -1, -1) if (a_hr is not None and b_hr is not None) else None,
                                             img_B_hr=b_hr)
                for i in range(len(chunk)):
                    warp_ab = pred["warp_AB"][i]
                    cert = pred["overlap_AB"][i].squeeze(-1).contiguous()
                    H, W = cert.shape
                    if self.two_channel:
                        out.append((warp_ab.contiguous(), cert))
                    else:
                        ax, ay = self.reference_axes(H, W)
                        grid = torch.stack([ax.view(1, W).expand(H, W), ay.view(H, 1).expand(H, W)], dim=-1)
                        out.append((torch.cat([grid, warp_ab], dim=-1).contiguous(), cert))
        finally:
            if keyed is not None:
                model.f = plain_f
        return out

    def match_grids(self, imA, imB):
        res = self.match_grids_batch(imA, [imB])
        if not res:
            raise RuntimeError("RoMaV2 returned no matches for the requested pair.")
        return res[0]

    def close(self) -> None:
        model = getattr(self, "model", None)
        if model is None:
            return
        try:
            model.to("cpu")
        except Exception:
            pass
        self.model = None
        self._axes.clear()
        gc.collect()
        if torch.cuda.is_available():
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            torch.cuda.empty_cache()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
