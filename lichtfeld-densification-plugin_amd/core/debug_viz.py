"""What the pipeline needs from the host's match-debug object, and nothing else.

Upstream's GUI owns a ``MatchDebugState`` (core/debug_viz.py:29-161 there) with list cursors, history and
visibility controls for its panel; that is GUI state and out of scope here.
``run_dense_pipeline`` only ever calls five methods on whatever object it is handed:

    is_enabled()  is_auto_step()  set_total_pairs(n)  submit_preview(MatchPreview)  release_waiters()

so the host's own object is accepted as it is (duck typing).  ``PreviewGate`` below is the small
stand-alone implementation of those five calls for callers without a GUI (the tests, scripts): a
one-slot mailbox with an optional turnstile - when manual stepping is on, ``submit_preview`` parks the
producer until the consumer calls ``step_once()`` (upstream's panel does the same to the pipeline thread).

``MatchPreview`` is the record handed over; its field names are what upstream's panel reads
(core/debug_viz.py:12-26 there).
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class MatchPreview:
    ref_id: int
    nbr_id: int
    ref_label: str
    nbr_label: str
    left_image: np.ndarray     # (h,w,3) u8
    right_image: np.ndarray    # (h,w,3) u8
    matches: np.ndarray        # (n,4) f32 [xA,yA,xB,yB] in match pixels
    cert_norm: np.ndarray      # (n,) f32 in [0,1]
    match_count: int
    pair_index: int
    total_pairs: int


class PreviewGate:
    """One-slot mailbox + turnstile.  Thread-safe; every wait is on one condition variable."""

    def __init__(self) -> None:
        self._cv = threading.Condition()
        self._on = False
        self._manual = False          # True: every submitted preview waits for a step
        self._slot: Optional[MatchPreview] = None
        self._pairs = 0
        self._tickets = 0             # steps granted so far
        self._open = False            # release_waiters(): nobody waits any more

    # -- the five calls of the pipeline ---------------------------------------------------------------
    def is_enabled(self) -> bool:
        with self._cv:
            return self._on

    def is_auto_step(self) -> bool:
        with self._cv:
            return not self._manual

    def set_total_pairs(self, total: int) -> None:
        with self._cv:
            self._pairs = int(total) if total > 0 else 0

    def submit_preview(self, preview: MatchPreview) -> None:
        with self._cv:
            if not self._on:
                return
            self._slot = preview
            if not self._manual or self._open:
                return
            want = self._tickets + 1
            self._cv.wait_for(lambda: self._tickets >= want or self._open or not self._manual or not self._on)

    def release_waiters(self) -> None:
        with self._cv:
            self._open = True
            self._cv.notify_all()

    # -- consumer side --------------------------------------------------------------------------------
    def set_enabled(self, enabled: bool) -> None:
        with self._cv:
            self._on = bool(enabled)
            if not self._on:
                self._slot = None
                self._manual = False
            self._cv.notify_all()

    def set_auto_step(self, auto: bool) -> None:
        with self._cv:
            self._manual = not bool(auto)
            self._cv.notify_all()

    def step_once(self) -> None:
        with self._cv:
            self._tickets += 1
            self._cv.notify_all()

    def latest(self) -> Optional[MatchPreview]:
        with self._cv:
            return self._slot

    def total_pairs(self) -> int:
        with self._cv:
            return self._pairs


# the name the upstream entry points use in their signatures (densify.py:148-153,248-255 there)
MatchDebugState = PreviewGate
