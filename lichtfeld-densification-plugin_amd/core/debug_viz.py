"""Live match-preview store shared between the pipeline thread and the GUI (upstream
core/debug_viz.py:12-161).  The pipeline only needs ``is_enabled / is_auto_step / set_total_pairs /
submit_preview / release_waiters``; ``submit_preview`` blocks the pipeline thread while manual
stepping is active."""
from __future__ import annotations

import threading
from collections import deque
from dataclasses import dataclass
from typing import Deque, List, Optional

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


class MatchDebugState:
    def __init__(self, max_history: int = 4) -> None:
        self._lock = threading.Lock()
        self._go = threading.Event()
        self._go.set()
        self._enabled = False
        self._auto = True
        self._latest: Optional[MatchPreview] = None
        self._history: Deque[MatchPreview] = deque(maxlen=max_history)
        self._total_pairs = 0
        self._max_visible = 0
        self._single = False
        self._cursor = 0

    # -- switches -------------------------------------------------------------------------------
    def set_enabled(self, enabled: bool) -> None:
        with self._lock:
            self._enabled = bool(enabled)
            if not enabled:
                self._history.clear()
                self._latest = None
                self._auto = True
                self._go.set()

    def is_enabled(self) -> bool:
        with self._lock:
            return self._enabled

    def set_auto_step(self, auto: bool) -> None:
        with self._lock:
            self._auto = bool(auto)
            if auto:
                self._go.set()

    def is_auto_step(self) -> bool:
        with self._lock:
            return self._auto

    def set_total_pairs(self, total: int) -> None:
        with self._lock:
            self._total_pairs = max(0, int(total))

    def total_pairs(self) -> int:
        with self._lock:
            return self._total_pairs

    # -- producer / consumer ------------------------------------------------------------------------
    def submit_preview(self, preview: MatchPreview) -> None:
        with self._lock:
            if not self._enabled:
                return
            self._latest = preview
            self._history.append(preview)
            auto = self._auto
        if not auto:
            self._go.clear()
            self._go.wait()

    def step_once(self) -> None:
        self._go.set()

    def release_waiters(self) -> None:
        self._go.set()

    def latest(self) -> Optional[MatchPreview]:
        with self._lock:
            return self._latest

    def history(self) -> List[MatchPreview]:
        with self._lock:
            return list(self._history)

    # -- which matches the panel draws ----------------------------------------------------------------
    def set_max_visible_matches(self, n: int) -> None:
        with self._lock:
            self._max_visible = max(0, int(n))

    def max_visible_matches(self) -> int:
        with self._lock:
            return self._max_visible

    def set_single_match_mode(self, enabled: bool) -> None:
        with self._lock:
            self._single = bool(enabled)
            if enabled:
                self._cursor = 0

    def is_single_match_mode(self) -> bool:
        with self._lock:
            return self._single

    def current_match_index(self) -> int:
        with self._lock:
            return self._cursor

    def set_current_match_index(self, idx: int) -> None:
        with self._lock:
            self._cursor = max(0, int(idx))

    def next_match(self, total: int) -> None:
        with self._lock:
            if total > 0:
                self._cursor = (self._cursor + 1) % total

    def prev_match(self, total: int) -> None:
        with self._lock:
            if total > 0:
                self._cursor = (self._cursor - 1) % total

    def visible_match_indices(self, total: int) -> List[int]:
        with self._lock:
            if total <= 0:
                return []
            if self._single:
                return [min(self._cursor, total - 1)]
            if self._max_visible <= 0 or self._max_visible >= total:
                return list(range(total))
            return list(range(self._max_visible))
