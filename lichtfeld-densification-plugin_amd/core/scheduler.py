"""Pair scheduler (N4): the order in which (reference, neighbour) pairs reach the matcher, and what is shared between them.

Upstream walks the reference list and, per reference, runs one RoMa forward per neighbour; the reference's backbone features
are computed once per reference and reused across its neighbours (core/matcher.py:172-181), every neighbour's features are
computed again for every reference that lists it (RoMaV2/src/romav2/romav2.py:163-180, ``f_B = self.f(img_B_lr)``).  With
``num_refs`` = 0.3 ... 0.8 of the cameras and k = 3 ... 8 neighbours a camera's image goes through the DINOv3 backbone
(1 + k) * n_refs / n_cams times - 2.7 x (GUI defaults) to 4.4 x (CLI) too often.  On a 288 GB part every camera's features fit
many times over (tens of MB each), so the schedule below keeps them resident exactly as long as a later pair needs them:

  * ``PairSchedule`` lists the work in upstream's order (references in ``refs_local`` order, each with its loaded neighbours -
    the unit the hot path consumes, because of the arg-max across a reference's neighbours) and knows, for every camera, the
    position of its LAST use;
  * ``FeatureCache`` holds backbone features by camera; an entry is dropped the moment the schedule says it will not be used
    again (no LRU guessing), so the cache never holds more than the cameras "in flight" between their first and last use.

The matcher mirror (core/matcher.py) consults the cache when the pipeline hands it camera keys; a matcher without that
capability is driven exactly as before.  Sharding (multi-GPU) deals whole references to ranks; each rank schedules its own.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Hashable, List, Optional, Sequence


@dataclasses.dataclass
class ScheduledReference:
    position: int                 # position in refs_local
    ref_index: int                # index into the camera records
    nbr_indices: List[int]        # neighbours that will be matched, in nn_table order (self-matches removed)


class PairSchedule:
    def __init__(self, refs_local: Sequence[int], nn_table, uids: Sequence[int], nns_per_ref: int,
                 positions: Optional[Sequence[int]] = None):
        """``positions``: the subset of ``refs_local`` positions this process works on (multi-GPU sharding); default all."""
        self.items: List[ScheduledReference] = []
        pos_list = list(range(len(refs_local))) if positions is None else [int(p) for p in positions]
        for p in pos_list:
            r = int(refs_local[p])
            nbrs = [int(n) for n in nn_table[r][:nns_per_ref] if uids[int(n)] != uids[r]]
            self.items.append(ScheduledReference(position=p, ref_index=r, nbr_indices=nbrs))
        self.last_use: Dict[int, int] = {}
        self.uses: Dict[int, int] = {}
        for step, it in enumerate(self.items):
            for cam in [it.ref_index] + it.nbr_indices:
                self.last_use[cam] = step
                self.uses[cam] = self.uses.get(cam, 0) + 1

    @property
    def n_pairs(self) -> int:
        return sum(len(it.nbr_indices) for it in self.items)

    @property
    def n_backbone_forwards_upstream(self) -> int:
        """Backbone passes upstream makes: one per reference + one per pair."""
        return sum(1 + len(it.nbr_indices) for it in self.items if it.nbr_indices)

    @property
    def n_backbone_forwards_shared(self) -> int:
        """... and with every camera's features computed once."""
        return len(self.uses)

    def peak_resident(self) -> int:
        """Largest number of cameras whose features are alive at once under last-use eviction."""
        first: Dict[int, int] = {}
        for step, it in enumerate(self.items):
            for cam in [it.ref_index] + it.nbr_indices:
                first.setdefault(cam, step)
        peak = 0
        for step in range(len(self.items)):
            peak = max(peak, sum(1 for c in first if first[c] <= step <= self.last_use[c]))
        return peak


class FeatureCache:
    """Backbone features by camera key, evicted by schedule position (an entry whose last use is behind us is dropped)."""

    def __init__(self, last_use: Optional[Dict[Hashable, int]] = None, max_entries: int = 0):
        self._last_use = dict(last_use) if last_use else {}
        self._max = int(max_entries)
        self._store: Dict[Hashable, object] = {}
        self.hits = 0
        self.misses = 0
        self.peak = 0

    def get_or_compute(self, key: Hashable, compute, variant: Hashable = ()):
        """``key`` is what the schedule counts uses of (a camera); ``variant`` distinguishes values of one key that must not be
        mixed (the matcher passes the backbone's input size) and lives and dies with the key."""
        slot = (key, variant)
        if slot in self._store:
            self.hits += 1
            return self._store[slot]
        self.misses += 1
        val = compute()
        if self._max <= 0 or len(self._store) < self._max:
            self._store[slot] = val
            self.peak = max(self.peak, len(self._store))
        return val

    def lookup(self, key: Hashable, variant: Hashable = ()):
        """The value held for (key, variant) or None; counted as a hit / a miss like get_or_compute."""
        slot = (key, variant)
        if slot in self._store:
            self.hits += 1
            return self._store[slot]
        self.misses += 1
        return None

    def store(self, key: Hashable, val, variant: Hashable = ()):
        """Keep ``val`` for (key, variant) (what get_or_compute does after computing); returns ``val``."""
        slot = (key, variant)
        if slot not in self._store and (self._max <= 0 or len(self._store) < self._max):
            self._store[slot] = val
            self.peak = max(self.peak, len(self._store))
        return val

    def advance(self, step: int) -> None:
        """Schedule position ``step`` is done: drop what no later position needs."""
        if not self._last_use:
            return
        for slot in [s for s in self._store if self._last_use.get(s[0], -1) <= step]:
            del self._store[slot]

    def keys(self):
        return [s[0] for s in self._store]

    def clear(self) -> None:
        self._store.clear()

    def __len__(self) -> int:
        return len(self._store)
