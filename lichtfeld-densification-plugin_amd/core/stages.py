"""Where a run's wall-clock goes: seconds per named stage of the driver loop (upstream core/pipeline.py:783-928 has no timing beyond
its `it/s` progress text).

Stages, as `bench.py`'s pipeline leg reports them: ``decode`` (PIL decode on the pack threads), ``prepare`` (resize / mask / black-out:
PIL on the pack threads or the HIP image kernels), ``match`` (the matcher's call), ``select`` (aggregate + coverage sampling), ``kernel``
(triangulation: the indexed kernels, the fused sampled call, the dense kernel), ``d2h`` (survivors or file payload crossing PCIe),
``write`` (file appends).  A ``StageClock`` is handed to ``run_dense_pipeline(stage_clock=...)``; without one every ``stage()`` is a shared
no-op context.

Two ways of reading it:
  * plain: host wall time between entering and leaving a stage.  Launches are asynchronous, so device time shows up where the host next
    waits (usually ``d2h``) - what the run costs, not who caused it;
  * ``StageClock(sync=torch.cuda.synchronize)``: the device is drained at the end of every stage, so each stage is charged its own
    device time.  That serialises the run: use it for the split, never for the throughput.
Stages entered on other threads (pack workers, the file writer) are accumulated under a lock and overlap the main loop: the sum of all
stages can exceed the run's wall time."""
from __future__ import annotations

import contextlib
import threading
import time
from typing import Callable, Dict, Optional


class _NullClock:
    _ctx = contextlib.nullcontext()

    def stage(self, name: str, sync: bool = True):
        return self._ctx

    def add(self, name: str, seconds: float, count: int = 1) -> None:
        pass

    def count(self, name: str, n: int) -> None:
        pass

    serialising = False


NULL_CLOCK = _NullClock()
_tls = threading.local()


def current():
    """The clock bound to this thread (``bound``), else the no-op clock: how code far from the driver (the image decoders on the pack
    threads) finds the run's clock without every signature carrying it."""
    return getattr(_tls, "clock", None) or NULL_CLOCK


@contextlib.contextmanager
def bound(clock):
    prev = getattr(_tls, "clock", None)
    _tls.clock = clock
    try:
        yield clock
    finally:
        _tls.clock = prev


class StageClock:
    def __init__(self, sync: Optional[Callable[[], None]] = None):
        self._sync = sync
        self._lock = threading.Lock()
        self.seconds: Dict[str, float] = {}
        self.calls: Dict[str, int] = {}
        self.counters: Dict[str, int] = {}       # things counted beside the time (bytes that crossed PCIe ...)

    @property
    def serialising(self) -> bool:
        """True when every stage drains the device: the driver then takes the unfused per-stage calls so that ``select`` and ``kernel`` can
        be told apart (the fused sampled call is one stage otherwise)."""
        return self._sync is not None

    def add(self, name: str, seconds: float, count: int = 1) -> None:
        with self._lock:
            self.seconds[name] = self.seconds.get(name, 0.0) + float(seconds)
            self.calls[name] = self.calls.get(name, 0) + int(count)

    def count(self, name: str, n: int) -> None:
        with self._lock:
            self.counters[name] = self.counters.get(name, 0) + int(n)

    @contextlib.contextmanager
    def stage(self, name: str, sync: bool = True):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if sync and self._sync is not None:
                self._sync()
            self.add(name, time.perf_counter() - t0)

    def report(self) -> Dict[str, dict]:
        with self._lock:
            rep = {k: {"seconds": round(v, 6), "calls": self.calls.get(k, 0)} for k, v in sorted(self.seconds.items())}
            rep.update({k: int(v) for k, v in self.counters.items()})
            return rep
