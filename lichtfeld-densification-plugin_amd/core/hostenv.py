"""Host-side environment of a run: how many cores the container may really use.

A container often sees every CPU of its host (``os.cpu_count()``: 256 on the MI355X boxes of this pool) while its cgroup grants a fraction of them
(16 here).  torch sizes its intra-op pool by the CPUs it SEES, and its OpenMP workers spin between parallel regions: 128 spinning threads burn a
16-core CFS quota in the first 12 ms of every 100 ms period, and the kernel then stops the WHOLE process for the rest of the period - a per-reference
host step of 0.3 ms becomes 6 ms as soon as it runs back to back for longer than that (profiles/r4/sampled_sustained.txt).  ``fit_threads_to_quota``
is what the command-line entry point and ``bench.py`` call; the library functions never touch process-wide settings on their own (inside LichtFeld
Studio the host application owns them).  No upstream counterpart."""
from __future__ import annotations

import os
from typing import Optional


def cpu_quota() -> Optional[float]:
    """Cores the cgroup of this process may use (cgroup v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # cgroup v2
            quota, period = fh.read().split()[:2]
            if quota != "max" and float(period) > 0:
                return float(quota) / float(period)
            return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:      # cgroup v1
            quota, period = float(fq.read()), float(fp.read())
            if quota > 0 and period > 0:
                return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cores() -> int:
    """min(CPUs this process may be scheduled on, cgroup quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q)))
    return max(1, n)


def fit_threads_to_quota(log=None) -> int:
    """torch's intra-op threads = min(what torch chose, usable_cores()) unless the user fixed them (OMP_NUM_THREADS / MKL_NUM_THREADS set).  Returns
    the count in force.  Note: upstream's sampling normaliser is a torch CPU f32 sum, whose last bits depend on this count (core/sampling.py): a run
    compared bit for bit against upstream must use upstream's setting - set OMP_NUM_THREADS, which this function respects."""
    import torch
    if os.environ.get("OMP_NUM_THREADS") or os.environ.get("MKL_NUM_THREADS"):
        return int(torch.get_num_threads())
    want = usable_cores()
    have = int(torch.get_num_threads())
    if want < have:
        torch.set_num_threads(want)
        if log is not None:
            log(f"torch intra-op threads {have} -> {want} (the container's CPU quota)")
    return int(torch.get_num_threads())
