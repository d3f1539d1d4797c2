"""How many host CPUs this process may actually burn.

`os.sched_getaffinity` (or `hardware_concurrency`) reports the CPUs the process may be scheduled ON; a
container's CFS quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`) caps how much CPU TIME it gets per
period.  Where the quota is below the mask -- a one-GPU slice of a 256-thread host with `cpu.max` =
"1600000 100000", i.e. 16 CPUs -- a burst of 64 tokenizer threads or a 256-thread OpenMP copy spends the
period's budget in a few milliseconds and the kernel then parks EVERY thread of the process until the next
period: the text ingest showed 70-150 ms stalls at random places (nr_throttled in cpu.stat), which is most of
what separated it from the token-id path.  Host-side pools are therefore sized by cpu_budget().
"""
from __future__ import annotations

import functools
import math
import os


def _cgroup_quota() -> float | None:
    try:   # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return int(quota) / int(period)
    except (OSError, ValueError):
        pass
    try:   # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


@functools.lru_cache(maxsize=1)
def cpu_budget() -> int:
    """min(CPUs in the affinity mask, CFS quota rounded up), at least 1.  Read once per process (a 256-CPU
    affinity mask and two cgroup files cost ~25 us per call: too much on the single-query path)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = _cgroup_quota()
    if q is not None:
        n = min(n, max(1, math.ceil(q)))
    return max(1, n)
