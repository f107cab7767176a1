"""Host-thread budget: cgroup CPU quota, affinity mask, at most 64 (same rule as include/nsk_threads.h)."""
import math
import os


def cpu_budget() -> int:
    try:
        budget = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        budget = os.cpu_count() or 1
    quota, period = -1.0, 100000.0
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, *rest = f.read().split()
            if q != "max":
                quota = float(q)
                period = float(rest[0]) if rest else period
    except OSError:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
        except OSError:
            pass
    if quota > 0 and period > 0:
        budget = min(budget, max(1, math.ceil(quota / period)))
    return max(1, min(budget, 64))
