"""Host-side settings that decide whether a GPU program on this stack runs smoothly (nothing here touches the device).

``cpu_quota()`` / ``limit_cpu_threads()``: torch sizes its OpenMP team from the MACHINE (128 threads on a 256-thread host); a container
with a CPU quota of 16 that lets such a team spin burns the quota of the 100 ms scheduler period in a few milliseconds, and the kernel
then suspends the whole process -- the HIP runtime's threads included -- until the period ends: 10-90 ms stalls at random places a moment
after any parallel CPU torch op (>= 32k elements).  Measured: ``scripts/micro/cpu_parallel_stall.py``, ``profiles/r06_notes.md`` section 10.
"""
from __future__ import annotations

import os

__all__ = ["cpu_quota", "limit_cpu_threads"]


def cpu_quota() -> int:
    """CPUs this process may use: the cgroup's quota (v2 ``cpu.max`` = "<quota> <period>" or "max"; v1 ``cpu.cfs_quota_us``), else the
    affinity mask, else the machine."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt and txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def limit_cpu_threads(local_world_size: int | None = None) -> int:
    """``torch.set_num_threads(min(current, quota // ranks on this node))``; returns the number set.  Call once at start-up (the ranks of
    one node share the quota: ``LOCAL_WORLD_SIZE`` is read when ``local_world_size`` is not given)."""
    import torch
    if local_world_size is None:
        local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    n = max(1, min(torch.get_num_threads(), cpu_quota() // max(1, int(local_world_size))))
    torch.set_num_threads(n)
    return n
