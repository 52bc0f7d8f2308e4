"""Does a PARALLEL CPU torch op (an OpenMP team of torch.get_num_threads() threads) stall the GPU work of the same process a moment later?
(profiles/r06_notes.md section 10: a container whose CPU quota is a fraction of the machine's hardware threads.)  For each mode, REPS
times: run the CPU work, then launch a tiny kernel + synchronise in a loop for 400 ms and record the longest single iteration.
usage: python scripts/micro/cpu_parallel_stall.py [REPS=8]"""
import os
import sys
import time

import torch

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
print("torch threads", torch.get_num_threads(), "os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cpu.max unreadable:", e)


def watch(ms=400.0):
    worst, n, t_end = 0.0, 0, time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        x.add_(1.0)
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
        n += 1
    return round(1e3 * worst, 2), n


def cpu_work(n):
    g = torch.Generator().manual_seed(1)
    p = torch.randperm(n, generator=g)
    m = torch.zeros(n, dtype=torch.bool)
    m[p[: n // 10]] = True
    r = torch.randn(n, 3, generator=g) * 0.004
    return float((r.abs() + 1).log().sum()) + float(m.sum())


watch(200.0)                                                     # (first launches: code objects)
for mode, n, threads in (("none", 0, None), ("cpu 6.9k", 6890, None), ("cpu 100k", 100000, None), ("cpu 1M", 1000000, None),
                         ("cpu 100k, 8 threads", 100000, 8), ("cpu 1M, 8 threads", 1000000, 8), ("none", 0, None)):
    if threads:
        torch.set_num_threads(threads)
    res = []
    for _ in range(REPS):
        if n:
            cpu_work(n)
        res.append(watch())
    print(f"{mode:22s}: longest iteration per repetition (ms) {[r[0] for r in res]}", flush=True)
