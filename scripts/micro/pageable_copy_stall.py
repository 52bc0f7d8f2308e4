"""Does a pageable host -> device copy of a buffer that is then freed stall the GPU a moment later?  (profiles/r06_notes.md section 10)
For each mode -- pageable / pinned staging / no copy -- REPS times: make a fresh CPU tensor of MB megabytes, upload it, drop both, then
launch a tiny kernel + synchronise in a loop for 400 ms and record the longest single iteration.
usage: python scripts/micro/pageable_copy_stall.py [MB=19] [REPS=8]"""
import sys
import time

import torch

MB = float(sys.argv[1]) if len(sys.argv) > 1 else 19.0
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()


def watch(ms=400.0):
    worst, n, t_end = 0.0, 0, time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        x.add_(1.0)
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
        n += 1
    return round(1e3 * worst, 2), n


for mode in ("none", "pageable", "pinned", "pageable", "pinned", "none"):
    res = []
    for _ in range(REPS):
        if mode != "none":
            h = torch.randn(int(MB * 262144))
            d = (h.pin_memory() if mode == "pinned" else h).to(dev)
            torch.cuda.synchronize()
            del h, d
        res.append(watch())
    print(f"{mode:9s} {MB} MB: longest iteration per repetition (ms) {[r[0] for r in res]}  iterations {[r[1] for r in res]}", flush=True)
