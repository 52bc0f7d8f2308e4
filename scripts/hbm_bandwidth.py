"""Achievable HBM bandwidth on this GPU (report the roofline against nominal AND achievable): device copy and triad
over buffers far larger than the 256 MB infinity cache."""
import json
import torch

n = 1 << 28                                    # 1 GiB of fp32 per buffer
a = torch.rand(n, device="cuda"); b = torch.rand(n, device="cuda"); c = torch.empty(n, device="cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t_copy = timed(lambda: c.copy_(a))
t_triad = timed(lambda: torch.add(a, b, alpha=2.0, out=c))
t_read = timed(lambda: a.sum())
print(json.dumps({"buffer_GiB": 1, "copy_GBps": round(2 * 4 * n / t_copy / 1e9, 1), "triad_GBps": round(3 * 4 * n / t_triad / 1e9, 1),
                  "read_sum_GBps": round(4 * n / t_read / 1e9, 1), "device": torch.cuda.get_device_name(0)}))
