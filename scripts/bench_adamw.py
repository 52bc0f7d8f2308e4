import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd.dist import GradBucket
from moss_amd.optim import FlatAdamW
dev = torch.device("cuda:0")
P = 100000
def run(capturable, pattern):
    xyz = torch.nn.Parameter(torch.randn(P, 3, device=dev)); op = torch.nn.Parameter(torch.randn(P, 1, device=dev))
    sc = torch.nn.Parameter(torch.randn(P, 3, device=dev)); ro = torch.nn.Parameter(torch.randn(P, 4, device=dev))
    if pattern:
        f = torch.nn.Parameter(torch.randn(P, 16, 3, device=dev))
        groups = [{"params": [xyz], "lr": 1e-4}, {"params": [f], "lr": 2.5e-3, "lr_pattern": (48, 3, 1.25e-4)},
                  {"params": [op], "lr": 0.05}, {"params": [sc], "lr": 5e-3}, {"params": [ro], "lr": 1e-3}]
        params = [xyz, f, op, sc, ro]
    else:
        dc = torch.nn.Parameter(torch.randn(P, 1, 3, device=dev)); rest = torch.nn.Parameter(torch.randn(P, 15, 3, device=dev))
        groups = [{"params": [xyz], "lr": 1e-4}, {"params": [dc], "lr": 2.5e-3}, {"params": [rest], "lr": 1.25e-4},
                  {"params": [op], "lr": 0.05}, {"params": [sc], "lr": 5e-3}, {"params": [ro], "lr": 1e-3}]
        params = [xyz, dc, rest, op, sc, ro]
    b = GradBucket(params); b.flat.normal_()
    o = FlatAdamW(groups, b, eps=1e-15, capturable=capturable)
    for _ in range(5): o.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): o.step()
    e1.record(); torch.cuda.synchronize()
    print("capturable=%s pattern=%s: %.1f us/step" % (capturable, pattern, e0.elapsed_time(e1) / 50 * 1e3))
for c in (False, True):
    for p in (False, True):
        run(c, p)
