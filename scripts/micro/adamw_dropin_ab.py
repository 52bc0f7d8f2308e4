import sys, time, torch
sys.path.insert(0, "/root/repo")
from moss_amd.optim import AdamW
dev = torch.device("cuda", 0)
P = 100000
shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
params = {k: torch.nn.Parameter(torch.randn(s, device=dev)) for k, s in shapes.items()}
for p in params.values():
    p.grad = torch.randn_like(p)
one = AdamW([{"params": [p], "lr": 1e-3, "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
six = [AdamW([{"params": [p], "lr": 1e-3, "name": k}], lr=0.0, eps=1e-15) for k, p in params.items()]
ref = torch.optim.AdamW([{"params": [p], "lr": 1e-3, "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
def run(f, n=500):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return round(1e6 * (t1 - t0) / n, 1), round(1e6 * (t2 - t0) / n, 1)
for rep in range(2):
    print("one launch for six tensors : host us/step, total us/step", run(one.step))
    print("six launches (one each)    :", run(lambda: [o.step() for o in six]))
    print("torch.optim.AdamW          :", run(ref.step))
