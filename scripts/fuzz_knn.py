"""Randomised check that the cell-grid k-NN equals the brute-force kernel bit for bit (indices and distances, ties included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_cuda import knn

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(n_cases):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: lo + (hi - lo) * float(torch.rand(1, generator=g))
    Nr = int(10 ** r(0, 4.5)); Nq = int(10 ** r(0, 4.3)); k = min(int(torch.randint(1, 5, (1,), generator=g)), Nr)
    kind = int(torch.randint(0, 6, (1,), generator=g))
    ref = torch.randn(Nr, 3, generator=g)
    if kind == 1: ref[:, 2] = 0.3                                            # coplanar
    if kind == 2: ref[:, 1:] = 0.0                                           # collinear
    if kind == 3: ref = ref[torch.randint(0, max(Nr // 3, 1), (Nr,), generator=g)]          # many duplicates
    if kind == 4: ref = ref * torch.tensor([1.0, 50.0, 0.02])                # very flat box
    if kind == 5: ref = torch.cat([ref[: Nr // 2] * 0.01, ref[Nr // 2:] * 10 + 100])       # two clusters far apart
    same = float(torch.rand(1, generator=g)) < 0.3
    query = ref[:Nq] if (same and Nq <= Nr) else torch.randn(Nq, 3, generator=g) * r(0.2, 3.0) + r(-1, 1)
    ref_c, q_c = ref.cuda().contiguous(), query.cuda().contiguous()
    d0, i0 = knn(ref_c[None], q_c[None], k, "brute")
    d1, i1 = knn(ref_c[None], q_c[None], k, "grid")
    if not (torch.equal(i0, i1) and torch.equal(d0, d1)):
        bad += 1
        print(f"seed {seed}: Nr={Nr} Nq={q_c.shape[0]} k={k} kind={kind}: {(i0 != i1).sum().item()} index / {(d0 != d1).sum().item()} distance mismatches")
print(f"{n_cases - bad} / {n_cases} cases identical")
sys.exit(1 if bad else 0)
