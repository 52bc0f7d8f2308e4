"""Timings of the densification-side ops on one GPU: k-NN (brute force vs cell grid), neighbour KL, per-step
statistics.  Reported SEPARATELY from the training-step metric (the KL-guided densify is amortised every 100 steps)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import scenes
from moss_amd.densify import DensifyStats, neighbour_kl, cal_kl
from moss_amd.knn_cuda import knn, KnnGrid


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    out = {}
    for cfg in ("cfg3", "cfg5"):
        sc = scenes.config3() if cfg == "cfg3" else scenes.config5()
        xyz = sc.means3D.cuda().contiguous(); P = xyz.shape[0]
        rot = sc.rotations.cuda().contiguous(); scaling = sc.scales.cuda().contiguous()
        body = scenes.body_points(6890, torch.Generator().manual_seed(99)).cuda().contiguous()
        smpl = body                                                   # stand-in for the 6 890 template vertices
        r = {"P": P}
        r["self_k2_brute_ms"] = timed(lambda: knn(xyz[None], xyz[None], 2, "brute"), reps=3, warm=1)
        r["self_k2_grid_build_plus_query_ms"] = timed(lambda: knn(xyz[None], xyz[None], 2, "grid"))
        grid = KnnGrid(xyz)
        r["self_k2_grid_query_only_ms"] = timed(lambda: grid.query(xyz, 2))
        r["smpl_k1_brute_ms"] = timed(lambda: knn(smpl[None], xyz[None], 1, "brute"))
        r["smpl_k1_grid_build_plus_query_ms"] = timed(lambda: knn(smpl[None], xyz[None], 1, "grid"))
        sg = KnnGrid(smpl)
        r["smpl_k1_grid_query_only_ms"] = timed(lambda: sg.query(xyz, 1))
        _, ids = grid.query(xyz, 2)
        r["neighbour_kl_ms"] = timed(lambda: neighbour_kl(xyz, rot, scaling, ids))
        r["cal_kl_total_ms"] = timed(lambda: cal_kl(xyz, rot, scaling))
        stats = DensifyStats(P)
        radii = torch.randint(0, 20, (P,), device="cuda", dtype=torch.int32); grad = torch.randn(P, 3, device="cuda")
        r["densify_stats_ms"] = timed(lambda: stats.add(radii, grad))

        def torch_stats():
            vis = radii > 0
            stats.max_radii2D[vis] = torch.max(stats.max_radii2D[vis], radii[vis].float())
            stats.xyz_gradient_accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
            stats.denom[vis] += 1
        r["densify_stats_torch_expressions_ms"] = timed(torch_stats)
        out[cfg] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
