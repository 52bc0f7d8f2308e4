"""Randomised parity sweep: many small random scenes (sizes, image shapes, cameras, scale / opacity extremes, input modes, SH
degrees, backgrounds): integer stages bit-exact; images 3e-4 of the largest value (random anisotropic Gaussians make
power = -1/2 (A dx^2 + C dy^2) - B dx dy a difference of terms 100-1000x its size, so the kernels' fmaf chain and the oracle's
unfused expression -- both legitimate fp32 evaluations; nvcc fuses too -- differ by up to ~1e-4 in alpha there); GRADIENTS under the
ONE rule of tests/helpers.py (RULE_K, RULE_EPS, reference_noise_floor), element by element against float64, the same constants for
every seed and for the BASELINE configurations (rounds 2-3: 2e-2 of the largest value here, plus a looser second rule for a list of seeds).
Also checks, per case, that switching the block-mask culling off changes no decision and no gradient bit.
Usage: python scripts/fuzz_parity.py [n_cases] [first_seed]"""
import math, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from types import SimpleNamespace
from moss_amd import scenes
from tests import helpers as hp
from tests import test_gpu_parity as tp

dev = torch.device("cuda:0")
from moss_amd import _lib
L = _lib.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0


from fuzz_scenes import random_scene  # noqa: E402


bad = 0
worst_ratio = 0.0
failures = []
for seed in range(first, first + n_cases):
    if (seed - first) % 100 == 0:
        print(f"[progress] seed {seed} ({seed - first} of {n_cases} done, {bad} flagged)", flush=True)
    s, mode, degree, colors = random_scene(seed)
    try:
        d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
        tp.IMG_TOL, tp.GRAD_TOL = 3e-4, 2e-2
        fw, t, e = tp._check_forward(d, dev, max_fragile=2e-2)
        if t.R > 0:
            fw64 = hp.oracle_forward64(d, fw)
            m = hp.stable_mask(d, fw, fw64, thr=1e-4)          # incoming gradients on the pixels where every implementation takes the same branches
            dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1))
            g = hp.hip_backward(d, t, dc * m, dd * m, da * m, dev)
            mass = hp.oracle_gradient_scales(d, fw, dc * m, dd * m, da * m)
            spread, ref64 = hp.reference_noise_floor(d, fw, fw64, dc * m, dd * m, da * m, seed=seed)
            for n in hp.RULE_NAMES:
                v = getattr(g, n, None)
                if v is None or not torch.is_tensor(v) or v.numel() == 0 or n not in mass:
                    continue
                assert torch.isfinite(v).all(), f"{n} not finite"
                ratio, k = hp.single_rule_ratio(v.cpu().numpy(), getattr(ref64, n), mass[n], spread[n])
                worst_ratio = max(worst_ratio, ratio)
                assert ratio <= hp.RULE_K, f"single rule: {n} element {k} at {ratio:.2f} x (spread + eps mass) from float64 (allowed {hp.RULE_K})"
            # the block masks must be conservative: with culling off the decisions and every gradient equals up to rounding
            nocull = 2                          # MOSS_DEBUG_NO_BLOCK_CULL on both calls (every entry point takes it since ABI 3)
            t0 = hp.hip_forward(d, dev, debug=nocull)
            g0 = hp.hip_backward(d, t0, dc * m, dd * m, da * m, dev, debug=nocull)
            for a, b in ((t.color, t0.color), (t.alpha, t0.alpha), (t.depth, t0.depth)):     # equal up to the order of the per-slot sums
                assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), "culling changed the image"
            # With culling off more (entry, block) pairs leave records, so the same terms are added in another order: the two runs differ
            # by rounding, and what rounding may amount to depends on the element -- an opacity gradient of -3.3 whose contributions
            # add up to a mass of 2e5 moves by 4e-5 (seed 14875: 1e-5 of the tensor's largest value, 2e-10 of its mass); everything
            # derived from dL/dconic through the covariance chain is amplified by the Gaussian's condition number (needles seen end-on:
            # 1e-2 of the mass on dL_dscales, seeds 8036 / 8314 / 8388 / 8679).  So the unculled run is held to the SAME single rule as
            # the culled one -- |g0 - f64| <= RULE_K (spread + RULE_EPS mass), element by element, every tensor -- instead of to a
            # fixed distance from it (rounds 2-3: 1e-5 of the largest value for the sums the kernels form themselves, half the
            # per-Gaussian bar in mass units for the rest).
            for k, v in vars(g).items():
                if v is not None and torch.is_tensor(v) and v.numel() > 0 and k in mass and k in hp.RULE_NAMES:
                    v0 = getattr(g0, k)                      # (segments are cut every 64 HITS: the unculled run cuts elsewhere -> rounding only)
                    ratio, el = hp.single_rule_ratio(v0.cpu().numpy(), getattr(ref64, k), mass[k], spread[k])
                    worst_ratio = max(worst_ratio, ratio)
                    assert ratio <= hp.RULE_K, f"culling off: {k} element {el} at {ratio:.2f} x (spread + eps mass) from float64 (allowed {hp.RULE_K})"
            # round 5: the ASYNCHRONOUS forward (capacity-bounded: keys bucketed by the preprocess kernel, the scan inside the sort kernel,
            # no scatter kernel) must render the same frame bit for bit, down to the sorted lists -- and so must the exact-math mode's
            # integers (its blend decisions are checked against the oracle by tests/test_gpu_exact.py)
            from moss_amd.diff_gaussian_rasterization import _C
            cx = _C.RasterContext(); cx.set_async(True)      # (capacity 0: the first call is synchronous and learns it -- instances AND record-pool cells)
            a_ = t.args; c_ = d.cam
            _C.rasterize_gaussians(a_["bg"], a_["means3D"], a_["colors"], a_["opacity"], a_["scales"], a_["rotations"], d.scale_modifier,
                                   a_["cov3D"], a_["view"], a_["proj"], c_.tanfovx, c_.tanfovy, c_.H, c_.W, a_["sh"], d.degree, a_["campos"],
                                   False, 0, a_["transforms"], 0, cx)
            assert cx.capacity >= 2 * t.R
            ra = _C.rasterize_gaussians(a_["bg"], a_["means3D"], a_["colors"], a_["opacity"], a_["scales"], a_["rotations"], d.scale_modifier,
                                        a_["cov3D"], a_["view"], a_["proj"], c_.tanfovx, c_.tanfovy, c_.H, c_.W, a_["sh"], d.degree, a_["campos"],
                                        False, 0, a_["transforms"], 0, cx)
            cx.check_status()
            assert cx.last_needed == t.R, ("async forward: instances", cx.last_needed, t.R)
            for name_, u_, v_ in (("color", ra[1], t.color), ("depth", ra[2], t.depth), ("alpha", ra[3], t.alpha), ("radii", ra[4], t.radii)):
                assert torch.equal(u_, v_), f"asynchronous forward: {name_} differs from the synchronous forward"
            ta = SimpleNamespace(R=t.R, color=ra[1], depth=ra[2], alpha=ra[3], radii=ra[4], geom=ra[5], binning=ra[6], img=ra[7])
            ea = hp.hip_export(d, ta, dev)
            assert np.array_equal(ea.point_list, e.point_list) and np.array_equal(ea.ranges, e.ranges) and np.array_equal(ea.n_contrib, e.n_contrib)
            ga = _C.rasterize_gaussians_backward(a_["bg"], a_["means3D"], ra[4], a_["colors"], a_["scales"], a_["rotations"], d.scale_modifier,
                                                 a_["cov3D"], a_["view"], a_["proj"], c_.tanfovx, c_.tanfovy, (dc * m).to(dev), (dd * m).to(dev),
                                                 (da * m).to(dev), a_["sh"], d.degree, a_["campos"], ra[5], ra[0], ra[6], ra[7], ra[3], 0,
                                                 a_["transforms"], 0, None, cx)
            for name_, u_ in zip(["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"], ga):
                assert torch.equal(u_, getattr(g, name_)), f"asynchronous path: {name_} differs from the synchronous path's"
    except Exception as ex:                                      # keep going: report every failing seed
        bad += 1
        failures.append({"seed": seed, "P": s.P, "W": s.camera.W, "H": s.camera.H, "mode": mode, "degree": degree, "colors": colors,
                         "error": f"{type(ex).__name__}: {str(ex)[:600]}"})
        print(f"seed {seed}: P={s.P} {s.camera.W}x{s.camera.H} mode={mode} deg={degree} colors={colors}: {type(ex).__name__}: {str(ex)[:300]}")
        if os.environ.get("FUZZ_TRACE"):
            traceback.print_exc()
import json
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"fuzz_failures_{first}_{n_cases}.json"), "w") as f:
    json.dump({"first": first, "cases": n_cases, "failed": failures}, f, indent=1)
print(f"{n_cases - bad} / {n_cases} random cases passed; worst single-rule ratio {worst_ratio:.2f} of the allowed {hp.RULE_K}")
sys.exit(1 if bad else 0)
