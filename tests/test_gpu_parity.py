"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star):
  * BIT-EXACT for integer work: radii, tiles_touched, depth bits, the sorted (tile<<32|depth) keys, the sorted
    Gaussian ids, the per-tile ranges; n_contrib on every pixel whose decisions are not within rounding of a threshold
    (the blend kernels evaluate exp() with the hardware's v_exp_f32 and 1/(1-alpha) with v_rcp_f32, ~1 ulp each, where the
    reference calls expf() and divides: a pixel whose test sits within FRAGILE of its threshold may take the other branch);
  * floating point, stated tolerances:
      preprocess outputs (means2D, conic, rgb, cov3D)  : exact equality (same operation order, no contraction)
      images (colour, depth, alpha, final_T)            : IMG_TOL  = 2e-5 of the largest value
      gradients, per tensor                             : GRAD_TOL = 2e-5 of the largest value AND 1 - cosine <= COS_GAP_TOL = 1e-9
      gradients, per Gaussian and element               : |hip - oracle| <= PER_GAUSSIAN_TOL x the element's CONTRIBUTION MASS
        (sum over pixels of the absolute terms behind it, oracle.gradient_scales): an fp32 sum carries an error proportional to
        that mass, not to the possibly cancelled result, so this is the scale at which a single Gaussian's gradient -- a small
        opacity, a degree-3 SH coefficient -- can be called right or wrong; elements without any contribution must be exactly 0.
        The float32 oracle itself sits up to ~1e-3 of the mass away from the float64 evaluation on cfg3 (T = T / (1 - alpha)
        chains through alpha ~ 0.99); tests/test_gpu_parity_hardened.py adjudicates both against float64.
"""
import numpy as np
import pytest
import torch

from moss_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu

IMG_TOL = 2e-5
GRAD_TOL = 2e-5           # round 1: 2e-4.  Measured (profiles/r02_parity_report.json): <= 8e-7 on every configuration
COS_GAP_TOL = 1e-9        # measured <= 3e-13
PER_GAUSSIAN_TOL = 2e-3   # measured <= 4e-4 (cfg5, rotation gradients through 1/(1-alpha) chains); typical 1e-6
FRAGILE = 2e-5      # pixels whose oracle decision margin is below this may legitimately flip a threshold
# What ONE flipped decision can move a pixel by, as a fraction of the largest per-entry value c_max of that image (colour: the largest
# |rgb| / depth: the largest depth / alpha: 1).  alpha >= 1/255 taken the other way: the entry's own alpha T c <= c_max / 255, and
# every later weight scales by (1 - 1/255): <= 2 c_max / 255.  T (1 - alpha) < 1e-4 taken the other way: the entry is blended (or
# not) with weight alpha T, where T = 1e-4 / (1 - alpha) <= 1e-2 at alpha <= 0.99: <= 0.0099 c_max, and everything behind it carries
# T <= 1e-4.  A pixel within FRAGILE of a threshold may take up to FLIPS_PER_PIXEL such flips (measured: 1).
FLIP_BOUND = 0.0099 + 1e-4
FLIPS_PER_PIXEL = 2
MAX_FLIP_RATE = 1e-4  # asserted: fraction of pixels whose n_contrib differs from the oracle's.  Measured (profiles/r03_parity_report.json): ONE pixel
                      # of cfg3 in scale/rotation mode (3.8e-6), none on cfg1 / cfg2 / cfg3 precomp / cfg3 lbs / cfg5
# backward with UNMASKED incoming gradients (fragile pixels included), oracle backward on the HIP forward's state: only the backward's
# own alpha >= 1/255 decisions can differ, on <= 0.01 % of the pixels -- whole-tensor bars only (one flipped entry is the whole
# contribution mass of a Gaussian that is seen by that pixel alone)
UNMASKED_GRAD_TOL = 2e-4   # measured <= 5.5e-5 (cfg5; <= 7e-7 on cfg1-3)
UNMASKED_COS_GAP = 1e-9    # measured <= 2.7e-11


def _stable_pixels(fw):
    return fw.margin > FRAGILE


def _check_forward(d, gpu, check_images=True, max_fragile=2e-3):
    fw = hp.oracle_forward(d)
    t = hp.hip_forward(d, gpu)
    e = hp.hip_export(d, t, gpu)
    # ---- K1 preprocess: integers exact, floats exact (same op order, contraction off on both sides)
    assert t.R == fw.num_rendered
    np.testing.assert_array_equal(e.radii, fw.radii)
    np.testing.assert_array_equal(e.tiles_touched, fw.tiles_touched)
    np.testing.assert_array_equal(e.depths.view(np.uint32), fw.depths.view(np.uint32))
    np.testing.assert_array_equal(e.means2D, fw.means2D)
    np.testing.assert_array_equal(e.conic_opacity, fw.conic_opacity)
    if d.colors_precomp is None:
        np.testing.assert_array_equal(e.rgb, fw.rgb)
        np.testing.assert_array_equal(e.clamped, fw.clamped)
    if d.cov3D_precomp is None:
        vis = fw.radii > 0
        np.testing.assert_array_equal(e.cov3D[vis], fw.cov3D[vis])
    # ---- K2-K5 binning: bit-exact keys / ids / ranges
    np.testing.assert_array_equal(e.point_list_keys, fw.point_list_keys)
    np.testing.assert_array_equal(e.point_list, fw.point_list)
    np.testing.assert_array_equal(e.ranges, fw.ranges)
    # ---- K6 blend
    if check_images:
        ok = _stable_pixels(fw)
        assert (~ok).mean() < max_fragile, "too many threshold-fragile pixels for a meaningful comparison"
        np.testing.assert_array_equal(e.n_contrib[ok], fw.n_contrib[ok])
        okc = ok.reshape(d.H, d.W)
        for name, a, b in (("color", e.color, fw.color), ("depth", e.depth, fw.depth), ("alpha", e.alpha, fw.alpha)):
            assert hp.rel_err(a[:, okc], b[:, okc]) < IMG_TOL, name
        assert hp.rel_err(e.final_T[ok], fw.final_T[ok]) < IMG_TOL
        # (callers that admit more fragile pixels -- fuzz scenes, opacities placed on the 1/255 threshold -- admit as many flips)
        fw.flip_stats = check_every_pixel(d, fw, e, max_flip_rate=MAX_FLIP_RATE if max_fragile <= 2e-3 else max_fragile)
    return fw, t, e


def check_every_pixel(d, fw, e, fragile=FRAGILE, max_flip_rate=MAX_FLIP_RATE):
    """The forward outputs over ALL pixels, the threshold-fragile ones included (they used to be excluded from
    every comparison).  (i) every pixel of every image is within IMG_TOL of the oracle's, plus -- only where the oracle's decision
    margin is below `fragile` -- what FLIPS_PER_PIXEL flipped decisions can move it by (FLIP_BOUND x the image's largest per-entry
    value); (ii) n_contrib differs from the oracle's ONLY on such pixels; (iii) the fraction of pixels where it differs is below
    `max_flip_rate`.  Returns the measured numbers (they go into the parity report)."""
    frag = ~(np.asarray(fw.margin) > fragile)
    fragc = frag.reshape(d.H, d.W)
    differs = e.n_contrib != fw.n_contrib
    assert not (differs & ~frag).any(), "n_contrib differs on a pixel whose decisions are all clear of their thresholds"
    flip_rate = float(differs.mean())
    assert flip_rate <= max_flip_rate, f"n_contrib differs from the oracle's on {flip_rate:.2e} of the pixels"
    vis = fw.radii > 0
    feat = np.asarray(fw.features)
    cmax = {"color": max(float(np.abs(feat[vis]).max()) if vis.any() else 0.0, float(np.abs(d.bg.numpy()).max())),
            "depth": float(fw.depths[vis].max()) if vis.any() else 0.0, "alpha": 1.0, "final_T": 1.0}
    worst = {}
    for name, a, b in (("color", e.color, fw.color), ("depth", e.depth, fw.depth), ("alpha", e.alpha, fw.alpha),
                       ("final_T", e.final_T.reshape(1, d.H, d.W), fw.final_T.reshape(1, d.H, d.W))):
        a = np.asarray(a, np.float64).reshape(-1, d.H, d.W); b = np.asarray(b, np.float64).reshape(-1, d.H, d.W)
        base = IMG_TOL * max(float(np.abs(b).max()), 1e-30)
        # final_T: a flipped stop decision leaves T at the value in front of the entry instead of behind it (or the reverse): <= 1e-2
        allowed = base + fragc[None] * (FLIPS_PER_PIXEL * FLIP_BOUND * cmax[name])
        diff = np.abs(a - b)
        assert (diff <= allowed).all(), (name, float(diff.max()), float((diff - allowed).max()))
        worst[name] = float(diff[:, fragc].max()) if fragc.any() else 0.0
    return {"fragile_fraction": float(frag.mean()), "n_contrib_flip_rate": flip_rate, "n_contrib_flips": int(differs.sum()),
            "worst_fragile_pixel_error": worst}


def check_gradients(got, ref, scales, tol=GRAD_TOL, per_gaussian=PER_GAUSSIAN_TOL, cos_gap=COS_GAP_TOL):
    """The three gradient bars of this suite (see the module docstring) for {name: array} against {name: array}.  Returns
    {name: (relative max error, 1 - cosine, largest |diff| / contribution mass)}."""
    errs, bad = {}, {}
    for name, r in ref.items():
        a = np.asarray(got[name])
        assert a.shape == r.shape, name
        assert np.isfinite(a).all(), name
        live, dead = hp.scaled_err(a, r, scales[name]) if name in scales else (0.0, 0.0)
        errs[name] = (hp.rel_err(a, r), hp.cosine_gap(a, r), live)
        if errs[name][0] > tol or errs[name][1] > cos_gap or live > per_gaussian or dead != 0.0:
            bad[name] = errs[name] + (dead,)
    assert not bad, f"gradient mismatch (rel max, 1-cos, per-Gaussian scaled, dead-element diff): {bad}"
    return errs


def gaussians_seen_by_fragile_pixels(d, fw, thr=1e-4):
    """(P,) bool: the Gaussians that can CONTRIBUTE to a pixel with a decision (power > 0, alpha >= 1/255, T (1 - alpha) < 1e-4) within
    `thr` (relative) of its threshold in the oracle -- the entries of that pixel's tile list whose alpha at the pixel reaches 0.9 / 255
    (a margin of 10 % around the blend's own 1 / 255 test).  Only on such pixels can the product path's v_exp_f32 / FMA / v_rcp_f32
    arithmetic take another branch than the reference's (tests/test_gpu_exact.py: fast and exact n_contrib differ on such pixels only);
    every other Gaussian receives its gradient exclusively from pixels on which both sides took IDENTICAL decisions, entry by entry."""
    from oracle import oracle
    frag = ~(np.asarray(fw.margin).reshape(-1) > thr)
    seen = np.zeros(d.P, bool)
    if frag.any():
        gx, _ = oracle.tile_grid(d.W, d.H)
        rg = np.asarray(fw.ranges).reshape(-1, 2)
        xy = np.asarray(fw.means2D, np.float64).reshape(-1, 2)
        co = np.asarray(fw.conic_opacity, np.float64).reshape(-1, 4)
        for pix in np.nonzero(frag)[0]:
            py, px = int(pix) // d.W, int(pix) % d.W
            tl = (py // 16) * gx + px // 16
            ids = np.asarray(fw.point_list[int(rg[tl, 0]):int(rg[tl, 1])], np.int64)
            if not ids.size:
                continue
            dx, dy = xy[ids, 0] - px, xy[ids, 1] - py
            power = -0.5 * (co[ids, 0] * dx * dx + co[ids, 2] * dy * dy) - co[ids, 1] * dx * dy
            alpha = co[ids, 3] * np.exp(np.minimum(power, 0.0))
            seen[ids[(alpha >= 0.9 / 255.0) & (power <= 1e-3)]] = True
    return seen, int(frag.sum())


def check_backward_unmasked(d, gpu, fw, t, e, zero_depth=False, tol=UNMASKED_GRAD_TOL, cos_gap=UNMASKED_COS_GAP, per_gaussian=PER_GAUSSIAN_TOL):
    """The backward with incoming gradients on EVERY pixel (no stable-pixel mask), against the oracle backward on the HIP forward's
    (final_T, n_contrib): whole-tensor bars at the stated looser tolerance -- and (round 6) the PER-GAUSSIAN bar of the masked tests,
    asserted on the PRODUCT kernels for every Gaussian that no threshold-fragile pixel sees (`gaussians_seen_by_fragile_pixels`: all
    but a few per cent): there kernel and oracle took identical decisions, so nothing but the rounding of the sums is left.  (Rounds
    1-5 asserted that bar with unmasked gradients on the MOSS_DEBUG_EXACT_MATH instantiations only.)  Returns {name: (relative max
    error, 1 - cosine, largest per-element error in units of the contribution mass over ALL Gaussians)}."""
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=zero_depth)
    g = hp.hip_backward(d, t, dc, dd, da, gpu)
    ref = hp.oracle_backward(d, hp.replace_forward_state(fw, e), dc, dd, da)
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    if getattr(d, "transforms", None) is not None:
        names.append("dL_dtransforms")
    # ... and the per-Gaussian level in units of the element's contribution mass (MEASURED and reported over all Gaussians, round 3's
    # review item 4; ASSERTED below on the Gaussians clear of every fragile pixel: one flipped alpha >= 1/255 decision on a fragile pixel
    # moves a few-pixel Gaussian's element by more than all rounding together)
    scales = hp.oracle_gradient_scales(d, hp.replace_forward_state(fw, e), dc, dd, da)
    seen, n_frag = gaussians_seen_by_fragile_pixels(d, fw)
    clear = ~seen
    assert clear.mean() > 0.2, f"{int(seen.sum())} of {d.P} Gaussians are seen by one of {n_frag} fragile pixels: nothing left to assert on"
    errs, bad = {}, {}
    for n in names:
        a, r = getattr(g, n).cpu().numpy(), getattr(ref, n)
        assert np.isfinite(a).all(), n
        live = hp.scaled_err(a, r, scales[n])[0] if n in scales and a.size else 0.0
        errs[n] = (hp.rel_err(a, r), hp.cosine_gap(a, r), live)
        if errs[n][0] > tol or errs[n][1] > cos_gap:
            bad[n] = errs[n]
        if n in scales and a.size and a.shape[0] == d.P:
            sc = np.asarray(scales[n]).reshape(a.shape)
            live_c, dead_c = hp.scaled_err(a[clear], np.asarray(r).reshape(a.shape)[clear], sc[clear])
            if live_c > per_gaussian or dead_c != 0.0:
                bad[n + " (per Gaussian, unmasked, clear of fragile pixels)"] = (live_c, dead_c)
            errs[n] = errs[n] + (live_c,)
    assert not bad, f"unmasked backward (rel max, 1-cos, per-Gaussian in mass units[, the same over the clear Gaussians]): {bad}"
    errs["gaussians_clear_of_fragile_pixels"] = (int(clear.sum()), int(d.P), n_frag)
    return errs


def _check_backward(d, gpu, fw, t, e, zero_depth=False, tol=GRAD_TOL, per_gaussian=PER_GAUSSIAN_TOL, cos_gap=COS_GAP_TOL):
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=zero_depth)
    # Only pixels whose every decision is clear of its threshold carry an incoming gradient: at the others the kernels (v_exp_f32,
    # fused multiply-adds) and the oracle (expf, one rounding per operation) may legitimately skip different entries, and one flipped
    # alpha >= 1/255 test moves a gradient by more than all rounding together (2-5e-5 of the largest value on cfg5).
    m = hp.stable_mask(d, fw, thr=1e-4)
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, gpu)
    # backward arithmetic in isolation: the oracle backward consumes the HIP forward's (final_T, n_contrib)
    ref = hp.oracle_backward(d, hp.replace_forward_state(fw, e), dc, dd, da)
    pairs = [("dL_dmeans2D", ref.dL_dmeans2D), ("dL_dcolors", ref.dL_dcolors), ("dL_dopacity", ref.dL_dopacity),
             ("dL_dmeans3D", ref.dL_dmeans3D), ("dL_dcov3D", ref.dL_dcov3D), ("dL_dsh", ref.dL_dsh),
             ("dL_dscales", ref.dL_dscales), ("dL_drotations", ref.dL_drotations)]
    if getattr(d, "transforms", None) is not None:            # n2 extension: transform applied inside the op
        pairs.append(("dL_dtransforms", ref.dL_dtransforms))
    errs = check_gradients({name: getattr(g, name).cpu().numpy() for name, _ in pairs}, dict(pairs),
                           hp.oracle_gradient_scales(d, hp.replace_forward_state(fw, e), dc, dd, da), tol=tol, per_gaussian=per_gaussian, cos_gap=cos_gap)
    g.errors = errs
    # culled Gaussians get exactly zero everywhere
    inv = torch.from_numpy(fw.radii <= 0)
    if inv.any():
        for name, _ in pairs:
            assert float(getattr(g, name).cpu()[inv].abs().sum()) == 0.0, name
    return g


@pytest.mark.parametrize("mode", ["scale_rot", "precomp", "lbs"])
@pytest.mark.parametrize("zero_depth", [True, False])
def test_cfg1_forward_backward(gpu, hip_lib, mode, zero_depth):
    d = hp.inputs_of(scenes.config1(), mode)
    fw, t, e = _check_forward(d, gpu)
    _check_backward(d, gpu, fw, t, e, zero_depth=zero_depth)


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_cfg1_lower_sh_degrees(gpu, hip_lib, degree):
    d = hp.inputs_of(scenes.config1(), "precomp", degree=degree)
    fw, t, e = _check_forward(d, gpu)
    g = _check_backward(d, gpu, fw, t, e)
    used = (degree + 1) ** 2
    assert float(g.dL_dsh[:, used:, :].abs().sum()) == 0.0


def test_cfg1_colors_precomp_and_background(gpu, hip_lib):
    d = hp.inputs_of(scenes.config1(), "scale_rot", colors=True, bg=[0.3, 0.6, 0.9])
    fw, t, e = _check_forward(d, gpu)
    _check_backward(d, gpu, fw, t, e)


def test_cfg2_body_init(gpu, hip_lib):
    d = hp.inputs_of(scenes.config2(), "precomp")
    fw, t, e = _check_forward(d, gpu)
    _check_backward(d, gpu, fw, t, e)


def test_cfg3_full_size(gpu, hip_lib):
    """BASELINE configs[2] at full size (100k Gaussians, 512x512): the oracle still finishes in ~1 s."""
    d = hp.inputs_of(scenes.config3(), "precomp")
    fw, t, e = _check_forward(d, gpu)
    _check_backward(d, gpu, fw, t, e)


def test_ragged_image_size(gpu, hip_lib):
    """W, H not multiples of 16: partial edge tiles (the reference reads out of bounds there, backward.cu:460-464)."""
    s = scenes.config1()
    s.camera = scenes.make_camera(100, 70, 120.0, 120.0, 52.0, 33.0, np.eye(3), np.array([0.0, 0.0, 3.0]))
    d = hp.inputs_of(s, "scale_rot")
    fw, t, e = _check_forward(d, gpu)
    _check_backward(d, gpu, fw, t, e)


def test_gradients_are_bitwise_reproducible(gpu, hip_lib):
    d = hp.inputs_of(scenes.config2(), "precomp")
    dc, dd, da = hp.image_grads(d.H, d.W)
    outs = []
    for _ in range(2):
        t = hp.hip_forward(d, gpu)
        g = hp.hip_backward(d, t, dc, dd, da, gpu)
        outs.append([getattr(g, n).cpu() for n in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dcov3D", "dL_dmeans2D")])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_cfg1_raw_parameter_entry_points_against_the_oracle(gpu):
    """moss_raster_forward_raw / _backward_raw: raw logits / log-scales / unnormalised quaternions in, image and RAW-parameter
    gradients out, against the C oracle run on the activated values (float32 getters in numpy) with the chain rule of the getters
    applied to its gradients in float64."""
    from moss_amd.diff_gaussian_rasterization import _C
    sc = scenes.config1()
    d = hp.inputs_of(sc, "scale_rot")
    g = torch.Generator().manual_seed(9)
    raw_opa = torch.randn(d.P, 1, generator=g)
    raw_scl = torch.log(d.scales)
    raw_rot = d.rotations * (0.3 + 2.0 * torch.rand(d.P, 1, generator=g))
    # the getters in float32, the way the kernels evaluate them
    d.opacities = (1.0 / (1.0 + torch.exp(-raw_opa))).float()
    d.scales = torch.exp(raw_scl).float()
    nrm = torch.sqrt((raw_rot * raw_rot).sum(1, keepdim=True)).clamp_min(1e-12)
    d.rotations = (raw_rot * (1.0 / nrm)).float()
    fw = hp.oracle_forward(d)
    dc, dd, da = hp.image_grads(d.H, d.W, seed=5)
    gb = hp.oracle_backward(d, fw, dc, dd, da)

    c = d.cam
    e = torch.Tensor([])
    dev = lambda t: t.to(gpu)
    R, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        dev(d.bg), dev(d.means3D), e, dev(raw_opa), dev(raw_scl), dev(raw_rot), 1.0, e, dev(c.viewmatrix), dev(c.projmatrix),
        c.tanfovx, c.tanfovy, c.H, c.W, dev(d.shs), d.degree, dev(c.campos), False, False, None, 7)
    assert R == fw.num_rendered
    np.testing.assert_array_equal(radii.cpu().numpy(), fw.radii)
    ok = _stable_pixels(fw).reshape(d.H, d.W)
    assert hp.rel_err(color.cpu().numpy()[:, ok], fw.color[:, ok]) < IMG_TOL
    assert hp.rel_err(alpha.cpu().numpy()[:, ok], fw.alpha[:, ok]) < IMG_TOL
    grads = _C.rasterize_gaussians_backward(
        dev(d.bg), dev(d.means3D), radii, e, dev(raw_scl), dev(raw_rot), 1.0, e, dev(c.viewmatrix), dev(c.projmatrix),
        c.tanfovx, c.tanfovy, dev(dc), dev(dd), dev(da), dev(d.shs), d.degree, dev(c.campos), geom, R, binning, img, alpha, False,
        None, 7, dev(raw_opa))
    dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot = [x.cpu().numpy() for x in grads]
    # chain rule of the getters on the oracle's gradients (float64)
    s = d.opacities.double().numpy()
    want_opa = gb.dL_dopacity.astype(np.float64).reshape(-1, 1) * (s * (1 - s))
    want_scl = gb.dL_dscales.astype(np.float64) * d.scales.double().numpy()
    y = d.rotations.double().numpy(); gq = gb.dL_drotations.astype(np.float64); n = nrm.double().numpy()
    want_rot = (gq - y * (y * gq).sum(1, keepdims=True)) / n
    assert hp.rel_err(dL_dopacity, want_opa) < GRAD_TOL
    assert hp.rel_err(dL_dscales, want_scl) < GRAD_TOL
    assert hp.rel_err(dL_drot, want_rot) < GRAD_TOL
    assert hp.rel_err(dL_dmeans3D, gb.dL_dmeans3D) < GRAD_TOL
    assert hp.rel_err(dL_dsh, gb.dL_dsh) < GRAD_TOL
