"""GPU parity, second line of defence (round 2): the checks that do not lean on the HIP forward's own state, the float64
adjudication, and the configurations / call modes the first suite did not reach.

* END TO END: HIP forward+backward against oracle forward+backward (each with its own final_T / n_contrib), on the pixels where
  every implementation provably takes the same branches (``helpers.stable_mask``: all decision margins > 1e-4 in the float32
  AND the float64 oracle; the incoming image gradients are zeroed elsewhere), at the bars of tests/test_gpu_parity.py.
* ONE RULE FOR EVERY GRADIENT ELEMENT (round 4; rounds 2-3 had two tolerances selected by a hand-kept list of seeds): the reference
  holds no vector for the blend / backward ("parity unpinned"), so where the HIP kernels and the float32
  restatement differ neither is right by definition.  The same C source compiled with float -> double (oracle/Makefile) referees, and
  the tolerance is what the reference's OWN float32 arithmetic scatters by at that element, measured by stochastic arithmetic
  (helpers.reference_noise_floor: the float32 restatement re-run on inputs moved by one float32 ulp, double and float32 accumulators):
      |HIP - f64|  <=  RULE_K * (spread + RULE_EPS * contribution mass)            per element, RULE_K = 8, RULE_EPS = 16 x 2^-24
  asserted on cfg1-3, cfg5 AND on every random scene of tests/golden/fuzz_outlier_seeds.json -- the 147 seeds earlier sweeps flagged
  (image-covering, 300:1 anisotropic Gaussians) are regression cases now, not exceptions: scripts/fuzz_rule.py applies the same
  rule with the same constants to every seed of a sweep (profiles/r04_fuzz_rule_*.log: pass rate).
* BASELINE configs[4] (300k Gaussians, 1024x1024) in cov3D_precomp mode and in the bench's raw-parameter scale/rotation mode;
  BASELINE configs[2] exactly the way bench.py calls the op (render(), raw parameters, gradient sinks into a GradBucket).
"""
import json
import os

import numpy as np
import pytest
import torch

from moss_amd import scenes
from tests import helpers as hp
from tests import test_gpu_parity as tp

pytestmark = pytest.mark.gpu

ADJ_FACTOR = 2.0            # images only (colour, alpha, final_T against float64): no further than 2x the float32 oracle + 2e-6
STABLE = 1e-4

_REPORT = {}


def _note(key, value):
    """Collect the measured error levels; written to gpurun_out/parity_report.json (copied into profiles/ per round)."""
    _REPORT[key] = value
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


GRAD_NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]


def _names(d):
    return GRAD_NAMES + (["dL_dtransforms"] if getattr(d, "transforms", None) is not None else [])


def _end_to_end(d, gpu, key, adjudicate=True, per_gaussian=tp.PER_GAUSSIAN_TOL):
    fw = hp.oracle_forward(d)
    fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=STABLE)
    assert float(1.0 - m.mean()) < 2e-2, "too few stable pixels for a meaningful comparison"
    t = hp.hip_forward(d, gpu)
    e = hp.hip_export(d, t, gpu)
    assert t.R == fw.num_rendered
    ok = m.numpy().astype(bool).reshape(-1)
    # forward, every stable pixel: the same stop index, images within IMG_TOL of the float32 oracle
    np.testing.assert_array_equal(e.n_contrib[ok], fw.n_contrib[ok])
    okc = ok.reshape(d.H, d.W)
    for name, a, b in (("color", e.color, fw.color), ("depth", e.depth, fw.depth), ("alpha", e.alpha, fw.alpha)):
        assert hp.rel_err(a[:, okc], b[:, okc]) < tp.IMG_TOL, name
    # ... and EVERY pixel, the fragile ones included: within what one or two flipped decisions can move it; n_contrib differs only
    # there, on a measured and bounded fraction of the image
    flips = tp.check_every_pixel(d, fw, e)
    img_adj = {}
    for name, a, b32, b64 in (("color", e.color, fw.color, fw64.color), ("alpha", e.alpha, fw.alpha, fw64.alpha),
                              ("final_T", e.final_T.reshape(1, d.H, d.W), fw.final_T.reshape(1, d.H, d.W), fw64.final_T.reshape(1, d.H, d.W))):
        img_adj[name] = (float(np.abs(a[:, okc] - b64[:, okc]).max()), float(np.abs(b32[:, okc] - b64[:, okc]).max()))
        if adjudicate:
            assert img_adj[name][0] <= ADJ_FACTOR * img_adj[name][1] + 2e-6, (name, img_adj[name])
    dc, dd, da = hp.image_grads(d.H, d.W)
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, gpu)
    ref = hp.oracle_backward(d, fw, dc, dd, da)             # the oracle's OWN forward state
    scales = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    got = {n: getattr(g, n).cpu().numpy() for n in _names(d)}
    errs = tp.check_gradients(got, {n: getattr(ref, n) for n in _names(d)}, scales, per_gaussian=per_gaussian)
    # the single rule: every element within RULE_K x (what the reference's own float32 arithmetic scatters by + the summation share)
    spread, ref64 = hp.reference_noise_floor(d, fw, fw64, dc, dd, da)
    adj = {}
    for n in _names(d):
        if n not in scales or not got[n].size:
            continue
        ratio, k = hp.single_rule_ratio(got[n], getattr(ref64, n), scales[n], spread[n])
        adj[n] = ratio
        if adjudicate:
            assert ratio <= hp.RULE_K_BASELINE, f"{n}: element {k} is {ratio:.2f} x (spread + eps mass) from float64 (rule for the BASELINE configurations: {hp.RULE_K_BASELINE})"
    unmasked = tp.check_backward_unmasked(d, gpu, fw, t, e)   # incoming gradients on every pixel, whole-tensor bars
    _note(key, {"fragile_pixels": float(1.0 - m.mean()), "pixels_excluded_by_the_margins": int(round(float((1.0 - m).sum()))), "pixels": int(m.numel()), "every_pixel": flips, "backward_unmasked (relmax, 1-cos, per-Gaussian in mass units)": unmasked,
                "images_vs_f64 (hip, oracle32)": img_adj,
                "grads_vs_oracle32 (relmax, 1-cos, per-Gaussian scaled)": errs, "single_rule_ratio (|hip - f64| / (spread + eps mass), <= RULE_K)": adj})
    return errs, adj


@pytest.mark.parametrize("mode", ["scale_rot", "precomp", "lbs"])
def test_cfg1_end_to_end_and_adjudicated(gpu, hip_lib, mode):
    _end_to_end(hp.inputs_of(scenes.config1(), mode), gpu, f"cfg1_{mode}")


def test_cfg2_end_to_end_and_adjudicated(gpu, hip_lib):
    _end_to_end(hp.inputs_of(scenes.config2(), "precomp"), gpu, "cfg2_precomp")


@pytest.mark.parametrize("mode", ["scale_rot", "precomp", "lbs"])
def test_cfg3_end_to_end_and_adjudicated(gpu, hip_lib, mode):
    """BASELINE configs[2] at full size, both covariance modes and MOSS's real data flow -- per-Gaussian LBS-like transforms applied
    inside the op (row n2, what `value_lbs_in_op` of the bench measures) -- HIP fwd+bwd against oracle fwd+bwd and against float64."""
    sc = scenes.config3()
    if mode == "lbs":
        gT = torch.Generator().manual_seed(1234)                # the transforms bench.py uses (lbs_transforms)
        sc.transforms = torch.eye(3) + 0.05 * torch.randn(sc.means3D.shape[0], 3, 3, generator=gT)
    _end_to_end(hp.inputs_of(sc, mode), gpu, f"cfg3_{mode}")


def test_cfg5_precomp_full_size(gpu, hip_lib):
    """BASELINE configs[4]: 300k Gaussians, 1024x1024 (4096 tiles: tile sort and LDS-occupancy stress), precomputed covariance."""
    d = hp.inputs_of(scenes.config5(), "precomp")
    fw, t, e = tp._check_forward(d, gpu)
    tp._check_backward(d, gpu, fw, t, e)
    _end_to_end(d, gpu, "cfg5_precomp")


def test_mid_size_runs_of_gradient_records(gpu, hip_lib):
    """MOSS's own regime -- tens of thousands of Gaussians on a 1024 x 1024 frame -- scaled down: 12 000 Gaussians at 768 x 768, where a
    Gaussian covers ~5 tiles and ~50 gradient-record cells and hundreds of them have 100-500.  Runs of more than 96 cells (and at most
    sixteen validity words) are summed four Gaussians at a time by sixteen lanes each (preprocess.hip, coop_gather<4>): the owner of a
    run is usually NOT among the lanes that sum it.  (P > 8192: the one-Gaussian-per-lane kernel, not the small-P one.)"""
    sc = scenes.body_scene(12_000, 768, 768, 810.0, init_like=False, name="mid_size_runs")
    d = hp.inputs_of(sc, "scale_rot")
    fw, t, e = tp._check_forward(d, gpu)
    # the scene must be what the docstring says
    al = lambda x: (x + 255) // 256 * 256
    off = al(64 * d.P) + al(4 * d.P) + al(4 * d.P)
    cells = t.geom.cpu().numpy()[off: off + 4 * d.P].view(np.uint32)
    assert int(((cells > 96) & (cells <= 480)).sum()) > 200, int((cells > 96).sum())
    tp._check_backward(d, gpu, fw, t, e)
    _end_to_end(d, gpu, "mid_size_runs")


# ---- raw-parameter mode (what bench.py runs): the op receives logits / log-scales / unnormalised quaternions -------------------

def _raw_parameters(scene, seed=9):
    """MOSS's raw parameters for a scene (GaussianModel: _opacity = inverse_sigmoid, _scaling = log, _rotation unnormalised)."""
    g = torch.Generator().manual_seed(seed)
    opa = scene.opacities.clamp(1e-4, 1 - 1e-4)
    raw_opa = torch.log(opa / (1 - opa)).float().contiguous()
    raw_scl = torch.log(scene.scales).float().contiguous()
    raw_rot = (scene.rotations * (0.3 + 2.0 * torch.rand(scene.rotations.shape[0], 1, generator=g))).float().contiguous()
    return raw_opa, raw_scl, raw_rot


def _raw_case(scene, gpu, key, sinks=False):
    """Raw-parameter forward/backward through the C ABI (or, sinks=True, through render() with a GradBucket exactly like bench.py)
    against the oracle.  The oracle cannot call the device's expf, so it is fed what the kernel actually built from the raw
    parameters -- the activated opacity and the 3-D covariance, read back from the geometry buffer -- in cov3D_precomp mode: every
    integer stage then has to match bit for bit, and the chain raw -> (opacity, covariance) is differentiated independently in
    float64 torch on the host."""
    from moss_amd.diff_gaussian_rasterization import _C
    raw_opa, raw_scl, raw_rot = _raw_parameters(scene)
    d = hp.inputs_of(scene, "scale_rot")
    c = d.cam
    E = torch.Tensor([])
    dev = lambda x: x.to(gpu)
    a = dict(bg=dev(d.bg), means3D=dev(d.means3D), opa=dev(raw_opa), scl=dev(raw_scl), rot=dev(raw_rot), view=dev(c.viewmatrix),
             proj=dev(c.projmatrix), sh=dev(d.shs), campos=dev(c.campos))
    R, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        a["bg"], a["means3D"], E, a["opa"], a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy, c.H, c.W,
        a["sh"], d.degree, a["campos"], False, False, None, 7)
    t = hp.SimpleNamespace(R=R, color=color, depth=depth, alpha=alpha, radii=radii, geom=geom, binning=binning, img=img)
    e = hp.hip_export(d, t, gpu)
    # the oracle, in cov3D_precomp mode on what the kernel built
    vis = e.radii > 0
    d2 = hp.inputs_of(scene, "precomp")
    cov = e.cov3D.copy(); cov[~vis] = scene.cov3D_precomp.numpy()[~vis]          # culled Gaussians: any finite value (never read past the cull)
    d2.cov3D_precomp = torch.from_numpy(cov)
    opa_act = e.conic_opacity[:, 3:4].copy()
    opa_host = (1.0 / (1.0 + np.exp(-raw_opa.numpy().astype(np.float32)))).astype(np.float32)
    assert np.abs(opa_act[vis] - opa_host[vis]).max() <= 2e-7, "sigmoid inside the op"
    opa_act[~vis] = opa_host[~vis]
    d2.opacities = torch.from_numpy(opa_act)
    # covariance the op built vs float64 from the raw parameters
    s64 = torch.exp(raw_scl.double()); q64 = torch.nn.functional.normalize(raw_rot.double())
    cov64 = scenes.covariance_precomp(s64, q64, 1.0, None).numpy()
    assert np.abs(e.cov3D[vis] - cov64[vis]).max() <= 4e-6 * np.abs(cov64[vis]).max(), "covariance inside the op"
    fw = hp.oracle_forward(d2)
    assert R == fw.num_rendered
    np.testing.assert_array_equal(e.radii, fw.radii)
    np.testing.assert_array_equal(e.point_list_keys, fw.point_list_keys)
    np.testing.assert_array_equal(e.point_list, fw.point_list)
    np.testing.assert_array_equal(e.ranges, fw.ranges)
    ok = tp._stable_pixels(fw)
    assert (~ok).mean() < 2e-3
    np.testing.assert_array_equal(e.n_contrib[ok], fw.n_contrib[ok])
    okc = ok.reshape(d.H, d.W)
    for name, x, y in (("color", e.color, fw.color), ("depth", e.depth, fw.depth), ("alpha", e.alpha, fw.alpha)):
        assert hp.rel_err(x[:, okc], y[:, okc]) < tp.IMG_TOL, name
    # backward: incoming gradients on the stable pixels only, oracle with its own forward state
    m = hp.stable_mask(d2, fw, thr=STABLE)
    dc, dd, da = hp.image_grads(d.H, d.W)
    dc, dd, da = dc * m, dd * m, da * m
    if sinks:
        got = _bench_mode_gradients(scene, gpu, raw_opa, raw_scl, raw_rot, dc, dd, da, color)
    else:
        grads = _C.rasterize_gaussians_backward(
            a["bg"], a["means3D"], radii, E, a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy, dev(dc), dev(dd),
            dev(da), a["sh"], d.degree, a["campos"], geom, R, binning, img, alpha, False, None, 7, a["opa"])
        got = dict(zip(["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"],
                       [x.cpu().numpy() for x in grads]))
    ref = hp.oracle_backward(d2, fw, dc, dd, da)
    scales = hp.oracle_gradient_scales(d2, fw, dc, dd, da)
    # chain rule of the getters in float64 on the host: raw -> (sigmoid, covariance(exp, normalize))
    rs = raw_scl.double().requires_grad_(True); rq = raw_rot.double().requires_grad_(True)
    cov_t = scenes.covariance_precomp(torch.exp(rs), torch.nn.functional.normalize(rq), 1.0, None)
    cov_t.backward(torch.from_numpy(ref.dL_dcov3D.astype(np.float64)))
    sg = torch.sigmoid(raw_opa.double()).numpy()
    want = {"dL_dmeans3D": ref.dL_dmeans3D, "dL_dsh": ref.dL_dsh, "dL_dopacity": ref.dL_dopacity.astype(np.float64) * sg * (1 - sg),
            "dL_dscales": rs.grad.numpy(), "dL_drotations": rq.grad.numpy()}
    if "dL_dmeans2D" in got:
        want["dL_dmeans2D"] = ref.dL_dmeans2D
    # per-Gaussian scales of the raw gradients: the covariance scales pushed through |d cov / d raw| (float64 autograd, 6 probes)
    sc = {"dL_dmeans3D": scales["dL_dmeans3D"], "dL_dsh": scales["dL_dsh"], "dL_dopacity": scales["dL_dopacity"] * np.abs(sg * (1 - sg))}
    if "dL_dmeans2D" in got:
        sc["dL_dmeans2D"] = scales["dL_dmeans2D"]
    acc_s = np.zeros_like(want["dL_dscales"]); acc_q = np.zeros_like(want["dL_drotations"])
    for j in range(6):
        rs2 = raw_scl.double().requires_grad_(True); rq2 = raw_rot.double().requires_grad_(True)
        cj = scenes.covariance_precomp(torch.exp(rs2), torch.nn.functional.normalize(rq2), 1.0, None)[:, j]
        cj.sum().backward()
        acc_s += np.abs(rs2.grad.numpy()) * scales["dL_dcov3D"][:, j:j + 1]
        acc_q += np.abs(rq2.grad.numpy()) * scales["dL_dcov3D"][:, j:j + 1]
    sc["dL_dscales"], sc["dL_drotations"] = acc_s, acc_q
    errs = tp.check_gradients(got, want, sc)
    _note(key, {"grads_vs_oracle32 (relmax, 1-cos, per-Gaussian scaled)": errs})


def _bench_mode_gradients(scene, gpu, raw_opa, raw_scl, raw_rot, dc, dd, da, color_direct):
    """The op exactly as bench.py drives it: render() on a GaussianSet holding the raw parameters, raw_parameters_in_op, one unified
    SH parameter, every parameter gradient written by the backward kernel into its slice of a GradBucket (gradient sinks)."""
    from types import SimpleNamespace
    from moss_amd import diff_gaussian_rasterization as dgr
    from moss_amd import dist as mdist
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import camera_view, render
    pc = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
    with torch.no_grad():
        pc._opacity.copy_(raw_opa.to(gpu)); pc._scaling.copy_(raw_scl.to(gpu)); pc._rotation.copy_(raw_rot.to(gpu))
    bucket = mdist.GradBucket(list(pc.parameters()))
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=True,
                           transforms_in_op=False, raw_parameters_in_op=True, grad_bucket=bucket)
    dgr.set_grad_sink(sh=lambda: bucket.sink_for(pc._features), opacity=lambda: bucket.sink_for(pc._opacity),
                      scales=lambda: bucket.sink_for(pc._scaling), rotations=lambda: bucket.sink_for(pc._rotation),
                      means3D=lambda: bucket.sink_for(pc._xyz))
    try:
        bucket.detach_grads()
        out = render(camera_view(scene.camera, gpu), pc, pipe, scene.bg.to(gpu))
        assert torch.equal(out["render"], color_direct), "render() and the direct C-ABI call must run the same kernels"
        loss = (out["render"] * dc.to(gpu)).sum() + (out["render_depth"] * dd.to(gpu)).sum() + (out["render_alpha"] * da.to(gpu)).sum()
        loss.backward()
        bucket.collect()
    finally:
        dgr.set_grad_sink()
    for p_ in pc.parameters():                               # every gradient sits in the bucket, no copy was needed
        off = bucket._offset[id(p_)]
        assert p_.grad.data_ptr() == bucket.flat[off:off + 1].data_ptr()
    return {"dL_dmeans3D": pc._xyz.grad.cpu().numpy(), "dL_dsh": pc._features.grad.cpu().numpy(),
            "dL_dopacity": pc._opacity.grad.cpu().numpy(), "dL_dscales": pc._scaling.grad.cpu().numpy(),
            "dL_drotations": pc._rotation.grad.cpu().numpy()}


def test_cfg3_exactly_as_the_bench_calls_it(gpu, hip_lib):
    """BASELINE configs[2], scale_rot + raw_flags = 7 + gradient sinks into a GradBucket through render(): bench.py's call path."""
    _raw_case(scenes.config3(), gpu, "cfg3_bench_mode", sinks=True)


def test_cfg3_raw_parameters_direct(gpu, hip_lib):
    _raw_case(scenes.config3(), gpu, "cfg3_raw")


def test_cfg5_raw_scale_rot_full_size(gpu, hip_lib):
    """BASELINE configs[4] in the bench's mode: raw parameters, covariance from scale / rotation inside the op."""
    _raw_case(scenes.config5(), gpu, "cfg5_raw")


# ---- the random scenes the round-1 fuzz sweep flagged (scripts/fuzz_parity.py: gradient differences above 2e-2 of the largest value) ----

def _fuzz_outlier_seeds():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_outlier_seeds.json")
    with open(path) as f:
        return json.load(f)["seeds"]


@pytest.mark.parametrize("seed", _fuzz_outlier_seeds())
def test_fuzz_scenes_hold_the_single_rule(gpu, hip_lib, seed):
    """Random scenes with 50-600:1 anisotropic Gaussians over dozens to hundreds of tiles -- the 147 seeds that rounds 2-3's sweeps
    flagged against their plain bars and then passed under a looser, second tolerance.  Round 4 holds them to the SAME rule as the
    BASELINE configurations (helpers.RULE_K, RULE_EPS; measured worst on these seeds: 3.9 of the allowed 8): the scale / rotation
    gradient of such a needle is a difference of terms 100-1000x its size, the reference's own float32 arithmetic scatters there by
    just that much, and the rule's tolerance is that scatter, measured per element.  Images: no further from float64 than 4x the
    float32 oracle; n_contrib exact on the stable pixels; nothing NaN."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_scenes", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_scenes.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    s, mode, degree, colors = fz.random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d)
    if fw.num_rendered == 0:
        pytest.skip("nothing rendered")
    fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=STABLE)
    t = hp.hip_forward(d, gpu)
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1))
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, gpu)
    scales = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    spread, ref64 = hp.reference_noise_floor(d, fw, fw64, dc, dd, da, seed=seed)
    got = {n: getattr(g, n).cpu().numpy() for n in _names(d)}
    e = hp.hip_export(d, t, gpu)
    ok = m.numpy().astype(bool)
    np.testing.assert_array_equal(e.n_contrib[ok.reshape(-1)], fw.n_contrib[ok.reshape(-1)])
    for name, a_, b32, b64 in (("color", e.color, fw.color, fw64.color), ("alpha", e.alpha, fw.alpha, fw64.alpha)):
        h, o = float(np.abs(a_ - b64)[:, ok].max()), float(np.abs(b32 - b64)[:, ok].max())
        assert h <= 4.0 * o + 2e-6, (name, h, o)              # images: no further from float64 than 4x the float32 oracle
    adj = {}
    for n in _names(d):
        if not got[n].size:
            continue
        assert np.isfinite(got[n]).all(), n
        ratio, k = hp.single_rule_ratio(got[n], getattr(ref64, n), scales[n], spread[n])
        adj[n] = ratio
        assert ratio <= hp.RULE_K, (n, k, ratio)
    _note(f"fuzz{seed}", {"single_rule_ratio": adj})


@pytest.mark.gpu
@pytest.mark.parametrize("P", [320, 1400])
def test_many_gradient_records_per_gaussian(gpu, hip_lib, P):
    """Wide, faint Gaussians: every one reaches ~9 tiles and nearly every 4x4 block of them without saturating a pixel, so a wave of
    the per-Gaussian backward owns thousands of gradient records (P=320: light tiles, up to 4 quadrant records per instance; P=1400:
    heavy tiles, up to 16 block records per instance) -- the wave-balanced gather (csrc/preprocess.hip) runs many passes of its
    record list and every row is a mixture of long runs.  Checked against the oracle like every other configuration, plus run-to-run
    bitwise reproducibility."""
    s = scenes.config1(P=P, seed=77)
    g = torch.Generator().manual_seed(123)
    s.scales = torch.exp(np.log(0.11) + 0.15 * torch.randn(P, 3, generator=g))
    s.opacities = 0.01 + 0.05 * torch.rand(P, 1, generator=g)
    s.cov3D_precomp = scenes.covariance_precomp(s.scales, s.rotations, 1.0, s.transforms)
    d = hp.inputs_of(s, "scale_rot")
    fw, t, e = tp._check_forward(d, gpu)
    tiles = e.tiles_touched.astype(np.int64)
    assert tiles[tiles > 0].mean() > 6.0                      # the scene does what it is meant to do
    g1 = tp._check_backward(d, gpu, fw, t, e)
    dc, dd, da = hp.image_grads(d.H, d.W)
    a = hp.hip_backward(d, t, dc, dd, da, gpu)
    b = hp.hip_backward(d, t, dc, dd, da, gpu)
    for k, v in vars(a).items():
        if torch.is_tensor(v) and v.numel() > 0:
            assert torch.equal(v, getattr(b, k)), k
