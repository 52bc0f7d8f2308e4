"""CPU tests that PIN THE ORACLE: golden vectors produced by the reference's own Python (tests/golden/make_golden.py),
closed-form known answers, and an independent float64 autograd restatement for the explicit backward."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from moss_amd import scenes
from oracle import autograd_rasterizer as ag
from oracle import oracle
from tests import helpers as hp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold(name):
    return np.load(os.path.join(GOLD, name))


def _preprocess_only(means, shs, degree, campos, scales=None, rots=None, cov=None, W=64, H=64, tan=2.0, t=(0, 0, 10.0), op=None,
                     transforms=None):
    """Run just the oracle's preprocess stage on a wide camera so that nothing is culled."""
    L = oracle.lib()
    P = means.shape[0]
    cam = scenes.make_camera(W, H, W / (2 * tan), H / (2 * tan), W / 2, H / 2, np.eye(3), np.array(t, dtype=float))
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    means, shs, scales, rots, cov = f(means), f(shs), f(scales), f(rots), f(cov)
    transforms = None if transforms is None else f(np.asarray(transforms).reshape(-1, 9))
    op = np.full(P, 0.5, np.float32) if op is None else f(op)
    view, proj, cp = f(cam.viewmatrix.numpy()), f(cam.projmatrix.numpy()), f(campos)
    radii = np.zeros(P, np.int32); xy = np.zeros((P, 2), np.float32); depths = np.zeros(P, np.float32)
    cov3D = np.zeros((P, 6), np.float32); rgb = np.zeros((P, 3), np.float32); conic = np.zeros((P, 4), np.float32)
    tiles = np.zeros(P, np.uint32); clamped = np.zeros((P, 3), np.uint8)
    M = 0 if shs is None else shs.shape[1]
    err = L.oracle_preprocess(C.c_int(P), C.c_int(degree), C.c_int(M), p(means), p(scales), C.c_float(1.0), p(rots), p(op), p(shs),
                              p(cov), None, p(view), p(proj), p(cp), C.c_int(W), C.c_int(H), C.c_float(cam.tanfovx),
                              C.c_float(cam.tanfovy), C.c_int(0), p(radii), p(xy), p(depths), p(cov3D), p(rgb), p(conic), p(tiles),
                              p(clamped), p(transforms))
    assert err == 0
    return dict(radii=radii, xy=xy, depths=depths, cov3D=cov3D, rgb=rgb, conic=conic, tiles=tiles, clamped=clamped, cam=cam)


# ------------------------------------------------------------------------------------------------ golden: SH -> RGB
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_to_rgb_matches_reference_eval_sh(deg):
    """oracle computeColorFromSH == clamp_min(eval_sh(reference) + 0.5, 0)  (gaussian_renderer/__init__.py:100-105)."""
    g = _gold("sh_eval.npz")
    dirs, sh = g["dirs"], g["sh"]                                    # sh: (P, 3, 16) reference layout
    means = dirs * 3.0                                               # campos = 0 -> view direction == dirs
    shs = np.ascontiguousarray(sh.transpose(0, 2, 1))                # op layout (P, 16, 3)
    cov = np.tile(np.array([0.05, 0, 0, 0.05, 0, 0.05], np.float32), (len(means), 1))
    r = _preprocess_only(means, shs, deg, np.zeros(3, np.float32), cov=cov)
    assert (r["radii"] > 0).all()
    want = g[f"rgb_deg{deg}"] + 0.5
    np.testing.assert_allclose(r["rgb"], np.maximum(want, 0.0), rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(r["clamped"].astype(bool)[np.abs(want) > 1e-5], (want < 0)[np.abs(want) > 1e-5])


def test_rgb2sh_constant():
    g = _gold("sh_eval.npz")
    np.testing.assert_allclose((g["rgb2sh_in"] - 0.5) / scenes.C0, g["rgb2sh_out"], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ golden: covariance
def test_cov3d_from_scale_rotation_matches_reference():
    """oracle computeCov3D (quaternion used as given) == reference build_covariance_from_scaling_rotation on the normalised
    quaternion (the reference's Python normalises, its kernel does not: Q5 of SURVEY appendix A)."""
    g = _gold("cov3d.npz")
    scales, rots = g["scales"], g["rots"]
    qn = rots / np.linalg.norm(rots, axis=1, keepdims=True)
    means = np.zeros((len(scales), 3), np.float32)
    shs = np.zeros((len(scales), 1, 3), np.float32)
    r = _preprocess_only(means, shs, 0, np.array([0, 0, -5.0], np.float32), scales=scales, rots=qn)
    np.testing.assert_allclose(r["cov3D"], g["cov_plain"], rtol=2e-5, atol=1e-9)


def test_cov3d_with_transform_inside_the_op_matches_reference():
    """n2 extension: oracle Sigma' = T Sigma T^T == the reference's build_covariance_from_scaling_rotation(..., transform)
    (scene/gaussian_model.py:37-44; fixture generated from the reference's own function)."""
    g = _gold("cov3d.npz")
    scales, rots, T = g["scales"], g["rots"], g["transforms"]
    qn = rots / np.linalg.norm(rots, axis=1, keepdims=True)
    means = np.zeros((len(scales), 3), np.float32)
    shs = np.zeros((len(scales), 1, 3), np.float32)
    r = _preprocess_only(means, shs, 0, np.array([0, 0, -5.0], np.float32), scales=scales, rots=qn, transforms=T)
    np.testing.assert_allclose(r["cov3D"], g["cov_T"], rtol=5e-5, atol=1e-9)


def test_covariance_precomp_helper_matches_reference():
    g = _gold("cov3d.npz")
    s, q, T = torch.from_numpy(g["scales"]), torch.from_numpy(g["rots"]), torch.from_numpy(g["transforms"])
    np.testing.assert_allclose(scenes.covariance_precomp(s, q).numpy(), g["cov_plain"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(scenes.covariance_precomp(s, q, 1.7).numpy(), g["cov_mod17"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(scenes.covariance_precomp(s, q, 1.0, T).numpy(), g["cov_T"], rtol=5e-5, atol=1e-9)
    np.testing.assert_allclose(scenes.quat_to_rot(q).numpy(), g["rotmat"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ golden: cameras
@pytest.mark.parametrize("i", [0, 1, 2])
def test_camera_matrices_match_reference(i):
    import math
    g = _gold("camera.npz")
    W, H, fx, fy, cx, cy, ang, tx, ty, tz = g[f"c{i}_params"]
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    cam = scenes.make_camera(int(W), int(H), fx, fy, cx, cy, R, np.array([tx, ty, tz]))
    np.testing.assert_allclose(cam.viewmatrix.numpy(), g[f"c{i}_w2v"].T, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cam.projmatrix.numpy(), g[f"c{i}_full"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cam.campos.numpy(), g[f"c{i}_campos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([cam.FoVx, cam.FoVy], g[f"c{i}_fov"], rtol=1e-7)
    np.testing.assert_allclose(scenes.projection_refine(cam.K, int(H), int(W)), g[f"c{i}_proj"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ golden: losses
@pytest.mark.parametrize("i", [0, 1])
def test_loss_functions_match_reference(i):
    from moss_amd import loss
    g = _gold("loss.npz")
    a = torch.from_numpy(g[f"l{i}_a"]).requires_grad_(True); b = torch.from_numpy(g[f"l{i}_b"])
    l1, l2, s = loss.l1_loss(a, b), loss.l2_loss(a, b), loss.ssim(a.unsqueeze(0), b.unsqueeze(0))
    assert abs(l1.item() - g[f"l{i}_l1"]) < 1e-12 and abs(l2.item() - g[f"l{i}_l2"]) < 1e-12
    assert abs(s.item() - g[f"l{i}_ssim"]) < 1e-7          # the reference builds its window in fp32
    total = l1 + 0.2 * (1.0 - s)
    total.backward()
    np.testing.assert_allclose(a.grad.numpy(), g[f"l{i}_grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_moss_loss_expression_matches_reference(i):
    """MOSS's own composition of the three rasterizer-facing terms (train_ZJU.py:108-119,131: L1 and mask L2 over the pixels of
    bound_mask, SSIM on the crop boundingRect(bound_mask)) -- ``loss.training_loss_moss`` and ``loss.bounding_rect`` -- against the numbers
    the reference's own functions gave for it (tests/golden/loss_moss.npz, float64; case 1: the mask touches the image's top edge; case 2:
    the mask is the whole frame, where the expression is ``training_loss``)."""
    from moss_amd import loss
    g = _gold("loss_moss.npz")
    img = torch.from_numpy(g[f"m{i}_image"]).double().requires_grad_(True); gt = torch.from_numpy(g[f"m{i}_gt"]).double()
    alpha = torch.from_numpy(g[f"m{i}_alpha"]).double().requires_grad_(True); bk = torch.from_numpy(g[f"m{i}_bkgd_mask"]).double()
    bound = torch.from_numpy(g[f"m{i}_bound_mask"])
    assert loss.bounding_rect(bound) == tuple(int(v) for v in g[f"m{i}_rect"])
    total = loss.training_loss_moss(img, alpha, gt, bk, bound)
    total.backward()
    assert abs(total.item() - float(g[f"m{i}_total"])) < 1e-7                  # (the reference builds its window in fp32)
    np.testing.assert_allclose(img.grad.numpy(), g[f"m{i}_grad_image"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(alpha.grad.numpy(), g[f"m{i}_grad_alpha"], rtol=1e-9, atol=1e-12)
    if i == 2:
        a2 = img.detach().clone().requires_grad_(True)
        assert abs(loss.training_loss(a2, alpha.detach(), gt, bk).item() - total.item()) < 1e-12


# ------------------------------------------------------------------------------------------------ known answers
def _single(opacity=0.8, sigma=0.02, z=0.0, color=(0.9, 0.3, 0.1), W=64, H=64, bg=(0.0, 0.0, 0.0)):
    s = scenes.config1(P=1, W=W, H=H)
    s.means3D = torch.tensor([[0.0, 0.0, z]])
    s.scales = torch.full((1, 3), sigma); s.rotations = torch.tensor([[1.0, 0, 0, 0]])
    s.opacities = torch.tensor([[opacity]])
    s.cov3D_precomp = scenes.covariance_precomp(s.scales, s.rotations)
    s.camera = scenes.make_camera(W, H, 96.0, 96.0, W / 2 + 0.5, H / 2 + 0.5, np.eye(3), np.array([0.0, 0.0, 3.0]))
    d = hp.inputs_of(s, "scale_rot", colors=True, bg=list(bg))
    d.colors_precomp = torch.tensor([list(color)])
    return d


def test_single_isotropic_gaussian_closed_form():
    """Centre projects exactly onto pixel (32,32): cov2D = (f*sigma/z)^2 + 0.3 on the diagonal, radius = ceil(3 sqrt(lambda)),
    alpha(centre) = opacity, colour(centre) = rgb*alpha + T*bg, alpha image = alpha, depth image = z*alpha."""
    d = _single(bg=(0.2, 0.4, 0.6))
    fw = hp.oracle_forward(d)
    var = (96.0 * 0.02 / 3.0) ** 2 + 0.3
    assert abs(fw.means2D[0, 0] - 32.0) < 1e-4 and abs(fw.means2D[0, 1] - 32.0) < 1e-4
    np.testing.assert_allclose(fw.conic_opacity[0], [1 / var, 0.0, 1 / var, 0.8], rtol=1e-5, atol=1e-7)
    # isotropic: mid^2 - det = 0, but the reference floors it at 0.1 (forward.cu:230-231) => lambda1 = var + sqrt(0.1)
    assert fw.radii[0] == int(np.ceil(3 * np.sqrt(var + np.sqrt(0.1))))
    r = fw.radii[0]
    x0, x1 = int((32 - r) / 16), int((32 + r + 15) / 16)
    assert fw.tiles_touched[0] == (x1 - x0) ** 2 == fw.num_rendered
    np.testing.assert_allclose(fw.alpha[0, 32, 32], 0.8, rtol=1e-5)
    np.testing.assert_allclose(fw.color[:, 32, 32], np.array([0.9, 0.3, 0.1]) * 0.8 + 0.2 * np.array([0.2, 0.4, 0.6]), rtol=1e-5)
    np.testing.assert_allclose(fw.depth[0, 32, 32], 3.0 * 0.8, rtol=1e-5)
    # two pixels to the right: alpha = o * exp(-0.5 * 4 / var)
    np.testing.assert_allclose(fw.alpha[0, 32, 34], 0.8 * np.exp(-2.0 / var), rtol=1e-4)
    # far away: background only, n_contrib 0
    np.testing.assert_allclose(fw.color[:, 2, 2], [0.2, 0.4, 0.6], rtol=1e-6)
    assert fw.n_contrib.reshape(64, 64)[2, 2] == 0 and fw.n_contrib.reshape(64, 64)[32, 32] == 1


def test_two_gaussians_blend_front_to_back_and_alpha_clamp():
    d = _single(opacity=1.0)                      # alpha clamps at 0.99
    d.means3D = torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 0.5]]); d.P = 2
    d.scales = torch.full((2, 3), 0.02); d.rotations = torch.tensor([[1.0, 0, 0, 0]] * 2)
    d.opacities = torch.tensor([[1.0], [0.5]])
    d.colors_precomp = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    fw = hp.oracle_forward(d)
    # the nearer Gaussian (z_view 3.0) comes first in every tile it shares with the farther one
    t = np.flatnonzero(fw.ranges[:, 1] - fw.ranges[:, 0] == 2)[0]
    np.testing.assert_array_equal(fw.point_list[fw.ranges[t, 0]:fw.ranges[t, 1]], [0, 1])
    a0 = 0.99
    px = fw.means2D[1]                            # second Gaussian's centre pixel (32, 32) as well (on the optical axis)
    var1 = (96.0 * 0.02 / 3.5) ** 2 + 0.3
    a1 = 0.5 * np.exp(-0.5 * ((px[0] - 32) ** 2 + (px[1] - 32) ** 2) / var1)
    np.testing.assert_allclose(fw.color[:, 32, 32], [a0, (1 - a0) * a1, 0.0], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(fw.final_T.reshape(64, 64)[32, 32], (1 - a0) * (1 - a1), rtol=1e-4)


def test_sort_is_stable_on_equal_depth():
    d = _single()
    d.means3D = torch.tensor([[0.01, 0.0, 0.0], [0.0, 0.01, 0.0], [-0.01, 0.0, 0.0]]); d.P = 3      # identical view depth
    d.scales = torch.full((3, 3), 0.02); d.rotations = torch.tensor([[1.0, 0, 0, 0]] * 3)
    d.opacities = torch.full((3, 1), 0.3); d.colors_precomp = torch.rand(3, 3)
    fw = hp.oracle_forward(d)
    assert len(set(fw.depths.view(np.uint32).tolist())) == 1
    for t in range(len(fw.ranges)):
        seg = fw.point_list[fw.ranges[t, 0]:fw.ranges[t, 1]]
        assert list(seg) == sorted(seg)
    assert (np.diff(fw.point_list_keys.astype(np.uint64)) >= 0).all()


def test_binning_invariants_cfg2():
    fw = hp.oracle_forward(hp.inputs_of(scenes.config2(), "precomp"))
    assert fw.num_rendered == int(fw.tiles_touched.sum()) == len(fw.point_list)
    assert (np.diff(fw.point_list_keys) >= 0).all()                                   # sorted by (tile, depth bits)
    lens = fw.ranges[:, 1].astype(np.int64) - fw.ranges[:, 0]
    assert lens.sum() == fw.num_rendered and (lens >= 0).all()
    tiles = (fw.point_list_keys >> np.uint64(32)).astype(np.int64)
    for t in np.flatnonzero(lens > 0)[:20]:
        assert (tiles[fw.ranges[t, 0]:fw.ranges[t, 1]] == t).all()
    assert oracle.get_higher_msb(1024) == 11 and oracle.get_higher_msb(4096) == 13


def test_dist2_known_answers():
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [10, 10, 10]], np.float32)
    d = oracle.dist2(pts)
    np.testing.assert_allclose(d[0], (1 + 4 + 9) / 3.0, rtol=1e-6)
    np.testing.assert_allclose(d[1], (1 + 5 + 10) / 3.0, rtol=1e-6)
    # fewer than 3 neighbours leaves a FLT_MAX term in the mean (simple_knn.cu:154,182): ~FLT_MAX/3
    np.testing.assert_allclose(oracle.dist2(pts[:3]), np.float32(3.4028235e38) / np.float32(3.0), rtol=1e-6)
    dup = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [2, 0, 0]], np.float32)
    np.testing.assert_allclose(oracle.dist2(dup)[0], (0 + 1 + 4) / 3.0, rtol=1e-6)


# ------------------------------------------------------------------------------------------------ backward vs autograd
@pytest.mark.parametrize("mode", ["scale_rot", "precomp", "lbs"])
def test_explicit_backward_matches_independent_autograd(mode):
    """The C oracle's explicit backward (restated backward.cu) against autograd of an independent float64 forward.
    Tolerance 2e-3 of each gradient's max: fp32 vs fp64 plus semantic (vii) above."""
    s = scenes.config1(P=96, W=64, H=64, seed=7)
    d = hp.inputs_of(s, mode)
    fw = hp.oracle_forward(d)
    dc, dd, da = hp.image_grads(d.H, d.W, seed=9)
    ref = hp.oracle_backward(d, fw, dc, dd, da)
    # pixels whose decisions sit on a threshold would make the two forwards differ discretely: require none
    assert fw.margin.min() > 1e-6
    c = d.cam
    f64 = lambda t: None if t is None else t.double()
    leaf = lambda t: None if t is None else t.double().clone().requires_grad_(True)
    means, opa, shs = leaf(d.means3D), leaf(d.opacities), leaf(d.shs)
    scales, rots, cov = leaf(d.scales), leaf(d.rotations), leaf(d.cov3D_precomp)
    tfm = None
    if mode == "lbs":
        # n2 extension: the op applies a per-Gaussian transform to the scale/rotation covariance.  The independent forward gets the
        # covariance the way MOSS's Python builds it (scene/gaussian_model.py:37-44) as a differentiable function of all three.
        tfm = leaf(d.transforms)
        r, x, y, z = rots.unbind(1)                         # quaternion used AS GIVEN, like the kernel (Q5): no normalisation here
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        Lm = tfm @ (R * scales[:, None, :])
        full = Lm @ Lm.transpose(1, 2)
        cov = torch.stack([full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2], full[:, 2, 2]], dim=1)
    ndc = torch.zeros(d.P, 2, dtype=torch.float64, requires_grad=True)
    col, dep, alp = ag.render(fw, means, opa, f64(c.viewmatrix), f64(c.projmatrix), f64(c.campos), c.tanfovx, c.tanfovy,
                              f64(d.bg), d.degree, shs=shs, scales=None if mode == "lbs" else scales,
                              rotations=None if mode == "lbs" else rots, cov3D_precomp=cov, ndc_offset=ndc)
    # forward agreement first (fp32 oracle vs fp64 autograd forward)
    assert hp.rel_err(fw.color, col.detach().numpy()) < 5e-5
    assert hp.rel_err(fw.alpha, alp.detach().numpy()) < 5e-5
    assert hp.rel_err(fw.depth, dep.detach().numpy()) < 5e-5
    loss = (col * dc.double()).sum() + (dep * dd.double()).sum() + (alp * da.double()).sum()
    loss.backward()
    checks = [("dL_dmeans3D", means.grad), ("dL_dopacity", opa.grad), ("dL_dsh", shs.grad), ("dL_dmeans2D", ndc.grad)]
    if mode == "scale_rot":
        checks += [("dL_dscales", scales.grad), ("dL_drotations", rots.grad)]
    elif mode == "lbs":
        checks += [("dL_dscales", scales.grad), ("dL_drotations", rots.grad), ("dL_dtransforms", tfm.grad)]
    else:
        checks += [("dL_dcov3D", cov.grad)]
    for name, gref in checks:
        got = getattr(ref, name)
        if name == "dL_dmeans2D":
            got = got[:, :2]
        err = hp.rel_err(got.reshape(gref.shape), gref.numpy())
        assert err < 2e-3, (name, err)


# ---------------------------------------------------------------- densification restatements: known answers
def test_kl_div_known_answers():
    rng = np.random.default_rng(7)
    P = 64
    mu0 = rng.normal(size=(P, 3)); q0 = rng.normal(size=(P, 4)); s0 = np.exp(rng.normal(size=(P, 3)) * 0.4)
    mu1 = mu0 + 0.3 * rng.normal(size=(P, 3)); q1 = rng.normal(size=(P, 4)); s1 = np.exp(rng.normal(size=(P, 3)) * 0.4)
    kl, mag = oracle.kl_div(mu0, q0, s0, mu0, q0, s0)
    assert np.abs(kl).max() < 1e-12                                         # a Gaussian against itself
    iso = np.full((P, 3), 0.7)
    kl, _ = oracle.kl_div(mu0, q0, iso, mu1, q1, iso)                        # equal isotropic covariances: |d|^2 / (2 s^2)
    np.testing.assert_allclose(kl, 0.5 * ((mu1 - mu0) ** 2).sum(1) / 0.49, rtol=1e-12)
    # the textbook formula evaluated with explicit covariance matrices, inverse and determinants
    kl, _ = oracle.kl_div(mu0, q0, s0, mu1, q1, s1)
    R0, R1 = oracle.build_rotation(q0), oracle.build_rotation(q1)
    for i in range(P):
        S0 = R0[i] @ np.diag(s0[i] ** 2) @ R0[i].T
        S1 = R1[i] @ np.diag(s1[i] ** 2) @ R1[i].T
        d = mu1[i] - mu0[i]
        want = 0.5 * (np.trace(np.linalg.inv(S1) @ S0) + d @ np.linalg.inv(S1) @ d - 3 + np.log(np.linalg.det(S1) / np.linalg.det(S0)))
        assert abs(kl[i] - want) < 1e-9 * max(1.0, abs(want))
    assert (kl > -1e-12).all()                                               # a KL divergence is non-negative


def test_densify_stats_and_exhaustive_knn_restatements():
    radii = np.array([3, 0, -1, 7, 2], dtype=np.int32)
    grad = np.array([[3, 4, 9], [1, 1, 1], [5, 12, 0], [0, 0, 5], [6, 8, 1]], dtype=np.float32)
    acc, den, mr = oracle.densify_stats(radii, grad, np.zeros(5), np.ones(5), np.array([1, 1, 1, 9, 1], dtype=np.float32))
    assert acc.tolist() == [5.0, 0.0, 0.0, 0.0, 10.0] and den.tolist() == [2.0, 1.0, 1.0, 2.0, 2.0] and mr.tolist() == [3.0, 1.0, 1.0, 9.0, 2.0]
    ref = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [1, 0, 0]], dtype=np.float32)
    d, i = oracle.knn_exhaustive(ref, np.array([[0.9, 0, 0], [0, 1.5, 0]], dtype=np.float32), 3)
    assert i.tolist() == [[1, 3, 0], [2, 0, 1]]                              # the duplicate keeps the lower index in front
    np.testing.assert_allclose(d[0], [0.1, 0.1, 0.9], rtol=1e-6)


def test_float64_adjudicator_build_agrees_with_the_float32_oracle():
    """oracle/_build/libmoss_oracle_f64.so is the same source with float -> double: on cfg1 (no pixel near a threshold) it takes the
    same decisions and its images / gradients differ from the float32 build by float32 rounding only."""
    from tests import helpers as hp
    for mode in ("scale_rot", "precomp", "lbs"):
        d = hp.inputs_of(scenes.config1(), mode)
        fw = hp.oracle_forward(d)
        fw64 = hp.oracle_forward64(d, fw)
        assert fw64.color.dtype == np.float64
        np.testing.assert_array_equal(fw64.radii64, fw.radii)
        m = hp.stable_mask(d, fw, fw64, thr=1e-4).numpy().astype(bool)
        assert m.mean() > 0.99
        np.testing.assert_array_equal(fw64.n_contrib[m.reshape(-1)], fw.n_contrib[m.reshape(-1)])
        assert np.abs(fw64.color - fw.color)[:, m].max() < 5e-6
        dc, dd, da = hp.image_grads(d.H, d.W)
        mt = torch.from_numpy(m.astype(np.float32))
        dc, dd, da = dc * mt, dd * mt, da * mt
        g32, g64 = hp.oracle_backward(d, fw, dc, dd, da), hp.oracle_backward(d, fw64, dc, dd, da)
        sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
        for name, scale in sc.items():
            a, b = getattr(g32, name), getattr(g64, name)
            if not b.size:
                continue
            live, dead = hp.scaled_err(a, b, scale)
            assert dead == 0.0, name                         # no contribution mass <=> exactly zero in both builds
            assert live < 1e-4, (name, live)                 # float32 rounding, in units of the contribution mass
            assert hp.cosine_gap(a, b) < 1e-10, name


def test_conditioning_probe_of_the_float64_backward():
    """``sum_noise_ulps`` multiplies the blend backward's per-Gaussian sums by 1 + eta (|eta| <= k * 2^-24) before the per-Gaussian stages:
    0 is the plain run bit for bit, and on a well-conditioned scene (cfg1) the final gradients move by about that much and no more --
    what profiles/r03_notes.md finding 24 used to tell an ill-conditioned PROBLEM (the float64 answer moves) from an unstable
    float32 EVALUATION (it does not: the scale gradient of a needle)."""
    from tests import helpers as hp
    d = hp.inputs_of(scenes.config1(), "scale_rot")
    fw = hp.oracle_forward(d)
    fw64 = hp.oracle_forward64(d, fw)
    dc, dd, da = hp.image_grads(d.H, d.W)
    g = hp.oracle_backward(d, fw64, dc, dd, da)
    g0 = hp.oracle_backward(d, fw64, dc, dd, da, sum_noise_ulps=0.0)
    g4 = hp.oracle_backward(d, fw64, dc, dd, da, sum_noise_ulps=4.0, noise_seed=1)
    sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    moved = 0.0
    for name, scale in sc.items():
        a, b, c = getattr(g, name), getattr(g0, name), getattr(g4, name)
        if not a.size:
            continue
        np.testing.assert_array_equal(a, b)
        live, dead = hp.scaled_err(c, a, scale)
        assert dead == 0.0 and live < 200 * 4.0 * 2.0 ** -24, (name, live)     # a few hundred times the noise at most, in mass units
        moved = max(moved, live)
    assert moved > 0.0                                                        # the probe did perturb something


def test_contribution_mass_bounds_the_gradient_and_is_zero_where_nothing_contributes():
    from tests import helpers as hp
    d = hp.inputs_of(scenes.config1(), "scale_rot")
    fw = hp.oracle_forward(d)
    dc, dd, da = hp.image_grads(d.H, d.W)
    g = hp.oracle_backward(d, fw, dc, dd, da)
    sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    for name in ("dL_dcolors", "dL_dopacity", "dL_dmeans2D", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dcov3D"):
        a = np.abs(getattr(g, name).astype(np.float64)); s = sc[name].reshape(a.shape)
        assert (a <= s * (1 + 1e-4) + 1e-30).all(), name     # |sum of terms| <= sum of |terms|
        assert (a[s == 0] == 0).all(), name
    untouched = sc["dL_dopacity"].reshape(-1) == 0            # Gaussians no stable pixel blends (cfg1 culls none)
    assert (sc["dL_dmeans3D"][untouched] == 0).all() and (sc["dL_dsh"][untouched] == 0).all()


def test_deterministic_expf_is_the_c_librarys_expf():
    """moss_expf_det (oracle/moss_oracle.c; csrc/blend.hip carries the same function for MOSS_DEBUG_EXACT_MATH) restates glibc's expf
    with one rounding per double operation.  It must BE the C library's expf for every argument the blend can ask for: counted on a grid
    of every 97th float of [-104, -0] (11.5 M arguments; the exhaustive run over all 1 120 927 745 of them finds ONE difference on this
    image's glibc 2.35, at x = -0x1.f8cbb2p+5 where the result is 2^-92), plus the known answers exp(0) = 1 and exp(-ln 2) = 1/2."""
    from oracle import oracle
    bits = np.arange(0x80000000, 0xC2D00001, 97, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    assert x.min() >= -104.0 and x.max() <= 0.0 and x.size > 11_000_000
    assert oracle.expf_det_mismatches(x) <= 2
    y = oracle.expf_det(np.array([0.0, -0.0, -0.6931471805599453, -1.0, -200.0, -np.inf], np.float32))
    assert y[0] == 1.0 and y[1] == 1.0 and y[2] == 0.5 and abs(float(y[3]) - 0.36787944117144233) < 3e-8 and y[4] == 0.0 and y[5] == 0.0
    # switching the oracle's blend to it changes no decision of cfg1 (and is reset afterwards)
    from moss_amd import scenes
    from tests import helpers as hp
    d = hp.inputs_of(scenes.config1(), "precomp")
    a, b = hp.oracle_forward(d), hp.oracle_forward(d, det_exp=True)
    assert np.array_equal(a.n_contrib, b.n_contrib) and np.array_equal(a.final_T, b.final_T)
