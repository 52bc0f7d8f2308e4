"""GPU tests of the remaining entry points and of the edge cases the domain offers."""
import numpy as np
import pytest
import torch

from moss_amd import scenes
from oracle import oracle
from tests import helpers as hp

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- fused loss
@pytest.mark.parametrize("shape", [(3, 128, 128), (3, 70, 100), (3, 512, 512)])
def test_fused_loss_matches_torch_reference(gpu, hip_lib, shape):
    """Tolerance: loss value 1e-6 absolute (fp32 sums of <= 786k terms), gradients 2e-5 of their max."""
    from moss_amd.loss import training_loss, training_loss_fused
    C, H, W = shape
    g = torch.Generator().manual_seed(3)
    img = torch.rand(C, H, W, generator=g); gt = scenes.synthetic_target(H, W)
    alpha = torch.rand(1, H, W, generator=g); mask = (torch.rand(1, H, W, generator=g) > 0.5).float()
    a = img.double().requires_grad_(True); b = alpha.double().requires_grad_(True)
    ref = training_loss(a, b, gt.double(), mask.double())           # float64 torch reference on the CPU
    ref.backward()
    x = img.to(gpu).requires_grad_(True); al = alpha.to(gpu).requires_grad_(True)
    out = training_loss_fused(x, al, gt.to(gpu), mask.to(gpu))
    (out * 1.0).backward()
    assert abs(float(out) - float(ref)) < 1e-6
    assert hp.rel_err(x.grad.cpu().numpy(), a.grad.numpy()) < 2e-5
    assert hp.rel_err(al.grad.cpu().numpy(), b.grad.numpy()) < 2e-5


# ---------------------------------------------------------------- distCUDA2
@pytest.mark.parametrize("P", [1, 3, 4, 5, 1000, 6890, 20000])
def test_dist2_bit_exact_vs_bruteforce_oracle(gpu, hip_lib, P):
    from moss_amd.simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(P)
    pts = scenes.body_points(P, g) if P >= 1000 else torch.randn(P, 3, generator=g)
    if P >= 1000:
        pts[7] = pts[3]                       # exact duplicates -> distance 0 counted
    got = distCUDA2(pts.to(gpu)).cpu().numpy()
    ref = oracle.dist2(pts.numpy())
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_dist2_empty(gpu, hip_lib):
    from moss_amd.simple_knn._C import distCUDA2
    assert distCUDA2(torch.zeros(0, 3, device=gpu)).shape == (0,)


# ---------------------------------------------------------------- markVisible
def test_mark_visible(gpu, hip_lib):
    from moss_amd.diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    s = scenes.config1()
    c = s.camera
    pts = s.means3D.clone(); pts[::3, 2] -= 3.0              # push a third behind the near plane
    rs = GaussianRasterizationSettings(c.H, c.W, c.tanfovx, c.tanfovy, s.bg.to(gpu), 1.0, c.viewmatrix.to(gpu),
                                       c.projmatrix.to(gpu), 3, c.campos.to(gpu), False, False)
    vis = GaussianRasterizer(rs).markVisible(pts.to(gpu)).cpu().numpy()
    ref = oracle.mark_visible(pts.numpy(), c.viewmatrix.numpy(), c.projmatrix.numpy())
    np.testing.assert_array_equal(vis, ref)
    assert 0 < vis.sum() < len(vis)


# ---------------------------------------------------------------- edge cases
def test_zero_gaussians(gpu, hip_lib):
    """P == 0 short-circuits both directions with zero-filled outputs (rasterize_points.cu:83,168)."""
    from moss_amd.diff_gaussian_rasterization import _C
    c = scenes.config1().camera
    z = lambda *s: torch.zeros(*s, device=gpu)
    R, color, depth, alpha, radii, gb, bb, ib = _C.rasterize_gaussians(
        torch.tensor([0.2, 0.3, 0.4], device=gpu), z(0, 3), torch.Tensor([]), z(0, 1), z(0, 3), z(0, 4), 1.0, torch.Tensor([]),
        c.viewmatrix.to(gpu), c.projmatrix.to(gpu), c.tanfovx, c.tanfovy, c.H, c.W, z(0, 16, 3), 3, c.campos.to(gpu), False, False)
    assert R == 0 and float(color.abs().sum()) == 0 and float(alpha.abs().sum()) == 0 and radii.numel() == 0
    grads = _C.rasterize_gaussians_backward(
        torch.zeros(3, device=gpu), z(0, 3), radii, torch.Tensor([]), z(0, 3), z(0, 4), 1.0, torch.Tensor([]), c.viewmatrix.to(gpu),
        c.projmatrix.to(gpu), c.tanfovx, c.tanfovy, z(3, c.H, c.W), z(1, c.H, c.W), z(1, c.H, c.W), z(0, 16, 3), 3,
        c.campos.to(gpu), gb, R, bb, ib, alpha, False)
    assert all(g.shape[0] == 0 for g in grads)


def test_everything_culled_gives_background(gpu, hip_lib):
    s = scenes.config1()
    s.means3D[:, 2] -= 10.0                                   # all behind the camera
    d = hp.inputs_of(s, "scale_rot", bg=[0.1, 0.5, 0.9])
    fw = hp.oracle_forward(d)
    t = hp.hip_forward(d, gpu)
    assert t.R == 0 == fw.num_rendered
    np.testing.assert_array_equal(t.color.cpu().numpy(), fw.color)
    np.testing.assert_array_equal(t.radii.cpu().numpy(), 0)
    dc, dd, da = hp.image_grads(d.H, d.W)
    g = hp.hip_backward(d, t, dc, dd, da, gpu)
    for name in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
        assert float(getattr(g, name).abs().sum()) == 0.0


def _stacked_scene(P, W=64, H=64, spread=0.02, scale=0.15, seed=0, equal_depth_every=0):
    """P big Gaussians piled onto a few tiles: long per-tile lists (sort classes, multi-batch blend)."""
    g = torch.Generator().manual_seed(seed)
    s = scenes.config1(P=P, W=W, H=H)
    s.means3D = torch.randn(P, 3, generator=g) * spread
    if equal_depth_every:
        s.means3D[::equal_depth_every, 2] = 0.0               # identical depths: order must fall back to the index
    s.scales = torch.full((P, 3), scale) * torch.exp(0.2 * torch.randn(P, 3, generator=g))
    s.opacities = torch.sigmoid(torch.randn(P, 1, generator=g) - 3.0)   # faint, so pixels do not saturate early
    s.cov3D_precomp = scenes.covariance_precomp(s.scales, s.rotations)
    s.camera = scenes.make_camera(W, H, 70.0, 70.0, W / 2, H / 2, np.eye(3), np.array([0.0, 0.0, 3.0]))
    return s


@pytest.mark.parametrize("P,ties", [(700, 0), (3000, 0), (3000, 7), (9500, 0)])
def test_long_tile_lists_and_depth_ties(gpu, hip_lib, P, ties):
    """3000 > 2048 and 9500 > 8192 entries per tile exercise the larger sort paths (LDS and global-memory network);
    equal depths must keep ascending Gaussian index (stability of the reference's radix sort)."""
    from tests.test_gpu_parity import _check_forward, _check_backward
    d = hp.inputs_of(_stacked_scene(P, equal_depth_every=ties), "precomp")
    # thousands of faint entries per pixel put many alphas within rounding of 1/255: allow more excluded pixels here
    fw, t, e = _check_forward(d, gpu, max_fragile=3e-2)
    assert (fw.ranges[:, 1] - fw.ranges[:, 0]).max() >= P * 0.9
    _check_backward(d, gpu, fw, t, e)


def test_prefiltered_trap_is_reported(gpu, hip_lib):
    s = scenes.config1()
    s.means3D[0, 2] = -10.0
    d = hp.inputs_of(s, "scale_rot")
    with pytest.raises(RuntimeError, match="prefiltered"):
        hp.hip_forward(d, gpu, prefiltered=True)


def test_debug_mode_runs_stage_checks(gpu, hip_lib):
    d = hp.inputs_of(scenes.config1(), "scale_rot")
    t = hp.hip_forward(d, gpu, debug=True)
    assert t.R > 0


# ---------------------------------------------------------------- Python surface, end to end
def test_render_binding_and_autograd(gpu, hip_lib):
    """render() (gaussian_renderer counterpart) -> autograd.Function -> .backward(): gradients land on the raw parameters and
    on the zero means2D sink (viewspace_points.grad, consumed by MOSS's densification, scene/gaussian_model.py:816-818)."""
    from types import SimpleNamespace
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render, camera_view
    s = scenes.config1()
    pc = GaussianSet(s, device=gpu)
    cam = camera_view(s.camera, gpu)
    outs = {}
    for cov_py in (True, False):
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=cov_py, debug=False)
        pc.zero_grad()
        out = render(cam, pc, pipe, torch.zeros(3, device=gpu))
        assert set(["render", "render_depth", "render_alpha", "viewspace_points", "visibility_filter", "radii"]) <= set(out)
        loss = out["render"].sum() + out["render_alpha"].sum()
        loss.backward()
        assert out["viewspace_points"].grad is not None and out["viewspace_points"].grad.abs().sum() > 0
        assert out["visibility_filter"].dtype == torch.bool
        outs[cov_py] = (out["render"].detach().cpu(), pc._xyz.grad.clone().cpu(), pc._scaling.grad.clone().cpu())
    # the two covariance input modes (Python-precomputed vs in-kernel) must agree (gaussian_renderer/__init__.py:88-93)
    assert hp.rel_err(outs[True][0].numpy(), outs[False][0].numpy()) < 1e-4
    assert hp.rel_err(outs[True][1].numpy(), outs[False][1].numpy()) < 1e-3
    assert hp.rel_err(outs[True][2].numpy(), outs[False][2].numpy()) < 1e-3


def test_convert_shs_python_matches_native(gpu, hip_lib):
    """pipe.convert_SHs_python (SH->RGB in torch, gaussian_renderer/__init__.py:100-105) vs the in-kernel SH path."""
    from types import SimpleNamespace
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render, camera_view
    s = scenes.config1()
    pc = GaussianSet(s, device=gpu)
    cam = camera_view(s.camera, gpu)
    imgs = []
    for sh_py in (True, False):
        pipe = SimpleNamespace(convert_SHs_python=sh_py, compute_cov3D_python=False, debug=False)
        with torch.no_grad():
            imgs.append(render(cam, pc, pipe, torch.zeros(3, device=gpu))["render"].cpu().numpy())
    assert hp.rel_err(imgs[0], imgs[1]) < 1e-5


def test_argument_validation(gpu, hip_lib):
    from moss_amd.diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    s = scenes.config1(); c = s.camera
    rs = GaussianRasterizationSettings(c.H, c.W, c.tanfovx, c.tanfovy, s.bg.to(gpu), 1.0, c.viewmatrix.to(gpu),
                                       c.projmatrix.to(gpu), 3, c.campos.to(gpu), False, False)
    r = GaussianRasterizer(rs)
    m, m2, o = s.means3D.to(gpu), torch.zeros_like(s.means3D).to(gpu), s.opacities.to(gpu)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m2, o, shs=None, colors_precomp=None, scales=s.scales.to(gpu), rotations=s.rotations.to(gpu))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m2, o, shs=s.shs.to(gpu), scales=s.scales.to(gpu), rotations=s.rotations.to(gpu), cov3D_precomp=s.cov3D_precomp.to(gpu))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        r(m[:, :2], m2, o, shs=s.shs.to(gpu), scales=s.scales.to(gpu), rotations=s.rotations.to(gpu))


# ---------------------------------------------------------------- flat fused AdamW
def test_flat_adamw_matches_torch_adamw(gpu, hip_lib):
    """Three steps of the flat HIP AdamW vs torch.optim.AdamW on the same gradients: parameters agree to 1e-6 relative."""
    from moss_amd.dist import GradBucket
    from moss_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes, lrs = [(1001, 3), (1001, 15, 3), (1001, 1), (1001, 4)], [0.00016, 0.000125, 0.05, 0.001]
    init = [torch.randn(*s) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(gpu)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(gpu)) for t in init]
    ref = torch.optim.AdamW([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    bucket = GradBucket(pa)
    opt = FlatAdamW([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], bucket, eps=1e-15)
    for it in range(3):
        bucket.attach()
        grads = [torch.randn(*s, device=gpu) * (it + 1) for s in shapes]
        for p, q, g in zip(pa, pb, grads):
            p.grad.copy_(g); q.grad = g.clone()
        opt.step(); ref.step()
    for p, q in zip(pa, pb):
        assert hp.rel_err(p.detach().cpu().numpy(), q.detach().cpu().numpy()) < 1e-6
