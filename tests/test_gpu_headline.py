"""GPU parity of the call path `bench.py` TIMES (VERDICT r4 "what's weak" 1 / "next round" 1): BASELINE configs[2] (and configs[4]) in
MOSS's data flow -- raw `_opacity / _scaling / _rotation`, canonical positions, a per-Gaussian LBS transform (bench's
``lbs_transforms()``, seed 1234), `mode="lbs"`, `pose_in_op` (MOSS_RAW_POSE), raw flags 7, gradient sinks into a `GradBucket`, once
as the plain backward and once with `FlatAdamW.fuse_into_backward`, the whole step replayed as a hipGraph.  Rounds 1-4 covered that
combination by composition only (lbs against the oracle without pose / raw; RAW_POSE against torch posing on 500 Gaussians; fused
against flat AdamW on cfg2); here the oracle checks the exact kernels and template instantiation
(`preprocess_backward_kernel<true, true>`) that produce `value`.

Reference lines this replaces: the posing `gaussian_renderer/__init__.py:74-77`, the covariance `scene/gaussian_model.py:37-44`
fed at `gaussian_renderer/__init__.py:88-93`, the getters `scene/gaussian_model.py:142-161`, the optimizer step
`train_ZJU.py:204-205`.

How the oracle gets in: it cannot call the device's expf / normalise, so it is run in cov3D_precomp mode on what the kernel BUILT
from the raw parameters (activated opacity and transformed covariance, read back from the geometry buffer) and on the means posed
with the kernel's expression (float32, one rounding per operation, left to right: numpy does the same) -- every integer stage then
has to match bit for bit.  The chain raw parameters -> (posed mean, covariance, opacity) is differentiated independently in float64
torch on the host and applied to the oracle's gradients; contribution masses and the noise floor of the single rule go through the
absolute Jacobian of the same chain.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from moss_amd import scenes
from tests import helpers as hp
from tests import test_gpu_parity as tp
from tests import test_gpu_parity_hardened as th

pytestmark = pytest.mark.gpu

RAW = 7          # _C.RAW_OPACITY | RAW_SCALE | RAW_ROTATION
POSE = 16        # _C.RAW_POSE


def bench_transforms(P):
    """bench.py: lbs_transforms()."""
    gT = torch.Generator().manual_seed(1234)
    return (torch.eye(3) + 0.05 * torch.randn(P, 3, 3, generator=gT)).float().contiguous()


def posed_float32(T, x, tl):
    """p = T x + t exactly as csrc/preprocess.hip: pose_point evaluates it (-ffp-contract=off: one rounding per operation, the row's
    three products added left to right, then the translation; t = 0 when there is none)."""
    Tn, xn = T.numpy().astype(np.float32), x.numpy().astype(np.float32)
    p = Tn[:, :, 0] * xn[:, 0:1]
    p = p + Tn[:, :, 1] * xn[:, 1:2]
    p = p + Tn[:, :, 2] * xn[:, 2:3]
    p = p + (tl.numpy().astype(np.float32) if tl is not None else np.float32(0.0))
    return np.ascontiguousarray(p.astype(np.float32))


def _chain64(x, T, tl, raw_scl, raw_rot, g_p, g_cov):
    """float64 chain rule of (x, T, t, raw scale, raw rotation) -> (posed mean, transformed covariance), applied to (g_p, g_cov)."""
    x64 = x.double().requires_grad_(True); T64 = T.double().requires_grad_(True)
    t64 = (torch.zeros_like(x) if tl is None else tl).double().requires_grad_(True)
    rs = raw_scl.double().requires_grad_(True); rq = raw_rot.double().requires_grad_(True)
    p = (T64 * x64[:, None, :]).sum(-1) + t64
    cov = scenes.covariance_precomp(torch.exp(rs), torch.nn.functional.normalize(rq), 1.0, T64)
    ((p * torch.from_numpy(np.asarray(g_p, np.float64))).sum() + (cov * torch.from_numpy(np.asarray(g_cov, np.float64))).sum()).backward()
    return {"dL_dmeans3D": x64.grad.numpy(), "dL_dtransforms": T64.grad.numpy(), "dL_dtranslation": t64.grad.numpy(),
            "dL_dscales": rs.grad.numpy(), "dL_drotations": rq.grad.numpy()}


def _chain_noise32(x, T, tl, raw_scl, raw_rot, g_p, g_cov, want64, probes=hp.RULE_PROBES, seed=0):
    """What FLOAT32 evaluation of the chain itself scatters by -- the reference evaluates it in float32 torch (get_covariance and the
    posing with their autograd mirrors, scene/gaussian_model.py:37-44, gaussian_renderer/__init__.py:74-77): the rotation gradient of
    an elongated Gaussian is a difference of terms many times its size there too.  Measured like helpers.reference_noise_floor: the
    chain in float32 on inputs (and incoming gradients) moved by -1 / 0 / +1 float32 ulp at random, `probes` times, against the float64
    chain; the per-element maximum joins the propagated noise floor of the single rule."""
    out = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in want64.items() if k in ("dL_dmeans3D", "dL_dtransforms", "dL_dtranslation", "dL_dscales", "dL_drotations")}
    for s_ in range(probes):
        rng = np.random.default_rng(7000 + 100 * seed + s_)
        pt = (lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)))) if s_ == 0 else (lambda a: torch.from_numpy(hp._ulp_perturbed(np.asarray(a, np.float32), rng)))
        x32 = pt(x.numpy()).requires_grad_(True); T32 = pt(T.numpy()).requires_grad_(True)
        t32 = pt((torch.zeros_like(x) if tl is None else tl).numpy()).requires_grad_(True)
        rs = pt(raw_scl.numpy()).requires_grad_(True); rq = pt(raw_rot.numpy()).requires_grad_(True)
        p = (T32 * x32[:, None, :]).sum(-1) + t32
        cov = scenes.covariance_precomp(torch.exp(rs), torch.nn.functional.normalize(rq), 1.0, T32)
        ((p * pt(g_p)).sum() + (cov * pt(g_cov)).sum()).backward()
        got = {"dL_dmeans3D": x32.grad, "dL_dtransforms": T32.grad, "dL_dtranslation": t32.grad, "dL_dscales": rs.grad, "dL_drotations": rq.grad}
        for k in out:
            out[k] = np.maximum(out[k], np.abs(got[k].double().numpy() - np.asarray(want64[k], np.float64)))
    return out


def _abs_chain(x, T, raw_scl, raw_rot, s_p, s_cov):
    """The same chain with ABSOLUTE Jacobians: per-element error scales (contribution masses, noise floors) of the oracle's
    (dL_dmeans3D, dL_dcov3D) pushed to the raw tensors.  Every row of the outputs depends on its own Gaussian only, so six probes
    (one per covariance entry) give |d cov_j / d .| for all Gaussians at once."""
    s_p = np.asarray(s_p, np.float64); s_cov = np.asarray(s_cov, np.float64)
    aT = np.abs(T.double().numpy()); ax = np.abs(x.double().numpy())
    out = {"dL_dmeans3D": np.einsum("pac,pa->pc", aT, s_p), "dL_dtranslation": s_p.copy(),
           "dL_dtransforms": s_p[:, :, None] * ax[:, None, :],
           "dL_dscales": np.zeros((x.shape[0], 3)), "dL_drotations": np.zeros((x.shape[0], 4))}
    for j in range(6):
        T64 = T.double().requires_grad_(True); rs = raw_scl.double().requires_grad_(True); rq = raw_rot.double().requires_grad_(True)
        scenes.covariance_precomp(torch.exp(rs), torch.nn.functional.normalize(rq), 1.0, T64)[:, j].sum().backward()
        out["dL_dtransforms"] += np.abs(T64.grad.numpy()) * s_cov[:, j, None, None]
        out["dL_dscales"] += np.abs(rs.grad.numpy()) * s_cov[:, j:j + 1]
        out["dL_drotations"] += np.abs(rq.grad.numpy()) * s_cov[:, j:j + 1]
    return out


def _headline_case(scene, gpu, key, with_translation=False, rule_k=hp.RULE_K_BASELINE, raw=None, T=None, degree=None):
    """Direct C-ABI calls of the headline's entry points (all outputs wanted) against the oracle; returns everything the render() /
    bucket / optimizer legs below compare themselves with.  ``raw`` / ``T``: the raw parameters and LBS transforms to use instead of
    the scene's defaults (tests/test_gpu_surgery.py: a model in the middle of a training run); ``degree``: the ACTIVE SH degree
    (MOSS trains at 0, 1, 2 for iterations 1-2999, train_ZJU.py:85-86)."""
    from moss_amd.diff_gaussian_rasterization import _C
    raw_opa, raw_scl, raw_rot = th._raw_parameters(scene) if raw is None else raw
    d = hp.inputs_of(scene, "scale_rot", degree=degree)
    c = d.cam
    P = d.P
    T = bench_transforms(P) if T is None else T
    tl = None
    if with_translation:
        tl = (0.01 * torch.randn(P, 3, generator=torch.Generator().manual_seed(77))).float().contiguous()
    E = torch.Tensor([])
    dev = lambda v: v.to(gpu)
    a = dict(bg=dev(d.bg), means3D=dev(d.means3D), opa=dev(raw_opa), scl=dev(raw_scl), rot=dev(raw_rot), view=dev(c.viewmatrix),
             proj=dev(c.projmatrix), sh=dev(d.shs), campos=dev(c.campos), T=dev(T), tl=None if tl is None else dev(tl))
    R, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        a["bg"], a["means3D"], E, a["opa"], a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy, c.H, c.W,
        a["sh"], d.degree, a["campos"], False, False, a["T"], RAW | POSE, None, a["tl"])
    t = SimpleNamespace(R=R, color=color, depth=depth, alpha=alpha, radii=radii, geom=geom, binning=binning, img=img)
    e = hp.hip_export(d, t, gpu)
    vis = e.radii > 0
    # ---- what the kernel built from the raw parameters, against float64 of the same expressions
    s64 = torch.exp(raw_scl.double()); q64 = torch.nn.functional.normalize(raw_rot.double())
    cov64 = scenes.covariance_precomp(s64, q64, 1.0, T.double()).numpy()
    assert np.abs(e.cov3D[vis] - cov64[vis]).max() <= 4e-6 * np.abs(cov64[vis]).max(), "T (R S S^T R^T) T^T inside the op"
    opa_host = (1.0 / (1.0 + np.exp(-raw_opa.numpy().astype(np.float32)))).astype(np.float32)
    opa_act = e.conic_opacity[:, 3:4].copy()
    assert np.abs(opa_act[vis] - opa_host[vis]).max() <= 2e-7, "sigmoid inside the op"
    opa_act[~vis] = opa_host[~vis]
    # ---- the oracle: cov3D_precomp mode on the kernel's covariance / opacity and on the means posed with the kernel's expression
    posed = posed_float32(T, d.means3D, tl)
    p64 = ((T.double() * d.means3D.double()[:, None, :]).sum(-1) + (0 if tl is None else tl.double())).numpy()
    assert np.abs(posed - p64).max() <= 1e-6 * max(1.0, np.abs(p64).max())
    d2 = hp.inputs_of(scene, "precomp", degree=degree)
    cov = e.cov3D.copy(); cov[~vis] = cov64[~vis].astype(np.float32)      # culled Gaussians: any finite value (never read past the cull)
    d2.cov3D_precomp = torch.from_numpy(cov)
    d2.opacities = torch.from_numpy(opa_act)
    d2.means3D = torch.from_numpy(posed)
    fw = hp.oracle_forward(d2)
    # integers: bit-exact, every stage
    assert R == fw.num_rendered
    np.testing.assert_array_equal(e.radii, fw.radii)
    np.testing.assert_array_equal(e.tiles_touched, fw.tiles_touched)
    np.testing.assert_array_equal(e.depths.view(np.uint32)[vis], fw.depths.view(np.uint32)[vis])
    np.testing.assert_array_equal(e.means2D[vis], fw.means2D[vis])
    np.testing.assert_array_equal(e.conic_opacity[vis], fw.conic_opacity[vis])
    np.testing.assert_array_equal(e.rgb[vis], fw.rgb[vis])
    np.testing.assert_array_equal(e.clamped[vis], fw.clamped[vis])
    np.testing.assert_array_equal(e.point_list_keys, fw.point_list_keys)
    np.testing.assert_array_equal(e.point_list, fw.point_list)
    np.testing.assert_array_equal(e.ranges, fw.ranges)
    # images: every stable pixel at IMG_TOL and the same stop index; every pixel within what flipped decisions can move it
    ok = tp._stable_pixels(fw)
    assert (~ok).mean() < 2e-3
    np.testing.assert_array_equal(e.n_contrib[ok], fw.n_contrib[ok])
    okc = ok.reshape(d.H, d.W)
    for name, u, v in (("color", e.color, fw.color), ("depth", e.depth, fw.depth), ("alpha", e.alpha, fw.alpha)):
        assert hp.rel_err(u[:, okc], v[:, okc]) < tp.IMG_TOL, name
    flips = tp.check_every_pixel(d2, fw, e)
    # ---- backward: incoming gradients on the stable pixels (float32 AND float64 oracle), the oracle with its OWN forward state
    fw64 = hp.oracle_forward64(d2, fw)
    m = hp.stable_mask(d2, fw, fw64, thr=th.STABLE)
    dc, dd, da = hp.image_grads(d.H, d.W)
    dc, dd, da = dc * m, dd * m, da * m
    grads = _C.rasterize_gaussians_backward(
        a["bg"], a["means3D"], radii, E, a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy, dev(dc), dev(dd),
        dev(da), a["sh"], d.degree, a["campos"], geom, R, binning, img, alpha, False, a["T"], RAW | POSE, a["opa"], None, a["tl"])
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
    if tl is not None:
        names.append("dL_dtranslation")
    assert len(grads) == len(names)
    got = {n: g_.cpu().numpy() for n, g_ in zip(names, grads)}
    got["dL_dtransforms"] = got["dL_dtransforms"].reshape(P, 3, 3)
    ref = hp.oracle_backward(d2, fw, dc, dd, da)
    mass = hp.oracle_gradient_scales(d2, fw, dc, dd, da)
    sg = torch.sigmoid(raw_opa.double()).numpy(); dsg = sg * (1 - sg)

    def targets(r):
        w = _chain64(d.means3D, T, tl, raw_scl, raw_rot, r.dL_dmeans3D, r.dL_dcov3D)
        w.update({"dL_dmeans2D": np.asarray(r.dL_dmeans2D), "dL_dsh": np.asarray(r.dL_dsh), "dL_dcov3D": np.asarray(r.dL_dcov3D),
                  "dL_dopacity": np.asarray(r.dL_dopacity, np.float64).reshape(P, 1) * dsg})
        if tl is None:
            w.pop("dL_dtranslation")
        return w

    def scales_of(s):
        o = _abs_chain(d.means3D, T, raw_scl, raw_rot, s["dL_dmeans3D"], s["dL_dcov3D"])
        o.update({"dL_dmeans2D": s["dL_dmeans2D"], "dL_dsh": s["dL_dsh"], "dL_dcov3D": s["dL_dcov3D"],
                  "dL_dopacity": np.asarray(s["dL_dopacity"], np.float64).reshape(P, 1) * np.abs(dsg)})
        if tl is None:
            o.pop("dL_dtranslation")
        return o

    want, sc = targets(ref), scales_of(mass)
    chk = [n for n in names if n != "dL_dcolors"]            # (no colours_precomp: dL_dcolors is the SH path's intermediate)
    # (dL_dcov3D: the gradient w.r.t. the TRANSFORMED covariance as stored -- the oracle's own output)
    errs = tp.check_gradients({n: got[n] for n in chk}, {n: want[n] for n in chk}, sc)
    # ---- the single float64 rule, every element of every tensor
    spread, ref64 = hp.reference_noise_floor(d2, fw, fw64, dc, dd, da)
    want64, sp = targets(ref64), scales_of(spread)
    # (+ what float32 evaluation of the chain raw parameters -> (posed mean, covariance) itself scatters by: the reference runs it in
    # float32 torch; measured by the same stochastic arithmetic)
    for n, v in _chain_noise32(d.means3D, T, tl, raw_scl, raw_rot, ref64.dL_dmeans3D, ref64.dL_dcov3D, want64).items():
        sp[n] = sp[n] + v
    adj = {}
    for n in chk:
        ratio, k = hp.single_rule_ratio(got[n], want64[n], sc[n], sp[n])
        adj[n] = ratio
        assert ratio <= rule_k, f"{n}: element {k} is {ratio:.2f} x (spread + eps mass) from float64 (rule: {rule_k})"
    # culled Gaussians: exactly zero everywhere
    inv = ~vis
    if inv.any():
        for n in chk:
            assert not np.abs(got[n][inv]).any(), n
    th._note(key, {"every_pixel": flips, "grads_vs_oracle32 (relmax, 1-cos, per-Gaussian scaled)": errs,
                   "single_rule_ratio (|hip - f64| / (spread + eps mass))": adj, "rule_k": rule_k})
    return SimpleNamespace(d=d, T=T, tl=tl, raw=(raw_opa, raw_scl, raw_rot), dc=dc, dd=dd, da=da, got=got, color=color, depth=depth,
                           alpha=alpha, R=R)


class _BenchStep:
    """One model driven exactly like bench.py's Harness drives the headline: GaussianSet with unified SH, render() with
    transforms_in_op / pose_in_op / raw_parameters_in_op, gradient sinks into a GradBucket, FlatAdamW (capturable), asynchronous
    forward on its own RasterContext -- and, `fused`, the AdamW step inside the backward kernel."""

    def __init__(self, scene, gpu, case, fused, requires_T_grad=False, degree=3):
        from moss_amd import dist as mdist
        from moss_amd.diff_gaussian_rasterization import _C, RasterContext
        from moss_amd.gaussian_model import GaussianSet
        from moss_amd.gaussian_renderer import camera_view, render
        from moss_amd.optim import FlatAdamW
        raw_opa, raw_scl, raw_rot = case.raw
        self.pc = pc = GaussianSet(scene, sh_degree=degree, device=gpu, unified_features=True)
        with torch.no_grad():
            pc._opacity.copy_(raw_opa.to(gpu)); pc._scaling.copy_(raw_scl.to(gpu)); pc._rotation.copy_(raw_rot.to(gpu))
        self.ctx = cx = RasterContext()
        cx.set_async(True)
        self.bucket = bucket = mdist.GradBucket(list(pc.parameters()))
        self.pipe = pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=True,
                                           transforms_in_op=True, pose_in_op=True, raw_parameters_in_op=True, raster_context=cx, grad_bucket=bucket)
        self.opt = opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True)
        cx.set_grad_sink(sh=lambda: bucket.sink_for(pc._features), opacity=lambda: bucket.sink_for(pc._opacity),
                         scales=lambda: bucket.sink_for(pc._scaling), rotations=lambda: bucket.sink_for(pc._rotation),
                         means3D=lambda: bucket.sink_for(pc._xyz))
        self.fused = fused
        if fused:
            opt.fuse_into_backward(cx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)
        cam, bg = camera_view(scene.camera, gpu), scene.bg.to(gpu)
        self.T = case.T.to(gpu).requires_grad_(requires_T_grad)
        self.tl = None if case.tl is None else case.tl.to(gpu).requires_grad_(requires_T_grad)
        dc, dd, da = case.dc.to(gpu), case.dd.to(gpu), case.da.to(gpu)
        self.take_step = True

        def compute():
            bucket.detach_grads()
            out = render(cam, pc, pipe, bg, transforms=self.T, translation=self.tl)
            loss = (out["render"] * dc).sum() + (out["render_depth"] * dd).sum() + (out["render_alpha"] * da).sum()
            loss.backward()
            if not fused:
                bucket.collect()
                if self.take_step:
                    # (bench.py: the step is guarded by the frame's status word whenever the forward was the asynchronous one)
                    img = cx.last_img_buffer
                    opt.step(skip_word=None if img is None else _C.frame_status_word(img))
            return {"render": out["render"].detach(), "depth": out["render_depth"].detach(), "alpha": out["render_alpha"].detach()}
        self.compute = compute


def _bucket_leg(scene, gpu, case, degree=3):
    """render() + sinks (no optimizer step): every gradient is where the bucket says it is and equals the direct call BIT FOR BIT."""
    b = _BenchStep(scene, gpu, case, fused=False, requires_T_grad=True, degree=degree)
    b.take_step = False
    b.compute()                                              # first forward: synchronous (sizes the capacity)
    b.T.grad = None                                          # (plain leaves: autograd would add the second pass to the first)
    if b.tl is not None:
        b.tl.grad = None
    out = b.compute()                                        # second: the capacity-bounded asynchronous forward the bench times
    torch.cuda.synchronize(gpu)
    b.ctx.check_status()
    assert torch.equal(out["render"], case.color) and torch.equal(out["depth"], case.depth) and torch.equal(out["alpha"], case.alpha)
    pc = b.pc
    for p_ in pc.parameters():
        off = b.bucket._offset[id(p_)]
        assert p_.grad.data_ptr() == b.bucket.flat[off:off + 1].data_ptr()
    pairs = {"dL_dmeans3D": pc._xyz.grad, "dL_dsh": pc._features.grad, "dL_dopacity": pc._opacity.grad, "dL_dscales": pc._scaling.grad,
             "dL_drotations": pc._rotation.grad, "dL_dtransforms": b.T.grad}
    if b.tl is not None:
        pairs["dL_dtranslation"] = b.tl.grad
    for n, g_ in pairs.items():
        assert np.array_equal(g_.cpu().numpy().reshape(case.got[n].shape), case.got[n]), f"{n}: render() + sinks differs from the direct call"
    return b.bucket.flat[:b.bucket.n_params].clone()


def _optimizer_leg(scene, gpu, case, bucket_grads, degree=3):
    """The step under hipGraph replay, flat and fused: after ONE replayed step from the same state, parameters, both moments and the
    step count agree bit for bit; the flat form's bucket holds exactly the oracle-checked gradients; and the parameters it leaves are
    AdamW's (float64 torch restatement of torch.optim.AdamW's rule, scene/gaussian_model.py:226: lr per group, eps 1e-15, wd 0.01)."""
    from moss_amd.graphs import GraphedStep
    res = {}
    for fused in (False, True):
        b = _BenchStep(scene, gpu, case, fused=fused, degree=degree)
        snap = b.opt.snapshot()                              # the state the direct calls of _headline_case saw: untouched parameters
        p0 = b.opt.flat_params.clone()
        b.compute()                                          # eager, synchronous forward: capacity (takes a step: undone below)
        torch.cuda.synchronize(gpu)
        step = GraphedStep(b.compute, warmup=2, device=gpu, context=b.ctx)
        torch.cuda.synchronize(gpu)
        b.opt.restore(snap)                                  # the eager run, the warm-up and the capture took steps: back to the start
        assert torch.equal(b.opt.flat_params, p0) and b.opt.step_count() == 0
        t0 = b.opt.step_count()
        out = step()                                         # ONE replay
        torch.cuda.synchronize(gpu)
        step.check()
        assert step.dropped_frames == 0
        assert b.opt.step_count() == t0 + 1
        assert torch.equal(out["render"], case.color)
        res[fused] = SimpleNamespace(b=b, p0=p0, t=t0 + 1, m0=snap[1], v0=snap[2])
    flat, fus = res[False], res[True]
    assert torch.equal(flat.p0, fus.p0)
    for name in ("flat_params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(flat.b.opt, name), getattr(fus.b.opt, name)), f"{name}: the fused step differs from backward -> bucket -> flat AdamW"
    # the flat form stepped on exactly the gradients the bucket leg compared with the oracle-checked direct call
    assert torch.equal(flat.b.bucket.flat[:flat.b.bucket.n_params], bucket_grads)
    # ... and the step is AdamW's: float64 restatement from the same start, gradients and step number
    opt = flat.b.opt
    g = bucket_grads.double().cpu(); p = flat.p0[:opt.n].double().cpu(); m = flat.m0.double().cpu(); v = flat.v0.double().cpu()
    lr = torch.zeros_like(p)
    start = 0
    for i in range(opt.nseg):
        end = int(opt.seg_end[i])
        seg = torch.full((end - start,), float(opt.seg_lr[i]), dtype=torch.float64)
        if int(opt.seg_period[i]):
            k = torch.arange(end - start) % int(opt.seg_period[i])
            seg = torch.where(k < int(opt.seg_split[i]), seg, torch.full_like(seg, float(opt.seg_lr2[i])))
        lr[start:end] = seg
        start = end
    b1, b2 = opt.betas
    t = flat.t
    p_ref = p * (1 - lr * opt.weight_decay)
    m_ref = b1 * m + (1 - b1) * g; v_ref = b2 * v + (1 - b2) * g * g
    p_ref = p_ref - lr / (1 - b1 ** t) * m_ref / ((v_ref / (1 - b2 ** t)).sqrt() + opt.eps)
    got_p = opt.flat_params[:opt.n].double().cpu()
    moved = (got_p - p).abs().max().item()
    assert moved > 0
    # (the update of an element is lr x O(1): compare the MOVE, relative to the largest one)
    assert ((got_p - p) - (p_ref - p)).abs().max().item() <= 2e-5 * (p_ref - p).abs().max().item()
    assert (opt.exp_avg.double().cpu() - m_ref).abs().max().item() <= 1e-6 * m_ref.abs().max().item()
    assert (opt.exp_avg_sq.double().cpu() - v_ref).abs().max().item() <= 1e-6 * v_ref.abs().max().item()


def test_cfg3_headline_path_against_the_oracle(gpu, hip_lib):
    """BASELINE configs[2] exactly as bench.py's headline drives the op, all three legs (see the module docstring)."""
    scene = scenes.config3()
    case = _headline_case(scene, gpu, "cfg3_headline")
    grads = _bucket_leg(scene, gpu, case)
    _optimizer_leg(scene, gpu, case, grads)


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_cfg3_headline_path_at_lower_sh_degrees(gpu, hip_lib, degree):
    """MOSS trains at SH degree 0, 1, 2 for iterations 1-2999 and at degree 3 for iteration 3000 only (train_ZJU.py:85-86,
    scene/gaussian_model.py:171-173): the headline's call path -- RAW | POSE, sinks, the AdamW step inside the backward kernel, graph
    replay -- under the oracle at every active degree (VERDICT r5 "what's weak" 1b: rounds 1-5 checked lower degrees on cfg1 only).
    All three legs: integers bit-exact, images, gradients under the bars and the single float64 rule, dL_dsh exactly zero above the
    active degree, render() + sinks == the direct call, fused == flat under replay and both == AdamW."""
    scene = scenes.config3()
    case = _headline_case(scene, gpu, f"cfg3_headline_degree{degree}", degree=degree)
    k = (degree + 1) ** 2
    assert not np.abs(case.got["dL_dsh"][:, k:, :]).any() and np.abs(case.got["dL_dsh"][:, :k, :]).max() > 0
    grads = _bucket_leg(scene, gpu, case, degree=degree)
    _optimizer_leg(scene, gpu, case, grads, degree=degree)


def test_cfg3_headline_path_with_a_translation(gpu, hip_lib):
    """The same with the (P,3) translation of gaussian_renderer/__init__.py:77: dL_dtranslation = the oracle's dL/d(posed mean),
    under the same bars and the single rule; render() + sinks bit-identical to the direct call."""
    scene = scenes.config3()
    case = _headline_case(scene, gpu, "cfg3_headline_translation", with_translation=True)
    _bucket_leg(scene, gpu, case)


def test_cfg5_headline_path_against_the_oracle(gpu, hip_lib):
    """BASELINE configs[4] (300k Gaussians, 1024x1024) through the same path: the per-Gaussian backward with the AdamW step inside runs
    its 4 688 blocks in more than one round there."""
    scene = scenes.config5()
    case = _headline_case(scene, gpu, "cfg5_headline")
    grads = _bucket_leg(scene, gpu, case)
    _optimizer_leg(scene, gpu, case, grads)


@pytest.mark.parametrize("activations", ["torch_getters", "fused_kernel"])
def test_pose_in_op_without_raw_parameters(gpu, hip_lib, activations):
    """`pipe.pose_in_op` ORs RAW_POSE into raw_flags, and render() used to pick the RAW tensors whenever ANY bit was
    set -- pose_in_op + transforms_in_op WITHOUT raw_parameters_in_op handed the op logits / log-scales / unnormalised quaternions as
    activated values (silently wrong image and gradients), and with `fused_activations` raised AttributeError.  Now the three
    activation bits alone decide: image and gradients equal the torch-posed path's to float32 rounding."""
    from moss_amd import dist as mdist
    from moss_amd.diff_gaussian_rasterization import RasterContext
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import camera_view, render
    scene = scenes.config2()
    P = scene.P
    T = bench_transforms(P).to(gpu)
    tl = (0.01 * torch.randn(P, 3, generator=torch.Generator().manual_seed(3))).to(gpu)
    cam, bg = camera_view(scene.camera, gpu), scene.bg.to(gpu)
    w = torch.rand(3, scene.camera.H, scene.camera.W, generator=torch.Generator().manual_seed(5)).to(gpu)
    res = {}
    for pose_in_op in (False, True):
        pc = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
        bucket = mdist.GradBucket(list(pc.parameters()))
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, raster_context=RasterContext(),
                               fused_activations=(activations == "fused_kernel"), transforms_in_op=True, pose_in_op=pose_in_op,
                               raw_parameters_in_op=False, grad_bucket=bucket)
        if pipe.fused_activations:
            bucket.detach_grads()
        out = render(cam, pc, pipe, bg, transforms=T, translation=tl)
        ((out["render"] * w).sum() + out["render_alpha"].sum()).backward()
        if pipe.fused_activations:
            bucket.collect()
        res[pose_in_op] = (out["render"].detach(), [p_.grad.detach().clone() for p_ in (pc._xyz, pc._features, pc._opacity, pc._scaling, pc._rotation)])
    (img0, g0), (img1, g1) = res[False], res[True]
    assert float(img0.abs().max()) > 0.1
    # posed by torch ops vs inside the op: p differs by a float32 rounding, so a threshold-fragile pixel may flip -- compared in norm
    # (the bug this test is for gave an image that has nothing to do with the other one)
    diff = (img0 - img1).abs()
    assert float(diff.mean()) <= 1e-5 * float(img0.abs().max()) and float((diff > 2e-4).float().mean()) <= 1e-4
    for name, u, v in zip(("xyz", "features", "opacity", "scaling", "rotation"), g0, g1):
        # (cfg2 is MOSS's initialisation -- isotropic Gaussians -- whose rotation gradient is exactly zero, in both paths)
        assert float(u.norm()) > 0 or name == "rotation"
        assert float((u - v).norm()) <= 2e-3 * float(u.norm()), name
