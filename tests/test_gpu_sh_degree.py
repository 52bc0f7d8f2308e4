"""Degree-aware SH traffic (VERDICT r5 "next round" 2): MOSS trains at SH degree 0 / 1 / 2 for iterations 1-2999 and at degree 3 for the
last one (train_ZJU.py:85-86, scene/gaussian_model.py:171-173).  Below degree 3 the forward stages only the active float4 of every SH
record, the backward reads only those, and the AdamW update -- the flat kernel and the one inside the per-Gaussian backward -- leaves the
moments of never-active coefficients alone (decay only; nothing at all when the parameters there are the zeros MOSS creates them as,
scene/gaussian_model.py:179-181).  None of it may change a bit of any result."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from moss_amd import scenes
from tests import test_gpu_headline as thl

pytestmark = pytest.mark.gpu


def _opt(gpu, P, degree, zero_rest, seed=0, capturable=True):
    """A six-tensor Gaussian model's optimizer with gradients that are exact zeros above `degree` (what the backward writes)."""
    from moss_amd import dist as mdist
    from moss_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(seed)
    k = (degree + 1) ** 2
    sh0 = torch.randn(P, 16, 3, generator=g)
    if zero_rest:
        sh0[:, k:, :] = 0
    ps = [torch.nn.Parameter(t.to(gpu)) for t in (torch.randn(P, 3, generator=g), sh0, torch.randn(P, 1, generator=g), torch.randn(P, 3, generator=g),
                                                   torch.randn(P, 4, generator=g))]
    groups = [{"params": [ps[0]], "lr": 1.6e-4}, {"params": [ps[1]], "lr": 2.5e-3, "lr_pattern": (48, 3, 2.5e-3 / 20)},
              {"params": [ps[2]], "lr": 5e-2}, {"params": [ps[3]], "lr": 5e-3}, {"params": [ps[4]], "lr": 1e-3}]
    bucket = mdist.GradBucket(ps)
    opt = FlatAdamW(groups, bucket, eps=1e-15, capturable=capturable)
    return ps, bucket, opt


@pytest.mark.parametrize("degree", [0, 1, 2])
@pytest.mark.parametrize("zero_rest", [True, False])
@pytest.mark.parametrize("capturable", [True, False])
def test_degree_aware_flat_adamw_equals_the_full_update(gpu, hip_lib, degree, zero_rest, capturable):
    P = 3001                                                 # (not a multiple of anything: the last float4 of the SH tensor, the tail thread)
    k = (degree + 1) ** 2
    runs = {}
    for aware in (False, True):
        ps, bucket, opt = _opt(gpu, P, degree, zero_rest, capturable=capturable)
        if aware:
            assert opt.set_active_sh_degree(degree) == degree and opt.sh_inactive_zero == zero_rest
        gg = torch.Generator().manual_seed(77)
        for step in range(4):
            bucket.flat[:bucket.n_params] = torch.randn(bucket.n_params, generator=gg).to(gpu)
            for n, off, nxt in zip(bucket.sizes, bucket.offsets, list(bucket.offsets[1:]) + [bucket.n_params]):
                bucket.flat[off + n:nxt] = 0
            bucket.views[1][:, k:, :] = 0                    # the backward writes exact zeros above the active degree
            if aware and step == 2:
                # ... and the degree-aware update never LOOKS at the float4 that hold no active coefficient (MOSS_SH_GRAD_ACTIVE_ONLY
                # leaves them unwritten; the float4 that is partly active is written -- zeros -- and read as a whole)
                bucket.views[1].view(P, 48)[:, 4 * ((3 * k + 3) // 4):] = 123.0
            opt.step()
        torch.cuda.synchronize(gpu)
        runs[aware] = (opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count())
    for a, b, what in zip(runs[False], runs[True], ("parameters", "exp_avg", "exp_avg_sq", "step count")):
        assert (a == b) if isinstance(a, int) else torch.equal(a, b), what
    sh_m = runs[True][1][bucket.offsets[1]:bucket.offsets[1] + P * 48].view(P, 16, 3)
    assert not bool(sh_m[:, k:, :].any()) and bool(sh_m[:, :k, :].any())


def test_the_degree_in_force_is_the_highest_ever_active(gpu, hip_lib):
    ps, bucket, opt = _opt(gpu, 500, 0, zero_rest=True)
    assert opt.set_active_sh_degree(0) == 0 and opt.sh_inactive_zero
    bucket.flat[:bucket.n_params] = 1.0
    bucket.views[1][:, 4:, :] = 0
    assert opt.set_active_sh_degree(1) == 1
    opt.step()                                               # coefficients 1..3 now have moments
    assert opt.set_active_sh_degree(0) == 1                  # ... so the degree cannot go back below 1
    assert opt.set_active_sh_degree(2) == 2 and opt.sh_inactive_zero
    with torch.no_grad():
        ps[1][7, 12, 1] = 0.5                                # a non-zero coefficient above the degree (e.g. an appended clone)
    opt._verify_sh_inactive()
    assert opt.sh_active_degree == 2 and not opt.sh_inactive_zero
    opt.exp_avg[bucket.offsets[1] + 47] = 1e-3               # a moment above the degree: somebody trained there -- everything is active
    opt._verify_sh_inactive()
    assert opt.sh_active_degree == 3 and not opt.sh_inactive_zero


def test_older_flat_entry_points_equal_the_struct_entry(gpu, hip_lib):
    """moss_adamw_flat / _devstep / _range / _guarded stay exported (ABI): the same bits as moss_adamw_flat_ex, which FlatAdamW now calls."""
    import ctypes as C
    L = hip_lib
    n = 10_007
    g = torch.Generator().manual_seed(3)
    p0, gr = torch.randn(n, generator=g).to(gpu), torch.randn(n, generator=g).to(gpu)
    ends, lrs = (C.c_longlong * 2)(4000, n), (C.c_float * 2)(1e-3, 5e-2)
    zi, zf = (C.c_int * 2)(0, 0), (C.c_float * 2)(0.0, 0.0)
    st = torch.cuda.current_stream(gpu).cuda_stream
    from moss_amd._lib import AdamWFlatArgs

    def ex(first, count, step, state, skip=None):
        p, m, v = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
        a = AdamWFlatArgs()
        a.first, a.count, a.params, a.grads = first, count, p[first:].data_ptr(), gr[first:].data_ptr()
        a.exp_avg, a.exp_avg_sq, a.num_segments = m[first:].data_ptr(), v[first:].data_ptr(), 2
        a.segment_end, a.segment_lr = C.addressof(ends), C.addressof(lrs)
        a.segment_period = a.segment_split = a.segment_lr2 = a.segment_active = None
        a.beta1, a.beta2, a.eps, a.weight_decay, a.step = 0.9, 0.999, 1e-15, 0.01, step
        a.step_state = None if state is None else state.data_ptr()
        a.skip_word, a.skip_mask = (None, 0) if skip is None else (skip.data_ptr(), 2)
        assert L.moss_adamw_flat_ex(C.addressof(a), st) == 0
        return p, m, v
    state = lambda: torch.zeros(int(L.moss_adamw_state_bytes()) // 4, dtype=torch.int32, device=gpu)
    # host step count
    p, m, v = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    assert L.moss_adamw_flat(n, p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 2, ends, lrs, zi, zi, zf, 0.9, 0.999, 1e-15, 0.01, 5, st) == 0
    for a_, b_ in zip((p, m, v), ex(0, n, 5, None)):
        assert torch.equal(a_, b_)
    # device step count
    p, m, v, s1 = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), state()
    assert L.moss_adamw_flat_devstep(n, p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 2, ends, lrs, zi, zi, zf, 0.9, 0.999, 1e-15, 0.01, s1.data_ptr(), st) == 0
    s2 = state()
    for a_, b_ in zip((p, m, v), ex(0, n, 1, s2)):
        assert torch.equal(a_, b_)
    assert int(s1[0]) == int(s2[0]) == 1
    # a range
    p, m, v = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    assert L.moss_adamw_flat_range(2000, 6000, p[2000:].data_ptr(), gr[2000:].data_ptr(), m[2000:].data_ptr(), v[2000:].data_ptr(), 2, ends, lrs, zi, zi, zf,
                                   0.9, 0.999, 1e-15, 0.01, 3, None, st) == 0
    for a_, b_ in zip((p, m, v), ex(2000, 6000, 3, None)):
        assert torch.equal(a_, b_)
    # the guard
    for word in (0, 2):
        skip = torch.tensor([word], dtype=torch.int32, device=gpu)
        p, m, v, s1 = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), state()
        assert L.moss_adamw_flat_guarded(0, n, p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 2, ends, lrs, zi, zi, zf, 0.9, 0.999, 1e-15, 0.01,
                                         s1.data_ptr(), skip.data_ptr(), 2, st) == 0
        s2 = state()
        for a_, b_ in zip((p, m, v), ex(0, n, 1, s2, skip)):
            assert torch.equal(a_, b_)
        assert int(s1[0]) == int(s2[0]) == (0 if word else 1) and torch.equal(p, p0) == bool(word)


class _Step(thl._BenchStep):
    """tests/test_gpu_headline.py's harness with the optimizer TOLD the active degree and -- `zero_rest` -- MOSS's own initial state."""

    def __init__(self, scene, gpu, case, fused, degree, aware, zero_rest):
        super().__init__(scene, gpu, case, fused=False, degree=degree)
        if zero_rest:
            with torch.no_grad():
                self.pc._features[:, (degree + 1) ** 2:, :] = 0
        if aware:
            self.opt.set_active_sh_degree(degree)
            self.ctx.sh_grad_active_only = True
        self.fused = fused
        if fused:
            pc = self.pc
            self.opt.fuse_into_backward(self.ctx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)
        # (the parent's compute() closure read `fused` at construction: rebuild it with this object's flag)
        from moss_amd.diff_gaussian_rasterization import _C
        from moss_amd.gaussian_renderer import camera_view, render
        cam, bg = camera_view(scene.camera, gpu), scene.bg.to(gpu)
        dc, dd, da = case.dc.to(gpu), case.dd.to(gpu), case.da.to(gpu)
        bucket, opt, cx, pc, pipe = self.bucket, self.opt, self.ctx, self.pc, self.pipe

        def compute():
            bucket.detach_grads()
            out = render(cam, pc, pipe, bg, transforms=self.T, translation=self.tl)
            loss = (out["render"] * dc).sum() + (out["render_depth"] * dd).sum() + (out["render_alpha"] * da).sum()
            loss.backward()
            if not fused:
                bucket.collect()
                img = cx.last_img_buffer
                opt.step(skip_word=None if img is None else _C.frame_status_word(img))
            return {"render": out["render"].detach()}
        self.compute = compute


@pytest.mark.parametrize("degree", [0, 1, 2])
@pytest.mark.parametrize("zero_rest", [True, False])
def test_degree_aware_training_steps_equal_the_full_ones_fused_and_flat(gpu, hip_lib, degree, zero_rest):
    """Five replayed training steps of the headline's form on BASELINE configs[1] at an active degree below 3: the four combinations
    (fused / flat optimizer) x (degree-aware / everything active) end with bit-identical parameters, moments and images -- with the
    coefficients above the degree random (decay-only path) and zero as MOSS creates them (untouched path)."""
    from moss_amd.graphs import GraphedStep
    scene = scenes.config2()
    raw = thl.th._raw_parameters(scene)
    dc, dd, da = thl.hp.image_grads(scene.camera.H, scene.camera.W)
    case = SimpleNamespace(raw=raw, T=thl.bench_transforms(scene.P), tl=None, dc=dc * 1e-3, dd=dd * 1e-3, da=da * 1e-3)
    res = {}
    for fused in (False, True):
        for aware in (False, True):
            b = _Step(scene, gpu, case, fused, degree, aware, zero_rest)
            snap = b.opt.snapshot()
            b.compute()                                      # capacity (a step: undone below)
            torch.cuda.synchronize(gpu)
            step = GraphedStep(b.compute, warmup=1, device=gpu, context=b.ctx)
            b.opt.restore(snap)
            for _ in range(5):
                out = step()
            torch.cuda.synchronize(gpu)
            step.check()
            assert step.dropped_frames == 0 and b.opt.step_count() == 5
            assert b.opt.sh_active_degree == (degree if aware else 3) and b.opt.sh_inactive_zero == (aware and zero_rest)
            res[(fused, aware)] = (b.opt.flat_params.clone(), b.opt.exp_avg.clone(), b.opt.exp_avg_sq.clone(), out["render"].clone())
    base = res[(False, False)]
    assert float(base[1].abs().max()) > 0
    for key, val in res.items():
        for a, c, what in zip(base, val, ("parameters", "exp_avg", "exp_avg_sq", "image")):
            assert torch.equal(a, c), (key, what)
    k = (degree + 1) ** 2
    off = b.bucket._offset[id(b.pc._features)]
    m_sh = base[1][off:off + scene.P * 48].view(scene.P, 16, 3)
    assert not bool(m_sh[:, k:, :].any())


@pytest.mark.parametrize("degree", [0, 2])
def test_sh_gradient_written_for_the_active_degree_only_on_request(gpu, hip_lib, degree):
    """MOSS_SH_GRAD_ACTIVE_ONLY: the raw backward writes the active coefficients of dL_dsh and leaves the rest of the destination alone;
    without the bit every element is written (zeros above the degree), as the reference's contract says."""
    from moss_amd.diff_gaussian_rasterization import _C, RasterContext
    scene = scenes.config2()
    raw_opa, raw_scl, raw_rot = thl.th._raw_parameters(scene)
    d = thl.hp.inputs_of(scene, "scale_rot", degree=degree)
    c = d.cam
    E = torch.Tensor([])
    dev = lambda t: t.to(gpu)
    a = dict(bg=dev(d.bg), means3D=dev(d.means3D), opa=dev(raw_opa), scl=dev(raw_scl), rot=dev(raw_rot), view=dev(c.viewmatrix), proj=dev(c.projmatrix),
             sh=dev(d.shs), campos=dev(c.campos))
    R, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        a["bg"], a["means3D"], E, a["opa"], a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy, c.H, c.W, a["sh"], degree,
        a["campos"], False, False, None, 7)
    dc, dd, da = (t.to(gpu) for t in thl.hp.image_grads(d.H, d.W))
    k = (degree + 1) ** 2
    outs = {}
    for only in (False, True):
        cx = RasterContext()
        sink = torch.full((d.P, 16, 3), 7.5, device=gpu)
        cx.set_grad_sink(sh=lambda: sink)
        cx.sh_grad_active_only = only
        g = _C.rasterize_gaussians_backward(a["bg"], a["means3D"], radii, E, a["scl"], a["rot"], 1.0, E, a["view"], a["proj"], c.tanfovx, c.tanfovy,
                                            dc, dd, da, a["sh"], degree, a["campos"], geom, R, binning, img, alpha, False, None, 7, a["opa"], cx)
        assert g[5].data_ptr() == sink.data_ptr()
        outs[only] = [t.clone() for t in g]
    full, part = outs[False][5].view(d.P, 48), outs[True][5].view(d.P, 48)
    n4 = 4 * ((3 * k + 3) // 4)                              # whole float4 are written: the partly active one with its zeros
    assert torch.equal(full[:, :n4], part[:, :n4]) and float(full[:, :3 * k].abs().max()) > 0
    assert not bool(full[:, 3 * k:].any()) and bool((part[:, n4:] == 7.5).all())
    for i in (0, 2, 3, 6, 7):                                # every other gradient is the same
        assert torch.equal(outs[False][i], outs[True][i])
