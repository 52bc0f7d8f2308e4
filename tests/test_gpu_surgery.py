"""Row surgery on the objects that produce the headline number, on the GPU (VERDICT r5 "next round" 1a).

MOSS prunes and appends Gaussians every 100 iterations between iterations 400 and 2000 and resets the opacities
(train_ZJU.py:171-186; scene/gaussian_model.py:314-317, :362-454).  Here the SAME scripted schedule -- clone, split + prune of the
split sources, prune, opacity reset, every 100 steps of 300 on BASELINE configs[1] -- is driven twice:

  A  the headline's form: ``GaussianSet`` with one SH tensor, ``GradBucket`` + gradient sinks, ``FlatAdamW`` with the step INSIDE the
     per-Gaussian backward kernel, the whole step replayed as a hipGraph; events through ``moss_amd.surgery.densification_event``
     (``FlatAdamW.append_rows / prune_rows / reset_rows``, bucket re-layout, capacity re-learn, graph re-capture);
  B  MOSS's own form: six separate parameter tensors, ``moss_amd.optim.AdamW`` (torch's state keys: the class patches/gaussian_model.diff
     puts in the place of torch.optim.AdamW) stepped eagerly, and the reference's ``cat_tensors_to_optimizer`` / ``_prune_optimizer`` /
     ``replace_tensor_to_optimizer`` restated below on ``optimizer.state`` -- new ``nn.Parameter`` objects and all.

Both must end BIT-IDENTICAL: every parameter and both moments of every parameter.  On the step after each event the gradients of form
A's model are checked against the CPU oracle (the headline's entry points, tests/test_gpu_headline.py: _headline_case).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from moss_amd import scenes
from tests import test_gpu_headline as thl

pytestmark = pytest.mark.gpu

GROUPS = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


def _target(scene_maker, gpu):
    """bench.py's ground truth: a render of a DIFFERENT random body through the same camera, mask = alpha > 0.5."""
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import camera_view, render
    sc = scene_maker(seed=scenes.SEED + 7)
    cam, bg = camera_view(sc.camera, gpu), torch.zeros(3, device=gpu)
    with torch.no_grad():
        o = render(cam, GaussianSet(sc, sh_degree=3, device=gpu), SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False), bg)
    return o["render"].detach().clamp(0, 1).contiguous(), (o["render_alpha"].detach() > 0.5).float().contiguous()


class FormA:
    """The headline's step (bench.py Harness with mode lbs, fused optimizer, one hipGraph)."""

    def __init__(self, scene, gpu, gt, mask, T, degree=3, graph=True):
        from moss_amd import dist as mdist
        from moss_amd import loss as mloss
        from moss_amd.densify import DensifyStats
        from moss_amd.diff_gaussian_rasterization import RasterContext
        from moss_amd.gaussian_model import GaussianSet
        from moss_amd.gaussian_renderer import camera_view, render
        from moss_amd.graphs import GraphedStep
        from moss_amd.optim import FlatAdamW
        self.gpu = gpu
        self.pc = pc = GaussianSet(scene, sh_degree=degree, device=gpu, unified_features=True)
        self.ctx = cx = RasterContext()
        cx.set_async(True)
        self.bucket = bucket = mdist.GradBucket(list(pc.parameters()))
        self.pipe = pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=True,
                                           transforms_in_op=True, pose_in_op=True, raw_parameters_in_op=True, raster_context=cx, grad_bucket=bucket)
        self.opt = opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True)
        cx.set_grad_sink(sh=lambda: bucket.sink_for(pc._features), opacity=lambda: bucket.sink_for(pc._opacity),
                         scales=lambda: bucket.sink_for(pc._scaling), rotations=lambda: bucket.sink_for(pc._rotation),
                         means3D=lambda: bucket.sink_for(pc._xyz))
        opt.fuse_into_backward(cx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)
        self.stats = DensifyStats(scene.P, device=gpu)
        self.cam, self.bg = camera_view(scene.camera, gpu), torch.zeros(3, device=gpu)
        self.T = T.to(gpu).contiguous()

        def compute():
            bucket.detach_grads()
            out = render(self.cam, pc, pipe, self.bg, transforms=self.T)
            loss = mloss.training_loss_fused(out["render"], out["render_alpha"], gt, mask, terms_out=bucket.loss_terms)
            mloss.backward_from_loss(loss)
            return {"radii": out["radii"]}
        self.compute = compute

        def probe():                                          # forward only, no side effect: sizes the capacity for the new set
            with torch.no_grad():
                render(self.cam, pc, pipe, self.bg, transforms=self.T)
        self.probe = probe
        self.steps = 0
        self.graphed = None
        if graph:
            compute(); self.steps += 1                       # the first (synchronous) forward sizes the capacity: a training step
            torch.cuda.synchronize(gpu)
            self.graphed = GraphedStep(compute, warmup=2, device=gpu, context=cx)
            self.steps += 2                                  # (the warm-up runs are training steps too; the capture itself executes nothing)

    def step(self):
        (self.graphed or self.compute)()
        self.steps += 1

    def tensors(self):
        pc = self.pc
        return {"xyz": pc._xyz.data, "f_dc": pc._features.data[:, :1], "f_rest": pc._features.data[:, 1:], "opacity": pc._opacity.data,
                "scaling": pc._scaling.data, "rotation": pc._rotation.data}

    def moments(self):
        opt = self.opt
        idx = {id(p): i for i, p in enumerate(opt.bucket.params)}
        out = {}
        for name, p in (("xyz", self.pc._xyz), ("opacity", self.pc._opacity), ("scaling", self.pc._scaling), ("rotation", self.pc._rotation)):
            out[name] = opt._moments_of(idx[id(p)])
        m, v = opt._moments_of(idx[id(self.pc._features)])
        out["f_dc"] = (m[:, :1], v[:, :1]); out["f_rest"] = (m[:, 1:], v[:, 1:])
        return out

    def event(self, ev):
        from moss_amd.surgery import densification_event
        rep = densification_event(self.pc, self.opt, append=ev["append"], prune=ev["prune"], reset_opacity=ev["reset_opacity"],
                                  stats=self.stats, context=self.ctx, graphed=self.graphed, probe=self.probe,
                                  per_gaussian={"T": self.T}, after_surgery=lambda pg: setattr(self, "T", pg["T"]))
        return rep


class FormB:
    """MOSS's own form: separate tensors, a torch-state optimizer, the reference's surgery functions."""

    def __init__(self, scene, gpu, gt, mask, T, degree=3):
        from moss_amd import loss as mloss
        from moss_amd.diff_gaussian_rasterization import RasterContext
        from moss_amd.gaussian_model import GaussianSet
        from moss_amd.gaussian_renderer import camera_view, render
        from moss_amd.optim import AdamW
        self.pc = pc = GaussianSet(scene, sh_degree=degree, device=gpu, unified_features=False)
        self.optimizer = AdamW(pc.param_groups(), lr=0.0, eps=1e-15)        # scene/gaussian_model.py:226 with the class swapped
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False,
                               transforms_in_op=True, pose_in_op=True, raw_parameters_in_op=True, raster_context=RasterContext())
        self.cam, self.bg = camera_view(scene.camera, gpu), torch.zeros(3, device=gpu)
        self.T = T.to(gpu).contiguous()

        def compute():
            self.optimizer.zero_grad(set_to_none=True)
            out = render(self.cam, pc, pipe, self.bg, transforms=self.T)
            loss = mloss.training_loss_fused(out["render"], out["render_alpha"], gt, mask)
            mloss.backward_from_loss(loss)
            self.optimizer.step()
        self.step = compute

    ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}

    def tensors(self):
        return {k: getattr(self.pc, a).data for k, a in self.ATTR.items()}

    def moments(self):
        return {k: (self.optimizer.state[getattr(self.pc, a)]["exp_avg"], self.optimizer.state[getattr(self.pc, a)]["exp_avg_sq"]) for k, a in self.ATTR.items()}

    # ---- scene/gaussian_model.py:362-375 / :377-394 / :413-434, restated on this optimizer (same statements, same order) -------------
    def replace_tensor_to_optimizer(self, tensor, name):
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                stored_state = self.optimizer.state.get(group["params"][0], None)
                stored_state["exp_avg"] = torch.zeros_like(tensor)
                stored_state["exp_avg_sq"] = torch.zeros_like(tensor)
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
                self.optimizer.state[group["params"][0]] = stored_state
                setattr(self.pc, self.ATTR[name], group["params"][0])

    def _prune_optimizer(self, mask):
        for group in self.optimizer.param_groups:
            stored_state = self.optimizer.state.get(group["params"][0], None)
            stored_state["exp_avg"] = stored_state["exp_avg"][mask]
            stored_state["exp_avg_sq"] = stored_state["exp_avg_sq"][mask]
            del self.optimizer.state[group["params"][0]]
            group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
            self.optimizer.state[group["params"][0]] = stored_state
            setattr(self.pc, self.ATTR[group["name"]], group["params"][0])

    def cat_tensors_to_optimizer(self, tensors_dict):
        for group in self.optimizer.param_groups:
            extension_tensor = tensors_dict[group["name"]]
            stored_state = self.optimizer.state.get(group["params"][0], None)
            stored_state["exp_avg"] = torch.cat((stored_state["exp_avg"], torch.zeros_like(extension_tensor)), dim=0)
            stored_state["exp_avg_sq"] = torch.cat((stored_state["exp_avg_sq"], torch.zeros_like(extension_tensor)), dim=0)
            del self.optimizer.state[group["params"][0]]
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], extension_tensor), dim=0).requires_grad_(True))
            self.optimizer.state[group["params"][0]] = stored_state
            setattr(self.pc, self.ATTR[group["name"]], group["params"][0])

    def event(self, ev):
        with torch.no_grad():
            for a in ev["append"]:
                self.cat_tensors_to_optimizer({"xyz": a["new_xyz"], "f_dc": a["new_features_dc"], "f_rest": a["new_features_rest"],
                                               "opacity": a["new_opacities"], "scaling": a["new_scaling"], "rotation": a["new_rotation"]})
                self.T = torch.cat((self.T, self.T[a["source"]]), dim=0).contiguous()
            if ev["prune"] is not None:
                keep = ~ev["prune"]
                self._prune_optimizer(keep)
                self.T = self.T[keep].contiguous()
            if ev["reset_opacity"]:
                opa = torch.sigmoid(self.pc._opacity)
                new = torch.min(opa, torch.ones_like(opa) * 0.01)
                self.replace_tensor_to_optimizer(torch.log(new / (1 - new)), "opacity")      # reset_opacity, :314-317 (inverse_sigmoid)


def scripted_event(tensors, step, gpu, reset_opacity):
    """moss_amd.scenes.scripted_densification: a deterministic clone + split + prune from the CURRENT parameters (form A's; form B's
    are the same bits) and a seeded generator."""
    return scenes.scripted_densification(tensors, step, gpu, reset_opacity=reset_opacity)


def _scene_of(form, scene, degree):
    """A scene namespace of form A's CURRENT model, for the oracle check (tests/test_gpu_headline.py: _headline_case)."""
    pc = form.pc
    with torch.no_grad():
        s = SimpleNamespace(name="mid_training", P=int(pc._xyz.shape[0]), sh_degree=degree, means3D=pc._xyz.detach().cpu().clone(),
                            scales=pc.get_scaling.detach().cpu().clone(), rotations=pc._rotation.detach().cpu().clone(),
                            opacities=pc.get_opacity.detach().cpu().clone(), shs=pc._features.detach().cpu().clone(), bg=torch.zeros(3),
                            camera=scene.camera)
        s.cov3D_precomp = torch.zeros(s.P, 6)                 # (replaced by what the kernel built, _headline_case)
        raw = (pc._opacity.detach().cpu().clone(), pc._scaling.detach().cpu().clone(), pc._rotation.detach().cpu().clone())
    return s, raw


@pytest.mark.parametrize("degree", [3, 1])
def test_densification_schedule_flat_fused_graph_equals_moss_style_surgery(gpu, hip_lib, degree):
    """300 steps on BASELINE configs[1] with an event every 100 (clone + split + prune at 100 and 200 -- with an opacity reset at 200 --
    and at 300), forms A and B side by side: bit-identical parameters and moments at the end; oracle-checked gradients after each event."""
    scene = scenes.config2()
    gt, mask = _target(scenes.config2, gpu)
    T = thl.bench_transforms(scene.P)
    A = FormA(scene, gpu, gt, mask, T, degree=degree)
    B = FormB(scene, gpu, gt, mask, T, degree=degree)
    for _ in range(A.steps):                                  # (A's capacity run and graph warm-up were training steps)
        B.step()
    done = A.steps
    sizes = []
    for target in (100, 200, 300):
        while done < target:
            A.step(); B.step(); done += 1
        torch.cuda.synchronize(gpu)
        ev = scripted_event(A.tensors(), target, gpu, reset_opacity=(target == 200))
        rep = A.event(ev)
        B.event(ev)
        sizes.append((rep["rows_before"], rep["rows_after"], rep["event_ms"]))
        assert rep["recaptured"] and A.graphed.captured_capacity == A.ctx.capacity > 0
        assert torch.equal(A.T, B.T) and A.T.shape[0] == rep["rows_after"] == A.stats.denom.shape[0]
        ta, tb = A.tensors(), B.tensors()
        for k in GROUPS:
            assert torch.equal(ta[k], tb[k]), (target, k)
        # ---- the step after the event: the new set's gradients against the oracle (direct C-ABI calls on the current parameters)
        if target < 300:
            s_now, raw = _scene_of(A, scene, degree)
            thl._headline_case(s_now, gpu, f"cfg2_after_event_{target}_deg{degree}", raw=raw, T=A.T.detach().cpu(), degree=degree,
                               rule_k=thl.hp.RULE_K)
    for _ in range(20):                                       # ... and the run goes on after the last event
        A.step(); B.step()
    torch.cuda.synchronize(gpu)
    A.graphed.check()
    assert A.graphed.dropped_frames == 0
    assert all(a != b for a, b, _ in sizes) and A.graphed.recaptures >= 3
    ta, tb, ma, mb = A.tensors(), B.tensors(), A.moments(), B.moments()
    for k in GROUPS:
        assert ta[k].shape == tb[k].shape and torch.equal(ta[k], tb[k]), f"{k}: parameters differ"
        assert torch.equal(ma[k][0], mb[k][0]) and torch.equal(ma[k][1], mb[k][1]), f"{k}: moments differ"
    assert A.opt.step_count() == done + 20 == int(B.optimizer.state[B.pc._xyz]["step"])
    assert float(ma["f_dc"][0].abs().max()) > 0 and bool(torch.isfinite(A.opt.flat_params).all())


def test_check_before_the_first_replay_reads_no_stale_status(gpu, hip_lib):
    """A capture executes no kernel: the captured forward's image buffer -- where the status words live -- is pool memory nobody has
    written.  ``GraphedStep.check()`` / ``RasterContext.check_status()`` straight after a capture or a re-capture must not take whatever
    the block held before for a frame's report (seen: the bits of 1.0f as the "needed capacity", 10^9 instances, and an 800 GB
    allocation at the next capture).  The pool is poisoned with 1.0f first so that stale words would be exactly that."""
    scene = scenes.config2()
    gt, mask = _target(scenes.config2, gpu)
    T = torch.eye(3).repeat(scene.P, 1, 1)
    junk = [torch.full((1 << 20,), 1.0, device=gpu) for _ in range(8)]      # (what freed blocks of the general pool hold)
    del junk
    a = FormA(scene, gpu, gt, mask, T, degree=3, graph=True)
    a.graphed.reserve_pool(256 << 20)
    cap = a.ctx.capacity
    assert a.graphed.check() is False and a.ctx.capacity == cap         # straight after the first capture
    for _ in range(3):
        a.step()
    assert a.graphed.check() is False and a.ctx.capacity == cap
    ev = scripted_event(a.tensors(), 1, gpu, False)
    a.event(ev)
    cap2 = a.ctx.capacity
    a.ctx.check_status()                                                  # straight after a RE-capture, before any replay
    assert a.graphed.check() is False and a.ctx.capacity == cap2 and a.graphed.dropped_frames == 0
    for _ in range(3):
        a.step()
    a.ctx.check_status()
    assert a.ctx.capacity == cap2 and 0 < a.ctx.last_needed < cap2


def test_opacity_reset_needs_no_recapture(gpu, hip_lib):
    """reset_opacity (scene/gaussian_model.py:314-317) changes no shape and no address: applied in place between two replays of the
    SAME captured graph, equal to the eager unfused step driven the same way."""
    scene = scenes.config2()
    gt, mask = _target(scenes.config2, gpu)
    T = thl.bench_transforms(scene.P)
    A = FormA(scene, gpu, gt, mask, T)
    B = FormB(scene, gpu, gt, mask, T)
    for _ in range(A.steps):
        B.step()
    for _ in range(5):
        A.step(); B.step()
    graph_before = A.graphed.graph
    rep = A.event({"append": [], "prune": None, "reset_opacity": True})
    B.event({"append": [], "prune": None, "reset_opacity": True})
    assert not rep["recaptured"] and A.graphed.graph is graph_before
    for _ in range(5):
        A.step(); B.step()
    torch.cuda.synchronize(gpu)
    ta, tb, ma, mb = A.tensors(), B.tensors(), A.moments(), B.moments()
    for k in GROUPS:
        assert torch.equal(ta[k], tb[k]) and torch.equal(ma[k][0], mb[k][0]) and torch.equal(ma[k][1], mb[k][1]), k
    assert float(torch.sigmoid(ta["opacity"]).max()) < 0.05   # (0.01 five steps ago)
