"""CPU tests of the row surgery of the headline's optimizer objects (VERDICT r5 "next round" 1):
``FlatAdamW.prune_rows / append_rows / reset_rows`` + ``GradBucket.relayout`` + ``GaussianSet.prune_points / densification_postfix /
reset_opacity`` against a restatement of what MOSS does to ``torch.optim`` state -- ``replace_tensor_to_optimizer``
(scene/gaussian_model.py:362-375), ``_prune_optimizer`` (:377-394), ``cat_tensors_to_optimizer`` (:413-434), ``densification_postfix``
(:436-454), ``reset_opacity`` (:314-317) -- and, under a 2-process gloo group, that data-parallel replicas stay bit-identical across an
event.  The update kernels need a GPU; here the AdamW rule is applied by a torch restatement directly to the optimizer's flat buffers
(what is under test is the bookkeeping, not the arithmetic: tests/test_gpu_surgery.py runs the real step)."""
import os

import numpy as np
import torch

from moss_amd import scenes

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


class MossStyleState:
    """The reference's optimizer surgery, restated on a plain dict of tensors {name: (param, exp_avg, exp_avg_sq)} (torch.optim keeps
    exactly these per parameter; `step` is untouched by all three functions)."""

    def __init__(self, tensors):
        self.t = {k: [v.clone(), torch.zeros_like(v), torch.zeros_like(v)] for k, v in tensors.items()}

    def replace_tensor_to_optimizer(self, tensor, name):                    # scene/gaussian_model.py:362-375
        self.t[name] = [tensor.clone(), torch.zeros_like(tensor), torch.zeros_like(tensor)]

    def prune(self, keep):                                                   # _prune_optimizer, :377-394
        for k in NAMES:
            self.t[k] = [x[keep] for x in self.t[k]]

    def cat(self, d):                                                        # cat_tensors_to_optimizer, :413-434
        for k in NAMES:
            p, m, v = self.t[k]
            e = d[k]
            self.t[k] = [torch.cat((p, e), 0), torch.cat((m, torch.zeros_like(e)), 0), torch.cat((v, torch.zeros_like(e)), 0)]


def _model(P=53, seed=3):
    from moss_amd import dist as mdist
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.optim import FlatAdamW
    sc = scenes.config1(P=P, seed=seed)
    pc = GaussianSet(sc, sh_degree=3, device="cpu", unified_features=True)
    bucket = mdist.GradBucket(list(pc.parameters()))
    opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=False)
    return pc, bucket, opt


def _moss_view(pc, opt):
    """{name: (param, m, v)} of the flat optimizer in MOSS's six-tensor terms."""
    out = {}
    idx = {id(p): i for i, p in enumerate(opt.bucket.params)}
    for name, p in (("xyz", pc._xyz), ("opacity", pc._opacity), ("scaling", pc._scaling), ("rotation", pc._rotation)):
        m, v = opt._moments_of(idx[id(p)])
        out[name] = (p.data, m, v)
    m, v = opt._moments_of(idx[id(pc._features)])
    out["f_dc"] = (pc._features.data[:, :1], m[:, :1], v[:, :1])
    out["f_rest"] = (pc._features.data[:, 1:], m[:, 1:], v[:, 1:])
    return out


def _fill_moments(opt, seed):
    g = torch.Generator().manual_seed(seed)
    opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g))
    opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g))
    # (the <= 3 alignment floats behind a tensor are zeros and stay zeros)
    b = opt.bucket
    for n, off, nxt in zip(b.sizes, b.offsets, list(b.offsets[1:]) + [b.n_params]):
        opt.exp_avg[off + n:nxt] = 0; opt.exp_avg_sq[off + n:nxt] = 0


def _new_rows(n, seed):
    g = torch.Generator().manual_seed(seed)
    return {"xyz": torch.randn(n, 3, generator=g), "f_dc": torch.randn(n, 1, 3, generator=g), "f_rest": torch.randn(n, 15, 3, generator=g),
            "opacity": torch.randn(n, 1, generator=g), "scaling": torch.randn(n, 3, generator=g), "rotation": torch.randn(n, 4, generator=g)}


def _postfix_args(d):
    return (d["xyz"], d["f_dc"], d["f_rest"], d["opacity"], d["scaling"], d["rotation"])


def _assert_same(pc, opt, ref):
    got = _moss_view(pc, opt)
    for k in NAMES:
        for a, b, what in zip(got[k], ref.t[k], ("param", "exp_avg", "exp_avg_sq")):
            assert a.shape == b.shape and torch.equal(a, b), (k, what)
    b = opt.bucket
    assert all(off % 4 == 0 for off in b.offsets) and b.tail % 4 == 0 and b.flat.numel() == b.tail + 4
    assert not bool(b.flat.any())                                            # a re-laid-out bucket is all-zero, like a fresh one
    for p in b.params:
        off = b._offset[id(p)]
        assert p.grad is None and p.data_ptr() == opt.flat_params[off:off + 1].data_ptr() and p.is_contiguous()
    assert opt.n == b.n_params == opt.flat_params.numel() == opt.exp_avg.numel() == opt.exp_avg_sq.numel()
    assert [int(e) for e in opt.seg_end] == list(b.offsets[1:]) + [b.n_params]
    # the alignment gaps of the moments hold zeros (the flat kernel updates them with zero gradients: they must stay zero)
    for n, off, nxt in zip(b.sizes, b.offsets, list(b.offsets[1:]) + [b.n_params]):
        assert not bool(opt.exp_avg[off + n:nxt].any()) and not bool(opt.exp_avg_sq[off + n:nxt].any()) and not bool(opt.flat_params[off + n:nxt].any())


def test_flat_adamw_row_surgery_equals_moss_optimizer_surgery():
    from moss_amd.densify import DensifyStats
    pc, bucket, opt = _model()
    _fill_moments(opt, 1)
    opt.t = 17
    objs = [id(p) for p in bucket.params]
    ref = MossStyleState({k: v[0] for k, v in _moss_view(pc, opt).items()})
    for k, v in _moss_view(pc, opt).items():
        ref.t[k][1], ref.t[k][2] = v[1].clone(), v[2].clone()
    stats = DensifyStats.__new__(DensifyStats)                               # (its kernels need a GPU; the bookkeeping does not)
    P = pc._xyz.shape[0]
    stats.xyz_gradient_accum, stats.denom, stats.max_radii2D = torch.rand(P, 1), torch.rand(P, 1), torch.rand(P)
    # --- clone: append 7 rows (densification_postfix)
    d = _new_rows(7, 10)
    pc.densification_postfix(*_postfix_args(d), opt, stats=stats)
    ref.cat(d)
    _assert_same(pc, opt, ref)
    assert stats.denom.shape == (P + 7, 1) and not bool(stats.denom.any()) and not bool(stats.max_radii2D.any())     # :451-454
    # --- split: append 2 x 5 rows, then prune their 5 sources (+ padding zeros for the new rows, like :526-527)
    d2 = _new_rows(10, 11)
    pc.densification_postfix(*_postfix_args(d2), opt, stats=stats)
    ref.cat(d2)
    g = torch.Generator().manual_seed(4)
    mask = torch.zeros(P + 17, dtype=torch.bool)
    mask[torch.randperm(P, generator=g)[:5]] = True
    stats.denom.copy_(torch.rand(P + 17, 1)); kept = stats.denom[~mask].clone()
    pc.prune_points(mask, opt, stats=stats)
    ref.prune(~mask)
    _assert_same(pc, opt, ref)
    assert torch.equal(stats.denom, kept)                                    # :408-410
    # --- reset_opacity: values replaced, both moments of the opacity zeroed, nothing else touched, no re-layout
    flat_before = opt.flat_params.data_ptr()
    pc.reset_opacity(opt)
    new = torch.log(torch.min(torch.sigmoid(ref.t["opacity"][0]), torch.ones_like(ref.t["opacity"][0]) * 0.01)
                    / (1 - torch.min(torch.sigmoid(ref.t["opacity"][0]), torch.ones_like(ref.t["opacity"][0]) * 0.01)))
    ref.replace_tensor_to_optimizer(new, "opacity")
    got = _moss_view(pc, opt)
    for k in NAMES:
        for a, b in zip(got[k], ref.t[k]):
            assert torch.equal(a, b), k
    assert opt.flat_params.data_ptr() == flat_before
    # the Parameter OBJECTS are the ones the model, the sinks and the fused step hold; the shared step count is untouched
    assert [id(p) for p in bucket.params] == objs and opt.t == 17 and pc._features_dc.shape == (P + 12, 1, 3)


def test_surgery_keeps_a_learning_rate_schedule_and_refuses_what_it_cannot_do():
    import pytest
    from moss_amd import dist as mdist
    from moss_amd.optim import FlatAdamW
    pc, bucket, opt = _model()
    opt.set_learning_rates({pc._xyz: 3e-5, pc._features: (1e-3, 2e-4)})
    pc.densification_postfix(*_postfix_args(_new_rows(3, 1)), opt)
    i = {id(p): k for k, p in enumerate(bucket.params)}
    assert abs(opt.seg_lr[i[id(pc._xyz)]] - 3e-5) < 1e-12 and abs(opt.seg_lr2[i[id(pc._features)]] - 2e-4) < 1e-10
    with pytest.raises(ValueError):
        opt.append_rows({pc._xyz: torch.zeros(2, 4)})
    with pytest.raises(ValueError):
        opt.reset_rows(pc._opacity, torch.zeros(3, 1))
    with pytest.raises(TypeError):
        pc.prune_points(torch.zeros(pc._xyz.shape[0], dtype=torch.bool), torch.optim.AdamW(pc.param_groups(), lr=0.0))
    # a sharded optimizer holds a row's moments on several ranks
    params = [torch.nn.Parameter(torch.zeros(8, 3)), torch.nn.Parameter(torch.zeros(8, 1))]
    b2 = mdist.GradBucket(params, world=2)
    o2 = FlatAdamW([{"params": [params[0]], "lr": 1e-3}, {"params": [params[1]], "lr": 1e-3}], b2, shard=(0, 2))
    with pytest.raises(RuntimeError):
        o2.prune_rows(torch.ones(8, dtype=torch.bool))


def _adamw_flat_(opt, g, t):
    """torch restatement of the flat update on the optimizer's own buffers (per-element learning rates from its segment table)."""
    n = opt.n
    lr = torch.zeros(n)
    start = 0
    for i in range(opt.nseg):
        end = int(opt.seg_end[i])
        seg = torch.full((end - start,), float(opt.seg_lr[i]))
        if int(opt.seg_period[i]):
            k = torch.arange(end - start) % int(opt.seg_period[i])
            seg = torch.where(k < int(opt.seg_split[i]), seg, torch.full_like(seg, float(opt.seg_lr2[i])))
        lr[start:end] = seg
        start = end
    b1, b2 = opt.betas
    p, m, v = opt.flat_params[:n], opt.exp_avg[:n], opt.exp_avg_sq[:n]
    p.mul_(1 - lr * opt.weight_decay)
    m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(lr / (1 - b1 ** t) * m / ((v / (1 - b2 ** t)).sqrt() + opt.eps))


def _scripted_event(pc, opt, stats, step, seed=0):
    """A deterministic clone / split / prune / opacity-reset from the CURRENT parameters and a seeded generator: every replica that
    holds the same parameters takes the same decision (MOSS: the densify RNG is seeded identically on all ranks)."""
    from moss_amd.surgery import densification_event
    g = torch.Generator().manual_seed(1000 * seed + step)
    P = pc._xyz.shape[0]
    src = torch.randperm(P, generator=g)[:max(P // 10, 1)]
    clone = {"new_xyz": pc._xyz.data[src].clone(), "new_features_dc": pc._features_dc.data[src].clone(),
             "new_features_rest": pc._features_rest.data[src].clone(), "new_opacities": pc._opacity.data[src].clone(),
             "new_scaling": pc._scaling.data[src].clone(), "new_rotation": pc._rotation.data[src].clone(), "source": src}
    src2 = torch.randperm(P, generator=g)[:max(P // 20, 1)]
    noise = torch.randn(2 * src2.numel(), 3, generator=g) * 0.01
    rep = lambda t: t.data[src2].repeat(2, *([1] * (t.dim() - 1))).clone()
    split = {"new_xyz": rep(pc._xyz) + noise, "new_features_dc": rep(pc._features_dc), "new_features_rest": rep(pc._features_rest),
             "new_opacities": rep(pc._opacity), "new_scaling": rep(pc._scaling) - float(np.log(1.6)), "new_rotation": rep(pc._rotation),
             "source": src2.repeat(2)}
    P2 = P + src.numel() + 2 * src2.numel()
    prune = torch.zeros(P2, dtype=torch.bool)
    prune[src2] = True                                                       # the split sources go (scene/gaussian_model.py:526-527)
    prune[torch.randperm(P2, generator=g)[:max(P2 // 50, 1)]] = True         # + a few "transparent" ones
    return densification_event(pc, opt, append=[clone, split], prune=prune, reset_opacity=(step % 2 == 0), stats=stats)


def _surgery_rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from moss_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    pc, bucket, opt = _model(P=41, seed=9)                                   # the same model on every rank
    sizes = []
    gr = torch.Generator().manual_seed(500 + rank)                           # every rank its own views
    t = 0
    for step in range(1, 10):
        n = bucket.n_params
        bucket.flat[:n] = torch.randn(n, generator=gr)
        # (the backward leaves exact zeros in the alignment gaps and above the active SH degree; degree 1 here)
        for nn_, off, nxt in zip(bucket.sizes, bucket.offsets, list(bucket.offsets[1:]) + [n]):
            bucket.flat[off + nn_:nxt] = 0
        sh = bucket.flat[bucket._offset[id(pc._features)]:][:pc._features.numel()].view_as(pc._features)
        sh[:, 4:, :] = 0
        bucket.loss_terms[:] = float(rank + 1)
        bucket.all_reduce_mean(None, world, sh_param=pc._features, active_sh_degree=1)
        t += 1
        _adamw_flat_(opt, bucket.flat[:n], t)
        if step % 3 == 0:
            rep = _scripted_event(pc, opt, None, step)
            sizes.append((rep["rows_before"], rep["rows_after"]))
            bucket = opt.bucket
    n = bucket.n_params
    q.put((rank, opt.flat_params[:n].numpy().copy(), opt.exp_avg[:n].numpy().copy(), opt.exp_avg_sq[:n].numpy().copy(), sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_stay_identical_across_densification_events_gloo_world2():
    """SURVEY 8e + VERDICT r5 1(c): two data-parallel replicas (one model, every rank its own gradients, all-reduce of the bucket with
    only the ACTIVE SH coefficients travelling, the same AdamW update) carry out the same scripted clone / split / prune / opacity-reset
    events: parameters and both moments stay bit-identical on both ranks, through three events that change the row count."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_surgery_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, m0, v0, s0), (_, p1, m1, v1, s1) = res
    assert s0 == s1 and len(s0) == 3 and all(a != b for a, b in s0)
    assert np.array_equal(p0, p1) and np.array_equal(m0, m1) and np.array_equal(v0, v1)
    assert np.abs(m0).max() > 0 and np.isfinite(p0).all()
