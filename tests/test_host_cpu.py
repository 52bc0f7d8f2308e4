"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol (no compute without a GPU),
the op fails loudly instead of falling back, the Python surface validates arguments like the reference, scene generators
are deterministic, and the frame-parallel gradient exchange is correct under a 2-process gloo group."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from moss_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    text = open(os.path.join(ROOT, "include", "moss_raster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(moss_[a-z0-9_]+)\s*\(", text)) - {"moss_alloc_fn"}
    assert len(names) >= 15
    for n in sorted(names):
        assert hasattr(hip_lib, n), f"{n} is declared in include/moss_raster.h but not exported"
    assert hip_lib.moss_abi_version() == 1
    assert hip_lib.moss_last_error() == b""


def test_scratch_size_functions_are_host_only_and_monotonic(hip_lib):
    g = [hip_lib.moss_raster_geometry_bytes(p) for p in (1, 1000, 100000)]
    assert g[0] < g[1] < g[2] and g[2] < 100000 * 120           # ~ 100 B per Gaussian
    b = [hip_lib.moss_raster_binning_bytes(r) for r in (0, 1000, 1000000)]
    assert b[0] <= b[1] < b[2]
    assert hip_lib.moss_raster_image_bytes(512, 512) >= 512 * 512 * 8
    assert hip_lib.moss_knn_workspace_bytes(6890) >= 6890 * 24
    assert hip_lib.moss_loss_workspace_bytes(3, 512, 512) >= 3 * 3 * 512 * 512 * 4


def test_missing_library_is_an_import_error_not_a_fallback(monkeypatch):
    from moss_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmoss_raster.so")
    with pytest.raises(ImportError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_no_product_module_imports_the_oracle():
    """The oracle is test infrastructure: nothing under moss_amd/ (nor the root alias packages) may import it."""
    for base in ("moss_amd", "diff_gaussian_rasterization", "simple_knn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "moss_amd", "csrc")):
        for f in files:
            assert "#include \"../../oracle" not in open(os.path.join(dirpath, f)).read()


def test_python_surface_validation_without_gpu(hip_lib):
    import diff_gaussian_rasterization as dgr                      # the root alias MOSS imports
    from moss_amd.diff_gaussian_rasterization import GaussianRasterizer as GR2
    assert dgr.GaussianRasterizer is GR2
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    s = scenes.config1(); c = s.camera
    rs = dgr.GaussianRasterizationSettings(c.H, c.W, c.tanfovx, c.tanfovy, s.bg, 1.0, c.viewmatrix, c.projmatrix, 3, c.campos,
                                           False, False)
    r = dgr.GaussianRasterizer(rs)
    m2 = torch.zeros_like(s.means3D)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(s.means3D, m2, s.opacities, shs=s.shs, colors_precomp=torch.rand(256, 3), scales=s.scales, rotations=s.rotations)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(s.means3D, m2, s.opacities, shs=s.shs)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(s.means3D, m2, s.opacities, shs=s.shs, scales=s.scales)          # rotations missing
    # CPU tensors are refused loudly (the reference would dereference host pointers on the device)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(s.means3D, m2, s.opacities, shs=s.shs, scales=s.scales, rotations=s.rotations)
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(s.means3D)


def test_cpu_deep_copy_tuple():
    from moss_amd.diff_gaussian_rasterization import cpu_deep_copy_tuple
    t = torch.arange(4.0)
    out = cpu_deep_copy_tuple((t, 3, "x"))
    assert out[1:] == (3, "x") and torch.equal(out[0], t) and out[0].data_ptr() != t.data_ptr()


def test_scene_generators_are_deterministic():
    a, b = scenes.config2(), scenes.config2()
    assert torch.equal(a.means3D, b.means3D) and torch.equal(a.scales, b.scales) and torch.equal(a.shs, b.shs)
    assert a.means3D.shape == (6890, 3) and a.camera.W == 512
    ring = scenes.look_at_ring(8)
    R0, t0 = ring[0]
    np.testing.assert_allclose(R0, np.eye(3), atol=1e-12); np.testing.assert_allclose(t0, [0, 0, 3.0], atol=1e-12)
    for R, t in ring:
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(R) - 1) < 1e-12
        np.testing.assert_allclose(np.linalg.norm(-R.T @ t), 3.0, atol=1e-12)       # camera centre on the ring


def test_spatial_order_is_a_permutation_along_a_curve():
    """densify.spatial_order (pure torch, any device): a permutation of the indices; consecutive Gaussians of the re-indexed set are
    close in space (a Morton curve), which is all the rasterizer's memory traffic asks of it; deterministic; degenerate inputs
    (all points equal, a single point) do not divide by zero."""
    from moss_amd.densify import spatial_order
    s = scenes.config2()
    x = s.means3D
    perm = spatial_order(x)
    assert perm.dtype == torch.int64 and sorted(perm.tolist()) == list(range(x.shape[0]))
    assert torch.equal(perm, spatial_order(x))
    step = lambda y: float((y[1:] - y[:-1]).norm(dim=1).mean())
    assert step(x[perm]) < 0.25 * step(x)                      # the generator's own order is uncorrelated with position
    assert sorted(spatial_order(torch.zeros(5, 3)).tolist()) == [0, 1, 2, 3, 4]
    assert spatial_order(torch.tensor([[1.0, 2.0, 3.0]])).tolist() == [0]


def test_header_declares_the_spatial_order_hint():
    """MOSS_HINT_SPATIAL_ORDER of include/moss_raster.h and its Python mirror are the same bit, distinct from the raw-parameter bits."""
    from moss_amd.diff_gaussian_rasterization import _C
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "moss_raster.h")).read()
    m = re.search(r"#define\s+MOSS_HINT_SPATIAL_ORDER\s+(\d+)", text)
    assert m and int(m.group(1)) == _C.HINT_SPATIAL_ORDER == 8
    assert _C.HINT_SPATIAL_ORDER & (_C.RAW_OPACITY | _C.RAW_SCALE | _C.RAW_ROTATION) == 0


def test_shard_views():
    from moss_amd.dist import shard_views
    parts = [shard_views(10, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10)) and parts[1] == [1, 5, 9]


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from moss_amd import dist as mdist
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    bucket = mdist.GradBucket(params)
    bucket.attach()
    # each rank produces a different "view" gradient through autograd, accumulating straight into the bucket
    loss = ((params[0] + 1.0) * (rank + 1)).sum() + (params[1] * float(10 * (rank + 1))).sum()
    loss.backward()
    assert params[0].grad.data_ptr() == bucket.views[0].data_ptr()          # no pack copy
    got_loss = bucket.all_reduce_mean(loss, world).clone()
    # numpy, not tensors: a tensor crosses a multiprocessing queue as a file descriptor served by THIS process, which may have
    # exited by the time the parent unpickles it
    q.put((rank, params[0].grad.numpy().copy(), params[1].grad.numpy().copy(), got_loss.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_parallel_gradient_bucket_gloo_world2():
    """N>1 path on CPU: two processes, gloo backend, ONE all-reduce of the flat bucket averages every gradient and the loss."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g0, g1, loss in res:
        g0, g1, loss = torch.from_numpy(g0), torch.from_numpy(g1), torch.from_numpy(loss)
        assert torch.allclose(g0, torch.full((5, 3), 1.5))                  # mean of 1 and 2
        assert torch.allclose(g1, torch.full((7,), 15.0))                   # mean of 10 and 20
        assert torch.allclose(loss, torch.tensor([(15.0 + 30.0) / 2]))      # mean of the two ranks' losses


def _stats_rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from moss_amd import dist as mdist
    from moss_amd.densify import DensifyStats
    mdist.init_from_env(backend="gloo")
    stats = DensifyStats(6, device="cpu")
    # what two steps of moss_densify_stats would have left on this rank (the kernel itself needs a GPU)
    stats.xyz_gradient_accum[:, 0] = torch.tensor([1.0, 0.0, 2.0, 0.0, 0.5, 0.0]) * (rank + 1)
    stats.denom[:, 0] = torch.tensor([2.0, 0.0, 1.0, 0.0, 1.0, 0.0]) + rank
    stats.max_radii2D[:] = torch.tensor([3.0, 0.0, 9.0, 0.0, 1.0, 0.0]) if rank == 0 else torch.tensor([4.0, 7.0, 2.0, 0.0, 1.0, 0.0])
    stats.sync()
    q.put((rank, stats.xyz_gradient_accum.numpy().copy(), stats.denom.numpy().copy(), stats.max_radii2D.numpy().copy(), stats.mean_grads().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_densification_statistics_sync_gloo_world2():
    """Frame-parallel replicas must take the same densification decision: sum / sum / max of the per-rank statistics."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_stats_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, acc, den, mr, mean in res:
        acc, den, mr, mean = (torch.from_numpy(x) for x in (acc, den, mr, mean))
        assert torch.equal(acc[:, 0], torch.tensor([3.0, 0.0, 6.0, 0.0, 1.5, 0.0]))
        assert torch.equal(den[:, 0], torch.tensor([5.0, 1.0, 3.0, 1.0, 3.0, 1.0]))
        assert torch.equal(mr, torch.tensor([4.0, 7.0, 9.0, 0.0, 1.0, 0.0]))
        assert torch.equal(mean[:, 0], torch.tensor([0.6, 0.0, 2.0, 0.0, 0.5, 0.0]))


def test_densify_ops_refuse_cpu_tensors(hip_lib):
    from moss_amd.densify import DensifyStats, neighbour_kl
    from knn_cuda import KnnGrid
    s = DensifyStats(4, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        s.add(torch.ones(4, dtype=torch.int32), torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        neighbour_kl(torch.zeros(4, 3), torch.ones(4, 4), torch.ones(4, 3), torch.zeros(4, 2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        KnnGrid(torch.zeros(4, 3))
    assert hip_lib.moss_knn_grid_workspace_bytes(100000) > 100000 * 24
    assert hip_lib.moss_knn_grid_build(0, None, None, 0, None) != 0          # argument validation happens before any launch
    assert hip_lib.moss_neighbour_kl(-1, 0, None, None, None, None, None, None) != 0


def test_bench_starts_its_own_ranks_and_reports_the_collective(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a torchrun environment must start two ranks itself (it used to run one GPU
    silently and print n_gpus: 1), relay rank 0's single JSON line, and carry rccl_ranks / allreduce_ms / adamw_ms /
    replicas_identical.  --dry-run-cpu swaps the GPU step for a host-side gradient bucket so the launcher, the rendezvous, the
    all-reduce (gloo) and the max-over-ranks timing run on a machine without GPUs."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MOSS_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["steps"] == 5 and res["scaling"] == "weak"
    assert res["replicas_identical"] is True and res["backend"] == "gloo"
    assert res["allreduce_ms"] >= 0.0 and "adamw_ms" in res and res["value"] > 0
    # a rank that fails takes the launcher down with a non-zero exit code and no JSON line
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)         # no GPU here: every rank asserts
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
