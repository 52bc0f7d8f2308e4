"""Synthetic stand-ins for the BASELINE.json configs (real ZJU-MoCap / MonoCap data is not in the repo).

Host-side numpy/torch-CPU generators only; everything is seeded (3407, the reference's seed,
utils/general_utils.py:141) so tests, bench and the CPU oracle see identical inputs.

Camera conventions follow the reference exactly (they are pinned by tests/golden/camera_*.npz, generated from
the reference's own ``getWorld2View2`` / ``getProjectionMatrix_refine``):
  * ``viewmatrix``  = world->view 4x4, TRANSPOSED (row-vector convention), scene/cameras.py:60
  * ``projmatrix``  = viewmatrix @ P^T with the off-centre principal point, scene/cameras.py:63-64,
                      utils/graphics_utils.py:83-103
  * ``campos``      = inverse(viewmatrix)[3, :3], scene/cameras.py:65
  * ``tanfovx/y``   = from the focal length only (scene/dataset_readers.py:656-659, gaussian_renderer/__init__.py:36-37)
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

SEED = 3407
C0 = 0.28209479177387814   # utils/sh_utils.py:24


def world2view(R_w2c: np.ndarray, t: np.ndarray) -> np.ndarray:
    """World->camera 4x4 (column-vector convention), float32.  Mirrors getWorld2View2 with
    translate=0, scale=1 (utils/graphics_utils.py:39-50), whose ``R`` argument is the transpose of R_w2c."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R_w2c
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_refine(K: np.ndarray, H: int, W: int, znear: float = 0.001, zfar: float = 1000.0) -> np.ndarray:
    """utils/graphics_utils.py:83-103 (getProjectionMatrix_refine), float32 arithmetic like the torch original."""
    K = torch.as_tensor(K, dtype=torch.float32)
    fx, fy, cx, cy, s = K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2 * fx / W
    P[0, 1] = 2 * s / W
    P[0, 2] = -1 + 2 * (cx / W)
    P[1, 1] = 2 * fy / H
    P[1, 2] = -1 + 2 * (cy / H)
    P[2, 2] = 1.0 * (zfar + znear) / (zfar - znear)
    P[2, 3] = -1 * 1.0 * 2 * zfar * znear / (zfar - znear)
    P[3, 2] = 1.0
    return P.numpy()


def make_camera(W, H, fx, fy, cx, cy, R_w2c=None, t=None, znear=0.001, zfar=1000.0) -> SimpleNamespace:
    R_w2c = np.eye(3) if R_w2c is None else np.asarray(R_w2c, dtype=np.float64)
    t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
    wvt = torch.tensor(world2view(R_w2c, t)).transpose(0, 1).contiguous()                 # cameras.py:60
    proj = torch.tensor(projection_refine(K, H, W, znear, zfar)).transpose(0, 1)          # cameras.py:63
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()              # cameras.py:64
    campos = wvt.inverse()[3, :3].contiguous()                                            # cameras.py:65
    fovx = 2 * math.atan(W / (2 * fx))      # focal2fov, graphics_utils.py:108
    fovy = 2 * math.atan(H / (2 * fy))
    return SimpleNamespace(W=W, H=H, K=K, viewmatrix=wvt, projmatrix=full, campos=campos,
                           tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5), FoVx=fovx, FoVy=fovy)


def look_at_ring(n: int, radius: float = 3.0, height: float = 0.0):
    """n world->camera poses on a horizontal ring around the origin, all looking at the origin (cfg4)."""
    poses = []
    for i in range(n):
        a = 2 * math.pi * i / n
        c = np.array([radius * math.sin(a), height, -radius * math.cos(a)])
        fwd = -c / np.linalg.norm(c)                     # camera +z
        down_hint = np.array([0.0, 1.0, 0.0])            # world y points down (head of the body is at y < 0)
        right = np.cross(down_hint, fwd); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)                      # x (right) cross y (down) = z (forward)
        R = np.stack([right, down, fwd])                 # rows = camera axes in world coords
        poses.append((R, -R @ c))
    return poses


def _gen(seed=SEED):
    return torch.Generator().manual_seed(seed)


def _rand_quat(P, g):
    q = torch.randn(P, 4, generator=g)
    return q / q.norm(dim=1, keepdim=True)


def _rand_sh(P, g, sigma=0.3):
    sh = torch.randn(P, 16, 3, generator=g) * sigma
    sh[:, 0, :] += (torch.rand(P, 3, generator=g) - 0.5) / C0          # RGB2SH, utils/sh_utils.py:114
    return sh


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrices of (normalised) quaternions (r,x,y,z), the math of utils/general_utils.py:78-101."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def covariance_precomp(scales, rots, scale_modifier=1.0, transforms=None) -> torch.Tensor:
    """(P,6) upper-triangular covariances the way MOSS feeds them: strip_symmetric(T (R S S^T R^T) T^T)
    (scene/gaussian_model.py:37-44, utils/general_utils.py:65-118)."""
    R = quat_to_rot(rots)
    L = R * (scale_modifier * scales)[:, None, :]
    if transforms is not None:
        # T (R S) ; cov = L L^T = T R S S^T R^T T^T.  (= transforms @ L without the batched 3x3 GEMM, see below)
        L = (transforms[:, :, :, None] * L[:, None, :, :]).sum(2)
    # six unique entries of L L^T as row dot products: elementwise kernels only (a batched 3x3 GEMM over 100k tiny
    # matrices costs ~0.9 ms per call through hipBLASLt on MI355X)
    r0, r1, r2 = L[:, 0, :], L[:, 1, :], L[:, 2, :]
    return torch.stack([(r0 * r0).sum(1), (r0 * r1).sum(1), (r0 * r2).sum(1),
                        (r1 * r1).sum(1), (r1 * r2).sum(1), (r2 * r2).sum(1)], dim=1).contiguous()


def config1(P=256, W=128, H=128, seed=SEED) -> SimpleNamespace:
    """BASELINE configs[0]: 256 synthetic Gaussians, 128x128."""
    g = _gen(seed)
    s = SimpleNamespace(name="cfg1", P=P, sh_degree=3)
    s.means3D = (torch.rand(P, 3, generator=g) * 1.2 - 0.6)
    s.scales = torch.exp(math.log(0.05) + 0.3 * torch.randn(P, 3, generator=g))
    s.rotations = _rand_quat(P, g)
    s.opacities = torch.sigmoid(torch.randn(P, 1, generator=g))
    s.shs = _rand_sh(P, g)
    s.bg = torch.zeros(3)
    A = torch.randn(P, 3, 3, generator=g) * 0.1 + torch.eye(3)           # per-Gaussian LBS-like 3x3 transforms
    s.transforms = A
    s.cov3D_precomp = covariance_precomp(s.scales, s.rotations, 1.0, A)
    s.camera = make_camera(W, H, 140.0, 140.0, W / 2, H / 2, np.eye(3), np.array([0.0, 0.0, 3.0]))
    return s


# capsule-union "body": (a, b, radius) segments, metres; total height ~1.7 m centred at the origin, y down
_BODY = [
    ((0.0, -0.25, 0.0), (0.0, 0.25, 0.0), 0.17),      # torso
    ((0.0, -0.62, 0.0), (0.0, -0.50, 0.0), 0.11),     # head
    ((-0.24, -0.32, 0.0), (-0.42, 0.22, 0.0), 0.055), # arms
    ((0.24, -0.32, 0.0), (0.42, 0.22, 0.0), 0.055),
    ((-0.10, 0.30, 0.0), (-0.14, 0.80, 0.0), 0.075),  # legs
    ((0.10, 0.30, 0.0), (0.14, 0.80, 0.0), 0.075),
]


def body_points(P: int, g: torch.Generator, jitter=0.005) -> torch.Tensor:
    """P points on the surface of the capsule union, area-weighted, jittered 5 mm."""
    areas = []
    for a, b, r in _BODY:
        L = float(np.linalg.norm(np.subtract(b, a)))
        areas.append(2 * math.pi * r * L + 4 * math.pi * r * r)
    probs = torch.tensor(areas) / sum(areas)
    which = torch.multinomial(probs, P, replacement=True, generator=g)
    pts = torch.zeros(P, 3)
    for i, (a, b, r) in enumerate(_BODY):
        m = which == i
        n = int(m.sum())
        if n == 0:
            continue
        a = torch.tensor(a); b = torch.tensor(b)
        axis = b - a
        L = axis.norm()
        axis = axis / L
        # local frame
        tmp = torch.tensor([1.0, 0.0, 0.0]) if abs(float(axis[0])) < 0.9 else torch.tensor([0.0, 0.0, 1.0])
        u = torch.linalg.cross(axis, tmp); u = u / u.norm()
        v = torch.linalg.cross(axis, u)
        cyl_area = 2 * math.pi * r * float(L)
        cap_area = 4 * math.pi * r * r
        on_cyl = torch.rand(n, generator=g) < cyl_area / (cyl_area + cap_area)
        phi = torch.rand(n, generator=g) * 2 * math.pi
        h = torch.rand(n, generator=g) * L
        p_cyl = a + h[:, None] * axis + r * (torch.cos(phi)[:, None] * u + torch.sin(phi)[:, None] * v)
        d = torch.randn(n, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
        along = d @ axis
        centre = torch.where(along[:, None] > 0, b.expand(n, 3), a.expand(n, 3))
        p_cap = centre + r * d
        pts[m] = torch.where(on_cyl[:, None], p_cyl, p_cap)
    pts += jitter * torch.randn(P, 3, generator=g)
    return pts


def _knn_dist2(points: torch.Tensor) -> torch.Tensor:
    """mean squared distance to the 3 nearest other points (distCUDA2 semantics) via scipy, scene generation only."""
    from scipy.spatial import cKDTree
    x = points.double().numpy()
    d, _ = cKDTree(x).query(x, k=4)
    return torch.from_numpy((d[:, 1:] ** 2).mean(axis=1)).float()


def body_scene(P: int, W: int, H: int, fx: float, init_like: bool, seed=SEED, cam_pose=None,
               principal_offset=(12.0, -9.0), name="body") -> SimpleNamespace:
    """cfg2 (init_like=True: isotropic scales = sqrt(dist2), opacity 0.1, identity rotations, as
    GaussianModel.create_from_pcd scene/gaussian_model.py:175-198) or cfg3/cfg5 (post-densify-like statistics)."""
    g = _gen(seed)
    s = SimpleNamespace(name=name, P=P, sh_degree=3)
    s.means3D = body_points(P, g)
    dist2 = torch.clamp_min(_knn_dist2(s.means3D), 1e-7)
    if init_like:
        s.scales = torch.sqrt(dist2)[:, None].repeat(1, 3)
        s.rotations = torch.zeros(P, 4); s.rotations[:, 0] = 1
        s.opacities = torch.full((P, 1), 0.1)
    else:
        s.scales = torch.sqrt(dist2)[:, None] * torch.exp(0.3 * torch.randn(P, 3, generator=g))
        s.rotations = _rand_quat(P, g)
        s.opacities = torch.sigmoid(1.0 + 1.5 * torch.randn(P, 1, generator=g))
    s.shs = _rand_sh(P, g)
    s.bg = torch.zeros(3)
    s.transforms = None
    s.cov3D_precomp = covariance_precomp(s.scales, s.rotations, 1.0, None)
    R, t = (np.eye(3), np.array([0.0, 0.0, 3.0])) if cam_pose is None else cam_pose
    s.camera = make_camera(W, H, fx, fx, W / 2 + principal_offset[0], H / 2 + principal_offset[1], R, t)
    return s


def config2(seed=SEED):
    """BASELINE configs[1]: ~6.9k SMPL-vertex-init Gaussians, 512x512."""
    return body_scene(6890, 512, 512, 540.0, init_like=True, seed=seed, name="cfg2")


def config3(seed=SEED, P=100_000, cam_pose=None):
    """BASELINE configs[2] (the config `metric` is quoted on): ~100k Gaussians, 512x512."""
    return body_scene(P, 512, 512, 540.0, init_like=False, seed=seed, cam_pose=cam_pose, name="cfg3")


def config5(seed=SEED, P=300_000):
    """BASELINE configs[4]: ~300k Gaussians, 1024x1024."""
    return body_scene(P, 1024, 1024, 1080.0, init_like=False, seed=seed, name="cfg5")


def synthetic_target(H, W, seed=SEED) -> torch.Tensor:
    """A smooth (3,H,W) ground-truth image in [0,1] for the L1+SSIM step (no dataset access)."""
    g = _gen(seed + 1)
    low = torch.rand(1, 3, H // 16 + 2, W // 16 + 2, generator=g)
    img = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)[0]
    return img.clamp(0, 1).contiguous()


def scripted_densification(tensors, step, device, reset_opacity=False, clone_frac=1 / 12, split_frac=1 / 25, prune_frac=1 / 40, seed=4242):
    """A deterministic densification EVENT for tests and the bench (a stand-in WORKLOAD: which Gaussians MOSS clones, splits or prunes
    is decided by its own control plane, scene/gaussian_model.py:456-620, from gradient statistics, KL and screen size -- out of scope).
    ``tensors``: {"xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"} -> the current raw parameters.  Returns the arguments of
    ``moss_amd.surgery.densification_event``: clones of a seeded selection (densify_and_clone: copies), two jittered children with
    scales / 1.6 for another selection whose sources are pruned (densify_and_split, N = 2: :466-475, :526-527), a few more pruned, and
    optionally the opacity reset.  Decisions depend on the parameters' SHAPES and the seed only: every replica takes the same one."""
    # Everything is drawn ON ``device``.  (Round 6 first drew on the CPU and copied.  CPU torch ops on >= 32k elements run on an OpenMP team
    # of torch.get_num_threads() threads -- 128 on the GPU boxes, whose containers have a CPU QUOTA of 16: the team burns the quota of
    # the 100 ms period in a few milliseconds and the kernel then suspends the WHOLE process, HIP runtime threads included, until the
    # period ends -- stalls of 10-90 ms at random places a moment later: scripts/micro/cpu_parallel_stall.py, profiles/r06_notes.md
    # section 10.  configs[1]'s tensors are below the parallel grain and its schedule had none.)
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed + step)
    P = tensors["xyz"].shape[0]
    src = torch.randperm(P, generator=g, device=device)[:max(int(P * clone_frac), 1)]
    pick = lambda idx: {"new_xyz": tensors["xyz"][idx].clone(), "new_features_dc": tensors["f_dc"][idx].clone(),
                        "new_features_rest": tensors["f_rest"][idx].clone(), "new_opacities": tensors["opacity"][idx].clone(),
                        "new_scaling": tensors["scaling"][idx].clone(), "new_rotation": tensors["rotation"][idx].clone(), "source": idx}
    clone = pick(src)
    src2 = torch.randperm(P, generator=g, device=device)[:max(int(P * split_frac), 1)]
    split = pick(src2.repeat(2))
    split["new_xyz"] = split["new_xyz"] + torch.randn(2 * src2.numel(), 3, generator=g, device=device) * 0.004
    split["new_scaling"] = split["new_scaling"] - math.log(1.6)          # get_scaling / (0.8 N) in log space, :474
    P2 = P + src.numel() + 2 * src2.numel()
    prune = torch.zeros(P2, dtype=torch.bool, device=device)
    prune[src2] = True                                                   # the split sources (:526-527)
    prune[torch.randperm(P2, generator=g, device=device)[:max(int(P2 * prune_frac), 1)]] = True
    return {"append": [clone, split], "prune": prune, "reset_opacity": bool(reset_opacity)}
