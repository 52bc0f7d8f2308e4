"""Oracle (b): an independent, naive PyTorch-AUTOGRAD rasterizer in float64.  TEST INFRASTRUCTURE ONLY.

It re-derives every gradient by automatic differentiation of the forward maths, so it checks the explicit backward of
the C oracle (oracle/moss_oracle.c, a restatement of backward.cu) without sharing any backward formula with it.
The forward encodes exactly the semantics the reference's backward assumes:
  (i)   a pixel's candidates are the entries of ITS TILE's sorted list (tile rect membership, (depth, index) order) --
        taken from the C oracle's binning, which is integer work and checked separately;
  (ii)  alpha = a + (min(a, 0.99) - a).detach()                      straight-through clamp (backward.cu:512,567,584)
  (iii) depth is detached inside the depth image                       (no dD/d depth_i, backward.cu:541-545)
  (iv)  the entry that would push T below 1e-4 is not blended and ends the pixel (forward.cu:351-356)
  (v)   entries with power > 0 or alpha < 1/255 are skipped             (forward.cu:341-350)
  (vi)  SH clamp passes gradient iff the un-clamped value is >= 0     (forward.cu:63-70, backward.cu:29-34)
  (vii) conic = inverse(cov2D); the reference's hand-written inverse backward uses 1/(det^2 + 1e-7) where autograd has
        1/det^2: relative difference <= 1e-7/det^2 <= 1.3e-5, inside the comparison tolerance
  (viii) culled Gaussians get zero gradients (they are in no list)
  (ix)  the means2D gradient is w.r.t. the NDC coordinate (factor 0.5*W, 0.5*H): a zero `ndc_offset` leaf added before
        ndc2Pix receives it, exactly like MOSS's screenspace_points sink (gaussian_renderer/__init__.py:29-33)
  (x)   when the +-1.3*tanfov clamp of computeCov2D is active the clamped coordinate is treated as a constant
        (backward.cu:175-176,262-264)
"""
from __future__ import annotations

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    """sh (P,M,3), dirs (P,3) -> (P,3), then +0.5 and clamp at 0."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_rot_raw(q):
    """rotation-like matrix of the quaternion AS GIVEN (not normalised), forward.cu:127-138 (row-major math matrix R)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    # glm::mat3(...) lists COLUMNS; the mathematical matrix is its transpose, but Sigma = M^T M with M = S*R_glm
    # equals R_math diag(s^2) R_math^T with R_math = R_glm^T:
    Rg = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return Rg


def render(fw, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, bg, degree,
           shs=None, colors_precomp=None, scales=None, rotations=None, scale_modifier=1.0, cov3D_precomp=None,
           ndc_offset=None):
    """All tensor inputs float64 (leaves may require grad).  `fw` = oracle.forward(...) namespace (binning only).
    Returns color (3,H,W), depth (1,H,W), alpha (1,H,W)."""
    W, H = fw.W, fw.H
    gx, gy = fw.grid
    P = means3D.shape[0]
    V = viewmatrix.T          # settings hold the TRANSPOSED matrices; V, PM act on column vectors
    PM = projmatrix.T
    ones = torch.ones(P, 1, dtype=means3D.dtype)
    hom = torch.cat([means3D, ones], dim=1)
    p_view = hom @ V.T
    p_hom = hom @ PM.T
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    if ndc_offset is not None:
        ndc = ndc + ndc_offset
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    depth = p_view[:, 2]

    if cov3D_precomp is not None:
        c = cov3D_precomp
    else:
        Rg = quat_to_rot_raw(rotations)                         # standard rotation matrix R (rows as listed)
        M = (scale_modifier * scales)[:, :, None] * Rg.transpose(1, 2)   # the reference's M = S * R^T (glm lists columns)
        Sigma = M.transpose(1, 2) @ M                            # = R S^2 R^T
        c = torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], dim=1)
    Vrk = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], dim=1).reshape(P, 3, 3)

    focal_x = W / (2.0 * tanfovx); focal_y = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    cx_ = (txtz.detach() < -limx) | (txtz.detach() > limx)
    cy_ = (tytz.detach() < -limy) | (tytz.detach() > limy)
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz, zero, -(focal_x * tx) / (tz * tz),
                     zero, focal_y / tz, -(focal_y * ty) / (tz * tz),
                     zero, zero, zero], dim=1).reshape(P, 3, 3)        # mathematical Jacobian (rows)
    Wm = V[:3, :3]
    Tm = J @ Wm
    cov2 = Tm @ Vrk @ Tm.transpose(1, 2)
    a_ = cov2[:, 0, 0] + 0.3; b_ = cov2[:, 0, 1]; c_ = cov2[:, 1, 1] + 0.3
    det = a_ * c_ - b_ * b_
    conic = torch.stack([c_ / det, -b_ / det, a_ / det], dim=1)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = sh_to_rgb(degree, shs, d)

    color = torch.zeros(3, H, W, dtype=means3D.dtype)
    dimg = torch.zeros(1, H, W, dtype=means3D.dtype)
    aimg = torch.zeros(1, H, W, dtype=means3D.dtype)
    color = color + bg[:, None, None]
    color_tiles, depth_tiles, alpha_tiles = {}, {}, {}
    plist = torch.from_numpy(fw.point_list.astype(np.int64))
    for t in range(gx * gy):
        r0, r1 = int(fw.ranges[t, 0]), int(fw.ranges[t, 1])
        if r1 <= r0:
            continue
        ids = plist[r0:r1]
        tx0, ty0 = (t % gx) * 16, (t // gx) * 16
        ys, xs = torch.meshgrid(torch.arange(ty0, min(ty0 + 16, H)), torch.arange(tx0, min(tx0 + 16, W)), indexing="ij")
        pxf = xs.reshape(-1).to(means3D.dtype); pyf = ys.reshape(-1).to(means3D.dtype)
        dx = pix[ids, 0][:, None] - pxf[None, :]
        dy = pix[ids, 1][:, None] - pyf[None, :]
        con = conic[ids]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        araw = opacities[ids].reshape(-1, 1) * torch.exp(power)
        a = araw + (torch.clamp(araw, max=0.99) - araw).detach()
        valid = (power.detach() <= 0) & (a.detach() >= 1.0 / 255.0)
        a = torch.where(valid, a, torch.zeros_like(a))
        om = 1.0 - a
        T_incl = torch.cumprod(om, dim=0)
        T_excl = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], dim=0)
        stop = valid & (T_incl.detach() < 1e-4)
        alive = torch.cumsum(stop.to(torch.int64), dim=0) == 0
        w = a * T_excl * alive
        Tf = torch.prod(torch.where(alive, om, torch.ones_like(om)), dim=0)
        col = (w[:, :, None] * rgb[ids][:, None, :]).sum(0) + Tf[:, None] * bg[None, :]
        dep = (w * depth[ids].detach()[:, None]).sum(0)
        alp = w.sum(0)
        hh, ww = ys.shape
        color_tiles[t] = col.T.reshape(3, hh, ww); depth_tiles[t] = dep.reshape(hh, ww); alpha_tiles[t] = alp.reshape(hh, ww)
    # assemble without in-place writes on a graph tensor
    rows_c, rows_d, rows_a = [], [], []
    for tyi in range(gy):
        rc, rd, ra = [], [], []
        for txi in range(gx):
            t = tyi * gx + txi
            hh = min(16, H - tyi * 16); ww = min(16, W - txi * 16)
            if t in color_tiles:
                rc.append(color_tiles[t]); rd.append(depth_tiles[t]); ra.append(alpha_tiles[t])
            else:
                rc.append(bg[:, None, None].expand(3, hh, ww).to(means3D.dtype)); rd.append(torch.zeros(hh, ww, dtype=means3D.dtype))
                ra.append(torch.zeros(hh, ww, dtype=means3D.dtype))
        rows_c.append(torch.cat(rc, dim=2)); rows_d.append(torch.cat(rd, dim=1)); rows_a.append(torch.cat(ra, dim=1))
    color = torch.cat(rows_c, dim=1)
    dimg = torch.cat(rows_d, dim=0)[None]
    aimg = torch.cat(rows_a, dim=0)[None]
    return color, dimg, aimg
