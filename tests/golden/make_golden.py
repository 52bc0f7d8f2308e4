#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (only possible in the build container, where
/root/reference is mounted).  The fixtures are data: seeded inputs and the outputs the reference functions produce.

    python tests/golden/make_golden.py

Reference functions exercised (all CPU-importable):
  utils/sh_utils.py       eval_sh (:57-112), RGB2SH (:114)
  utils/graphics_utils.py getWorld2View2 (:39-50), getProjectionMatrix_refine (:83-103), focal2fov (:108)
  utils/loss_utils.py     l1_loss (:41), l2_loss (:44), ssim (:57-87)    (+ torch autograd gradients)
  utils/general_utils.py  build_scaling_rotation (:108-118), strip_symmetric (:76-77) -- these hard-code device='cuda';
                          the generator redirects that ONE keyword to the CPU while calling them (the arithmetic is the
                          reference's), and composes them exactly as scene/gaussian_model.py:37-44 does.
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from utils import sh_utils, graphics_utils, loss_utils, general_utils  # noqa: E402


class cuda_kw_to_cpu:
    """While active, torch.zeros(..., device='cuda') allocates on the CPU (nothing else changes)."""

    def __enter__(self):
        self._zeros = torch.zeros

        def zeros(*a, **k):
            if str(k.get("device", "")) == "cuda":
                k["device"] = "cpu"
            return self._zeros(*a, **k)
        torch.zeros = zeros

    def __exit__(self, *exc):
        torch.zeros = self._zeros


def main():
    g = torch.Generator().manual_seed(3407)

    # ---- SH -> RGB ------------------------------------------------------------------------------------------
    P = 200
    sh = torch.randn(P, 3, 16, generator=g) * 0.5                      # reference layout for eval_sh: (..., C, coeffs)
    dirs = torch.randn(P, 3, generator=g); dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = sh_utils.eval_sh(deg, sh, dirs).numpy()
    out["rgb2sh_in"] = torch.rand(10, 3, generator=g).numpy()
    out["rgb2sh_out"] = sh_utils.RGB2SH(torch.from_numpy(out["rgb2sh_in"])).numpy()
    np.savez(os.path.join(OUT, "sh_eval.npz"), **out)

    # ---- camera matrices --------------------------------------------------------------------------------------
    cams = {}
    for i, (W, H, fx, fy, cx, cy, ang, t) in enumerate([
            (128, 128, 140.0, 140.0, 64.0, 64.0, 0.0, (0.0, 0.0, 3.0)),
            (512, 512, 540.0, 540.0, 268.0, 247.0, 0.6, (0.2, -0.1, 3.0)),
            (1024, 1024, 1080.0, 1075.0, 500.0, 530.0, -2.1, (0.0, 0.3, 2.5))]):
        R_w2c = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
        # scene/cameras.py:60 passes R such that Rt[:3,:3] = R.transpose(); i.e. R = R_w2c^T
        w2v = graphics_utils.getWorld2View2(R_w2c.T, np.array(t), np.array([0.0, 0.0, 0.0]), 1.0)
        proj = graphics_utils.getProjectionMatrix_refine(torch.tensor(K), H, W, 0.001, 1000).numpy()
        wvt = torch.tensor(w2v).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(torch.tensor(proj).transpose(0, 1).unsqueeze(0))).squeeze(0)
        cams.update({f"c{i}_params": np.array([W, H, fx, fy, cx, cy, ang, *t]), f"c{i}_w2v": w2v, f"c{i}_proj": proj,
                     f"c{i}_full": full.numpy(), f"c{i}_campos": wvt.inverse()[3, :3].numpy(),
                     f"c{i}_fov": np.array([graphics_utils.focal2fov(fx, W), graphics_utils.focal2fov(fy, H)])})
    np.savez(os.path.join(OUT, "camera.npz"), **cams)

    # ---- losses -------------------------------------------------------------------------------------------------
    losses = {}
    for i, (H, W) in enumerate([(48, 40), (64, 64)]):
        a = torch.rand(3, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
        b = torch.rand(3, H, W, generator=g, dtype=torch.float64)
        l1 = loss_utils.l1_loss(a, b); l2 = loss_utils.l2_loss(a, b); s = loss_utils.ssim(a.unsqueeze(0), b.unsqueeze(0))
        total = l1 + 0.2 * (1.0 - s)                                  # train_ZJU.py:131 (rasterizer-facing terms)
        total.backward()
        losses.update({f"l{i}_a": a.detach().numpy(), f"l{i}_b": b.numpy(), f"l{i}_l1": l1.item(), f"l{i}_l2": l2.item(),
                       f"l{i}_ssim": s.item(), f"l{i}_total": total.item(), f"l{i}_grad": a.grad.numpy(),
                       f"l{i}_ssim_f32": loss_utils.ssim(a.detach().float().unsqueeze(0), b.float().unsqueeze(0)).item()})
    np.savez(os.path.join(OUT, "loss.npz"), **losses)

    # ---- covariance from scale / rotation (+ per-Gaussian transform) ---------------------------------------------
    P = 64
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.4 - 3.0)
    rots = torch.randn(P, 4, generator=g)                              # build_rotation normalises internally
    T = torch.randn(P, 3, 3, generator=g) * 0.2 + torch.eye(3)
    with cuda_kw_to_cpu():
        def build_cov(scaling, modifier, rotation, transform=None):     # scene/gaussian_model.py:37-44
            L = general_utils.build_scaling_rotation(modifier * scaling, rotation)
            cov = L @ L.transpose(1, 2)
            if transform is not None:
                cov = transform @ cov
                cov = cov @ transform.transpose(1, 2)
            return general_utils.strip_symmetric(cov)
        cov_plain = build_cov(scales, 1.0, rots)
        cov_mod = build_cov(scales, 1.7, rots)
        cov_T = build_cov(scales, 1.0, rots, T)
        Rm = general_utils.build_rotation(rots)
    np.savez(os.path.join(OUT, "cov3d.npz"), scales=scales.numpy(), rots=rots.numpy(), transforms=T.numpy(),
             cov_plain=cov_plain.numpy(), cov_mod17=cov_mod.numpy(), cov_T=cov_T.numpy(), rotmat=Rm.numpy())
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


def bounding_rect(mask2d):
    """cv2.boundingRect of a 0/1 mask (cv2 is not installed here): x, y = the first non-zero column / row, w, h = the extent."""
    ys, xs = np.nonzero(mask2d)
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def loss_moss():
    """loss_moss.npz: MOSS's OWN loss expression for the rasterizer-facing terms, train_ZJU.py:108-119,131 -- the L1 and mask terms over
    the pixels of `bound_mask`, SSIM on the crop boundingRect(bound_mask) -- composed from the reference's l1_loss / l2_loss / ssim
    exactly as the script does (lpips / s3im / nll are other subsystems), values and autograd gradients in float64."""
    g = torch.Generator().manual_seed(3407 + 1)
    out = {}
    for i, (H, W, poly) in enumerate([(72, 60, (9, 14, 47, 61)), (56, 88, (30, 0, 81, 40)), (48, 48, (0, 0, 48, 48))]):
        # (float32-representable inputs: the kernels take float32; stored as float32, evaluated in float64)
        image = torch.rand(3, H, W, generator=g, dtype=torch.float32).double().requires_grad_(True)
        gt_image = torch.rand(3, H, W, generator=g, dtype=torch.float32).double()
        alpha = torch.rand(1, H, W, generator=g, dtype=torch.float32).double().requires_grad_(True)
        bkgd_mask = (torch.rand(1, H, W, generator=g) > 0.5).double()
        # bound_mask: the projected 3-D box of the body -- a convex region; here an ellipse inside the rectangle poly = (x0, y0, x1, y1)
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        x0, y0, x1, y1 = poly
        cx, cy, rx, ry = (x0 + x1 - 1) / 2, (y0 + y1 - 1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
        bound_mask = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0).to(torch.uint8)[None]
        if i == 2:
            bound_mask = torch.ones(1, H, W, dtype=torch.uint8)              # the whole frame: the expression without a crop
        Ll1 = loss_utils.l1_loss(image.permute(1, 2, 0)[bound_mask[0] == 1], gt_image.permute(1, 2, 0)[bound_mask[0] == 1])      # train_ZJU.py:111
        mask_loss = loss_utils.l2_loss(alpha[bound_mask == 1], bkgd_mask[bound_mask == 1])                                          # :112
        x, y, w, h = bounding_rect(bound_mask[0].numpy())                                                                             # :115
        img_pred = image[:, y:y + h, x:x + w].unsqueeze(0)
        img_gt = gt_image[:, y:y + h, x:x + w].unsqueeze(0)
        ssim_loss = loss_utils.ssim(img_pred, img_gt)                                                                                 # :119
        total = Ll1 + 0.5 * mask_loss + 0.2 * (1.0 - ssim_loss)                                                                        # :131
        total.backward()
        f32 = lambda t: t.detach().numpy().astype(np.float32)
        out.update({f"m{i}_image": f32(image), f"m{i}_gt": f32(gt_image), f"m{i}_alpha": f32(alpha),
                    f"m{i}_bkgd_mask": bkgd_mask.numpy().astype(np.uint8), f"m{i}_bound_mask": bound_mask.numpy(), f"m{i}_rect": np.array([x, y, w, h]),
                    f"m{i}_l1": Ll1.item(), f"m{i}_mask": mask_loss.item(), f"m{i}_ssim": ssim_loss.item(), f"m{i}_total": total.item(),
                    f"m{i}_grad_image": image.grad.numpy(), f"m{i}_grad_alpha": alpha.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "loss_moss.npz"), **out)
    print("wrote loss_moss.npz")


if __name__ == "__main__":
    if "--loss-moss" in sys.argv:                            # (round 6: added without regenerating the other fixtures)
        loss_moss()
    else:
        main()
        loss_moss()
