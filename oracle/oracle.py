"""numpy front-end of the C oracle (``oracle/moss_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

``forward()`` / ``backward()`` compose the stage functions in the order of the reference's
``CudaRasterizer::Rasterizer::forward`` / ``::backward``
(submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-341, :345-447) with the
allocation / zero-fill / P==0 behaviour of ``RasterizeGaussiansCUDA`` / ``RasterizeGaussiansBackwardCUDA``
(submodules/diff-gaussian-rasterization/rasterize_points.cu:35-119, :121-206).
Every intermediate (radii, tiles_touched, offsets, unsorted/sorted keys, ranges, final_T, n_contrib ...)
is returned so the HIP path can be compared stage by stage.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmoss_oracle.so")
_LIB64_PATH = os.path.join(_HERE, "_build", "libmoss_oracle_f64.so")   # the same source compiled with float -> double (Makefile)
_lib = None
_lib64 = None

BLOCK_X = 16
BLOCK_Y = 16


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the path of the shared object."""
    src = os.path.join(_HERE, "moss_oracle.c")
    stale = lambda p: not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src)
    if force or stale(_LIB_PATH) or stale(_LIB64_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_preprocess.restype = C.c_int
        _lib.oracle_inclusive_sum.restype = C.c_int
        _lib.oracle_get_higher_msb.restype = C.c_uint32
        _lib.oracle_get_higher_msb.argtypes = [C.c_uint32]
    return _lib


def lib64() -> C.CDLL:
    """The float64 build of the same source: every `float` of moss_oracle.c is a `double` here (arrays AND scalars)."""
    global _lib64
    if _lib64 is None:
        build()
        _lib64 = C.CDLL(_LIB64_PATH)
        _lib64.oracle_preprocess.restype = C.c_int
    return _lib64


def _p(a):
    """ctypes pointer of a numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _opt(a):
    """reference convention: an absent tensor is an EMPTY tensor whose data pointer is null
    (submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:200-210)."""
    if a is None:
        return None
    a = np.asarray(a)
    return None if a.size == 0 else _f32(a)


class det_exp:
    """``with oracle.det_exp(on):`` -- the blend forward / backward of the float32 oracle evaluate exp() with ``moss_expf_det`` (the
    deterministic restatement of glibc's expf that csrc/blend.hip carries too, moss_oracle.c) instead of the C library's expf: the
    oracle side of the EXACT-MATH parity mode (MOSS_DEBUG_EXACT_MATH).  The float64 adjudicator is not affected."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        lib().oracle_set_det_exp(C.c_int(1 if self.on else 0))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_det_exp(C.c_int(0))
        return False


def expf_det_mismatches(x) -> int:
    """How many of the float32 values x have moss_expf_det(x) != expf(x) (the C library's), bit for bit."""
    x = _f32(x).reshape(-1)
    L = lib()
    L.oracle_expf_det_mismatches.restype = C.c_int
    return int(L.oracle_expf_det_mismatches(_p(x), C.c_int(x.size)))


def expf_det(x):
    x = _f32(x).reshape(-1)
    y = np.zeros_like(x)
    lib().oracle_expf_det_array(_p(x), C.c_int(x.size), _p(y))
    return y


def tile_grid(W, H):
    return (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y


def get_higher_msb(n: int) -> int:
    return int(lib().oracle_get_higher_msb(C.c_uint32(n)))


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    present = np.zeros(P, dtype=np.uint8)
    if P:
        lib().oracle_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(present))
    return present.astype(bool)


def dist2(points):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, dtype=np.float32)
    if P:
        lib().oracle_dist2(C.c_int(P), _p(points), _p(out))
    return out


def forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
            viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered=False,
            want_margin=True, transforms=None, f64=False, binning_from=None):
    """Returns a namespace with the reference's outputs (color (3,H,W), depth (1,H,W), alpha (1,H,W),
    radii (P,), num_rendered) and every intermediate.

    ``f64=True`` (the ADJUDICATOR, not the reference's arithmetic): the float64 build of the same source evaluates preprocess and
    the blend in double precision.  The integer structure of the frame -- which Gaussians are in which tile, in which order --
    is taken from ``binning_from`` (a float32 forward of the same inputs: radii, tiles_touched, point_list, ranges), so that the
    float64 run differentiates THE SAME function of the inputs as the float32 paths it adjudicates."""
    if f64:
        return _forward64(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                          viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, transforms, binning_from)
    L = lib()
    bg = _f32(bg); means3D = _f32(means3D); opacities = _f32(opacities)
    viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos)
    colors_precomp = _opt(colors_precomp); scales = _opt(scales); rotations = _opt(rotations)
    cov3D_precomp = _opt(cov3D_precomp); sh = _opt(sh)
    transforms = None if transforms is None else np.ascontiguousarray(_f32(transforms).reshape(-1, 9))   # n2 extension: (P,3,3)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
    P = means3D.shape[0]
    M = 0 if sh is None else sh.shape[1]
    N = W * H
    gx, gy = tile_grid(W, H)
    o = SimpleNamespace(P=P, W=W, H=H, M=M, D=degree, grid=(gx, gy))
    o.color = np.zeros((3, H, W), np.float32)
    o.depth = np.zeros((1, H, W), np.float32)
    o.alpha = np.zeros((1, H, W), np.float32)
    o.radii = np.zeros(P, np.int32)
    o.num_rendered = 0
    if P == 0:                                                              # rasterize_points.cu:83
        return o
    o.means2D = np.zeros((P, 2), np.float32)
    o.depths = np.zeros(P, np.float32)
    o.cov3D = np.zeros((P, 6), np.float32)
    o.rgb = np.zeros((P, 3), np.float32)
    o.conic_opacity = np.zeros((P, 4), np.float32)
    o.tiles_touched = np.zeros(P, np.uint32)
    o.clamped = np.zeros((P, 3), np.uint8)
    err = L.oracle_preprocess(
        C.c_int(P), C.c_int(degree), C.c_int(M), _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations),
        _p(opacities), _p(sh), _p(cov3D_precomp), _p(colors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
        C.c_int(W), C.c_int(H), C.c_float(tan_fovx), C.c_float(tan_fovy), C.c_int(int(prefiltered)),
        _p(o.radii), _p(o.means2D), _p(o.depths), _p(o.cov3D), _p(o.rgb), _p(o.conic_opacity),
        _p(o.tiles_touched), _p(o.clamped), _p(transforms))
    if err:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    o.point_offsets = np.zeros(P, np.uint32)
    R = L.oracle_inclusive_sum(C.c_int(P), _p(o.tiles_touched), _p(o.point_offsets))
    o.num_rendered = R
    o.keys_unsorted = np.zeros(R, np.uint64)
    o.values_unsorted = np.zeros(R, np.uint32)
    L.oracle_duplicate_with_keys(C.c_int(P), _p(o.means2D), _p(o.depths), _p(o.point_offsets),
                                 _p(o.keys_unsorted), _p(o.values_unsorted), _p(o.radii), C.c_int(W), C.c_int(H))
    o.sort_bits = 32 + get_higher_msb(gx * gy)
    o.point_list_keys = np.zeros(R, np.uint64)
    o.point_list = np.zeros(R, np.uint32)
    L.oracle_sort_pairs(C.c_int(R), _p(o.keys_unsorted), _p(o.values_unsorted), _p(o.point_list_keys),
                        _p(o.point_list), C.c_int(o.sort_bits))
    o.ranges = np.zeros((gx * gy, 2), np.uint32)
    L.oracle_identify_tile_ranges(C.c_int(R), _p(o.point_list_keys), _p(o.ranges), C.c_int(gx * gy))
    o.final_T = np.zeros(N, np.float32)
    o.n_contrib = np.zeros(N, np.uint32)
    o.margin = np.ones(N, np.float32) if want_margin else None
    o.features = colors_precomp if colors_precomp is not None else o.rgb    # rasterizer_impl.cu:323
    L.oracle_render_forward(C.c_int(W), C.c_int(H), _p(o.ranges), _p(o.point_list), _p(o.means2D),
                            _p(np.ascontiguousarray(o.features)), _p(o.depths), _p(o.conic_opacity), _p(bg),
                            _p(o.color), _p(o.depth), _p(o.alpha), _p(o.final_T), _p(o.n_contrib), _p(o.margin))
    return o


def _forward64(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
               viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, transforms, fw32):
    """float64 preprocess + blend over the float32 run's tile lists (see forward(f64=True))."""
    assert fw32 is not None, "the float64 adjudicator needs the float32 forward whose tile lists it shares (binning_from=)"
    L = lib64()
    f64 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32), dtype=np.float64)   # fp32 INPUTS, exactly
    opt = lambda a: None if (a is None or np.asarray(a).size == 0) else f64(a)
    bg = f64(bg); means3D = f64(means3D); opacities = f64(opacities)
    viewmatrix = f64(viewmatrix); projmatrix = f64(projmatrix); campos = f64(campos)
    colors_precomp = opt(colors_precomp); scales = opt(scales); rotations = opt(rotations); cov3D_precomp = opt(cov3D_precomp); sh = opt(sh)
    transforms = None if transforms is None else np.ascontiguousarray(f64(transforms).reshape(-1, 9))
    P = means3D.shape[0]
    M = 0 if sh is None else sh.shape[1]
    N = W * H
    d = C.c_double
    o = SimpleNamespace(P=P, W=W, H=H, M=M, D=degree, grid=fw32.grid, f64=True)
    o.color = np.zeros((3, H, W)); o.depth = np.zeros((1, H, W)); o.alpha = np.zeros((1, H, W))
    o.radii = np.zeros(P, np.int32)
    o.num_rendered = fw32.num_rendered
    if P == 0:
        return o
    o.means2D = np.zeros((P, 2)); o.depths = np.zeros(P); o.cov3D = np.zeros((P, 6)); o.rgb = np.zeros((P, 3))
    o.conic_opacity = np.zeros((P, 4)); o.tiles_touched = np.zeros(P, np.uint32); o.clamped = np.zeros((P, 3), np.uint8)
    L.oracle_preprocess(
        C.c_int(P), C.c_int(degree), C.c_int(M), _p(means3D), _p(scales), d(float(np.float32(scale_modifier))), _p(rotations),
        _p(opacities), _p(sh), _p(cov3D_precomp), _p(colors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
        C.c_int(W), C.c_int(H), d(float(np.float32(tan_fovx))), d(float(np.float32(tan_fovy))), C.c_int(0),
        _p(o.radii), _p(o.means2D), _p(o.depths), _p(o.cov3D), _p(o.rgb), _p(o.conic_opacity),
        _p(o.tiles_touched), _p(o.clamped), _p(transforms))
    # the frame's integer structure is the float32 run's: same visible set, same tile lists, same order
    o.radii64 = o.radii
    o.radii = fw32.radii; o.tiles_touched = fw32.tiles_touched
    o.point_list = fw32.point_list; o.ranges = fw32.ranges; o.point_list_keys = fw32.point_list_keys
    o.final_T = np.zeros(N); o.n_contrib = np.zeros(N, np.uint32); o.margin = np.ones(N)
    o.features = colors_precomp if colors_precomp is not None else o.rgb
    L.oracle_render_forward(C.c_int(W), C.c_int(H), _p(o.ranges), _p(o.point_list), _p(o.means2D),
                            _p(np.ascontiguousarray(o.features)), _p(o.depths), _p(o.conic_opacity), _p(bg),
                            _p(o.color), _p(o.depth), _p(o.alpha), _p(o.final_T), _p(o.n_contrib), _p(o.margin))
    return o


def backward(fw, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
             projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha, sh, degree, campos, transforms=None,
             f32_accumulators=False, sum_noise_ulps=0.0, noise_seed=0, chain_noise_ulps=0.0):
    """``fw`` is the namespace returned by :func:`forward`.  Returns the reference's 8 gradient arrays
    (rasterize_points.cu:205) plus dL_dconic as a namespace.  A float64 ``fw`` (forward(f64=True)) runs the float64 build.
    ``f32_accumulators``: the blend backward adds its per-pixel terms in float32 in loop order -- one of the orders the reference's
    atomicAdd can take -- instead of in double (the default: the centre of that distribution).
    ``sum_noise_ulps`` (a conditioning probe, meant for the float64 build): the per-Gaussian sums the blend backward hands to the
    per-Gaussian stages (dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors -- BACKWARD::render's atomicAdd targets, backward.cu:566-588)
    are multiplied by 1 + eta, eta uniform in +-sum_noise_ulps * 2^-24, before those stages run: what ANY float32 evaluation of the
    sums does to them at the least.  The change of the final gradients against the unperturbed run is how far float32 rounding of
    the sums alone is entitled to move them (a 300:1 anisotropic Gaussian amplifies it by 10^3-10^4 in its scale gradient).
    ``chain_noise_ulps`` (the same kind of probe, one stage further): the six-vector dL/dSigma3 that computeCov2DCUDA hands to
    computeCov3D's backward (backward.cu:268-273 -> :278-341) is multiplied by 1 + eta likewise -- the reference STORES that vector in
    float32 between its two kernels, so its own arithmetic rounds exactly there; for a needle whose long axis points along the
    viewing ray the scale gradient r^T dSigma3 r is a difference of terms 10^3 times its size and that rounding is what it is
    entitled to (round 3's finding 24)."""
    is64 = bool(getattr(fw, "f64", False))
    L = lib64() if is64 else lib()
    dt = np.float64 if is64 else np.float32
    cf = (lambda v: C.c_double(float(np.float32(v)))) if is64 else (lambda v: C.c_float(v))
    if is64:
        _f = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32), dtype=np.float64)
        _o = lambda a: None if (a is None or np.asarray(a).size == 0) else _f(a)
    else:
        _f, _o = _f32, _opt
    bg = _f(bg); means3D = _f(means3D)
    viewmatrix = _f(viewmatrix); projmatrix = _f(projmatrix); campos = _f(campos)
    colors_precomp = _o(colors_precomp); scales = _o(scales); rotations = _o(rotations)
    cov3D_precomp = _o(cov3D_precomp); sh = _o(sh)
    transforms = None if transforms is None else np.ascontiguousarray(_f(transforms).reshape(-1, 9))
    P = means3D.shape[0]
    M = 0 if sh is None else sh.shape[1]
    H, W = fw.H, fw.W
    g = SimpleNamespace()
    g.dL_dmeans3D = np.zeros((P, 3), dt)                            # rasterize_points.cu:158-166
    g.dL_dmeans2D = np.zeros((P, 3), dt)
    g.dL_dcolors = np.zeros((P, 3), dt)
    g.dL_dconic = np.zeros((P, 2, 2), dt)
    g.dL_dopacity = np.zeros((P, 1), dt)
    g.dL_dcov3D = np.zeros((P, 6), dt)
    g.dL_dsh = np.zeros((P, M, 3), dt)
    g.dL_dscales = np.zeros((P, 3), dt)
    g.dL_drotations = np.zeros((P, 4), dt)
    g.dL_dtransforms = np.zeros((P, 3, 3), dt)
    if P == 0:
        return g
    dpix = _f(dL_dout_color).reshape(3, H, W)
    ddep = _f(dL_dout_depth).reshape(H * W)
    dalp = _f(dL_dout_alpha).reshape(H * W)
    color_ptr = colors_precomp if colors_precomp is not None else fw.rgb    # rasterizer_impl.cu:397
    (L.oracle_render_backward_f32acc if f32_accumulators else L.oracle_render_backward)(
                             C.c_int(P), C.c_int(W), C.c_int(H), _p(fw.ranges), _p(fw.point_list), _p(bg),
                             _p(fw.means2D), _p(fw.conic_opacity), _p(np.ascontiguousarray(color_ptr)), _p(fw.depths),
                             _p(fw.final_T), _p(fw.n_contrib), _p(dpix), _p(ddep), _p(dalp),
                             _p(g.dL_dmeans2D), _p(g.dL_dconic), _p(g.dL_dopacity), _p(g.dL_dcolors))
    if sum_noise_ulps:
        rng = np.random.default_rng(noise_seed)
        for a in (g.dL_dmeans2D, g.dL_dconic, g.dL_dopacity, g.dL_dcolors):
            a *= (1.0 + (rng.random(a.shape) * 2.0 - 1.0) * float(sum_noise_ulps) * 2.0 ** -24).astype(dt)
    cov3D_ptr = cov3D_precomp if cov3D_precomp is not None else fw.cov3D    # rasterizer_impl.cu:424
    if is64:
        focal_y = H / (2.0 * float(np.float32(tan_fovy))); focal_x = W / (2.0 * float(np.float32(tan_fovx)))
        fx_c, fy_c = C.c_double(focal_x), C.c_double(focal_y)
    else:
        focal_y = np.float32(H) / (np.float32(2.0) * np.float32(tan_fovy))      # rasterizer_impl.cu:388-389
        focal_x = np.float32(W) / (np.float32(2.0) * np.float32(tan_fovx))
        fx_c, fy_c = C.c_float(focal_x), C.c_float(focal_y)
    L.oracle_compute_cov2d_backward(C.c_int(P), _p(means3D), _p(fw.radii), _p(np.ascontiguousarray(cov3D_ptr)),
                                    fx_c, fy_c, cf(tan_fovx), cf(tan_fovy),
                                    _p(viewmatrix), _p(g.dL_dconic), _p(g.dL_dmeans3D), _p(g.dL_dcov3D))
    kept_cov3D = None
    if chain_noise_ulps:
        rng = np.random.default_rng(noise_seed + 7919)
        kept_cov3D = g.dL_dcov3D.copy()                      # (the returned dL_dcov3D stays the unperturbed one: only its CONSUMER sees the noise)
        g.dL_dcov3D *= (1.0 + (rng.random(g.dL_dcov3D.shape) * 2.0 - 1.0) * float(chain_noise_ulps) * 2.0 ** -24).astype(dt)
    L.oracle_preprocess_backward(C.c_int(P), C.c_int(degree), C.c_int(M), _p(means3D), _p(fw.radii), _p(sh),
                                 _p(fw.clamped), _p(scales), _p(rotations), cf(scale_modifier), _p(projmatrix),
                                 _p(campos), _p(g.dL_dmeans2D), _p(g.dL_dmeans3D), _p(g.dL_dcolors), _p(g.dL_dcov3D),
                                 _p(g.dL_dsh), _p(g.dL_dscales), _p(g.dL_drotations), _p(transforms), _p(g.dL_dtransforms))
    if kept_cov3D is not None:
        g.dL_dcov3D = kept_cov3D
    return g


# ---------------------------------------------------------------------------------------------------------------------
# Densification bookkeeping and the KL test of the KL-guided densify: numpy restatements.
# Parity unpinned by the reference for kl_div: GaussianModel.kl_div allocates with .to('cuda') (scene/gaussian_model.py:796)
# and cannot run in a container without CUDA; it is restated here step by step in float64 and pinned by known answers
# (tests/test_oracle_cpu.py: identical Gaussians -> 0, isotropic closed form, scipy-free direct matrix evaluation).

def densify_stats(radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D):
    """train_ZJU.py:171-174 + GaussianModel.add_densification_stats (scene/gaussian_model.py:815-817), out of place, fp32."""
    vis = np.asarray(radii) > 0
    acc = np.array(xyz_gradient_accum, dtype=np.float32, copy=True).reshape(-1)
    den = np.array(denom, dtype=np.float32, copy=True).reshape(-1)
    mr = np.array(max_radii2D, dtype=np.float32, copy=True)
    g = np.asarray(viewspace_grad, dtype=np.float32)
    mr[vis] = np.maximum(mr[vis], np.asarray(radii)[vis].astype(np.float32))                     # train_ZJU.py:173
    acc[vis] += np.sqrt(g[vis, 0] * g[vis, 0] + g[vis, 1] * g[vis, 1]).astype(np.float32)      # gaussian_model.py:816
    den[vis] += 1.0                                                                             # :817
    return acc, den, mr


def build_rotation(q):
    """utils/general_utils.py:79-100 (float64)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.sqrt((q * q).sum(-1, keepdims=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def kl_div(mu_0, rotation_0_q, scaling_0_diag, mu_1, rotation_1_q, scaling_1_diag):
    """GaussianModel.kl_div, scene/gaussian_model.py:773-813, in float64; returns (kl, magnitude) where magnitude is the sum
    of the absolute values of the four terms (the scale the fp32 kernel's rounding error is measured against)."""
    s0 = np.asarray(scaling_0_diag, dtype=np.float64); s1 = np.asarray(scaling_1_diag, dtype=np.float64)
    R0 = build_rotation(rotation_0_q)                                   # :776
    L0 = R0 * s0[:, None, :]                                            # rotation_0 @ build_scaling(s0)   :779
    cov0 = L0 @ L0.transpose(0, 2, 1)                                   # :780
    R1 = build_rotation(rotation_1_q)                                   # :783
    L1i = R1 * (1.0 / s1)[:, None, :]                                   # rotation_1 @ build_scaling(1/s1) :786
    cov1_inv = L1i @ L1i.transpose(0, 2, 1)                             # :787
    d = np.asarray(mu_1, dtype=np.float64) - np.asarray(mu_0, dtype=np.float64)    # :790
    kl0 = np.trace(cov1_inv @ cov0, axis1=1, axis2=2)                   # :796-804 (trace of each product)
    kl1 = np.einsum("pi,pij,pj->p", d, cov1_inv, d)                     # :806
    kl2 = np.log(np.prod((s1 / s0) ** 2, axis=1))                       # :807
    return 0.5 * (kl0 + kl1 + kl2 - 3), np.abs(kl0) + np.abs(kl1) + np.abs(kl2) + 3   # :808


def knn_exhaustive(ref, query, k):
    """Exhaustive k nearest references per query (float32 arithmetic of moss_knn_query: dx*dx + dy*dy + dz*dz, no fused
    multiply-add; ties -> lower index).  O(Nq Nr) memory in chunks; for small cases."""
    ref = np.asarray(ref, dtype=np.float32); query = np.asarray(query, dtype=np.float32)
    dist = np.empty((query.shape[0], k), dtype=np.float32); idx = np.empty((query.shape[0], k), dtype=np.int64)
    for s in range(0, query.shape[0], 2048):
        q = query[s:s + 2048]
        dx = ref[None, :, 0] - q[:, None, 0]; dy = ref[None, :, 1] - q[:, None, 1]; dz = ref[None, :, 2] - q[:, None, 2]
        d = (dx * dx + dy * dy) + dz * dz
        order = np.argsort(d, axis=1, kind="stable")[:, :k]
        idx[s:s + 2048] = order
        dist[s:s + 2048] = np.sqrt(np.take_along_axis(d, order, axis=1))
    return dist, idx


# ---------------------------------------------------------------------------------------------------------------------
# Per-Gaussian error scales (test infrastructure).  A gradient component of one Gaussian is a sum over pixels of terms that may
# cancel; an fp32 evaluation of it -- the reference's atomics, this repository's kernels, the float32 oracle -- carries an error
# proportional to the sum of the ABSOLUTE terms, whatever the order.  gradient_scales() returns that sum for every output element:
# the blend-level masses from oracle_render_backward_mass, pushed through each Gaussian's (linear) cov2D / projection / SH /
# cov3D backward by nine unit probes.  Tests state per-Gaussian bars as  |a - b| <= rel * scale  with these scales.

def gradient_scales(fw, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha, sh, degree, campos, transforms=None):
    """{name: float64 array shaped like the gradient} for the names of backward()'s namespace (dL_dconic excluded)."""
    assert not getattr(fw, "f64", False), "scales come from the float32 restatement"
    L = lib()
    P, H, W = fw.P, fw.H, fw.W
    bgf = _f32(bg)
    color_ptr = _opt(colors_precomp) if _opt(colors_precomp) is not None else fw.rgb
    mass = np.zeros((P, 9), np.float64)
    scratch = [np.zeros((P, 3), np.float32), np.zeros((P, 2, 2), np.float32), np.zeros((P, 1), np.float32), np.zeros((P, 3), np.float32)]
    L.oracle_render_backward_mass(C.c_int(P), C.c_int(W), C.c_int(H), _p(fw.ranges), _p(fw.point_list), _p(bgf),
                                  _p(fw.means2D), _p(fw.conic_opacity), _p(np.ascontiguousarray(color_ptr)), _p(fw.depths),
                                  _p(fw.final_T), _p(fw.n_contrib), _p(_f32(dL_dout_color).reshape(3, H, W)),
                                  _p(_f32(dL_dout_depth).reshape(H * W)), _p(_f32(dL_dout_alpha).reshape(H * W)),
                                  _p(scratch[0]), _p(scratch[1]), _p(scratch[2]), _p(scratch[3]), _p(mass))
    names = ["dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
    out = {"dL_dcolors": mass[:, 0:3].copy(), "dL_dmeans2D": np.concatenate([mass[:, 3:5], np.zeros((P, 1))], 1),
           "dL_dopacity": mass[:, 8:9].copy()}
    acc = None
    means3D32 = _f32(means3D); view = _f32(viewmatrix); proj = _f32(projmatrix); cam = _f32(campos)
    sc = _opt(scales); rt = _opt(rotations); c3 = _opt(cov3D_precomp); shs = _opt(sh)
    tf = None if transforms is None else np.ascontiguousarray(_f32(transforms).reshape(-1, 9))
    M = 0 if shs is None else shs.shape[1]
    cov3D_ptr = c3 if c3 is not None else fw.cov3D
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(tan_fovy)); focal_x = np.float32(W) / (np.float32(2.0) * np.float32(tan_fovx))
    for j in range(8):                                       # unit probes: colour 0-2, mean2D 3-4, conic 5-7 (opacity passes through)
        dcol = np.zeros((P, 3), np.float32); dm2 = np.zeros((P, 3), np.float32); dcon = np.zeros((P, 2, 2), np.float32)
        if j < 3: dcol[:, j] = 1.0
        elif j < 5: dm2[:, j - 3] = 1.0
        else: dcon.reshape(P, 4)[:, (0, 1, 3)[j - 5]] = 1.0
        g = {"dL_dmeans3D": np.zeros((P, 3), np.float32), "dL_dcov3D": np.zeros((P, 6), np.float32), "dL_dsh": np.zeros((P, M, 3), np.float32),
             "dL_dscales": np.zeros((P, 3), np.float32), "dL_drotations": np.zeros((P, 4), np.float32), "dL_dtransforms": np.zeros((P, 3, 3), np.float32)}
        L.oracle_compute_cov2d_backward(C.c_int(P), _p(means3D32), _p(fw.radii), _p(np.ascontiguousarray(cov3D_ptr)),
                                        C.c_float(focal_x), C.c_float(focal_y), C.c_float(tan_fovx), C.c_float(tan_fovy),
                                        _p(view), _p(dcon), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]))
        L.oracle_preprocess_backward(C.c_int(P), C.c_int(degree), C.c_int(M), _p(means3D32), _p(fw.radii), _p(shs),
                                     _p(fw.clamped), _p(sc), _p(rt), C.c_float(scale_modifier), _p(proj),
                                     _p(cam), _p(dm2), _p(g["dL_dmeans3D"]), _p(dcol), _p(g["dL_dcov3D"]),
                                     _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]), _p(tf), _p(g["dL_dtransforms"]))
        if acc is None:
            acc = {n: np.zeros(g[n].shape, np.float64) for n in names}
        mj = mass[:, j]
        for n in names:
            acc[n] += np.abs(g[n].astype(np.float64)) * mj.reshape((P,) + (1,) * (g[n].ndim - 1))
    out.update(acc)
    return out
