// torch_binding.cpp -- the PyTorch-ROCm extension module `moss_amd.lib._moss_C`: the compiled counterpart of the reference's
// `diff_gaussian_rasterization._C` (pybind11 exports DGR/ext.cpp:15-18; torch glue DGR/rasterize_points.cu:35-227,
// DGR/rasterize_points.h:18-70).  Same three callables, same positional arguments, same return tuples.
//
// Host code only: it allocates the outputs and the three opaque scratch tensors (grown through a callback exactly like
// resizeFunctional, rasterize_points.cu:27-33), unwraps every tensor to a raw device pointer and calls the C ABI of
// include/moss_raster.h (libmoss_raster.so) on the CURRENT torch HIP stream of the inputs' device.  No kernel lives here and no
// torch type crosses into the library.  The additions over the reference's signature are trailing and optional:
//   transforms (P,3,3)   per-Gaussian covariance transforms applied inside the op          (moss_raster_forward_tf / _backward_tf)
//   raw_flags            which of opacity / scales / rotations are raw parameters            (moss_raster_forward_raw / _backward_raw)
//   capacity             >= 0: asynchronous forward without the host read-back of num_rendered (moss_raster_forward_async)
//   sinks                caller-provided tensors the backward writes five of its gradients into (e.g. slices of a flat bucket)
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <string>
#include <tuple>
#include <vector>

#include "moss_raster.h"

namespace {

constexpr int NUM_CHANNELS = 3;   // DGR/cuda_rasterizer/config.h:14

thread_local std::string g_alloc_error;

char* grow(void* user, size_t nbytes)                     // resizeFunctional, rasterize_points.cu:27-33
{
    // an exception (torch's out-of-memory error) must not unwind through the C ABI: NULL makes the library return MOSS_ERR_ALLOC
    try {
        auto* t = static_cast<torch::Tensor*>(user);
        t->resize_({static_cast<int64_t>(nbytes)});
        return reinterpret_cast<char*>(t->data_ptr());
    } catch (const std::exception& e) {
        g_alloc_error = std::string("allocating ") + std::to_string(nbytes) + " bytes of rasterizer scratch failed (the binning buffer takes "
                        "~360 B per (Gaussian, tile) instance: 66 B of tables + 48 B per gradient-record cell, see INTEGRATION.md): " + e.what();
        return nullptr;
    }
}

[[noreturn]] void raise(int rc, const char* what)
{
    static const char* names[] = {"", "invalid argument", "HIP error", "allocation failed", "prefiltered point culled", "unsupported"};
    const int k = (-rc >= 1 && -rc <= 5) ? -rc : 0;
    if (rc == MOSS_ERR_ALLOC && !g_alloc_error.empty()) {
        const std::string msg = g_alloc_error;
        g_alloc_error.clear();
        TORCH_CHECK(false, what, ": ", names[k], ": ", msg);
    }
    TORCH_CHECK(false, what, ": ", names[k], ": ", moss_last_error());
}

// Raw device pointer of an optional tensor.  An EMPTY tensor means "absent" and maps to NULL
// (DGR/diff_gaussian_rasterization/__init__.py:200-210, rasterize_points.cu:96-108).  Contiguous copies are kept alive in `keep`.
template <typename T = float>
T* ptr(const torch::Tensor& t, const char* name, std::vector<torch::Tensor>& keep, c10::ScalarType dtype = torch::kFloat32)
{
    if (!t.defined() || t.numel() == 0) return nullptr;
    TORCH_CHECK(t.scalar_type() == dtype, name, ": expected ", dtype, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_cuda(), name, " must live on the GPU (got ", t.device(), "); this op has no CPU path");
    keep.push_back(t.contiguous());
    return reinterpret_cast<T*>(keep.back().data_ptr());
}
template <typename T = float>
T* ptr(const c10::optional<torch::Tensor>& t, const char* name, std::vector<torch::Tensor>& keep, c10::ScalarType dtype = torch::kFloat32)
{
    return t.has_value() ? ptr<T>(*t, name, keep, dtype) : nullptr;
}

// a gradient sink is used only if it has exactly the shape / dtype / device / layout the kernel writes
torch::Tensor out_or_sink(const c10::optional<torch::Tensor>& sink, at::IntArrayRef shape, const torch::TensorOptions& opts, bool zeros)
{
    if (sink.has_value() && sink->defined() && shape[0] != 0 && sink->sizes() == shape && sink->scalar_type() == torch::kFloat32 &&
        sink->device() == opts.device() && sink->is_contiguous())
        return *sink;
    return zeros ? torch::zeros(shape, opts) : torch::empty(shape, opts);
}

}  // namespace

// RasterizeGaussiansCUDA, rasterize_points.cu:35-119
std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations, double scale_modifier,
                    const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                    double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, const torch::Tensor& sh,
                    int64_t degree, const torch::Tensor& campos, bool prefiltered, int64_t debug /* bool in the reference; MOSS_DEBUG_* bits */,
                    const c10::optional<torch::Tensor>& transforms, int64_t raw_flags, int64_t capacity,
                    const c10::optional<torch::Tensor>& frame_state, const c10::optional<torch::Tensor>& translation)
{
    TORCH_CHECK(means3D.ndimension() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");   // :57-59
    TORCH_CHECK(means3D.is_cuda(), "means3D must live on the GPU; this op has no CPU path");
    const int P = static_cast<int>(means3D.size(0)), H = static_cast<int>(image_height), W = static_cast<int>(image_width);
    const bool has_tf = transforms.has_value() && transforms->defined();
    if (has_tf) {
        TORCH_CHECK(transforms->numel() == 9 * (int64_t)P && scales.numel() != 0 && rotations.numel() != 0 && cov3D_precomp.numel() == 0,
                    "transforms must be (P,3,3) and comes with scales and rotations (no cov3D_precomp)");
    }
    if (raw_flags)
        TORCH_CHECK(scales.numel() != 0 && rotations.numel() != 0 && cov3D_precomp.numel() == 0,
                    "raw_flags comes with scales and rotations (no cov3D_precomp)");
    const c10::DeviceGuard guard(means3D.device());
    const auto fopts = means3D.options().dtype(torch::kFloat32);
    // every element is written by the kernels (or memset by the library when P == 0): no zero-fill pass (the reference: torch::full x4)
    torch::Tensor out_color = torch::empty({NUM_CHANNELS, H, W}, fopts);
    torch::Tensor out_depth = torch::empty({1, H, W}, fopts);
    torch::Tensor out_alpha = torch::empty({1, H, W}, fopts);
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    const auto bopts = means3D.options().dtype(torch::kByte);
    torch::Tensor geom = torch::empty({0}, bopts), binning = torch::empty({0}, bopts), img = torch::empty({0}, bopts);

    const int M = sh.numel() != 0 ? static_cast<int>(sh.size(1)) : 0;                       // :85-89
    std::vector<torch::Tensor> keep;
    keep.reserve(16);
    void* stream = c10::hip::getCurrentHIPStream(means3D.device().index()).stream();
    const bool use_async = capacity >= 0 && !(debug & MOSS_DEBUG_SYNC) && P > 0;   // (the reference's debug flag synchronises after every launch)
    const float* p_bg = ptr(background, "background", keep);
    const float* p_means = ptr(means3D, "means3D", keep);
    const float* p_sh = ptr(sh, "sh", keep);
    const float* p_col = ptr(colors, "colors_precomp", keep);
    const float* p_opa = ptr(opacity, "opacity", keep);
    const float* p_scl = ptr(scales, "scales", keep);
    const float* p_rot = ptr(rotations, "rotations", keep);
    const float* p_cov = ptr(cov3D_precomp, "cov3D_precomp", keep);
    const float* p_tf = has_tf ? ptr(*transforms, "transforms", keep) : nullptr;
    const bool has_tl = translation.has_value() && translation->defined() && translation->numel() != 0;
    if (has_tl) TORCH_CHECK((raw_flags & MOSS_RAW_POSE) && translation->numel() == 3 * (int64_t)P, "translation must be (P,3) and comes with MOSS_RAW_POSE");
    const float* p_tl = has_tl ? ptr(*translation, "translation", keep) : nullptr;
    const float* p_view = ptr(viewmatrix, "viewmatrix", keep);
    const float* p_proj = ptr(projmatrix, "projmatrix", keep);
    const float* p_cam = ptr(campos, "campos", keep);
    int* p_radii = P ? reinterpret_cast<int*>(radii.data_ptr()) : nullptr;
    float *oc = reinterpret_cast<float*>(out_color.data_ptr()), *od = reinterpret_cast<float*>(out_depth.data_ptr()),
          *oa = reinterpret_cast<float*>(out_alpha.data_ptr());
    const int cap = use_async ? static_cast<int>(capacity) : -1;
    char* p_fs = nullptr;                                   // the caller's frame state: an argument of the asynchronous forwards (ABI 2)
    if (use_async && frame_state.has_value() && frame_state->defined()) {
        TORCH_CHECK(frame_state->is_cuda() && frame_state->is_contiguous() && frame_state->scalar_type() == torch::kByte &&
                    frame_state->device() == means3D.device() &&
                    (size_t)frame_state->numel() >= moss_raster_frame_state_bytes(W, H), "frame_state: a zero-initialised byte tensor of moss_raster_frame_state_bytes on the GPU");
        p_fs = reinterpret_cast<char*>(frame_state->data_ptr());
    }
    int rc;
    if (raw_flags)
        rc = moss_raster_forward_raw(grow, &geom, grow, &binning, grow, &img, P, (int)degree, M, p_bg, W, H, p_means, p_sh, p_col, p_opa,
                                     p_scl, (float)scale_modifier, p_rot, p_tf, p_tl, p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy,
                                     prefiltered ? 1 : 0, oc, od, oa, p_radii, (int)raw_flags, cap, p_fs, (int)debug, stream);
    else if (has_tf)
        rc = moss_raster_forward_tf(grow, &geom, grow, &binning, grow, &img, P, (int)degree, M, p_bg, W, H, p_means, p_sh, p_col, p_opa,
                                    p_scl, (float)scale_modifier, p_rot, p_tf, p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy,
                                    prefiltered ? 1 : 0, oc, od, oa, p_radii, cap, p_fs, (int)debug, stream);
    else if (use_async)
        rc = moss_raster_forward_async(grow, &geom, grow, &binning, grow, &img, P, (int)degree, M, p_bg, W, H, p_means, p_sh, p_col,
                                       p_opa, p_scl, (float)scale_modifier, p_rot, p_cov, p_view, p_proj, p_cam, (float)tan_fovx,
                                       (float)tan_fovy, prefiltered ? 1 : 0, oc, od, oa, p_radii, cap, p_fs, (int)debug, stream);
    else
        rc = moss_raster_forward(grow, &geom, grow, &binning, grow, &img, P, (int)degree, M, p_bg, W, H, p_means, p_sh, p_col, p_opa,
                                 p_scl, (float)scale_modifier, p_rot, p_cov, p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy,
                                 prefiltered ? 1 : 0, oc, od, oa, p_radii, (int)debug, stream);
    if (rc < 0) raise(rc, "rasterize_gaussians");
    return std::make_tuple((int64_t)rc, out_color, out_depth, out_alpha, radii, geom, binning, img);
}

// RasterizeGaussiansBackwardCUDA, rasterize_points.cu:121-206.  Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D,
// dL_dcov3D, dL_dsh, dL_dscales, dL_drotations) -- plus dL_dtransforms when `transforms` was given.
std::vector<torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                             const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                             double scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                             const torch::Tensor& projmatrix, double tan_fovx, double tan_fovy,
                             const c10::optional<torch::Tensor>& dL_dout_color, const c10::optional<torch::Tensor>& dL_dout_depth,
                             const c10::optional<torch::Tensor>& dL_dout_alpha, const torch::Tensor& sh, int64_t degree,
                             const torch::Tensor& campos, const torch::Tensor& geomBuffer, int64_t R, const torch::Tensor& binningBuffer,
                             const torch::Tensor& imageBuffer, const torch::Tensor& alphas, int64_t debug,
                             const c10::optional<torch::Tensor>& transforms, int64_t raw_flags, const c10::optional<torch::Tensor>& opacities,
                             const c10::optional<torch::Tensor>& sink_means3D, const c10::optional<torch::Tensor>& sink_opacity,
                             const c10::optional<torch::Tensor>& sink_sh, const c10::optional<torch::Tensor>& sink_scales,
                             const c10::optional<torch::Tensor>& sink_rotations, const c10::optional<torch::Tensor>& translation,
                             int64_t fused_adamw /* address of a host moss_fused_adamw the caller keeps alive, or 0 */,
                             bool all_outputs /* false: gradients nobody can receive are not computed into memory */)
{
    const int P = static_cast<int>(means3D.size(0));
    // The tensors whose AdamW update the backward kernel applies itself (moss_raster_backward_raw_adamw): their gradients stay inside
    // the kernel and come back as None.
    const moss_fused_adamw* opt = reinterpret_cast<const moss_fused_adamw*>(static_cast<intptr_t>(fused_adamw));
    const uint32_t fused = (opt != nullptr && P != 0) ? opt->tensors : 0u;
    TORCH_CHECK(!fused || raw_flags, "the fused AdamW update needs the raw-parameter backward (raw_flags)");
    const int H = static_cast<int>(alphas.size(-2)), W = static_cast<int>(alphas.size(-1));   // incoming gradients may be absent (= zeros)
    const int M = sh.numel() != 0 ? static_cast<int>(sh.size(1)) : 0;
    const bool has_tf = transforms.has_value() && transforms->defined();
    const c10::DeviceGuard guard(means3D.device());
    const auto fopts = means3D.options().dtype(torch::kFloat32);
    // The reference zero-fills nine tensors here (300 B per Gaussian, rasterize_points.cu:158-166); the HIP backward writes every
    // element exactly once, so plain allocations suffice.  P == 0 keeps the reference's zeros.
    const bool z = P == 0;
    auto mk = [&](at::IntArrayRef shape) { return z ? torch::zeros(shape, fopts) : torch::empty(shape, fopts); };
    torch::Tensor dL_dmeans3D = (fused & MOSS_OPT_MEANS) ? torch::Tensor() : out_or_sink(sink_means3D, {P, 3}, fopts, z);
    torch::Tensor dL_dmeans2D = mk({P, 3});
    // gradients nobody can receive are not computed into memory: dL_dcolors without colors_precomp, dL_dcov3D without cov3D_precomp
    // (their inputs are absent: autograd drops whatever is returned for them), dL_dconic always (the reference allocates and fills it,
    // rasterize_points.cu:160, and returns it to nobody)
    // (all_outputs = false, what the autograd function asks for)
    torch::Tensor dL_dcolors = (all_outputs || colors.numel() != 0 || z) ? mk({P, NUM_CHANNELS}) : torch::Tensor();
    torch::Tensor dL_dconic = all_outputs ? mk({P, 2, 2}) : torch::Tensor();
    torch::Tensor dL_dopacity = (fused & MOSS_OPT_OPACITY) ? torch::Tensor() : out_or_sink(sink_opacity, {P, 1}, fopts, z);
    torch::Tensor dL_dcov3D = (all_outputs || cov3D_precomp.numel() != 0 || z) ? mk({P, 6}) : torch::Tensor();
    torch::Tensor dL_dsh = (fused & MOSS_OPT_SH) ? torch::Tensor() : M != 0 ? out_or_sink(sink_sh, {P, M, 3}, fopts, z) : mk({P, M, 3});
    torch::Tensor dL_dscales = (fused & MOSS_OPT_SCALES) ? torch::Tensor() : out_or_sink(sink_scales, {P, 3}, fopts, z);
    torch::Tensor dL_drotations = (fused & MOSS_OPT_ROTATIONS) ? torch::Tensor() : out_or_sink(sink_rotations, {P, 4}, fopts, z);
    torch::Tensor dL_dtransforms = has_tf ? mk({P, 3, 3}) : torch::Tensor();
    const bool has_tl = translation.has_value() && translation->defined() && translation->numel() != 0;
    torch::Tensor dL_dtranslation = has_tl ? mk({P, 3}) : torch::Tensor();
    if (P != 0) {
        std::vector<torch::Tensor> keep;
        keep.reserve(20);
        void* stream = c10::hip::getCurrentHIPStream(means3D.device().index()).stream();
        auto f = [](torch::Tensor& t) { return t.defined() ? reinterpret_cast<float*>(t.data_ptr()) : nullptr; };
        const float* p_bg = ptr(background, "background", keep);
        const float* p_means = ptr(means3D, "means3D", keep);
        const float* p_sh = ptr(sh, "sh", keep);
        const float* p_col = ptr(colors, "colors_precomp", keep);
        const float* p_scl = ptr(scales, "scales", keep);
        const float* p_rot = ptr(rotations, "rotations", keep);
        const float* p_cov = ptr(cov3D_precomp, "cov3D_precomp", keep);
        const float* p_tf = has_tf ? ptr(*transforms, "transforms", keep) : nullptr;
        const float* p_view = ptr(viewmatrix, "viewmatrix", keep);
        const float* p_proj = ptr(projmatrix, "projmatrix", keep);
        const float* p_cam = ptr(campos, "campos", keep);
        char* p_geom = ptr<char>(geomBuffer, "geomBuffer", keep, torch::kByte);
        char* p_bin = ptr<char>(binningBuffer, "binningBuffer", keep, torch::kByte);
        char* p_img = ptr<char>(imageBuffer, "imageBuffer", keep, torch::kByte);
        const float* g_c = ptr(dL_dout_color, "dL_dout_color", keep);
        const float* g_d = ptr(dL_dout_depth, "dL_dout_depth", keep);
        const float* g_a = ptr(dL_dout_alpha, "dL_dout_alpha", keep);
        float* p_dsh = M ? f(dL_dsh) : nullptr;
        int rc;
        if (fused) {
            // the parameters are updated IN PLACE: they must be the caller's own contiguous float32 tensors, not copies made here
            TORCH_CHECK(opacities.has_value() && opacities->defined(), "the raw-parameter backward needs the raw opacities");
            for (const torch::Tensor* t : { &means3D, &sh, &*opacities, &scales, &rotations })
                TORCH_CHECK(t->is_contiguous() && t->scalar_type() == torch::kFloat32, "fused AdamW update: the parameters must be contiguous float32 tensors");
            rc = moss_raster_backward_raw_adamw(P, (int)degree, M, (int)R, p_bg, W, H, const_cast<float*>(p_means), const_cast<float*>(p_sh), p_col,
                                                const_cast<float*>(ptr(*opacities, "opacity", keep)), const_cast<float*>(p_scl),
                                                (float)scale_modifier, const_cast<float*>(p_rot), p_tf, has_tl ? ptr(*translation, "translation", keep) : nullptr,
                                                p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy, p_geom,
                                                p_bin, p_img, g_c, g_d, g_a, f(dL_dmeans2D), f(dL_dconic), f(dL_dopacity), f(dL_dcolors), f(dL_dmeans3D),
                                                f(dL_dcov3D), p_dsh, f(dL_dscales), f(dL_drotations), has_tf ? f(dL_dtransforms) : nullptr,
                                                has_tl ? f(dL_dtranslation) : nullptr, opt, (int)raw_flags, (int)debug, stream);
        } else if (raw_flags) {
            TORCH_CHECK(opacities.has_value() && opacities->defined(), "the raw-parameter backward needs the raw opacities");
            rc = moss_raster_backward_raw(P, (int)degree, M, (int)R, p_bg, W, H, p_means, p_sh, p_col, ptr(*opacities, "opacity", keep), p_scl,
                                          (float)scale_modifier, p_rot, p_tf, has_tl ? ptr(*translation, "translation", keep) : nullptr,
                                          p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy, p_geom,
                                          p_bin, p_img, g_c, g_d, g_a, f(dL_dmeans2D), f(dL_dconic), f(dL_dopacity), f(dL_dcolors), f(dL_dmeans3D),
                                          f(dL_dcov3D), p_dsh, f(dL_dscales), f(dL_drotations), has_tf ? f(dL_dtransforms) : nullptr,
                                          has_tl ? f(dL_dtranslation) : nullptr, (int)raw_flags, (int)debug, stream);
        } else if (has_tf) {
            rc = moss_raster_backward_tf(P, (int)degree, M, (int)R, p_bg, W, H, p_means, p_sh, p_col, p_scl, (float)scale_modifier, p_rot, p_tf,
                                         p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy, p_geom, p_bin, p_img, g_c, g_d, g_a,
                                         f(dL_dmeans2D), f(dL_dconic), f(dL_dopacity), f(dL_dcolors), f(dL_dmeans3D), f(dL_dcov3D), p_dsh,
                                         f(dL_dscales), f(dL_drotations), f(dL_dtransforms), (int)debug, stream);
        } else {
            rc = moss_raster_backward(P, (int)degree, M, (int)R, p_bg, W, H, p_means, p_sh, p_col, ptr(alphas, "alphas", keep), p_scl,
                                      (float)scale_modifier, p_rot, p_cov, p_view, p_proj, p_cam, (float)tan_fovx, (float)tan_fovy,
                                      ptr<int>(radii, "radii", keep, torch::kInt32), p_geom, p_bin, p_img, g_c, g_d, g_a, f(dL_dmeans2D),
                                      f(dL_dconic), f(dL_dopacity), f(dL_dcolors), f(dL_dmeans3D), f(dL_dcov3D), p_dsh, f(dL_dscales),
                                      f(dL_drotations), (int)debug, stream);
        }
        if (rc < 0) raise(rc, "rasterize_gaussians_backward");
    }
    std::vector<torch::Tensor> res = {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations};
    if (has_tf) res.push_back(dL_dtransforms);
    if (has_tl) res.push_back(dL_dtranslation);                // (only with MOSS_RAW_POSE, which needs transforms: always the tenth)
    return res;
}

// markVisible, rasterize_points.cu:208-227: bool (P,), True where z_view > 0.2
torch::Tensor mark_visible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix)
{
    TORCH_CHECK(means3D.is_cuda(), "means3D must live on the GPU; this op has no CPU path");
    const int P = static_cast<int>(means3D.size(0));
    const c10::DeviceGuard guard(means3D.device());
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        std::vector<torch::Tensor> keep;
        const int rc = moss_raster_mark_visible(P, ptr(means3D, "means3D", keep), ptr(viewmatrix, "viewmatrix", keep),
                                                ptr(projmatrix, "projmatrix", keep), reinterpret_cast<uint8_t*>(present.data_ptr()),
                                                c10::hip::getCurrentHIPStream(means3D.device().index()).stream());
        if (rc < 0) raise(rc, "mark_visible");
    }
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    namespace py = pybind11;
    m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"), py::arg("opacity"),
          py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"),
          py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"),
          py::arg("degree"), py::arg("campos"), py::arg("prefiltered"), py::arg("debug"), py::arg("transforms") = py::none(),
          py::arg("raw_flags") = 0, py::arg("capacity") = -1, py::arg("frame_state") = py::none(), py::arg("translation") = py::none());
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"), py::arg("radii"),
          py::arg("colors"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"),
          py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"),
          py::arg("dL_dout_depth"), py::arg("dL_dout_alpha"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"),
          py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("alphas"), py::arg("debug"),
          py::arg("transforms") = py::none(), py::arg("raw_flags") = 0, py::arg("opacities") = py::none(),
          py::arg("sink_means3D") = py::none(), py::arg("sink_opacity") = py::none(), py::arg("sink_sh") = py::none(),
          py::arg("sink_scales") = py::none(), py::arg("sink_rotations") = py::none(), py::arg("translation") = py::none(),
          py::arg("fused_adamw") = 0, py::arg("all_outputs") = true);
    m.def("mark_visible", &mark_visible);
    // the version of the header THIS module was compiled against (not the library's answer: moss_amd/_lib.py compares the two, so a
    // stale _moss_C.so next to a rebuilt libmoss_raster.so refuses to load instead of passing arguments in the old layout)
    m.def("abi_version", []() { return (int)MOSS_ABI_VERSION; });
}
