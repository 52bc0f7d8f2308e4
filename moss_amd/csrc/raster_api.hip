// raster_api.hip -- the extern "C" entry points declared in include/moss_raster.h (host orchestration only).
// Stage order follows CudaRasterizer::Rasterizer::forward / ::backward
// (DGR/cuda_rasterizer/rasterizer_impl.cu:198-341, :345-447); the stages themselves are this library's own.
#include "common.h"
#include "adamw.h"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>

using namespace moss;

namespace moss { thread_local StageEvents g_stage_events; }

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MOSS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// CHECK_CUDA equivalent (auxiliary.h:166-173): with debug, synchronise and surface errors after every launch.
#define STAGE_CHECK(name)                                                                              \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(s);                                   \
        if (e_ != hipSuccess) return fail(MOSS_ERR_HIP, "stage '%s' failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-stage timing with HIP events recorded on the launch stream (moss_raster_profile_*) ----------
struct ProfRec { int stage; hipEvent_t a, b; };
struct Profiler {
    std::mutex m;
    uint32_t mask = 0;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Profiler g_prof;

struct StageTimer {
    int stage; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    StageTimer(int stage_, hipStream_t s_) : stage(stage_), s(s_)
    {
        if (g_prof.mask & (1u << stage)) {
            std::lock_guard<std::mutex> lk(g_prof.m);
            a = g_prof.get(); b = g_prof.get();
            if (a) (void)hipEventRecord(a, s);
            // single-kernel stages: the launcher may attach the two events to the kernel itself (MOSS_LAUNCH_TIMED)
            if (a && b) { g_stage_events.start = a; g_stage_events.stop = b; g_stage_events.used = false; }
        }
    }
    ~StageTimer()
    {
        if (a && b) {
            if (!g_stage_events.used) (void)hipEventRecord(b, s);      // multi-kernel stage, or nothing was launched
            g_stage_events = StageEvents();
            std::lock_guard<std::mutex> lk(g_prof.m);
            g_prof.recs.push_back({stage, a, b});
        }
    }
};

// ---- MOSS_DEBUG_TRACE: roctx ranges around the stage launchers (the reference has torch.cuda.Event pairs around its
// whole step only, train_ZJU.py:43-44).  rocprofv3 --marker-trace reads the ranges of the rocprofiler-sdk roctx library; it is
// resolved with dlopen at the first traced call (no link-time dependency; the older libroctx64 as a fallback), and without the debug
// bit nothing of this is touched.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        for (const char* name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
struct TraceRange {
    bool on;
    TraceRange(bool enabled, const char* name) : on(enabled && roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~TraceRange() { if (on) (void)roctx().pop(); }
};

// Small pinned host block for the header read-back, one per host thread.
struct Pinned {
    uint32_t* p = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); }
};
thread_local Pinned g_pinned;

FrameParams make_params(int P, int D, int M, int W, int H, float tan_fovx, float tan_fovy, float scale_modifier, int prefiltered,
                        const float* view, const float* proj, const float* campos, const float* bg)
{
    FrameParams fp;
    fp.P = P; fp.D = D; fp.M = M; fp.W = W; fp.H = H;
    fp.gx = (W + TILE - 1) / TILE; fp.gy = (H + TILE - 1) / TILE;
    fp.tan_fovx = tan_fovx; fp.tan_fovy = tan_fovy;
    fp.focal_y = H / (2.0f * tan_fovy);         // rasterizer_impl.cu:224-225
    fp.focal_x = W / (2.0f * tan_fovx);
    fp.scale_modifier = scale_modifier; fp.prefiltered = prefiltered; fp.raw = 0; fp.no_block_cull = 0; fp.exact_math = 0; fp.forward_only = 0;
    fp.view_dev = view; fp.proj_dev = proj; fp.campos_dev = campos; fp.bg_dev = bg;
    return fp;
}

}  // namespace

static int forward_impl(
    moss_alloc_fn geometry_alloc, void* geometry_user,
    moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int debug_flags, void* stream, long long capacity,
    const float* transforms, int raw_flags = 0, char* frame_state = nullptr, const float* translation = nullptr)
{
    g_err[0] = 0;
    hipStream_t s = (hipStream_t)stream;
    const int debug = debug_flags & MOSS_DEBUG_SYNC;
    if (P < 0 || width <= 0 || height <= 0) return fail(MOSS_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (!out_color || !out_depth || !out_alpha || !background) return fail(MOSS_ERR_INVALID_ARG, "null output/background pointer");
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(MOSS_ERR_INVALID_ARG, "null allocator callback");
    if ((width + TILE - 1) / TILE > 65535 || (height + TILE - 1) / TILE > 65535) return fail(MOSS_ERR_UNSUPPORTED, "image too large");
    const size_t N = (size_t)width * height;

    if (P == 0) {
        // rasterize_points.cu:68-83: zero-filled outputs, nothing else happens.  (Outputs here are caller-allocated
        // and possibly uninitialised, so fill them; the background is NOT composited in the reference either.)
        // (with kernels, not hipMemsetAsync: memset nodes did not re-execute on hipGraph replay with ROCm 7.2, see launch_clear)
        launch_zero_floats(out_color, 3 * N, s);
        launch_zero_floats(out_depth, N, s);
        launch_zero_floats(out_alpha, N, s);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos) return fail(MOSS_ERR_INVALID_ARG, "null required input");
    if (!colors_precomp && !shs) return fail(MOSS_ERR_UNSUPPORTED, "provide SHs or precomputed colours");
    if (!colors_precomp && (M <= 0 || (D + 1) * (D + 1) > M || D < 0 || D > 3)) return fail(MOSS_ERR_INVALID_ARG, "SH degree %d does not fit M=%d", D, M);
    if (!cov3D_precomp && (!scales || !rotations)) return fail(MOSS_ERR_INVALID_ARG, "provide scales+rotations or cov3D_precomp");

    char* geom_ptr = geometry_alloc(geometry_user, GeomView::bytes(P));
    if (!geom_ptr) return fail(MOSS_ERR_ALLOC, "geometry allocator returned NULL");
    char* img_ptr = image_alloc(image_user, ImageView::bytes(width, height));
    if (!img_ptr) return fail(MOSS_ERR_ALLOC, "image allocator returned NULL");
    GeomView g = GeomView::at(geom_ptr, P);
    ImageView im = ImageView::at(img_ptr, width, height);
    FrameParams fp = make_params(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier, prefiltered,
                                 viewmatrix, projmatrix, cam_pos, background);
    fp.raw = cov3D_precomp ? (raw_flags & RAW_OPACITY) : raw_flags;     // scales / rotations are not read with a precomputed covariance
    fp.no_block_cull = (debug_flags & MOSS_DEBUG_NO_BLOCK_CULL) ? 1 : 0;
    fp.exact_math = (debug_flags & MOSS_DEBUG_EXACT_MATH) ? 1 : 0;
    // MOSS_FORWARD_ONLY: an evaluation render (render_ZJU.py:56-72) -- the caller promises that no backward follows.  Same images, bit for
    // bit; no depth-segment state, no gradient-record cells, no validity bits, and a binning buffer of 62 B per instance (ids, block
    // masks, records, sort keys) instead of ~370.  The keys then take the scan -> scatter chain into exact ranges (the per-tile buckets
    // of the training forward live in the record pool's address space, which this buffer does not have).
    const bool fwd_only = (debug_flags & MOSS_FORWARD_ONLY) != 0;
    fp.forward_only = fwd_only ? 1 : 0;
    const bool trace = (debug_flags & MOSS_DEBUG_TRACE) != 0;
    const int T = fp.gx * fp.gy;

    // The counters kernels ADD to (tile histogram, tile cursors, error flags) must be zero here.  With the caller's frame state
    // (the `frame_state` argument: all-zero between calls, re-zeroed by the sort / merge kernel) nothing is launched for that; without it, a clear.
    const size_t fs_bytes = ImageView::frame_state_bytes(width, height);
    if (frame_state) im.use_frame_state(frame_state, width, height);
    else launch_clear(im.header, im.clear_bytes(), s);                   // header + tile histogram + tile cursors
    // (a forward that ends before its sort kernel has run leaves the frame state dirty: clean it on those paths)
    auto abandon_frame_state = [&]() { if (frame_state) clear_frame_state(frame_state, fs_bytes, s); };
    // Asynchronous forward with the tile histogram in LDS: the PREPROCESS kernel writes the sort keys itself, into per-tile buckets of the
    // key area (preprocess.hip, scatter mode) -- the binning buffer is sized for the caller's capacity, so it exists before the first
    // kernel -- the sort workgroups derive their chunk tables from the tile counts, and the scan rides along with the sort kernel as one
    // extra block: preprocess, sort, merge, blend = FOUR launches (rounds 2-4: five, with a scatter kernel of its own; the reference: 7+).
    // (a capacity so small that a bucket would hold no key at all keeps the scan -> scatter chain, whose only bound is the total)
    uint32_t key_stride = (capacity > 0 && !fwd_only && forward_buckets_keys(fp)) ? bucket_key_stride(BinView::at(nullptr, (int)capacity), T) : 0u;
    if (key_stride < 2u) key_stride = 0u;
    const bool bucketed = key_stride != 0u;
    int R = 0, total_chunks = 0;
    long long pool_cells = -1;                               // (-1: the default pool of a buffer for R instances, BinView::default_pool_cells)
    char* bin_ptr = nullptr;
    BinView b;
    if (bucketed) {
        R = (int)capacity;
        total_chunks = (int)(capacity / 1024) + T;
        bin_ptr = binning_alloc(binning_user, BinView::bytes(R));
        if (!bin_ptr) { abandon_frame_state(); return fail(MOSS_ERR_ALLOC, "binning allocator returned NULL"); }
        b = BinView::at(bin_ptr, R);
    }
    { StageTimer tm(MOSS_STAGE_PREPROCESS_FWD, s); TraceRange tr(trace, bucketed ? "moss:preprocess_fwd+scatter" : "moss:preprocess_fwd");
      launch_preprocess_forward(fp, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, transforms, translation, g, im, radii, s,
                                bucketed ? b.keys : nullptr, key_stride); }
    STAGE_CHECK("preprocess");
    // (asynchronous and un-bucketed -- the MOSS_FORWARD_ONLY renders: the scan rides along with the scatter kernel, no launch of its own)
    const bool fold_scan = !bucketed && capacity >= 0 && scatter_folds_scan(fp);
    if (!bucketed) {
        if (!fold_scan) {
            { StageTimer tm(MOSS_STAGE_SCAN, s); TraceRange tr(trace, "moss:scan"); launch_scan(P, g, im, T, capacity, s, fwd_only); }
            STAGE_CHECK("scan");
        }
        if (capacity < 0) {
            // The one host round trip of the forward pass: R sizes the binning buffer (rasterizer_impl.cu:283).
            if (!g_pinned.p) HIP_TRY(hipHostMalloc((void**)&g_pinned.p, 64, hipHostMallocDefault));
            // ... and the frame's gradient-record cells (header[9]) size the record pool at the buffer's end exactly
            HIP_TRY(hipMemcpyAsync(g_pinned.p, im.header, 40, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            R = (int)g_pinned.p[0];
            total_chunks = (int)g_pinned.p[4];
            pool_cells = (long long)g_pinned.p[9];
            if (g_pinned.p[2] & ERRFLAG_PREFILTERED) {
                abandon_frame_state();
                return fail(MOSS_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
            }
            if (g_pinned.p[2] & ERRFLAG_OVERFLOW) {                      // (the only overflow a synchronous forward knows)
                abandon_frame_state();
                return fail(MOSS_ERR_UNSUPPORTED, "the frame needs more than 80M gradient-record cells (4x4 pixel blocks under the Gaussians' "
                                                  "bounding boxes): the record pool is addressed with 32-bit byte offsets");
            }
        } else {
            // Asynchronous: no read-back.  Buffers and grids are sized for the caller's capacity; kernels bound themselves with the
            // device-side R; a frame that needs more renders nothing and sets the overflow flag (moss_raster_read_status).
            R = (int)capacity;
            total_chunks = (int)(capacity / 1024) + T;
        }
        bin_ptr = binning_alloc(binning_user, BinView::bytes(R, pool_cells, fwd_only));
        if (!bin_ptr) { abandon_frame_state(); return fail(MOSS_ERR_ALLOC, "binning allocator returned NULL"); }
        b = BinView::at(bin_ptr, R, pool_cells, fwd_only);
    }

    if (R > 0) {
        if (!bucketed) {
            { StageTimer tm(MOSS_STAGE_SCATTER, s); TraceRange tr(trace, fold_scan ? "moss:scatter+scan" : "moss:scatter"); launch_scatter(fp, g, im, b, s, fold_scan, capacity); }
            STAGE_CHECK("scatter");
        }
#ifdef MOSS_DIAG
        {   // timing experiment (scripts/exp_atomics.py): with the scatter's reservation atomics off the keys are garbage -- stop here
            static const bool stop = (knob("MOSS_EXPERIMENT", 0) & 2) != 0;
            if (stop) { abandon_frame_state(); return R; }
        }
#endif
        { StageTimer tm(MOSS_STAGE_TILE_SORT, s); TraceRange tr(trace, bucketed ? "moss:chunk_sort+scan" : "moss:chunk_sort");
          launch_tile_sort(fp, g, im, b, R, total_chunks, s, frame_state, fs_bytes, 0, key_stride, capacity); }
        { StageTimer tm(MOSS_STAGE_MERGE_GATHER, s); TraceRange tr(trace, "moss:merge_gather");
          launch_tile_sort(fp, g, im, b, R, total_chunks, s, frame_state, fs_bytes, 1, key_stride, capacity); }
        STAGE_CHECK("tile_sort");
    } else abandon_frame_state();                                        // (nothing rendered: no sort kernel to re-zero it)
    { StageTimer tm(MOSS_STAGE_BLEND_FWD, s); TraceRange tr(trace, "moss:blend_fwd"); launch_blend_forward(fp, g, im, b, out_color, out_depth, out_alpha, s); }
    STAGE_CHECK("blend_forward");
    return R;
}

extern "C" {

int moss_raster_forward(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int debug, void* stream)
{
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background,
                        width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii,
                        debug, stream, -1, nullptr);
}

int moss_raster_forward_async(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int capacity, char* frame_state, int debug, void* stream)
{
    if (capacity < 0) return fail(MOSS_ERR_INVALID_ARG, "capacity must be >= 0");
    if (debug & MOSS_DEBUG_SYNC) return fail(MOSS_ERR_INVALID_ARG, "MOSS_DEBUG_SYNC needs the synchronous forward");
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background,
                        width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii,
                        debug, stream, capacity, nullptr, 0, frame_state);
}

// n2 extension: like moss_raster_forward / _async (capacity < 0: synchronous, debug off) with a per-Gaussian 3x3
// transform applied to the scale/rotation covariance inside the op.
int moss_raster_forward_tf(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int capacity, char* frame_state, int debug, void* stream)
{
    if (P > 0 && (!scales || !rotations || !transforms)) return fail(MOSS_ERR_INVALID_ARG, "scales, rotations and transforms are required");
    if (capacity >= 0 && (debug & MOSS_DEBUG_SYNC)) return fail(MOSS_ERR_INVALID_ARG, "MOSS_DEBUG_SYNC needs the synchronous forward (capacity < 0)");
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background,
                        width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, nullptr,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii,
                        debug, stream, capacity < 0 ? -1 : capacity, transforms, 0, frame_state);
}

int moss_raster_read_status(const char* image_buffer, uint32_t* host_pinned_out /* 8 words */, void* stream)
{
    // enqueue a copy of {R rendered, longest list, error flags, -, chunks, non-empty tiles, instances needed, -} to host memory
    if (!image_buffer || !host_pinned_out) return fail(MOSS_ERR_INVALID_ARG, "null pointer");
    hipError_t e = hipMemcpyAsync(host_pinned_out, image_buffer, 32, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail(MOSS_ERR_HIP, "status copy failed: %s", hipGetErrorString(e));
}

size_t moss_raster_frame_state_bytes(int width, int height) { return ImageView::frame_state_bytes(width, height); }

int moss_abi_version(void) { return MOSS_ABI_VERSION; }
const char* moss_last_error(void) { return g_err; }

size_t moss_raster_geometry_bytes(int P) { return GeomView::bytes(P > 0 ? P : 1); }
size_t moss_raster_image_bytes(int width, int height) { return ImageView::bytes(width, height); }
size_t moss_raster_binning_bytes(int R) { return BinView::bytes(R); }
size_t moss_raster_binning_bytes_forward_only(int R) { return BinView::bytes(R, -1, true); }

static int backward_impl(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug_flags, void* stream,
    const float* transforms, float* dL_dtransforms, const float* opacities = nullptr, int raw_flags = 0,
    const float* translation = nullptr, float* dL_dtranslation = nullptr, const moss_fused_adamw* opt = nullptr)
{
    (void)alphas; (void)radii;
    const int debug = debug_flags & MOSS_DEBUG_SYNC;
    g_err[0] = 0;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(MOSS_ERR_INVALID_ARG, "bad sizes");
    if (P == 0) return 0;                                                 // rasterize_points.cu:168
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(MOSS_ERR_INVALID_ARG, "null scratch buffer");
    if (!dL_dpix && !dL_ddepths && !dL_dalphas) return fail(MOSS_ERR_INVALID_ARG, "all three incoming gradients are NULL");
    // (the gradient of a tensor whose AdamW update this call applies itself may stay inside the kernel)
    const uint32_t fused = opt ? opt->tensors : 0u;
    // dL_dconic (an intermediate the reference also exposes), dL_dcolor and dL_dcov3D (gradients of the OPTIONAL inputs colors_precomp /
    // cov3D_precomp) may be NULL = not wanted: 52 bytes per Gaussian that a caller working from SH and scales / rotations never reads
    if (!dL_dmean2D || (!dL_dopacity && !(fused & OPT_OPACITY)) || (!dL_dmean3D && !(fused & OPT_MEANS)) ||
        (!dL_dscale && !(fused & OPT_SCALES)) || (!dL_drot && !(fused & OPT_ROTATIONS)))
        return fail(MOSS_ERR_INVALID_ARG, "null gradient output");
    if (shs && !dL_dsh && !(fused & OPT_SH)) return fail(MOSS_ERR_INVALID_ARG, "dL_dsh is NULL although shs is given");
    FusedAdam fa;
    if (fused) {
        if (fused & ~(OPT_MEANS | OPT_SH | OPT_OPACITY | OPT_SCALES | OPT_ROTATIONS)) return fail(MOSS_ERR_INVALID_ARG, "unknown bits in moss_fused_adamw.tensors");
        if ((raw_flags & (RAW_OPACITY | RAW_SCALE | RAW_ROTATION)) != (RAW_OPACITY | RAW_SCALE | RAW_ROTATION) || cov3D_precomp)
            return fail(MOSS_ERR_INVALID_ARG, "the fused AdamW update works on the raw parameters: MOSS_RAW_OPACITY | MOSS_RAW_SCALE | MOSS_RAW_ROTATION");
        if ((fused & OPT_MEANS) && transforms && !(raw_flags & RAW_POSE))
            return fail(MOSS_ERR_INVALID_ARG, "MOSS_OPT_MEANS: with transforms the position parameter is only what the op sees under MOSS_RAW_POSE");
        if (!opt->step_state) return fail(MOSS_ERR_INVALID_ARG, "moss_fused_adamw.step_state is NULL");
        const float* params[5] = { means3D, shs, opacities, scales, rotations };
        for (int i = 0; i < 5; i++) {
            if (!(fused & (1u << i))) continue;
            if (!params[i] || !opt->exp_avg[i] || !opt->exp_avg_sq[i]) return fail(MOSS_ERR_INVALID_ARG, "a tensor named in moss_fused_adamw.tensors has a NULL parameter or moment array");
            fa.p[i] = const_cast<float*>(params[i]); fa.m[i] = opt->exp_avg[i]; fa.v[i] = opt->exp_avg_sq[i]; fa.lr[i] = opt->lr[i];
            if (opt->lr_segment[i] < -1 || opt->lr_segment[i] > 7) return fail(MOSS_ERR_INVALID_ARG, "moss_fused_adamw.lr_segment must be -1 or 0..7");
            fa.lr_segment[i] = opt->lr_segment[i];
        }
        if (fused & OPT_SH) {
            if (M != 16 || ((reinterpret_cast<uintptr_t>(shs) | reinterpret_cast<uintptr_t>(opt->exp_avg[1]) | reinterpret_cast<uintptr_t>(opt->exp_avg_sq[1]) |
                             reinterpret_cast<uintptr_t>(dL_dsh)) & 15u))
                return fail(MOSS_ERR_INVALID_ARG, "MOSS_OPT_SH needs M == 16 and 16-byte aligned SH, moment and gradient arrays");
        }
        if ((fused & OPT_ROTATIONS) && ((reinterpret_cast<uintptr_t>(rotations) | reinterpret_cast<uintptr_t>(opt->exp_avg[4]) | reinterpret_cast<uintptr_t>(opt->exp_avg_sq[4])) & 15u))
            return fail(MOSS_ERR_INVALID_ARG, "MOSS_OPT_ROTATIONS needs 16-byte aligned rotation and moment arrays");
        {   // degree-aware SH update: float4 parts of a record that hold an ever-active coefficient (the call's own degree is the floor)
            const int da = std::max(std::min((int)opt->sh_active_degree, 3), std::max(std::min(D, 3), 0));
            // (eps = 0 would make the full update of an all-zero element 0 x rcp(0) = NaN: no shortcut then)
            fa.sh_active_parts = opt->eps > 0.0f ? (3 * (da + 1) * (da + 1) + 3) / 4 : 12;
            fa.sh_inactive_zero = (opt->sh_inactive_zero != 0 && fa.sh_active_parts < 12) ? 1 : 0;
        }
        fa.tensors = fused; fa.lr_sh_rest = opt->lr_sh_rest;
        fa.betas = AdamBetas(opt->beta1, opt->beta2); fa.eps = opt->eps; fa.weight_decay = opt->weight_decay;
        fa.step_state = reinterpret_cast<const float*>(opt->step_state);
    }
    if (!means3D || !viewmatrix || !projmatrix || !campos || !background) return fail(MOSS_ERR_INVALID_ARG, "null required input");

    GeomView g = GeomView::at(geom_buffer, P);
    ImageView im = ImageView::at(image_buffer, width, height);
    BinView b = BinView::at(binning_buffer, R);
    FrameParams fp = make_params(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier, 0,
                                 viewmatrix, projmatrix, campos, background);
    fp.raw = cov3D_precomp ? (raw_flags & RAW_OPACITY) : raw_flags;
    fp.no_block_cull = (debug_flags & MOSS_DEBUG_NO_BLOCK_CULL) ? 1 : 0;
    fp.exact_math = (debug_flags & MOSS_DEBUG_EXACT_MATH) ? 1 : 0;
    const bool trace = (debug_flags & MOSS_DEBUG_TRACE) != 0;
    if ((fp.raw & RAW_OPACITY) && !opacities) return fail(MOSS_ERR_INVALID_ARG, "raw opacities are required to chain through the sigmoid");
    TraceRange tr_all(trace, "moss:raster_backward");
    if (R > 0) {
        { StageTimer tm(MOSS_STAGE_BLEND_BWD, s); TraceRange tr(trace, "moss:blend_bwd"); launch_blend_backward(fp, g, im, b, dL_dpix, dL_ddepths, dL_dalphas, s); }
        STAGE_CHECK("blend_backward");
    }
    { StageTimer tm(MOSS_STAGE_PREPROCESS_BWD, s); TraceRange tr(trace, fused ? "moss:preprocess_bwd+adamw" : "moss:preprocess_bwd");
      launch_preprocess_backward(fp, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, b, im.header, im.queues,
                                 dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                 transforms, dL_dtransforms, translation, dL_dtranslation, s, fused ? &fa : nullptr); }
    STAGE_CHECK("preprocess_backward");
    return 0;
}

int moss_raster_backward(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                         image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, nullptr, nullptr);
}

// n2 extension: backward of moss_raster_forward_tf; dL_dtransforms (P,9) is written for every Gaussian (zeros if culled).
// dL_dcov3D is the gradient w.r.t. the TRANSFORMED covariance (as stored), dL_dscale / dL_drot already include the transform.
int moss_raster_backward_tf(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtransforms, int debug, void* stream)
{
    if (P > 0 && (!scales || !rotations || !transforms || !dL_dtransforms))
        return fail(MOSS_ERR_INVALID_ARG, "scales, rotations, transforms and dL_dtransforms are required");
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, nullptr, scales, scale_modifier, rotations,
                         nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, nullptr, geom_buffer, binning_buffer,
                         image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, transforms, dL_dtransforms);
}

// Raw-parameter variant (the getters of GaussianModel applied inside preprocess, see include/moss_raster.h).
int moss_raster_forward_raw(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms, const float* translation,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int raw_flags, int capacity, char* frame_state, int debug, void* stream)
{
    if (raw_flags & ~(RAW_OPACITY | RAW_SCALE | RAW_ROTATION | HINT_SPATIAL_ORDER | RAW_POSE)) return fail(MOSS_ERR_INVALID_ARG, "unknown raw_flags bits");
    if (P > 0 && (!scales || !rotations)) return fail(MOSS_ERR_INVALID_ARG, "scales and rotations are required");
    if (P > 0 && (raw_flags & RAW_POSE) && !transforms) return fail(MOSS_ERR_INVALID_ARG, "MOSS_RAW_POSE needs the transforms");
    if (translation && !(raw_flags & RAW_POSE)) return fail(MOSS_ERR_INVALID_ARG, "a translation comes with MOSS_RAW_POSE");
    if (capacity >= 0 && (debug & MOSS_DEBUG_SYNC)) return fail(MOSS_ERR_INVALID_ARG, "MOSS_DEBUG_SYNC needs the synchronous forward (capacity < 0)");
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background,
                        width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, nullptr,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii,
                        debug, stream, capacity < 0 ? -1 : capacity, transforms, raw_flags, frame_state, translation);
}

int moss_raster_backward_raw(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms, const float* translation,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtransforms, float* dL_dtranslation,
    int raw_flags, int debug, void* stream)
{
    if (raw_flags & ~(RAW_OPACITY | RAW_SCALE | RAW_ROTATION | HINT_SPATIAL_ORDER | RAW_POSE | SH_GRAD_ACTIVE_ONLY)) return fail(MOSS_ERR_INVALID_ARG, "unknown raw_flags bits");
    if (P > 0 && (raw_flags & RAW_POSE) && !transforms) return fail(MOSS_ERR_INVALID_ARG, "MOSS_RAW_POSE needs the transforms");
    if ((translation || dL_dtranslation) && !(raw_flags & RAW_POSE)) return fail(MOSS_ERR_INVALID_ARG, "a translation comes with MOSS_RAW_POSE");
    if (P > 0 && (!scales || !rotations || (transforms && !dL_dtransforms)))
        return fail(MOSS_ERR_INVALID_ARG, "scales and rotations (and dL_dtransforms with transforms) are required");
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, nullptr, scales, scale_modifier, rotations,
                         nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, nullptr, geom_buffer, binning_buffer,
                         image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, transforms, transforms ? dL_dtransforms : nullptr, opacities,
                         raw_flags, translation, dL_dtranslation);
}

// The raw-parameter backward that also takes the AdamW step of the parameters named in opt->tensors (include/moss_raster.h).
int moss_raster_backward_raw_adamw(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    float* means3D, float* shs, const float* colors_precomp, float* opacities,
    float* scales, float scale_modifier, float* rotations, const float* transforms, const float* translation,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtransforms, float* dL_dtranslation,
    const moss_fused_adamw* opt, int raw_flags, int debug, void* stream)
{
    if (raw_flags & ~(RAW_OPACITY | RAW_SCALE | RAW_ROTATION | HINT_SPATIAL_ORDER | RAW_POSE | SH_GRAD_ACTIVE_ONLY)) return fail(MOSS_ERR_INVALID_ARG, "unknown raw_flags bits");
    if (P > 0 && (raw_flags & RAW_POSE) && !transforms) return fail(MOSS_ERR_INVALID_ARG, "MOSS_RAW_POSE needs the transforms");
    if ((translation || dL_dtranslation) && !(raw_flags & RAW_POSE)) return fail(MOSS_ERR_INVALID_ARG, "a translation comes with MOSS_RAW_POSE");
    if (P > 0 && (!scales || !rotations || (transforms && !dL_dtransforms)))
        return fail(MOSS_ERR_INVALID_ARG, "scales and rotations (and dL_dtransforms with transforms) are required");
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, nullptr, scales, scale_modifier, rotations,
                         nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, nullptr, geom_buffer, binning_buffer,
                         image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, transforms, transforms ? dL_dtransforms : nullptr, opacities,
                         raw_flags, translation, dL_dtranslation, (opt && opt->tensors) ? opt : nullptr);
}

int moss_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                             uint8_t* present, void* stream)
{
    (void)projmatrix;
    g_err[0] = 0;
    if (P < 0) return fail(MOSS_ERR_INVALID_ARG, "bad P");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(MOSS_ERR_INVALID_ARG, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    launch_mark_visible(P, means3D, viewmatrix, present, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOSS_ERR_HIP, "mark_visible launch failed: %s", hipGetErrorString(e));
    return 0;
}

int moss_raster_export_geometry(const char* geom_buffer, int P,
    float* depths, float* means2D, float* conic_opacity, float* rgb, uint32_t* tiles_touched,
    uint8_t* clamped, float* cov3D, void* stream)
{
    g_err[0] = 0;
    if (P <= 0 || !geom_buffer) return fail(MOSS_ERR_INVALID_ARG, "bad arguments");
    GeomView g = GeomView::at(const_cast<char*>(geom_buffer), P);
    launch_export_geometry(P, g, depths, means2D, conic_opacity, rgb, tiles_touched, clamped, cov3D, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOSS_ERR_HIP, "export_geometry launch failed: %s", hipGetErrorString(e));
    return 0;
}

int moss_raster_export_binning(const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    int P, int R, int width, int height,
    uint64_t* point_list_keys, uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, void* stream)
{
    g_err[0] = 0;
    if (P <= 0 || !geom_buffer || !binning_buffer || !image_buffer) return fail(MOSS_ERR_INVALID_ARG, "bad arguments");
    GeomView g = GeomView::at(const_cast<char*>(geom_buffer), P);
    ImageView im = ImageView::at(const_cast<char*>(image_buffer), width, height);
    BinView b = BinView::at(const_cast<char*>(binning_buffer), R);
    const FrameParams fp = make_params(P, 0, 0, width, height, 1.f, 1.f, 1.f, 0, nullptr, nullptr, nullptr, nullptr);
    launch_export_binning(fp, g, im, b, R, point_list_keys, point_list, ranges, final_T, n_contrib, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOSS_ERR_HIP, "export_binning launch failed: %s", hipGetErrorString(e));
    return 0;
}

#ifdef MOSS_DIAG
void moss_raster_debug_set_stamps(unsigned long long* device_buffer) { moss::g_stamps = device_buffer; }
void moss_raster_debug_set_bwd_stamps(unsigned long long* device_buffer) { moss::g_bwd_stamps = device_buffer; }
#endif
int moss_build_has_diagnostics(void)
{
#ifdef MOSS_DIAG
    return 1;
#else
    return 0;
#endif
}

void moss_raster_profile_enable(uint32_t stage_mask)
{
    std::lock_guard<std::mutex> lk(g_prof.m);
    g_prof.mask = stage_mask;
}

int moss_raster_profile_read(float* ms_sum, uint32_t* count)
{
    std::lock_guard<std::mutex> lk(g_prof.m);
    for (int i = 0; i < MOSS_NUM_STAGES; i++) { ms_sum[i] = 0.0f; count[i] = 0; }
    int rc = 0;
    for (const ProfRec& r : g_prof.recs) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) rc = MOSS_ERR_HIP;
        else if (r.stage >= 0 && r.stage < MOSS_NUM_STAGES) { ms_sum[r.stage] += ms; count[r.stage]++; }
        g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b);
    }
    g_prof.recs.clear();
    return rc;
}

}  // extern "C"
