/*
 * moss_raster.h -- C ABI of the MI355X (gfx950) differentiable Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary.  Each entry point replaces one function of the reference's native API
 * (3DHumanRehab/MOSS, submodules/diff-gaussian-rasterization = "DGR/", submodules/simple-knn = "SKNN/"),
 * with the same arguments in the same order and the same meaning; only C++-isms are flattened:
 *   - std::function<char*(size_t)> buffer growers  ->  a C callback + user pointer (moss_alloc_fn),
 *   - exceptions                                    ->  negative return code + moss_last_error(),
 *   - the implicit CUDA legacy default stream       ->  an explicit hipStream_t passed as void*.
 * All pointers are DEVICE pointers to contiguous fp32 (int32 for radii) unless stated otherwise; an absent
 * optional input is NULL (DGR/rasterize_points.cu passes the null data_ptr() of an empty tensor).
 * No torch types appear here.  The rasterizer entry points are bound by the compiled PyTorch-ROCm extension
 * moss_amd/csrc/torch_binding.cpp (the counterpart of DGR/rasterize_points.cu); the side kernels (k-NN, loss, AdamW,
 * densification statistics) by ctypes in moss_amd/_lib.py.
 *
 * The three scratch buffers are opaque to the caller exactly as in the reference (their internal layout is
 * this library's own); the caller must keep them alive and unmodified between forward and
 * backward, as the reference's autograd ctx does (DGR/diff_gaussian_rasterization/__init__.py:97).
 */
#ifndef MOSS_RASTER_H
#define MOSS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version 2 (round 3): the frame state is an ARGUMENT of the asynchronous forwards (version 1 armed it per host thread through
 * moss_raster_frame_state(), removed); the `debug` argument is a bit set (MOSS_DEBUG_*); moss_adamw_state_bytes() replaces the
 * caller's knowledge of the device-step block's size (288 bytes in early version-1 builds, 9216 later); diagnostics
 * (environment knobs, stamp buffers) exist only in -DMOSS_DIAG builds.  A binding compiled against another version must refuse to
 * load: compare ITS compile-time MOSS_ABI_VERSION with moss_abi_version(). */
/* ABI 5 (round 5): two more bits of `debug` -- MOSS_DEBUG_EXACT_MATH, MOSS_DEBUG_TRACE (below) -- and moss_raster_binning_bytes
 * follows the slimmer gradient-record layout.  No signature changed. */
/* ABI 6 (round 6): MOSS_FORWARD_ONLY (a bit of the forward entry points' `debug` argument) + moss_raster_binning_bytes_forward_only;
 * moss_fused_adamw gained sh_active_degree / sh_inactive_zero and moss_adamw_flat_ex takes the same two per segment (degree-aware SH
 * traffic: coefficients above the highest degree ever active are never read or written when they are known to be zero);
 * moss_photometric_loss_roi (MOSS's own loss expression: bound_mask selection, bounding-rectangle crop) and moss_adamw_multi (up to eight
 * tensors with buffers of their own in one launch) are new entry points. */
#define MOSS_ABI_VERSION 6
/* Version 3 (round 4): EVERY forward / backward entry point takes the `debug` bit set (version 2: only moss_raster_forward /
 * moss_raster_backward did, so MOSS_DEBUG_NO_BLOCK_CULL was silently dropped on the _async / _tf / _raw paths: last argument before
 * `stream`); moss_adamw_flat_guarded (an optimizer step that a dropped frame turns into a no-op); MOSS_RAW_POSE and the
 * `translation` / `dL_dtranslation` arguments of the _raw entry points (the canonical positions are posed inside the op). */

/* error codes (negative returns) */
#define MOSS_ERR_INVALID_ARG   (-1)   /* bad shape / null where required (AT_ERROR in DGR/rasterize_points.cu:57-59) */
#define MOSS_ERR_HIP           (-2)   /* a HIP runtime call or kernel failed (CHECK_CUDA, DGR/cuda_rasterizer/auxiliary.h:166-173) */
#define MOSS_ERR_ALLOC         (-3)   /* an allocation callback returned NULL */
#define MOSS_ERR_PREFILTERED   (-4)   /* a point was culled although `prefiltered` was set (__trap in auxiliary.h:156-160) */
#define MOSS_ERR_UNSUPPORTED   (-5)   /* e.g. no colours for NUM_CHANNELS != 3 (DGR/cuda_rasterizer/rasterizer_impl.cu:244-247) */

/* Replaces std::function<char*(size_t)> (DGR/cuda_rasterizer/rasterizer.h:32-34; grown by resizeFunctional,
 * DGR/rasterize_points.cu:27-33).  Must return a device pointer to at least nbytes, 256-byte aligned, or NULL. */
typedef char* (*moss_alloc_fn)(void* user, size_t nbytes);

int moss_abi_version(void);

/* Text of the last error on the calling thread ("" if none). */
const char* moss_last_error(void);

/*
 * Replaces CudaRasterizer::Rasterizer::forward (DGR/cuda_rasterizer/rasterizer.h:31-55,
 * DGR/cuda_rasterizer/rasterizer_impl.cu:198-341).
 *   P  number of Gaussians, D active SH degree (0..3), M stored SH coefficients per Gaussian (0 if shs==NULL).
 *   background (3), means3D (P,3), shs (P,M,3) | colors_precomp (P,3), opacities (P),
 *   scales (P,3) + rotations (P,4) | cov3D_precomp (P,6), viewmatrix/projmatrix (16, transposed = column-major),
 *   cam_pos (3).   Outputs: out_color (3,H,W), out_depth (H,W), out_alpha (H,W), radii (P) int32 (may be NULL).
 *   Outputs need NOT be pre-zeroed (the reference requires zero-filled tensors; every element is written here).
 * Returns num_rendered >= 0 (the number of (Gaussian, tile) instances), or a negative error code.
 * Performs ONE stream synchronisation (to size the binning buffer), like the reference's blocking read
 * at rasterizer_impl.cu:283.
 * `debug` is a bit set.  MOSS_DEBUG_SYNC (1, the reference's `debug = true`): the stream is synchronised and checked after every
 * launch (CHECK_CUDA, auxiliary.h:166-173).  MOSS_DEBUG_NO_BLOCK_CULL (2; forward AND the matching backward call): the blend kernels
 * ignore the per-instance 4x4-block masks and test every list entry against every block.  The masks only SKIP (entry, block)
 * pairs that cannot reach alpha >= 1/255: final_T and n_contrib are bit-identical either way, images and gradients equal up to
 * float32 summation order (tests/test_gpu_ops.py::test_block_mask_culling_never_changes_a_result).  Per call, no global state.
 * In the BACKWARD the masks' box is also an addressing contract (a pair's gradient record has a cell only inside the box): a pair
 * outside it leaves no record in this mode either (light tiles: visited, record dropped; heavy tiles: the masks are followed) --
 * whether such a pair could have reached 1/255 is what the no-cull FORWARD image shows.
 * MOSS_DEBUG_EXACT_MATH (4; ABI 5; forward AND the matching backward call): the blend kernels evaluate what decides a pixel's list
 * -- the exponent, exp(), alpha, the transmittance chain -- exactly as the reference's SOURCE reads (forward.cu:336-356,
 * backward.cu:504-516): `power = -0.5f * (A dx dx + C dy dy) - B dx dy` with one rounding per operation (the fast path spells two
 * of them as FMAs), exp() as a correctly defined function (the restatement of glibc's expf that oracle/moss_oracle.c carries as
 * moss_expf_det: the same bits on CPU and GPU; the fast path uses v_exp_f32, ~1 ulp), and T = T / (1 - alpha) as a chain of IEEE
 * divisions (the fast path: v_rcp_f32 and a prefix product).  Under it n_contrib and final_T equal the CPU oracle's (with the same
 * exp) BIT FOR BIT on every pixel -- which turns "n_contrib may differ on <= 1e-4 of the pixels" of the fast path into a checked
 * statement: fast and exact differ only where a decision sits within rounding of its threshold.  2-3x slower blend kernels; a
 * checking mode, not a product path.
 * MOSS_DEBUG_TRACE (8; ABI 5): every stage launcher is wrapped in a roctx range ("moss:preprocess_fwd", "moss:scatter", ...), so
 * that a `rocprofv3 --kernel-trace --marker-trace` timeline of the CALLER's program shows the op's stages.  The
 * roctx library (librocprofiler-sdk-roctx.so) is resolved with dlopen at the first traced call: no link-time dependency, nothing
 * happens without the bit.  Ranges bracket the LAUNCH calls on the host (inside a captured hipGraph they are recorded once, at capture).
 */
#define MOSS_DEBUG_SYNC          1
#define MOSS_DEBUG_NO_BLOCK_CULL 2
#define MOSS_DEBUG_EXACT_MATH    4
#define MOSS_DEBUG_TRACE         8
/* MOSS_FORWARD_ONLY (16; ABI 6; a bit of the same argument of every FORWARD entry point -- not a diagnostic: the one call option the
 * reference's signature has no room for): the caller promises that NO backward call follows this forward -- an evaluation render
 * (render_ZJU.py:56-72: `render(view, gaussians, pipeline, background)` under torch.no_grad()).  The outputs are the
 * training forward's BIT FOR BIT (the blend folds its sums at the same list positions); what is not produced is the state only the
 * backward reads: depth-segment cuts, per-block tails, gradient-record cells and their validity bits.  The binning buffer then holds
 * ids, block masks, the 48-byte records and the 8-byte sort keys only: moss_raster_binning_bytes_forward_only(R), 62 B per instance
 * (training: ~370 B).  With a capacity (asynchronous variants) the keys go through the scan -> scatter chain into exact ranges -- six
 * launches -- because the per-tile key buckets of the four-launch training forward live in the record pool's address space.
 * A backward call over such buffers is a no-op that returns ZERO gradients and takes no optimizer step (status flag
 * MOSS_STATUS_FORWARD_ONLY; the kernels check it on the device, like a capacity overflow): never out-of-bounds. */
#define MOSS_FORWARD_ONLY        16
int moss_raster_forward(
    moss_alloc_fn geometry_alloc, void* geometry_user,
    moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,
    float* out_depth,
    float* out_alpha,
    int* radii,
    int debug,
    void* stream);

/*
 * Asynchronous variant of moss_raster_forward for launch-bound training loops and hipGraph capture: NO host read-back.
 * The caller states an upper bound `capacity` on the number of (Gaussian, tile) instances (e.g. 2x the value a previous,
 * synchronous call returned); scratch buffers and launch grids are sized for it and every kernel bounds itself with the
 * device-side count.  Returns `capacity` (>= 0) -- pass that as R to moss_raster_backward -- or a negative error code.
 * If a frame needs more instances than `capacity`, nothing is rendered (outputs = background, gradients = 0) and the
 * overflow bit is set in the status words; poll them with moss_raster_read_status once the stream has advanced.
 * `debug`: MOSS_DEBUG_NO_BLOCK_CULL is honoured; MOSS_DEBUG_SYNC is refused (MOSS_ERR_INVALID_ARG: it synchronises by definition).
 *
 * `frame_state` (optional, may be NULL; no counterpart in the reference, which memsets its buffers in every forward): a caller-owned
 * device block of moss_raster_frame_state_bytes(width, height) bytes, zero-initialised ONCE.  With it this call keeps the per-frame
 * counters its kernels add to (tile histogram, tile cursors, error flags) in that block instead of in the image buffer and returns
 * the block all-zero again (its sort kernel re-zeroes it; error paths clean it too), so no clear kernel runs in front of the
 * preprocess kernel -- one launch less per frame (4 us inside a captured graph).  One block per concurrent user (stream); the same
 * block serves every call of that user, also across image sizes up to the one it was sized for.  The library keeps NO state between
 * calls: the block is an argument (ABI version 1 armed it per host thread).  Its address is a kernel argument, so a block handed
 * to a call that was captured into a hipGraph must stay alive for as long as that graph is replayed.
 * The 32-bit word MOSS_FRAME_STATE_DROPPED_WORD of the block is a STICKY counter: the library adds 1 for every frame that
 * overflowed its capacity (and therefore rendered nothing) and never clears it; the caller reads it whenever it likes (e.g. once per
 * few hundred replays of a captured step -- the status words only describe the LAST frame) and writes 0 back.
 */
#define MOSS_FRAME_STATE_DROPPED_WORD 4
size_t moss_raster_frame_state_bytes(int width, int height);
int moss_raster_forward_async(
    moss_alloc_fn geometry_alloc, void* geometry_user,
    moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int capacity, char* frame_state, int debug, void* stream);

/* Enqueue (on `stream`) a copy of the forward's 8 status words from the image buffer to pinned host memory:
 * [0] instances rendered  [1] longest tile list  [2] flags: bit0 prefiltered-point culled, bit1 capacity overflow
 * [3] the capacity that holds this frame: >= [6] -- the buffer's gradient-record pool (6 cells of 48 B per instance of capacity; a
 *     frame of wide Gaussians uses up to 16 per instance) and the per-tile key buckets are sized from the capacity too; a caller's
 *     policy grows the capacity from THIS word (moss_amd/diff_gaussian_rasterization/_C.py: margin x [3])
 * [4] sort chunks  [5] non-empty tiles  [6] instances the frame needed. */
#define MOSS_STATUS_PREFILTERED 1u
#define MOSS_STATUS_OVERFLOW    2u
#define MOSS_STATUS_FORWARD_ONLY 4u  /* the forward ran with MOSS_FORWARD_ONLY: its buffers carry no backward state */
int moss_raster_read_status(const char* image_buffer, uint32_t* host_pinned_out, void* stream);

/*
 * Replaces CudaRasterizer::Rasterizer::backward (DGR/cuda_rasterizer/rasterizer.h:57-89,
 * DGR/cuda_rasterizer/rasterizer_impl.cu:345-447).  R is the value forward returned.
 *   dL_dpix (3,H,W), dL_ddepths (H,W), dL_dalphas (H,W): incoming gradients; any (not all) of them may be NULL, meaning
 *   zeros (an output that did not take part in the loss), which saves the caller a zero-filled image.
 *   Gradient outputs: dL_dmean2D (P,3), dL_dconic (P,4 = 2x2), dL_dopacity (P), dL_dcolor (P,3),
 *   dL_dmean3D (P,3), dL_dcov3D (P,6), dL_dsh (P,M,3) (may be NULL if M==0), dL_dscale (P,3), dL_drot (P,4).
 *   Every element of every output is written (zeros for culled Gaussians, for dL_dscale/dL_drot when
 *   scales==NULL and for SH coefficients above the active degree), so outputs need NOT be pre-zeroed.
 *   dL_dconic (an intermediate), dL_dcolor and dL_dcov3D (gradients of the optional inputs colors_precomp / cov3D_precomp) may be
 *   NULL = not wanted (ABI 4): 52 bytes per Gaussian a caller working from SH and scales / rotations never reads.
 *   `alphas` and `radii` are accepted for signature parity and ignored (the reference ignores alphas too,
 *   DGR/cuda_rasterizer/backward.cu:410).
 * Gradients are bitwise reproducible run to run (no float atomics), unlike the reference.
 * Returns 0 or a negative error code.
 */
int moss_raster_backward(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* alphas,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* image_buffer,
    const float* dL_dpix,
    const float* dL_ddepths,
    const float* dL_dalphas,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_dmean3D,
    float* dL_dcov3D,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    int debug,
    void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (DGR/cuda_rasterizer/rasterizer.h:24-29,
 * rasterizer_impl.cu:141-153).  present: (P) bytes, 1 if z_view > 0.2. */
int moss_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                             uint8_t* present, void* stream);

/*
 * Replaces SimpleKNN::knn (SKNN/simple_knn.h:16-19, SKNN/simple_knn.cu:185-221) behind distCUDA2
 * (SKNN/spatial.cu:16-25): mean_dists[i] = mean of the squared distances from point i to its 3 nearest
 * other points.  points (P,3), mean_dists (P).  workspace: device scratch of moss_knn_workspace_bytes(P).
 * Fully asynchronous on `stream` (the reference blocks twice to fetch the scene bounding box, simple_knn.cu:197,200;
 * here the box is reduced and consumed on the device).
 */
size_t moss_knn_workspace_bytes(int P);
int moss_knn_dist2(int P, const float* points, float* mean_dists, char* workspace, size_t workspace_bytes, void* stream);

/*
 * Fused photometric loss of the training step and its gradient (the consumer side of the rasterizer's outputs):
 *   loss = mean|image - gt| + lambda_mask * mean((alpha - mask)^2) + lambda_dssim * (1 - mean SSIM(image, gt))
 * Replaces the torch graph built from l1_loss / l2_loss / ssim (utils/loss_utils.py:41-87) as combined in
 * train_ZJU.py:111-112,119,131 (lambda_dssim 0.2, lambda_mask 0.5) and its autograd backward.
 *   image, gt (C,H,W); alpha, mask (H,W) or both NULL; loss_out: 4 device floats {total, l1, ssim, mask_l2};
 *   dL_dimage (C,H,W), dL_dalpha (H,W; NULL iff alpha NULL): gradients of `total`, every element written.
 *   workspace: device scratch of moss_loss_workspace_bytes(C,H,W).  Asynchronous on `stream`, deterministic.
 */
size_t moss_loss_workspace_bytes(int C, int H, int W);
int moss_photometric_loss(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                          float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage, float* dL_dalpha,
                          char* workspace, size_t workspace_bytes, void* stream);
/* The same with a weight on the L1 term too (ABI 4): total = lambda_l1 L1 + lambda_mask maskL2 + lambda_dssim (1 - SSIM); lambda_l1 = 1
 * gives moss_photometric_loss bit for bit.  With lambda_l1 = 0, lambda_dssim = 1 and no alpha, loss_out[2] is the reference's
 * ssim(img1, img2) (utils/loss_utils.py:47-87: 11x11 window, zero padding, mean over all elements) and -dL_dimage its gradient:
 * a drop-in for MOSS's own ssim() call (train_ZJU.py:119) without changing its loss expression. */
int moss_photometric_loss_weighted(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                   float lambda_l1, float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage,
                                   float* dL_dalpha, char* workspace, size_t workspace_bytes, void* stream);
/* MOSS's OWN loss expression for these three terms (ABI 6; train_ZJU.py:108-119,131): the L1 and the mask term are means over the
 * pixels of the view's `bound_mask` -- Ll1 = l1_loss(image[bound], gt[bound]) (:111), mask_loss = l2_loss(alpha[bound], mask[bound])
 * (:112) -- and SSIM is taken on the crop x, y, w, h = cv2.boundingRect(bound_mask) of both images (:115-119: zero padding at the
 * CROP's edges, mean over C*w*h).  total = lambda_l1 Ll1 + lambda_mask mask_loss + lambda_dssim (1 - ssim).
 *   bound: (H,W) bytes, non-zero = counted; NULL = every pixel of the rectangle.
 *   rect:  FIVE int32 in DEVICE memory: x, y, w, h, and the number of non-zero bytes of `bound` (ignored when bound is NULL).  Device
 *          memory so that a step captured in a hipGraph changes view by rewriting them (and `bound`, `gt`, `mask`) in place.  The
 *          rectangle is clipped to the image.  Pixels of `bound` outside it count for nothing (MOSS's rectangle is the bounding box of
 *          the mask: there are none).  An empty mask gives NaN means, like torch's mean of an empty selection (MOSS's own expression
 *          raises there: its ssim() is handed a 0 x 0 crop).
 *   dL_dimage, dL_dalpha: written for the WHOLE image (zero off the crop / off the mask), so they are the gradients of `total`
 *          w.r.t. the full-size tensors the rasterizer produced.  Same workspace, same two launches, deterministic. */
int moss_photometric_loss_roi(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                              const unsigned char* bound, const int* rect, float lambda_l1, float lambda_dssim, float lambda_mask,
                              float* loss_out, float* dL_dimage, float* dL_dalpha, char* workspace, size_t workspace_bytes, void* stream);

/*
 * Flat fused AdamW (torch.optim.AdamW semantics, amsgrad off) over `n` contiguous fp32 parameters with their gradients and
 * moments; up to 8 learning-rate segments: parameter i belongs to the first segment s with i < segment_end[s] (host arrays).
 * Replaces the per-group optimizer step of scene/gaussian_model.py:215-226 for the Gaussian parameters.  `step` counts from 1.
 * Optional periodic pattern per segment (all three arrays NULL = none): where segment_period[s] > 0, element j of segment s uses
 * segment_lr[s] if j % segment_period[s] < segment_split[s], else segment_lr2[s] -- e.g. the SH coefficients stored as ONE
 * (P,16,3) tensor with the DC term's learning rate on the first 3 of every 48 floats and lr/20 on the rest (the reference's
 * separate f_dc / f_rest groups, without the per-step torch.cat of get_features).
 */
int moss_adamw_flat(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    int num_segments, const long long* segment_end, const float* segment_lr,
                    const int* segment_period, const int* segment_split, const float* segment_lr2,
                    double beta1, double beta2, float eps, float weight_decay, int step, void* stream);
/* Same update with the step counter kept on the device: `step_state` is moss_adamw_state_bytes() (= MOSS_ADAMW_STATE_BYTES of the
 * header the library was built from; ask the library, a binding's copy of the constant can be stale) zero-initialised device bytes
 * (32-bit words: [0] = int step, advanced by one per call by the update kernel itself; [8..11] = the bias corrections of the
 * current / next step, double-buffered by step parity; [64] and [128 + 64 g], g < 32 = its two-level block-completion counters,
 * each on a 256-byte line of its own).  n must be > 0.  Nothing in the call depends on a host-side
 * counter, so a captured hipGraph of a training step replays correctly.
 * LEARNING RATES ON THE DEVICE (ABI 4): when word MOSS_ADAMW_LR_VALID_WORD of the block is non-zero, every kernel that is given the
 * block (the _devstep / _range / _guarded updates and moss_raster_backward_raw_adamw) takes segment s's learning rate from float
 * word MOSS_ADAMW_LR_WORD0 + s and its second rate (periodic pattern) from MOSS_ADAMW_LR2_WORD0 + s instead of from segment_lr /
 * segment_lr2: a schedule -- MOSS decays the position rate every iteration, scene/gaussian_model.py:263-268 -- is then a 64-byte
 * host-to-device copy between two replays of a captured step, not a re-capture. */
#define MOSS_ADAMW_STATE_BYTES 9216
#define MOSS_ADAMW_LR_VALID_WORD 12
#define MOSS_ADAMW_LR_WORD0 16
#define MOSS_ADAMW_LR2_WORD0 24
size_t moss_adamw_state_bytes(void);
int moss_adamw_flat_devstep(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            int num_segments, const long long* segment_end, const float* segment_lr,
                            const int* segment_period, const int* segment_split, const float* segment_lr2,
                            double beta1, double beta2, float eps, float weight_decay, void* step_state, void* stream);

/* The same update on a SHARD of the flat buffers: the arrays hold the elements [first, first + count) (first a multiple of 4) of the
 * buffers the segment table -- global indices, as above -- describes.  step_state != NULL: device-side step counter (then `step` is
 * ignored), else `step` counts from 1.  For N ranks that reduce-scatter the gradient bucket, update their 1/N of the parameters
 * (moments memory and update time / N) and all-gather the result (moss_amd/dist.py ShardedStep). */
int moss_adamw_flat_range(long long first, long long count, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                          int num_segments, const long long* segment_end, const float* segment_lr,
                          const int* segment_period, const int* segment_split, const float* segment_lr2,
                          double beta1, double beta2, float eps, float weight_decay, int step, void* step_state, void* stream);

/* moss_adamw_flat_range with a GUARD: if (*skip_word & skip_mask) != 0 when the kernel runs, the call is a no-op -- parameters,
 * moments and the device-side step counter stay bit for bit what they were.  `skip_word`: a device word, e.g. the frame's status word
 * (image buffer, 32-bit word 2; skip_mask = 2: the capacity-overflow bit of moss_raster_forward_async) -- a frame that overflowed
 * its capacity rendered nothing and left zero gradients, and inside a captured step nobody is there to skip the optimizer: without
 * the guard such a frame is a weight-decay-only step that also decays the moments.  Or a float that is non-zero when ANY rank
 * dropped its frame (the flag averaged with the gradient bucket, skip_mask = 0x7fffffff) so that replicas skip together.
 * Needs the device-side step counter (step_state != NULL): a host-side count cannot know about the skipped step. */
int moss_adamw_flat_guarded(long long first, long long count, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            int num_segments, const long long* segment_end, const float* segment_lr,
                            const int* segment_period, const int* segment_split, const float* segment_lr2,
                            double beta1, double beta2, float eps, float weight_decay, void* step_state,
                            const uint32_t* skip_word, uint32_t skip_mask, void* stream);

/* Every form of the flat update behind one argument block (ABI 6), plus the DEGREE-AWARE update of an SH tensor: MOSS trains at SH
 * degree 0 / 1 / 2 for iterations 1-2999 and at degree 3 for the last one (train_ZJU.py:85-86, scene/gaussian_model.py:171-173); the
 * coefficients above the highest degree that has ever been active have received no gradient, so their moments are exactly zero and their
 * AdamW step is the decoupled weight decay alone.
 *   segment_active (NULL = none; else one int per segment, meaningful where segment_period[s] > 0 and a multiple of 4): of every
 *     `period` elements of segment s only the first segment_active[s] are ACTIVE (0 = all).  Inactive elements: gradient and moments are
 *     neither read nor written; the parameter takes p <- p (1 - lr wd), which is bit for bit what the full update gives for g = m = v = 0.
 *   inactive_zero != 0: the caller also knows the inactive PARAMETERS to be exactly zero (MOSS initialises features_rest with zeros,
 *     scene/gaussian_model.py:179-181, and 0 x decay = 0): they are not read or written at all.
 *   first / count as moss_adamw_flat_range; step_state NULL: `step` counts from 1, else the device-side counter; skip_word (with
 *   step_state) as moss_adamw_flat_guarded.  The results equal those of the corresponding older entry point bit for bit. */
typedef struct moss_adamw_flat_args {
    long long first, count;
    float* params; const float* grads; float* exp_avg; float* exp_avg_sq;
    int num_segments; const long long* segment_end; const float* segment_lr;
    const int* segment_period; const int* segment_split; const float* segment_lr2;
    const int* segment_active; int inactive_zero;
    double beta1, beta2; float eps, weight_decay;
    int step; void* step_state; const uint32_t* skip_word; uint32_t skip_mask;
    /* B views rendered for ONE optimizer step on one device (moss_amd/multiview.py: the data-parallel step of SURVEY 8e with the "ranks"
     * on one GPU): up to three more gradient buffers of the same layout as `grads` (for a range: its elements [first, first + count)).
     * The step's gradient is ((grads + grads_extra[0]) + grads_extra[1] ...) x grad_scale, added in that order, one float32 rounding
     * per operation -- bit for bit what accumulating the views one after the other into one buffer and scaling it gives.
     * num_grads_extra = 0: `grads` as it is (grad_scale is then ignored). */
    int num_grads_extra; const float* grads_extra[3]; float grad_scale;
} moss_adamw_flat_args;
int moss_adamw_flat_ex(const moss_adamw_flat_args* args, void* stream);

/* Up to eight parameter tensors with buffers OF THEIR OWN in one launch (ABI 6): what a torch.optim-style optimizer holds -- MOSS's six
 * single-tensor Gaussian groups (scene/gaussian_model.py:215-226), each with its `exp_avg` / `exp_avg_sq` state tensors and its own
 * step count (densification surgery keeps them per tensor).  Per element the arithmetic of moss_adamw_flat with one segment and the
 * tensor's `lr` and `step`: bit-identical to num_tensors calls of it.  Pointers 16-byte aligned; an entry with numel 0 is skipped.
 * moss_amd.optim.AdamW -- the drop-in for `torch.optim.AdamW` of patches/gaussian_model.diff -- steps its single-tensor groups with it. */
typedef struct moss_adamw_multi_args {
    int32_t num_tensors;                     /* 1..8 */
    long long numel[8];
    float* params[8]; const float* grads[8]; float* exp_avg[8]; float* exp_avg_sq[8];
    float lr[8]; int32_t step[8];            /* step counts from 1, like moss_adamw_flat */
    double beta1, beta2; float eps, weight_decay;
} moss_adamw_multi_args;
int moss_adamw_multi(const moss_adamw_multi_args* args, void* stream);

/*
 * k nearest reference points of every query point, 3-D, exact, k = 1..4: replaces the third-party
 * `knn_cuda.KNN(k, transpose_mode=True)(ref, query)` MOSS calls at scene/gaussian_model.py:85-86,586,657,759,827 (a CUDA-only
 * binary wheel, not in the repository; parity unpinned by the reference).
 *   ref (Nr,3), query (Nq,3) fp32 device arrays; dist_out (Nq,k) Euclidean distances ascending; idx_out (Nq,k) int64 reference
 *   indices.  Nr >= k.  Ties: the lower reference index first.  Asynchronous on `stream`.
 */
int moss_knn_query(int Nr, int Nq, int k, const float* ref, const float* query, float* dist_out, long long* idx_out, void* stream);

/*
 * The same query through a uniform cell grid over the references (exact; results identical to moss_knn_query bit for bit, ties
 * included).  Built once per reference set (8 small launches), queried any number of times: MOSS queries the SAME template
 * vertices every step (scene/gaussian_model.py:827) and the Gaussians themselves when densifying (:586,759; 100k x 100k is 7.7 ms
 * by brute force).  workspace: moss_knn_grid_workspace_bytes(Nr) device bytes, owned by the caller, read-only for queries.
 * Cost grows with the distance between a query and its k-th neighbour measured in cells (about two cells per reference): meant
 * for queries that lie among the references; far outliers stay exact but approach brute-force cost.  Asynchronous on `stream`.
 */
size_t moss_knn_grid_workspace_bytes(int Nr);
int moss_knn_grid_build(int Nr, const float* ref, char* workspace, size_t workspace_bytes, void* stream);
int moss_knn_grid_query(int Nr, int Nq, int k, const char* workspace, size_t workspace_bytes, const float* query,
                        float* dist_out, long long* idx_out, void* stream);

/*
 * Densification bookkeeping, one launch, no host synchronisation.  Replaces, for vis = radii > 0,
 *   gaussians.max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                          (train_ZJU.py:173)
 *   xyz_gradient_accum[vis] += norm(viewspace_points.grad[vis,:2], dim=-1); denom[vis] += 1  (scene/gaussian_model.py:815-817)
 * radii (P) int32 from the forward; viewspace_grad (P, grad_stride >= 2) = dL_dmeans2D of the backward;
 * xyz_gradient_accum (P), denom (P), max_radii2D (P, may be NULL) fp32, updated in place.
 */
int moss_densify_stats(int P, const int* radii, const float* viewspace_grad, int grad_stride,
                       float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream);
/*
 * KL divergence between pairs of Gaussians, with the gather fused in: replaces GaussianModel.kl_div
 * (scene/gaussian_model.py:773-813) applied to the pairs picked by the k = 2 self-query (:586-597, :759-770).
 *   xyz (Nsrc,3), rotation (Nsrc,4) raw quaternions (normalised inside like build_rotation, utils/general_utils.py:79-100),
 *   scaling (Nsrc,3) ACTIVATED scales; pair_idx (P,2) int64 = the kNN result (column 0: Gaussian "0", column 1: Gaussian "1");
 *   kl_out (P) = 0.5 (tr(S1^-1 S0) + (mu1-mu0)^T S1^-1 (mu1-mu0) + ln prod((s1/s0)^2) - 3).  An out-of-range index gives NaN.
 */
int moss_neighbour_kl(int P, int Nsrc, const float* xyz, const float* rotation, const float* scaling,
                      const long long* pair_idx, float* kl_out, void* stream);

/*
 * Extension: covariance with a per-Gaussian 3x3 transform INSIDE the op.
 * MOSS feeds cov3D_precomp = strip_symmetric(T (R S S^T R^T) T^T) built by torch ops (scene/gaussian_model.py:37-44,168-169;
 * gaussian_renderer/__init__.py:88-91) because the LBS transform T of each Gaussian changes every frame; on MI355X that Python
 * path costs more than the whole rasterizer.  These two entry points take (scales, rotations, transforms (P,3,3) row-major)
 * instead and return gradients for all three.  Same semantics otherwise as moss_raster_forward[_async] (capacity < 0:
 * synchronous sizing of the binning buffer, >= 0: asynchronous with that capacity) and moss_raster_backward; `debug` as there
 * (MOSS_DEBUG_SYNC only with capacity < 0; MOSS_DEBUG_NO_BLOCK_CULL on the forward AND the matching backward call).
 *   dL_dcov3D (P,6): gradient w.r.t. the transformed covariance (what the op stores); dL_dscale / dL_drot include the transform;
 *   dL_dtransforms (P,9): written for every Gaussian (zeros for culled ones).
 */
int moss_raster_forward_tf(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int capacity, char* frame_state, int debug, void* stream);
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
    float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtransforms, int debug, void* stream);

/*
 * Extension (caller side of the boundary): the GaussianModel getters applied INSIDE the op.
 * MOSS hands the rasterizer get_opacity = sigmoid(_opacity), get_scaling = exp(_scaling), get_rotation = normalize(_rotation)
 * (scene/gaussian_model.py:142-161, gaussian_renderer/__init__.py:77-93): five torch ops forward and a dozen backward per step.
 * These entry points take the RAW parameters for the inputs named in raw_flags and return the gradients w.r.t. the raw
 * parameters; otherwise they are moss_raster_forward_tf / moss_raster_backward_tf (transforms may be NULL here = none).
 *   raw_flags: MOSS_RAW_OPACITY (opacities are logits) | MOSS_RAW_SCALE (scales are logarithms) | MOSS_RAW_ROTATION
 *   (rotations are not normalised; normalised as x / max(|x|, 1e-12) like torch.nn.functional.normalize).
 *   The backward needs the raw opacities again (the reference backward does not take opacities at all).
 *   MOSS_HINT_SPATIAL_ORDER may be OR-ed in: "neighbours in index are neighbours in space" (the caller re-indexed its Gaussians along
 *   a space-filling curve, e.g. moss_amd.densify.spatial_order).  It changes no result beyond the order of some float32 sums, only how the per-Gaussian backward deals
 *   Gaussians to its workgroups (groups of 16 from places spread over the index range, so that no workgroup is all-heavy).
 *   MOSS_RAW_POSE (needs `transforms`): means3D are the CANONICAL positions x and the op poses them itself, p = T x (+ translation
 *   (P,3), may be NULL), rows of T times x summed left to right -- what MOSS's caller does with torch ops before the call
 *   (gaussian_renderer/__init__.py:74-77: torch.matmul(transforms, means3D[..., None]).squeeze(-1) + translation).  dL_dmean3D is
 *   then the gradient w.r.t. x (= T^T dL/dp: it can be written straight into the position parameter's gradient), dL_dtransforms
 *   gains dL/dp x^T, and dL_dtranslation (P,3; may be NULL) = dL/dp.
 *   MOSS_SH_GRAD_ACTIVE_ONLY (ABI 6; backward entry points): dL_dsh is written for the coefficients of the ACTIVE degree only, (D+1)^2 of
 *   M per Gaussian; the rest of the destination is left untouched.  For a caller whose destination already holds zeros there and whose
 *   consumers never read them -- a gradient sink into a zero-initialised bucket consumed by the degree-aware flat AdamW
 *   (moss_adamw_flat_ex) and the active-degree exchange.  Without the bit every element is written (zeros above the degree), as ever.
 */
#define MOSS_RAW_OPACITY 1
#define MOSS_RAW_SCALE 2
#define MOSS_RAW_ROTATION 4
#define MOSS_HINT_SPATIAL_ORDER 8
#define MOSS_RAW_POSE 16
#define MOSS_SH_GRAD_ACTIVE_ONLY 32
int moss_raster_forward_raw(
    moss_alloc_fn geometry_alloc, void* geometry_user, moss_alloc_fn binning_alloc, void* binning_user,
    moss_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transforms, const float* translation,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_depth, float* out_alpha, int* radii, int raw_flags, int capacity, char* frame_state, int debug, void* stream);
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
    int raw_flags, int debug, void* stream);

/*
 * Extension (ABI 4): the raw-parameter backward that also TAKES THE OPTIMIZER STEP.  The Gaussian parameters
 * of MOSS receive their gradients from this op alone (every loss term of train_ZJU.py:111-131 goes through the rendered image; the
 * exception is the position, which also feeds the LBS-weight network), and torch.optim.AdamW then streams parameter, gradient and
 * both moments through the device once more (scene/gaussian_model.py:215-226, train_ZJU.py:204-205).  Here the per-Gaussian backward
 * kernel, which holds a Gaussian's gradients in registers / LDS when it finishes, applies the AdamW update of the tensors named in
 * opt->tensors itself: the parameters (= the op's inputs, hence not const here) are updated in place, the moments in
 * opt->exp_avg / exp_avg_sq, and the gradient of an updated tensor need not be written at all (its dL_d* pointer may be NULL).
 * Same arithmetic, bit for bit, as moss_adamw_flat_devstep on the same values (both use csrc/adamw.h); a frame that overflowed its
 * capacity (asynchronous forward) takes no step, like moss_adamw_flat_guarded on the frame's status word.
 *   Requirements: raw_flags contains MOSS_RAW_OPACITY | MOSS_RAW_SCALE | MOSS_RAW_ROTATION (the inputs must be the parameters
 *   themselves, not activated copies); MOSS_OPT_SH needs M == 16 and 16-byte aligned shs / moments; MOSS_OPT_MEANS only if the
 *   means the op sees are the parameter (no transforms, or MOSS_RAW_POSE) and nothing else contributes to its gradient.
 *   opt->step_state: moss_adamw_state_bytes() zeroed device bytes owned by this optimizer (not shared with a moss_adamw_flat_* call
 *   of the same step: each launch that is given the block advances the count).
 * opt == NULL or opt->tensors == 0: exactly moss_raster_backward_raw.
 */
#define MOSS_OPT_MEANS 1
#define MOSS_OPT_SH 2
#define MOSS_OPT_OPACITY 4
#define MOSS_OPT_SCALES 8
#define MOSS_OPT_ROTATIONS 16
typedef struct moss_fused_adamw {
    uint32_t tensors;            /* MOSS_OPT_* bits: which parameters this call updates */
    float* exp_avg[5];           /* first moments, same shapes as the parameters; order: means, sh, opacity, scales, rotations */
    float* exp_avg_sq[5];        /* second moments */
    float lr[5];                 /* learning rates, same order; sh: of a Gaussian's first 3 floats (MOSS's features_dc group) */
    float lr_sh_rest;            /* sh: of the other 45 floats of a record (features_rest) */
    double beta1, beta2;         /* doubles (ABI 4): the kernels use float(beta) and float(1 - beta), rounded independently like torch's */
    float eps, weight_decay;
    void* step_state;
    int32_t lr_segment[5];       /* per tensor: its entry s (0..7) in the step-state block's learning-rate table, or -1; when word
                                  * MOSS_ADAMW_LR_VALID_WORD of step_state is non-zero the kernel reads lr from float word
                                  * MOSS_ADAMW_LR_WORD0 + s (sh: lr_sh_rest from MOSS_ADAMW_LR2_WORD0 + s) instead of from this struct */
    int32_t sh_active_degree;    /* ABI 6.  The HIGHEST SH degree that has ever been active for these parameters (0..3; the call's own
                                  * D is the floor; 3 = everything is active, the behaviour before ABI 6).  Coefficients above it have
                                  * never received a gradient: their moments are exactly zero and are neither read nor written, and
                                  * their parameters take the weight decay alone -- bit for bit the full update's result.  MOSS: degree
                                  * 0 / 1 / 2 for iterations 1-2999 (train_ZJU.py:85-86). */
    int32_t sh_inactive_zero;    /* != 0: the caller also knows those parameters to be exactly ZERO (features_rest starts as zeros,
                                  * scene/gaussian_model.py:179-181; 0 x decay = 0): they are not read or written at all */
} moss_fused_adamw;
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
    const moss_fused_adamw* opt, int raw_flags, int debug, void* stream);

/*
 * Gaussian parameter activations, forward and backward, one launch each (the rasterizer-facing getters of MOSS's GaussianModel,
 * scene/gaussian_model.py:46-53 and :134-166: get_xyz identity, get_features = cat(_features_dc, _features_rest, dim=1),
 * get_opacity = sigmoid, get_scaling = exp, get_rotation = F.normalize (eps 1e-12)).  K = SH coefficients per channel
 * ((max_sh_degree+1)^2); features_dc is (P,1,3), features_rest (P,K-1,3), out_features (P,K,3); all fp32, contiguous.
 * K = 0: the features are not touched (callers that keep them as one (P,K,3) tensor need no concatenation).
 * Backward: a NULL incoming gradient means that output was unused (its parameter gets zeros); every element of every d_*
 * array is written exactly once, so destinations (e.g. slices of a flat gradient bucket) need no zero fill.
 */
int moss_gaussian_activate_forward(int P, int K, const float* xyz, const float* features_dc, const float* features_rest,
                                   const float* opacity, const float* scaling, const float* rotation,
                                   float* out_xyz, float* out_features, float* out_opacity, float* out_scaling,
                                   float* out_rotation, void* stream);
int moss_gaussian_activate_backward(int P, int K, const float* rotation, const float* out_opacity, const float* out_scaling,
                                    const float* g_xyz, const float* g_features, const float* g_opacity,
                                    const float* g_scaling, const float* g_rotation,
                                    float* d_xyz, float* d_features_dc, float* d_features_rest, float* d_opacity,
                                    float* d_scaling, float* d_rotation, void* stream);

/* ---- inspection entry points (used by the parity tests; not needed by a caller of the op) ---------------- */

/* Scratch sizes this library will request for a given problem (host-only arithmetic, no GPU touched). */
size_t moss_raster_geometry_bytes(int P);
size_t moss_raster_image_bytes(int width, int height);
size_t moss_raster_binning_bytes(int R);
size_t moss_raster_binning_bytes_forward_only(int R);   /* the binning buffer of a MOSS_FORWARD_ONLY forward (ABI 6) */

/*
 * Re-express the opaque geometry buffer in the reference's GeometryState terms
 * (DGR/cuda_rasterizer/rasterizer_impl.cu:155-170); any output pointer may be NULL.
 *   depths (P), means2D (P,2), conic_opacity (P,4), rgb (P,3), tiles_touched (P) u32, clamped (P,3) bytes,
 *   cov3D (P,6; only meaningful when forward computed it from scales/rotations).
 * Entries of culled Gaussians read as zero.
 */
int moss_raster_export_geometry(const char* geom_buffer, int P,
    float* depths, float* means2D, float* conic_opacity, float* rgb, uint32_t* tiles_touched,
    uint8_t* clamped, float* cov3D, void* stream);

/*
 * Re-express the opaque binning + image buffers in the reference's BinningState / ImageState terms
 * (rasterizer_impl.cu:172-194): the SORTED 64-bit keys (tile << 32 | depth bits), the sorted Gaussian ids,
 * the per-tile ranges (tiles,2) u32, and per pixel final_T / n_contrib.  Any output pointer may be NULL.
 */
int moss_raster_export_binning(const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    int P, int R, int width, int height,
    uint64_t* point_list_keys, uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib,
    void* stream);

/*
 * Per-stage device timing, measured with HIP events recorded on the stream each stage is launched on (so it sees exactly
 * what a profiler's kernel trace sees, minus nothing).  Enable a set of stages (bit i = stage i), run, then read:
 * ms_sum[i] = summed duration of stage i over count[i] executions since the last read; reading synchronises those events.
 */
#define MOSS_STAGE_PREPROCESS_FWD 0
#define MOSS_STAGE_SCAN           1
#define MOSS_STAGE_SCATTER        2
#define MOSS_STAGE_TILE_SORT      3   /* the chunk sort kernel */
#define MOSS_STAGE_BLEND_FWD      4
#define MOSS_STAGE_BLEND_BWD      5
#define MOSS_STAGE_PREPROCESS_BWD 6
#define MOSS_STAGE_MERGE_GATHER   7   /* the second kernel of the tile sort: rank merge + per-instance emit */
#define MOSS_NUM_STAGES           8
void moss_raster_profile_enable(uint32_t stage_mask);
int moss_raster_profile_read(float* ms_sum /* [MOSS_NUM_STAGES] */, uint32_t* count /* [MOSS_NUM_STAGES] */);

/* 1 if the library was built with -DMOSS_DIAG (python -m moss_amd.build --diag -> moss_amd/lib_diag/): it then reads MOSS_*
 * environment knobs that select kernel variants for A/B timing -- some give WRONG results on purpose -- and exports the stamp
 * entry points below.  The product build returns 0, reads no environment variable and does not export them. */
int moss_build_has_diagnostics(void);
#ifdef MOSS_DIAG
/* register a device buffer of 8 x (4 * padded tile count) uint64; while set, the forward blend kernel stores per item
 * {total, list length, starve, first trip, begin, trip cycles, rounds, trips} of its blender wave.  NULL = off. */
void moss_raster_debug_set_stamps(unsigned long long* device_buffer);
/* the same for the backward blend kernel: 16 words per wave (start, end of the segment phase, end, item counts and cycle sums) */
void moss_raster_debug_set_bwd_stamps(unsigned long long* device_buffer);
#endif

#ifdef __cplusplus
}
#endif
#endif /* MOSS_RASTER_H */
