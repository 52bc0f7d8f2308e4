// loss.hip -- fused photometric loss of the training step and its gradient:
//     loss = mean|x - y| + lambda_mask * mean((alpha - mask)^2) + lambda_dssim * (1 - mean SSIM(x, y))
// x = rendered image (C,H,W), y = ground truth; SSIM exactly as the reference's utils/loss_utils.py:47-87: 11x11 Gaussian
// window (sigma 1.5, normalised in fp32), zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2; the combination and weights
// are those of train_ZJU.py:111-112,119,131.
//
// The reference runs five depthwise 11x11 convolutions + ~20 elementwise kernels forward and their autograd mirror
// backward (measured here with MIOpen: 8 x 178 us + ~60 launches per step).  Here: two launches.
//   pass 1: per 32x32 tile and channel, x and y tiles (+5 px halo) go to LDS once, the five windowed moments
//           (E[x], E[y], E[x^2], E[y^2], E[xy]) are formed separably (horizontal then vertical 11-tap), SSIM and its partial
//           derivatives w.r.t. the three x-dependent moments are evaluated per pixel; the derivative maps are written
//           (3 floats per pixel-channel) and the block's loss sums go to a partials array (fixed-order => deterministic);
//   pass 2: the derivative maps are filtered with the same (symmetric) window, again separably through LDS, and combined
//           into dL/dx = -lambda/N * (w*D_mu + 2x (w*D_xx) + y (w*D_xy)) + sign(x-y)/N; dL/dalpha alongside; block 0 folds
//           the partials into the scalar losses.
// Both passes are HBM-streaming: ~ (2 + 3) * 4 B read/written per pixel-channel in pass 1, (3 + 2 + 1) * 4 B in pass 2.
//
// ROI instantiation (moss_photometric_loss_roi): MOSS's OWN expression, train_ZJU.py:108-119 -- the L1 and the mask term are means over
// the pixels of the view's `bound_mask` (the projected 3-D box of the body), SSIM is taken on the crop cv2.boundingRect(bound_mask) of
// both images (zero padding at the CROP's edges).  Same tiles (image-aligned), same two launches: "inside" means inside the crop, the
// sums are weighted by the mask, tiles off the crop leave at once with zero gradients.  The rectangle and the mask's pixel count are
// read from DEVICE memory (5 ints), so that a captured step can change view by updating them in place.
#include "common.h"

namespace moss {

namespace {

constexpr int LT = 32;              // tile edge: 32 x 32 outputs per 256-thread workgroup
constexpr int HALO = 5;             // window 11
constexpr int LP = LT + 2 * HALO;   // 42
constexpr int SEG = 8;              // outputs per thread in the horizontal pass (a run of 8 in one row: 18 inputs)
constexpr int VR = 4;               // outputs per thread in the vertical pass (4 consecutive rows of one column: 14 inputs)
static_assert(LT % SEG == 0 && (LT / SEG) * LP <= 256 && LT * (LT / VR) == 256, "work split of a 256-thread workgroup");

struct Win { float g[11]; };
typedef float v2f __attribute__((ext_vector_type(2)));

// the crop [x0, x1) x [y0, y1) clipped to the image, its pixel count and the number of `bound` pixels (= the crop's when there is no mask)
struct Crop { int x0, y0, x1, y1; float n_pix, n_bound; };
__device__ __forceinline__ Crop load_crop(const int* __restrict__ rect, int W, int H, bool has_bound)
{
    Crop r;
    const int x = rect[0], y = rect[1], w = max(rect[2], 0), h = max(rect[3], 0);
    const long long xe = (long long)x + w, ye = (long long)y + h;     // (the rectangle's far edges, before clipping: x may be negative)
    r.x0 = min(max(x, 0), W); r.y0 = min(max(y, 0), H);
    r.x1 = (int)min(max(xe, (long long)r.x0), (long long)W); r.y1 = (int)min(max(ye, (long long)r.y0), (long long)H);
    r.n_pix = (float)(r.x1 - r.x0) * (float)(r.y1 - r.y0);
    r.n_bound = has_bound ? (float)rect[4] : r.n_pix;
    return r;
}
__device__ __forceinline__ bool in_crop(const Crop& r, int x, int y) { return x >= r.x0 && x < r.x1 && y >= r.y0 && y < r.y1; }

__device__ __forceinline__ float ld0(const float* __restrict__ p, int x, int y, int W, int H)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.0f;
}


// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2: with the natural
// order, a tile's four neighbours -- which re-read its 5-pixel halo -- run on four other XCDs and every halo is fetched from HBM
// again (PMC: 39 / 52 MB per launch against 16 / 25 MB of distinct data).  Here XCD k takes the k-th contiguous eighth of the
// (channel, row, column) tile sequence, so neighbouring tiles share an L2.
struct TileId { int bx, by, c; };
__device__ __forceinline__ TileId xcd_tile()
{
    const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
    const int lin = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    const int q = total / 8, r = total % 8, xcd = lin % 8, j = lin / 8;
    const int t = xcd * q + min(xcd, r) + j;
    TileId id;
    id.bx = t % (int)gridDim.x; id.by = (t / (int)gridDim.x) % (int)gridDim.y; id.c = t / (int)(gridDim.x * gridDim.y);
    return id;
}

// The crop instantiations: the tiles that meet the crop FIRST.  Workgroups are dispatched in linear-id order as slots free up; at 1024 x 1024
// the grid takes three residency rounds and, in image order, the crop's tiles (a person in the middle of the frame) are handed out behind
// the empty ones in front of them.  Here XCD x (linear id % 8) is dealt, in its turns j = id / 8, first the x-th contiguous eighth of the
// crop's tiles -- (channel, row, column) order inside the crop's tile rectangle: neighbours still share an L2 -- and then its share of the
// tiles off the crop (which leave at once).
__device__ __forceinline__ TileId roi_tile(const Crop& cr)
{
    const int gx = (int)gridDim.x, gy = (int)gridDim.y, C = (int)gridDim.z;
    const int total = gx * gy * C;
    const int lin = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    const bool some = cr.x1 > cr.x0 && cr.y1 > cr.y0;
    const int tx0 = cr.x0 / LT, ty0 = cr.y0 / LT;
    const int ncx = some ? (cr.x1 - 1) / LT - tx0 + 1 : 0, ncy = some ? (cr.y1 - 1) / LT - ty0 + 1 : 0;
    const int n_crop = ncx * ncy, n_in = n_crop * C;
    const int x = lin % 8, j = lin / 8;
    const int q = n_in / 8, r = n_in % 8, qt = total / 8, rt = total % 8;
    const int cnt = q + (x < r ? 1 : 0);                     // crop tiles dealt to this XCD
    TileId id;
    if (j < cnt) {
        const int ci = x * q + min(x, r) + j;
        const int rem = ci % n_crop;
        id.c = ci / n_crop; id.by = ty0 + rem / ncx; id.bx = tx0 + rem % ncx;
        return id;
    }
    // the (oi)-th tile off the crop, in (channel, row, column) order: the rows above the crop, the crop's rows without its columns, the rows below
    const int oi = (x * qt + min(x, rt)) - (x * q + min(x, r)) + (j - cnt);
    const int n_off = gx * gy - n_crop;                      // per channel (> 0 here: some block is left over for this branch)
    id.c = oi / n_off;
    int k = oi % n_off;
    const int above = ty0 * gx, side = gx - ncx;
    if (!some || k < above) { id.by = k / gx; id.bx = k % gx; return id; }
    k -= above;
    if (side > 0 && k < ncy * side) {
        const int cc = k % side;
        id.by = ty0 + k / side; id.bx = cc < tx0 ? cc : cc + ncx;
        return id;
    }
    k -= ncy * side;
    id.by = ty0 + ncy + k / gx; id.bx = k % gx;
    return id;
}

// Both passes are separable 11-tap filters through LDS with SLIDING WINDOWS in registers (round 2: one output per thread and pass,
// 11 LDS reads per output and moment, products recomputed per tap -- 7.8 wave-instructions per pixel-channel, 30 % of the wave
// cycles waiting on LDS).  Horizontal: a thread owns a run of SEG outputs of one row, walks its SEG + 10 inputs once and adds each
// into the outputs whose window holds it.  Vertical: a thread owns VR consecutive rows of one column.  32 x 32 tiles: the halo costs
// 1.7x instead of 2.6x.

template <bool SKIP_EMPTY, bool ROI>   // (SKIP_EMPTY: see EMPTY TILES below, the instantiation for grids of more than one residency round)
__global__ void __launch_bounds__(256)
ssim_pass1_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt, Win win,
                  float* __restrict__ dmap /* [3][C][H][W] */, float* __restrict__ partials /* [blocks][2] */,
                  const float* __restrict__ alpha, const float* __restrict__ mask, float lambda_mask,
                  float* __restrict__ dL_dalpha, float* __restrict__ mask_partials /* [tiles] */,
                  unsigned long long* __restrict__ stamps /* diagnostics (MOSS_LOSS_STAMPS, -DMOSS_DIAG builds): 8 words per workgroup, else NULL */,
                  const unsigned char* __restrict__ bound /* ROI: (H,W) 0/1 or NULL */, const int* __restrict__ rect /* ROI: device, x y w h count */)
{
#define LSTAMP(i) if (stamps && threadIdx.x == 0) stamps[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
    LSTAMP(0);
    __shared__ float s_x[LP][LP + 1];
    __shared__ float s_y[LP][LP + 1];
    __shared__ float s_h[4][LP][LT + 1];                 // (four moments, below: 36.7 KB of LDS in all -- FOUR workgroups per CU; five: 42.2 KB, three)
    __shared__ float s_red[3][4];

    Crop crop = {};
    if constexpr (ROI) crop = load_crop(rect, W, H, bound != nullptr);
    const TileId tile = ROI ? roi_tile(crop) : xcd_tile();
    const int c = tile.c;
    const int x0 = tile.bx * LT, y0 = tile.by * LT;
    const int tid = threadIdx.x;
    const float* xc = img + (size_t)c * H * W;
    const float* yc = gt + (size_t)c * H * W;
    bool nonzero = false;
    if constexpr (ROI) {
        if (x0 >= crop.x1 || x0 + LT <= crop.x0 || y0 >= crop.y1 || y0 + LT <= crop.y0) {
            // a tile off the crop: no SSIM term, no pixel of the mask (MOSS's rectangle is the mask's bounding box; a mask pixel outside
            // the rectangle handed in here counts for nothing): zero sums, zero alpha gradient; pass 2 never reads its derivative maps
            if (c == 0 && alpha != nullptr) {
                const int lx_ = tid % LT, ly_ = (tid / LT) * VR;
#pragma unroll
                for (int j = 0; j < VR; j++)
                    if (x0 + lx_ < W && y0 + ly_ + j < H) dL_dalpha[(size_t)(y0 + ly_ + j) * W + x0 + lx_] = 0.0f;
            }
            if (tid == 0) {
                const size_t b = ((size_t)tile.c * gridDim.y + tile.by) * gridDim.x + tile.bx;
                partials[2 * b] = 0.0f; partials[2 * b + 1] = 0.0f;
                if (c == 0 && alpha != nullptr) mask_partials[tile.by * gridDim.x + tile.bx] = 0.0f;
            }
            return;
        }
    }

    {
        // the tile + halo: all loads first (clamped addresses, no branch: they are in flight together), then the LDS stores
        constexpr int NLD = (LP * LP + 255) / 256;
        float vx[NLD], vy[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + 256 * k, r = i / LP, q = i % LP;
            const size_t o = (size_t)min(max(y0 + r - HALO, 0), H - 1) * W + min(max(x0 + q - HALO, 0), W - 1);
            vx[k] = xc[o]; vy[k] = yc[o];
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + 256 * k, r = i / LP, q = i % LP;
            const int gx_ = x0 + q - HALO, gy_ = y0 + r - HALO;
            const bool in = ROI ? in_crop(crop, gx_, gy_) : (gx_ >= 0 && gx_ < W && gy_ >= 0 && gy_ < H);
            if (i < LP * LP) {
                const float a = in ? vx[k] : 0.0f, b = in ? vy[k] : 0.0f;
                s_x[r][q] = a; s_y[r][q] = b;
                if constexpr (SKIP_EMPTY) nonzero |= (a != 0.0f) || (b != 0.0f);       // (NaN counts as content)
            }
        }
    }
    LSTAMP(1);
    // EMPTY TILES.  MOSS's frames are masked people on black: three quarters of a frame's tiles are exactly zero in BOTH images, halo
    // included.  Every windowed moment of such a tile is exactly +0 (sums of w * 0 from +0), so the two filter passes are skipped and the
    // epilogue runs on zero moments: the same bits, the same stores.  At 512 x 512 all 768 workgroups are resident at once and the
    // kernel lasts as long as one non-empty workgroup either way (measured in rounds 4 and 5: 18.8 -> 19.1 us for the two kernels, a dense
    // frame 19.1 -> 19.9): there the plain instantiation runs.  At 1024 x 1024 -- MOSS's ZJU-MoCap resolution, BASELINE configs[4] -- the
    // grid takes four rounds and the empty tiles leave at once: 56.9 -> 50.5 us on a masked frame (a dense one: 60.1 -> 61.2).
    bool tile_nonzero = true;
    if constexpr (SKIP_EMPTY) tile_nonzero = __syncthreads_or(nonzero ? 1 : 0) != 0;
    else __syncthreads();
    LSTAMP(2);
    if (tile_nonzero && tid < (LT / SEG) * LP) {                         // horizontal 11-tap for the 5 moments: rows run along the lanes (odd row
        const int r = tid % LP, q0 = (tid / LP) * SEG;   // stride: conflict-free LDS reads), a thread owns columns q0 .. q0 + SEG - 1
        // FOUR windowed moments, in two pairs (v_pk_fma_f32 does a pair per instruction): E[x], E[y], E[x^2 + y^2], E[xy].  SSIM needs
        // the two variances only as their SUM (b2 = sigma1^2 + sigma2^2 + C2), and so do its derivatives (dS/dsigma1^2 = dS/db2): rounds
        // 1-4 filtered x^2 and y^2 separately -- a fifth moment, a third instruction per tap, and the 5.5 KB of LDS that kept this kernel
        // at three workgroups per CU.
        v2f a01[SEG], a23[SEG];
#pragma unroll
        for (int j = 0; j < SEG; j++) { a01[j] = v2f{0.f, 0.f}; a23[j] = v2f{0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < SEG + 10; i++) {
            const float a = s_x[r][q0 + i], b = s_y[r][q0 + i];
            const v2f ab = v2f{a, b}, sq = ab * ab;
            const v2f sx = v2f{sq.x + sq.y, a * b};                                // (x^2 + y^2, x y)
#pragma unroll
            for (int j = 0; j < SEG; j++) {
                if (i - j >= 0 && i - j <= 10) {
                    const float w = win.g[i - j];
                    const v2f w2 = v2f{w, w};
                    a01[j] = __builtin_elementwise_fma(w2, ab, a01[j]); a23[j] = __builtin_elementwise_fma(w2, sx, a23[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SEG; j++) {
            s_h[0][r][q0 + j] = a01[j].x; s_h[1][r][q0 + j] = a01[j].y; s_h[2][r][q0 + j] = a23[j].x; s_h[3][r][q0 + j] = a23[j].y;
        }
    }
    __syncthreads();
    LSTAMP(3);
    const int lx = tid % LT, ly0 = (tid / LT) * VR;      // vertical: column lx, rows ly0 .. ly0 + VR - 1
    v2f m01[VR], m23[VR];
#pragma unroll
    for (int j = 0; j < VR; j++) { m01[j] = v2f{0.f, 0.f}; m23[j] = v2f{0.f, 0.f}; }
    if (tile_nonzero) {
#pragma unroll
    for (int i = 0; i < VR + 10; i++) {
        const v2f v01 = v2f{s_h[0][ly0 + i][lx], s_h[1][ly0 + i][lx]}, v23 = v2f{s_h[2][ly0 + i][lx], s_h[3][ly0 + i][lx]};
#pragma unroll
        for (int j = 0; j < VR; j++) {
            if (i - j >= 0 && i - j <= 10) {
                const float w = win.g[i - j];
                const v2f w2 = v2f{w, w};
                m01[j] = __builtin_elementwise_fma(w2, v01, m01[j]); m23[j] = __builtin_elementwise_fma(w2, v23, m23[j]);
            }
        }
    }
    }
    float ssim_v = 0.f, l1_v = 0.f, mask_v = 0.f;
    const float inv_hw = ROI ? 1.0f / crop.n_bound : 1.0f / ((float)H * (float)W);
    const int px = x0 + lx;
#pragma unroll
    for (int j = 0; j < VR; j++) {
        const int ly = ly0 + j, py = y0 + ly;
        if (ROI ? in_crop(crop, px, py) : (px < W && py < H)) {
            const float mu1 = m01[j].x, mu2 = m01[j].y, ess = m23[j].x /* E[x^2 + y^2] */, exy = m23[j].y;
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s12 = exy - mu12;
            // sigma1^2 + sigma2^2 = E[x^2 + y^2] - mu1^2 - mu2^2 (utils/loss_utils.py:73-75 forms the two variances and adds them)
            const float a1 = 2.f * mu12 + C1, a2 = 2.f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = (ess - mu1_sq - mu2_sq) + C2;
            // (hardware reciprocals, ~1 ulp: this file is compiled with correctly rounded division, ten instructions apiece, and the
            // four quotients per pixel-channel were an eighth of this kernel's vector instructions; b1 >= C1, b2 ~ C2 + variances)
            const float rb1 = __builtin_amdgcn_rcpf(b1), rb2 = __builtin_amdgcn_rcpf(b2), inv = rb1 * rb2;
            const float S = a1 * a2 * inv;
            // partial derivatives of S w.r.t. (mu1 | sigma1^2 | sigma12), then total derivatives w.r.t. the three filtered
            // moments E[x], E[x^2], E[xy] (sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2)
            const float dS_ds1 = -S * rb2;
            const float dS_ds12 = 2.f * a1 * inv;
            const float dS_dmu1 = 2.f * mu2 * a2 * inv - S * 2.f * mu1 * rb1 + dS_ds1 * (-2.f * mu1) + dS_ds12 * (-mu2);
            const size_t o = ((size_t)c * H + py) * W + px, plane3 = (size_t)C * H * W;
            dmap[o] = dS_dmu1; dmap[plane3 + o] = dS_ds1; dmap[2 * plane3 + o] = dS_ds12;
            ssim_v += S;
            float wb = 1.0f;                                                       // (ROI: the L1 and mask terms count the mask's pixels only)
            if constexpr (ROI) { if (bound != nullptr) wb = bound[(size_t)py * W + px] ? 1.0f : 0.0f; }
            l1_v += wb * fabsf(s_x[ly + HALO][lx + HALO] - s_y[ly + HALO][lx + HALO]);
            // the alpha-vs-mask L2 term rides on channel 0's tiles; it is computed HERE (not in pass 2, where its gradient would
            // fit just as well) so that everything pass 2's closing fold reads was written by an earlier kernel
            if (c == 0 && alpha != nullptr) {
                const size_t oa = (size_t)py * W + px;
                const float da = wb * (alpha[oa] - mask[oa]);
                mask_v += da * da;
                dL_dalpha[oa] = lambda_mask * 2.f * da * inv_hw;
            }
        } else if (ROI && px < W && py < H && c == 0 && alpha != nullptr) {
            dL_dalpha[(size_t)py * W + px] = 0.0f;
        }
    }
    LSTAMP(4);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ssim_v += __shfl_xor(ssim_v, d); l1_v += __shfl_xor(l1_v, d); mask_v += __shfl_xor(mask_v, d); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = ssim_v; s_red[1][tid >> 6] = l1_v; s_red[2][tid >> 6] = mask_v; }
    __syncthreads();
    LSTAMP(5);
    if (tid == 0) {
        const size_t b = ((size_t)tile.c * gridDim.y + tile.by) * gridDim.x + tile.bx;
        partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        if (c == 0 && alpha != nullptr) mask_partials[tile.by * gridDim.x + tile.bx] = (s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3]);
    }
}

template <bool SKIP_EMPTY, bool ROI>
__global__ void __launch_bounds__(256)
ssim_pass2_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                  const float* __restrict__ alpha, const float* __restrict__ mask, Win win,
                  const float* __restrict__ dmap, const float* __restrict__ partials, int nblocks,
                  float lambda_dssim, float lambda_mask, float* __restrict__ dL_dimg, float* __restrict__ dL_dalpha,
                  const float* __restrict__ mask_partials, float* __restrict__ loss_out, float lambda_l1,
                  unsigned long long* __restrict__ stamps, const unsigned char* __restrict__ bound, const int* __restrict__ rect)
{

    LSTAMP(0);
    __shared__ float s_d[3][LP][LP + 1];
    __shared__ float s_h[3][LP][LT + 1];
    __shared__ float s_red[3][4];

    Crop crop = {};
    if constexpr (ROI) crop = load_crop(rect, W, H, bound != nullptr);
    const TileId tile = ROI ? roi_tile(crop) : xcd_tile();
    const int c = tile.c;
    const int x0 = tile.bx * LT, y0 = tile.by * LT;
    const int tid = threadIdx.x;
    const size_t plane3 = (size_t)C * H * W;
    bool off_crop = false;                               // (block-uniform)
    if constexpr (ROI) {
        off_crop = x0 >= crop.x1 || x0 + LT <= crop.x0 || y0 >= crop.y1 || y0 + LT <= crop.y0;
    }
    const float N = ROI ? (float)C * crop.n_pix : (float)C * (float)H * (float)W;           // pixels-channels of the SSIM mean
    const float N1 = ROI ? (float)C * crop.n_bound : N;                                      // ... of the L1 mean

    // this thread's output pixels (column lx, rows ly0 .. ly0 + VR - 1): their x and y are requested with the maps (one round trip)
    const int lx = tid % LT, ly0 = (tid / LT) * VR;
    const int px = x0 + lx;
    float xs[VR] = {}, ys[VR] = {};
    bool content = false;
    if (!off_crop) {
        constexpr int NLD = (LP * LP + 255) / 256;
        float v[3][NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + 256 * k, r = i / LP, q = i % LP;
            const size_t o = (size_t)c * H * W + (size_t)min(max(y0 + r - HALO, 0), H - 1) * W + min(max(x0 + q - HALO, 0), W - 1);
            v[0][k] = dmap[o]; v[1][k] = dmap[plane3 + o]; v[2][k] = dmap[2 * plane3 + o];
        }
        if constexpr (SKIP_EMPTY) {
#pragma unroll
            for (int j = 0; j < VR; j++) {
                const size_t o = ((size_t)c * H + min(y0 + ly0 + j, H - 1)) * W + min(px, W - 1);
                xs[j] = img[o]; ys[j] = gt[o];
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + 256 * k, r = i / LP, q = i % LP;
            const int gx_ = x0 + q - HALO, gy_ = y0 + r - HALO;
            const bool in = ROI ? in_crop(crop, gx_, gy_) : (gx_ >= 0 && gx_ < W && gy_ >= 0 && gy_ < H);   // (ROI: pass 1 wrote the maps inside the crop only)
            if (i < LP * LP) {
                const float d0 = in ? v[0][k] : 0.0f;
                s_d[0][r][q] = d0; s_d[1][r][q] = in ? v[1][k] : 0.0f; s_d[2][r][q] = in ? v[2][k] : 0.0f;
                if constexpr (SKIP_EMPTY) content |= d0 != 0.0f;       // (NaN counts as content)
            }
        }
        if constexpr (SKIP_EMPTY) {
#pragma unroll
            for (int j = 0; j < VR; j++) content |= (xs[j] != 0.0f) || (ys[j] != 0.0f);
        }
    }
    LSTAMP(1);
    // EMPTY TILES (see pass 1).  Where the first derivative map is zero on the whole window (the moments mu1, mu2 vanish there: nothing
    // but black within 10 pixels) and x = y = 0 on the tile, the gradient is  (w * 0) + 2 * 0 * (w * D_xx) + 0 * (w * D_xy)  =  +0  for any
    // finite second and third map -- and the filters below, run on such a window, give exactly that (a sum of w * (+-0) from +0 is +0;
    // +0 + (+-0) = +0): skipped, same bits.
    bool tile_content = true;
    if constexpr (SKIP_EMPTY) tile_content = __syncthreads_or(content ? 1 : 0) != 0;
    else { __syncthreads(); if constexpr (ROI) tile_content = !off_crop; }      // (a tile off the crop loaded nothing: zeros are written below)
    LSTAMP(2);
    if (tile_content && tid < (LT / SEG) * LP) {         // horizontal pass of the three derivative maps (see pass 1)
        const int r = tid % LP, q0 = (tid / LP) * SEG;
        v2f a01[SEG]; float a2[SEG];
#pragma unroll
        for (int j = 0; j < SEG; j++) { a01[j] = v2f{0.f, 0.f}; a2[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < SEG + 10; i++) {
            const v2f v01 = v2f{s_d[0][r][q0 + i], s_d[1][r][q0 + i]};
            const float v2 = s_d[2][r][q0 + i];
#pragma unroll
            for (int j = 0; j < SEG; j++) {
                if (i - j >= 0 && i - j <= 10) {
                    const float w = win.g[i - j];
                    a01[j] = __builtin_elementwise_fma(v2f{w, w}, v01, a01[j]); a2[j] = __fmaf_rn(w, v2, a2[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SEG; j++) { s_h[0][r][q0 + j] = a01[j].x; s_h[1][r][q0 + j] = a01[j].y; s_h[2][r][q0 + j] = a2[j]; }
    }
    __syncthreads();
    LSTAMP(3);
    v2f f01[VR]; float f2[VR];
#pragma unroll
    for (int j = 0; j < VR; j++) { f01[j] = v2f{0.f, 0.f}; f2[j] = 0.f; }
    if (tile_content) {
#pragma unroll
    for (int i = 0; i < VR + 10; i++) {
        const v2f v01 = v2f{s_h[0][ly0 + i][lx], s_h[1][ly0 + i][lx]};
        const float v2 = s_h[2][ly0 + i][lx];
#pragma unroll
        for (int j = 0; j < VR; j++) {
            if (i - j >= 0 && i - j <= 10) {
                const float w = win.g[i - j];
                f01[j] = __builtin_elementwise_fma(v2f{w, w}, v01, f01[j]); f2[j] = __fmaf_rn(w, v2, f2[j]);
            }
        }
    }
    }
#pragma unroll
    for (int j = 0; j < VR; j++) {
        const int py = y0 + ly0 + j;
        if (px < W && py < H) {
            const size_t o = ((size_t)c * H + py) * W + px;
            const float x = SKIP_EMPTY ? xs[j] : img[o], y = SKIP_EMPTY ? ys[j] : gt[o];
            const float dssim = f01[j].x + 2.f * x * f01[j].y + y * f2[j];     // d(sum SSIM)/dx
            const float d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            if constexpr (ROI) {
                const float wb = (bound == nullptr || bound[(size_t)py * W + px]) ? 1.0f : 0.0f;
                dL_dimg[o] = in_crop(crop, px, py) ? (lambda_l1 * sgn * wb) / N1 - lambda_dssim * dssim / N : 0.0f;
            } else {
                dL_dimg[o] = (lambda_l1 * sgn) / N - lambda_dssim * dssim / N; // (lambda_l1 = 1: the same bits as sgn / N)
            }
        }
    }
    LSTAMP(4);
    // Block (0,0,0) folds pass 1's partials into the four loss terms (no separate "finish" launch: a minimal launch costs 4-5 us).
    // Everything it reads was written by pass 1, an earlier kernel -- which is why the mask term is computed there.
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        const int nmask = alpha != nullptr ? (int)(gridDim.x * gridDim.y) : 0;
        float a = 0.f, b = 0.f, m = 0.f;
        for (int i = tid; i < nblocks; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
        for (int i = tid; i < nmask; i += 256) m += mask_partials[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); m += __shfl_xor(m, d); }
        __syncthreads();
        if ((tid & 63) == 0) { s_red[0][tid >> 6] = m; s_red[1][tid >> 6] = a; s_red[2][tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) {
            const float ssim_mean = ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3])) / N;
            const float l1_mean = ((s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3])) / N1;
            const float mask_mean = ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) / (ROI ? crop.n_bound : (float)H * (float)W);
            const float lm = alpha != nullptr ? lambda_mask : 0.0f;
            loss_out[1] = l1_mean; loss_out[2] = ssim_mean; loss_out[3] = mask_mean;
            loss_out[0] = lambda_l1 * l1_mean + lm * mask_mean + lambda_dssim * (1.0f - ssim_mean);
        }
    }
}

#undef LSTAMP

int loss_device_cus()
{
    static const int n = [] {
        int dev = 0; hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return n;
}

Win make_window()
{
    // utils/loss_utils.py:47-49: gauss = Tensor([exp(-(x-5)^2 / (2*1.5^2))]) / sum, evaluated in fp32 like torch.Tensor
    Win w; float sum = 0.f;
    for (int i = 0; i < 11; i++) { w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += w.g[i]; }
    for (int i = 0; i < 11; i++) w.g[i] /= sum;
    return w;
}

}  // anonymous namespace
}  // namespace moss

using namespace moss;

extern "C" size_t moss_loss_workspace_bytes(int C, int H, int W)
{
    const size_t gx = (W + LT - 1) / LT, gy = (H + LT - 1) / LT;
    return align_up(3 * (size_t)C * H * W * 4) + align_up(gx * gy * C * 2 * 4) + align_up(gx * gy * 4);
}

extern "C" int moss_photometric_loss_weighted(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                              float lambda_l1, float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage,
                                              float* dL_dalpha, char* workspace, size_t workspace_bytes, void* stream);

extern "C" int moss_photometric_loss(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                     float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage, float* dL_dalpha,
                                     char* workspace, size_t workspace_bytes, void* stream)
{
    return moss_photometric_loss_weighted(C, H, W, image, gt, alpha, mask, 1.0f, lambda_dssim, lambda_mask, loss_out, dL_dimage, dL_dalpha,
                                          workspace, workspace_bytes, stream);
}

// The same two kernels with a weight on the L1 term as well: total = lambda_l1 L1 + lambda_mask maskL2 + lambda_dssim (1 - SSIM).
// lambda_l1 = 0, lambda_dssim = 1, no alpha: loss_out[2] is the reference's ssim(img1, img2) (utils/loss_utils.py:47-87) and
// -dL_dimage its gradient -- what moss_amd.loss.ssim_fused hands MOSS's own loss expression in place of five MIOpen convolutions.
extern "C" int moss_photometric_loss_weighted(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                              float lambda_l1, float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage,
                                              float* dL_dalpha, char* workspace, size_t workspace_bytes, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !image || !gt || !loss_out || !dL_dimage || !workspace) return MOSS_ERR_INVALID_ARG;
    if ((alpha == nullptr) != (mask == nullptr) || (alpha && !dL_dalpha)) return MOSS_ERR_INVALID_ARG;
    if (workspace_bytes < moss_loss_workspace_bytes(C, H, W)) return MOSS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + LT - 1) / LT, gy = (H + LT - 1) / LT;
    char* p = workspace;
    float* dmap = carve<float>(p, 3 * (size_t)C * H * W);
    float* partials = carve<float>(p, (size_t)gx * gy * C * 2);
    float* mask_partials = carve<float>(p, (size_t)gx * gy);
    static const Win win = make_window();
    unsigned long long* const loss_stamps = (g_stamps && knob("MOSS_LOSS_STAMPS", 0)) ? g_stamps : nullptr;      // (product build: constant NULL)
    const dim3 grid(gx, gy, C);
    // more workgroups than are resident at once (three per CU: 42 KB of LDS each): the instantiations that let empty tiles leave early
    if ((size_t)gx * gy * C > (size_t)3 * loss_device_cus()) {
        hipLaunchKernelGGL((ssim_pass1_kernel<true, false>), grid, dim3(256), 0, s, C, H, W, image, gt, win, dmap, partials, alpha, mask, lambda_mask, dL_dalpha, mask_partials, loss_stamps,
                           (const unsigned char*)nullptr, (const int*)nullptr);
        hipLaunchKernelGGL((ssim_pass2_kernel<true, false>), grid, dim3(256), 0, s, C, H, W, image, gt, alpha, mask, win, dmap, partials, gx * gy * C,
                           lambda_dssim, lambda_mask, dL_dimage, dL_dalpha, mask_partials, loss_out, lambda_l1, loss_stamps ? loss_stamps + 8 * 4096 : nullptr,
                           (const unsigned char*)nullptr, (const int*)nullptr);
    } else {
        hipLaunchKernelGGL((ssim_pass1_kernel<false, false>), grid, dim3(256), 0, s, C, H, W, image, gt, win, dmap, partials, alpha, mask, lambda_mask, dL_dalpha, mask_partials, loss_stamps,
                           (const unsigned char*)nullptr, (const int*)nullptr);
        hipLaunchKernelGGL((ssim_pass2_kernel<false, false>), grid, dim3(256), 0, s, C, H, W, image, gt, alpha, mask, win, dmap, partials, gx * gy * C,
                           lambda_dssim, lambda_mask, dL_dimage, dL_dalpha, mask_partials, loss_out, lambda_l1, loss_stamps ? loss_stamps + 8 * 4096 : nullptr,
                           (const unsigned char*)nullptr, (const int*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

// MOSS's own loss expression (train_ZJU.py:108-119,131): Ll1 = l1_loss(image[bound], gt[bound]), mask_loss = l2_loss(alpha[bound], mask[bound]),
// ssim on the crop boundingRect(bound) of both images.  `bound`: (H,W) bytes, non-zero = counted, or NULL (every pixel of the
// rectangle); `rect`: FIVE ints in DEVICE memory -- x, y, w, h of the crop and the number of non-zero bytes of `bound` (ignored when
// bound is NULL) -- which a captured step may rewrite between replays.  Pixels of `bound` outside the rectangle count for nothing
// (MOSS's rectangle is the mask's bounding box: there are none).  Gradients are written for the WHOLE image (zero off the crop).
extern "C" int moss_photometric_loss_roi(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                         const unsigned char* bound, const int* rect, float lambda_l1, float lambda_dssim, float lambda_mask,
                                         float* loss_out, float* dL_dimage, float* dL_dalpha, char* workspace, size_t workspace_bytes, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !image || !gt || !loss_out || !dL_dimage || !workspace || !rect) return MOSS_ERR_INVALID_ARG;
    if ((alpha == nullptr) != (mask == nullptr) || (alpha && !dL_dalpha)) return MOSS_ERR_INVALID_ARG;
    if (workspace_bytes < moss_loss_workspace_bytes(C, H, W)) return MOSS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + LT - 1) / LT, gy = (H + LT - 1) / LT;
    char* p = workspace;
    float* dmap = carve<float>(p, 3 * (size_t)C * H * W);
    float* partials = carve<float>(p, (size_t)gx * gy * C * 2);
    float* mask_partials = carve<float>(p, (size_t)gx * gy);
    static const Win win = make_window();
    const dim3 grid(gx, gy, C);
    // (as for the full frame: the instantiations that let empty tiles leave early only where the grid takes more than one residency round --
    // at 512 x 512 every workgroup is resident and the block-wide OR costs more than it saves: 19.2 -> 18.x us, scripts/loss_times_roi.py)
    if ((size_t)gx * gy * C > (size_t)3 * loss_device_cus()) {
        hipLaunchKernelGGL((ssim_pass1_kernel<true, true>), grid, dim3(256), 0, s, C, H, W, image, gt, win, dmap, partials, alpha, mask, lambda_mask, dL_dalpha,
                           mask_partials, (unsigned long long*)nullptr, bound, rect);
        hipLaunchKernelGGL((ssim_pass2_kernel<true, true>), grid, dim3(256), 0, s, C, H, W, image, gt, alpha, mask, win, dmap, partials, gx * gy * C,
                           lambda_dssim, lambda_mask, dL_dimage, dL_dalpha, mask_partials, loss_out, lambda_l1, (unsigned long long*)nullptr, bound, rect);
    } else {
        hipLaunchKernelGGL((ssim_pass1_kernel<false, true>), grid, dim3(256), 0, s, C, H, W, image, gt, win, dmap, partials, alpha, mask, lambda_mask, dL_dalpha,
                           mask_partials, (unsigned long long*)nullptr, bound, rect);
        hipLaunchKernelGGL((ssim_pass2_kernel<false, true>), grid, dim3(256), 0, s, C, H, W, image, gt, alpha, mask, win, dmap, partials, gx * gy * C,
                           lambda_dssim, lambda_mask, dL_dimage, dL_dalpha, mask_partials, loss_out, lambda_l1, (unsigned long long*)nullptr, bound, rect);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
