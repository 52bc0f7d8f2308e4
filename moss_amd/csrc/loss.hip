// loss.hip -- fused photometric loss of the training step and its gradient:
//     loss = mean|x - y| + lambda_mask * mean((alpha - mask)^2) + lambda_dssim * (1 - mean SSIM(x, y))
// x = rendered image (C,H,W), y = ground truth; SSIM exactly as the reference's utils/loss_utils.py:47-87: 11x11 Gaussian
// window (sigma 1.5, normalised in fp32), zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2; the combination and weights
// are those of train_ZJU.py:111-112,119,131.
//
// The reference runs five depthwise 11x11 convolutions + ~20 elementwise kernels forward and their autograd mirror
// backward (measured here with MIOpen: 8 x 178 us + ~60 launches per step).  Here: two launches.
//   pass 1: per 16x16 tile and channel, x and y tiles (+5 px halo) go to LDS once, the five windowed moments
//           (E[x], E[y], E[x^2], E[y^2], E[xy]) are formed separably (horizontal then vertical 11-tap), SSIM and its partial
//           derivatives w.r.t. the three x-dependent moments are evaluated per pixel; the derivative maps are written
//           (3 floats per pixel-channel) and the block's loss sums go to a partials array (fixed-order => deterministic);
//   pass 2: the derivative maps are filtered with the same (symmetric) window, again separably through LDS, and combined
//           into dL/dx = -lambda/N * (w*D_mu + 2x (w*D_xx) + y (w*D_xy)) + sign(x-y)/N; dL/dalpha alongside; block 0 folds
//           the partials into the scalar losses.
// Both passes are HBM-streaming: ~ (2 + 3) * 4 B read/written per pixel-channel in pass 1, (3 + 2 + 1) * 4 B in pass 2.
#include "common.h"

namespace moss {

namespace {

constexpr int LT = 16;              // tile edge
constexpr int HALO = 5;             // window 11
constexpr int LP = LT + 2 * HALO;   // 26

struct Win { float g[11]; };

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

__global__ void __launch_bounds__(256)
ssim_pass1_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt, Win win,
                  float* __restrict__ dmap /* [3][C][H][W] */, float* __restrict__ partials /* [blocks][2] */,
                  const float* __restrict__ alpha, const float* __restrict__ mask, float lambda_mask,
                  float* __restrict__ dL_dalpha, float* __restrict__ mask_partials /* [tiles] */)
{
    __shared__ float s_x[LP][LP + 1];
    __shared__ float s_y[LP][LP + 1];
    __shared__ float s_h[5][LP][LT + 1];
    __shared__ float s_red[3][4];

    const TileId tile = xcd_tile();
    const int c = tile.c;
    const int x0 = tile.bx * LT, y0 = tile.by * LT;
    const int tid = threadIdx.x;
    const float* xc = img + (size_t)c * H * W;
    const float* yc = gt + (size_t)c * H * W;

    for (int i = tid; i < LP * LP; i += 256) {
        const int r = i / LP, q = i % LP;
        s_x[r][q] = ld0(xc, x0 + q - HALO, y0 + r - HALO, W, H);
        s_y[r][q] = ld0(yc, x0 + q - HALO, y0 + r - HALO, W, H);
    }
    __syncthreads();
    for (int i = tid; i < LP * LT; i += 256) {          // horizontal 11-tap for the 5 moments
        // lanes run down a column (r fastest): consecutive lanes are LP + 1 = 27 words apart, an odd stride, so the 11-tap reads
        // below hit distinct LDS banks (row-major lanes put rows r and r + 1 of a half-wave on the same banks: PMC showed
        // 30% of this kernel's wave-cycles waiting on LDS with 2.6M bank-conflict cycles)
        const int r = i % LP, q = i / LP;
        float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = s_x[r][q + k], b = s_y[r][q + k], w = win.g[k];
            m1 = __fmaf_rn(w, a, m1); m2 = __fmaf_rn(w, b, m2);
            xx = __fmaf_rn(w, a * a, xx); yy = __fmaf_rn(w, b * b, yy); xy = __fmaf_rn(w, a * b, xy);
        }
        s_h[0][r][q] = m1; s_h[1][r][q] = m2; s_h[2][r][q] = xx; s_h[3][r][q] = yy; s_h[4][r][q] = xy;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int px = x0 + lx, py = y0 + ly;
    float ssim_v = 0.f, l1_v = 0.f, mask_v = 0.f;
    if (px < W && py < H) {
        float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.g[k];
            mu1 = __fmaf_rn(w, s_h[0][ly + k][lx], mu1); mu2 = __fmaf_rn(w, s_h[1][ly + k][lx], mu2);
            exx = __fmaf_rn(w, s_h[2][ly + k][lx], exx); eyy = __fmaf_rn(w, s_h[3][ly + k][lx], eyy);
            exy = __fmaf_rn(w, s_h[4][ly + k][lx], exy);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
        const float a1 = 2.f * mu12 + C1, a2 = 2.f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = s1 + s2 + C2;
        const float inv = 1.0f / (b1 * b2);
        const float S = a1 * a2 * inv;
        // partial derivatives of S w.r.t. (mu1 | sigma1^2 | sigma12), then total derivatives w.r.t. the three filtered
        // moments E[x], E[x^2], E[xy] (sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2)
        const float dS_ds1 = -S / b2;
        const float dS_ds12 = 2.f * a1 * inv;
        const float dS_dmu1 = 2.f * mu2 * a2 * inv - S * 2.f * mu1 / b1 + dS_ds1 * (-2.f * mu1) + dS_ds12 * (-mu2);
        const size_t o = ((size_t)c * H + py) * W + px, plane3 = (size_t)C * H * W;
        dmap[o] = dS_dmu1; dmap[plane3 + o] = dS_ds1; dmap[2 * plane3 + o] = dS_ds12;
        ssim_v = S;
        l1_v = fabsf(s_x[ly + HALO][lx + HALO] - s_y[ly + HALO][lx + HALO]);
        // the alpha-vs-mask L2 term rides on channel 0's tiles; it is computed HERE (not in pass 2, where its gradient would
        // fit just as well) so that everything pass 2's closing fold reads was written by an earlier kernel
        if (c == 0 && alpha != nullptr) {
            const size_t oa = (size_t)py * W + px;
            const float da = alpha[oa] - mask[oa];
            mask_v = da * da;
            dL_dalpha[oa] = lambda_mask * 2.f * da / ((float)H * (float)W);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ssim_v += __shfl_xor(ssim_v, d); l1_v += __shfl_xor(l1_v, d); mask_v += __shfl_xor(mask_v, d); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = ssim_v; s_red[1][tid >> 6] = l1_v; s_red[2][tid >> 6] = mask_v; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)tile.c * gridDim.y + tile.by) * gridDim.x + tile.bx;
        partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        if (c == 0 && alpha != nullptr) mask_partials[tile.by * gridDim.x + tile.bx] = (s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3]);
    }
}

__global__ void __launch_bounds__(256)
ssim_pass2_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                  const float* __restrict__ alpha, const float* __restrict__ mask, Win win,
                  const float* __restrict__ dmap, const float* __restrict__ partials, int nblocks,
                  float lambda_dssim, float lambda_mask, float* __restrict__ dL_dimg, float* __restrict__ dL_dalpha,
                  const float* __restrict__ mask_partials, float* __restrict__ loss_out)
{
    __shared__ float s_d[3][LP][LP + 1];
    __shared__ float s_h[3][LP][LT + 1];
    __shared__ float s_red[3][4];

    const TileId tile = xcd_tile();
    const int c = tile.c;
    const int x0 = tile.bx * LT, y0 = tile.by * LT;
    const int tid = threadIdx.x;
    const size_t plane3 = (size_t)C * H * W;
    const float N = (float)C * (float)H * (float)W;

    for (int i = tid; i < 3 * LP * LP; i += 256) {
        const int m = i / (LP * LP), j = i % (LP * LP);
        const int r = j / LP, q = j % LP;
        s_d[m][r][q] = ld0(dmap + m * plane3 + (size_t)c * H * W, x0 + q - HALO, y0 + r - HALO, W, H);
    }
    __syncthreads();
    for (int i = tid; i < 3 * LP * LT; i += 256) {
        const int m = i / (LP * LT), j = i % (LP * LT);
        const int r = j % LP, q = j / LP;                    // column-major lanes: conflict-free LDS reads (see pass 1)
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = __fmaf_rn(win.g[k], s_d[m][r][q + k], acc);
        s_h[m][r][q] = acc;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (px < W && py < H) {
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.g[k];
            f0 = __fmaf_rn(w, s_h[0][ly + k][lx], f0); f1 = __fmaf_rn(w, s_h[1][ly + k][lx], f1); f2 = __fmaf_rn(w, s_h[2][ly + k][lx], f2);
        }
        const size_t o = ((size_t)c * H + py) * W + px;
        const float x = img[o], y = gt[o];
        const float dssim = f0 + 2.f * x * f1 + y * f2;                    // d(sum SSIM)/dx
        const float d = x - y;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        dL_dimg[o] = sgn / N - lambda_dssim * dssim / N;
    }
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
            const float l1_mean = ((s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3])) / N;
            const float mask_mean = ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) / ((float)H * (float)W);
            const float lm = alpha != nullptr ? lambda_mask : 0.0f;
            loss_out[1] = l1_mean; loss_out[2] = ssim_mean; loss_out[3] = mask_mean;
            loss_out[0] = l1_mean + lm * mask_mean + lambda_dssim * (1.0f - ssim_mean);
        }
    }
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

extern "C" int moss_photometric_loss(int C, int H, int W, const float* image, const float* gt, const float* alpha, const float* mask,
                                     float lambda_dssim, float lambda_mask, float* loss_out, float* dL_dimage, float* dL_dalpha,
                                     char* workspace, size_t workspace_bytes, void* stream)
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
    const dim3 grid(gx, gy, C);
    hipLaunchKernelGGL(ssim_pass1_kernel, grid, dim3(256), 0, s, C, H, W, image, gt, win, dmap, partials, alpha, mask, lambda_mask, dL_dalpha, mask_partials);
    hipLaunchKernelGGL(ssim_pass2_kernel, grid, dim3(256), 0, s, C, H, W, image, gt, alpha, mask, win, dmap, partials, gx * gy * C,
                       lambda_dssim, lambda_mask, dL_dimage, dL_dalpha, mask_partials, loss_out);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
