// blend.hip -- the 16x16-tile alpha blend, forward (DGR/cuda_rasterizer/forward.cu:261-383) and backward
// (DGR/cuda_rasterizer/backward.cu:399-587), written for wave64 / CDNA4.
//
// Shape of both kernels: one 256-thread workgroup per tile = 4 wavefronts, wave w owns the 8x8-pixel quadrant
// (w&1, w>>1) of the tile (an 8x8 block has a smaller bounding box than the reference's 16x4 thread rows, which
// is what makes the per-wave culling below effective).  The tile's sorted entry list is walked in batches:
//   * the whole workgroup stages a batch (three 16-byte records per entry: position+conic, conic/opacity/depth,
//     colour -- ALL per-entry data, the reference leaves colour and depth in global memory in its forward loop,
//     forward.cu:360,362) into LDS with coalescable dwordx4 gathers;
//   * each wave tests 64 entries at a time, ONE ENTRY PER LANE, against its quadrant (conservative bounding box of
//     the alpha >= 1/255 ellipse computed by the preprocess kernel), __ballot()s the result into a 64-bit scalar
//     mask and then visits only the set bits in order (s_ff1): entries that cannot touch the quadrant cost one
//     lane-test instead of 64 lane-evaluations;
//   * LDS reads of the visited entry are wave-uniform broadcasts.
// Early out: a wave stops visiting when all its 64 pixels are done; the workgroup leaves when all four waves are.
//
// Backward: the reference issues 9 global float atomics per (pixel, Gaussian) pair (backward.cu:538,574-584).  Here the
// 9 partial gradients are summed across the wave's 64 pixels with DPP row operations (6 v_add_f32_dpp per value),
// the four wave sums are combined through LDS in a fixed order, and each (tile, entry) instance stores ONE 48-byte
// record with plain coalesced stores; the per-Gaussian kernel (preprocess.hip) gathers a Gaussian's records in a
// fixed order.  No float atomics anywhere => gradients are bitwise reproducible, and no accumulator needs zeroing.
//
// Arithmetic: the power/alpha evaluation is ONE shared inline function used by both kernels (explicit fmaf chain),
// so forward and backward take bit-identical skip decisions for every (pixel, entry) pair.
#include "common.h"

namespace moss {

namespace {

constexpr int FWD_BATCH = 256;
constexpr int BWD_BATCH = 128;
constexpr int NPART = 12;           // 9 partial gradients padded to 12 floats (48 B) per (wave, entry)

struct PairEval { float power, G, alpha; bool ok; };

// alpha of one (pixel, entry) pair; `ok` == the pair passes both skip tests of forward.cu:340-350 / backward.cu:507-514.
__device__ __forceinline__ PairEval eval_pair(float dx, float dy, float A, float B, float C, float opacity)
{
    PairEval r;
    // power = -0.5*(A dx^2 + C dy^2) - B dx dy
    const float q = __fmaf_rn(A * dx, dx, (C * dy) * dy);
    r.power = __fmaf_rn(-0.5f, q, -(B * dx) * dy);
    r.G = __expf(r.power);
    r.alpha = fminf(0.99f, opacity * r.G);
    r.ok = (r.power <= 0.0f) && (r.alpha >= 1.0f / 255.0f);
    return r;
}

__device__ __forceinline__ bool quad_hit(float4 a, float hx, float hy, float bx0, float by0)
{
    // bounding box of the entry's alpha >= 1/255 region vs. the wave's 8x8 pixel block [bx0,bx0+7] x [by0,by0+7]
    return (a.x + hx >= bx0) && (a.x - hx <= bx0 + 7.0f) && (a.y + hy >= by0) && (a.y - hy <= by0 + 7.0f);
}

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
blend_forward_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const float4* __restrict__ geo_a, const float4* __restrict__ geo_b, const float4* __restrict__ geo_c,
                     const float* __restrict__ bg_color, float* __restrict__ out_color, float* __restrict__ out_depth,
                     float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int use_cull)
{
    __shared__ float4 s_a[FWD_BATCH];
    __shared__ float4 s_b[FWD_BATCH];
    __shared__ float4 s_c[FWD_BATCH];

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qx = tx * TILE + (wv & 1) * 8, qy = ty * TILE + (wv >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float bx0 = (float)qx, by0 = (float)qy;

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, weight = 0.f, Dacc = 0.f;
    uint32_t last_contributor = 0;
    bool done = !inside;

    for (int base = 0; base < n; base += FWD_BATCH) {
        if (__syncthreads_and(done)) break;
        const int cnt = min(FWD_BATCH, n - base);
        if (tid < cnt) {
            const uint32_t id = point_list[rg.x + base + tid];
            s_a[tid] = geo_a[id]; s_b[tid] = geo_b[id]; s_c[tid] = geo_c[id];
        }
        __syncthreads();
        if (__ballot(!done) == 0ull) continue;          // this wave is finished; keep meeting the barriers
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            bool hit = j < cnt;
            if (hit && use_cull) hit = quad_hit(s_a[j], s_b[j].w, s_c[j].w, bx0, by0);
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int e = c0 + bit;
                const float4 a = s_a[e], b = s_b[e], c = s_c[e];
                if (!done) {
                    const PairEval pe = eval_pair(a.x - pixx, a.y - pixy, a.z, a.w, b.x, b.y);
                    if (pe.ok) {
                        const float test_T = T * (1.0f - pe.alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            const float wgt = pe.alpha * T;
                            Cr = __fmaf_rn(c.x, wgt, Cr); Cg = __fmaf_rn(c.y, wgt, Cg); Cb = __fmaf_rn(c.z, wgt, Cb);
                            weight += wgt;
                            Dacc = __fmaf_rn(b.z, wgt, Dacc);
                            T = test_T;
                            last_contributor = (uint32_t)(base + e + 1);
                        }
                    }
                }
                if (__ballot(!done) == 0ull) { mask = 0ull; c0 = cnt; }
            }
        }
    }

    if (inside) {
        const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        out_color[pix_id] = __fmaf_rn(T, bg_color[0], Cr);
        out_color[plane + pix_id] = __fmaf_rn(T, bg_color[1], Cg);
        out_color[2 * plane + pix_id] = __fmaf_rn(T, bg_color[2], Cb);
        out_alpha[pix_id] = weight;
        out_depth[pix_id] = Dacc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave; the total is valid in lane 63.  6 DPP adds:
// quad butterfly (x2), row_half_mirror, row_mirror (now every lane holds its 16-lane row sum), then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
#define DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    DPP_ADD(0xB1, 0xf);     // quad_perm:[1,0,3,2]
    DPP_ADD(0x4E, 0xf);     // quad_perm:[2,3,0,1]
    DPP_ADD(0x141, 0xf);    // row_half_mirror
    DPP_ADD(0x140, 0xf);    // row_mirror
    DPP_ADD(0x142, 0xa);    // row_bcast:15 -> rows 1,3
    DPP_ADD(0x143, 0xc);    // row_bcast:31 -> rows 2,3
#undef DPP_ADD
    return v;
}

__global__ void __launch_bounds__(256)
blend_backward_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float4* __restrict__ geo_a, const float4* __restrict__ geo_b, const float4* __restrict__ geo_c,
                      const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpixels, const float* __restrict__ dL_ddepths, const float* __restrict__ dL_dalphas,
                      float4* __restrict__ inst_grad, int use_cull)
{
    __shared__ float4 s_a[BWD_BATCH];
    __shared__ float4 s_b[BWD_BATCH];
    __shared__ float4 s_c[BWD_BATCH];
    __shared__ __attribute__((aligned(16))) float s_part[4][BWD_BATCH][NPART];
    __shared__ uint32_t s_nmax;

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qx = tx * TILE + (wv & 1) * 8, qy = ty * TILE + (wv >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float bx0 = (float)qx, by0 = (float)qy;
    const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n == 0) return;

    const float T_final = inside ? final_Ts[pix_id] : 0.0f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    float gpr = 0.f, gpg = 0.f, gpb = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        gpr = dL_dpixels[pix_id]; gpg = dL_dpixels[plane + pix_id]; gpb = dL_dpixels[2 * plane + pix_id];
        gpd = dL_ddepths[pix_id]; gpa = dL_dalphas[pix_id];
    }
    const float bg_dot = bg_color[0] * gpr + bg_color[1] * gpg + bg_color[2] * gpb;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_a = 0.f;
    float last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f, last_d = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // entries at list positions >= n_eff are behind every pixel's last contributor: nobody visits them
    if (tid == 0) s_nmax = 0;
    __syncthreads();
    {
        int m = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        if (lane == 0) atomicMax(&s_nmax, (uint32_t)m);
    }
    __syncthreads();
    const int n_eff = (int)s_nmax;
    int wave_max = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_max = max(wave_max, __shfl_xor(wave_max, d));

    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pos = n_eff + tid; pos < n; pos += 256) {
        float4* dst = inst_grad + 3 * ((size_t)rg.x + pos);
        dst[0] = z4; dst[1] = z4; dst[2] = z4;
    }

    for (int base = 0; base < n_eff; base += BWD_BATCH) {
        const int cnt = min(BWD_BATCH, n_eff - base);
        __syncthreads();                                   // previous batch fully flushed
        if (tid < cnt) {
            const int pos = n_eff - 1 - (base + tid);      // back to front
            const uint32_t id = point_list[rg.x + pos];
            s_a[tid] = geo_a[id]; s_b[tid] = geo_b[id]; s_c[tid] = geo_c[id];
        }
        {
            float4* zp = reinterpret_cast<float4*>(&s_part[0][0][0]);
            for (int i = tid; i < 4 * BWD_BATCH * NPART / 4; i += 256) zp[i] = z4;
        }
        __syncthreads();

        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            bool hit = (j < cnt) && (n_eff - 1 - (base + j) < wave_max);
            if (hit && use_cull) hit = quad_hit(s_a[j], s_b[j].w, s_c[j].w, bx0, by0);
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int e = c0 + bit;
                const int pos = n_eff - 1 - (base + e);
                const float4 a = s_a[e], b = s_b[e], c = s_c[e];
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
                bool contrib = false;
                if (pos < last_contributor) {              // backward.cu:499-501
                    const float dx = a.x - pixx, dy = a.y - pixy;
                    const PairEval pe = eval_pair(dx, dy, a.z, a.w, b.x, b.y);
                    if (pe.ok) {
                        contrib = true;
                        const float alpha = pe.alpha, G = pe.G;
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        // suffix blends (backward.cu:529-549)
                        acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r; last_r = c.x;
                        acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g; last_g = c.y;
                        acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b; last_b = c.z;
                        float dL_dopa = (c.x - acc_r) * gpr + (c.y - acc_g) * gpg + (c.z - acc_b) * gpb;
                        v0 = dchannel_dcolor * gpr; v1 = dchannel_dcolor * gpg; v2 = dchannel_dcolor * gpb;
                        acc_d = last_alpha * last_d + (1.f - last_alpha) * acc_d; last_d = b.z;
                        dL_dopa += (b.z - acc_d) * gpd;
                        acc_a = last_alpha + (1.f - last_alpha) * acc_a;
                        dL_dopa += (1.f - acc_a) * gpa;
                        dL_dopa *= T;
                        last_alpha = alpha;
                        dL_dopa += (-T_final / (1.f - alpha)) * bg_dot;
                        const float dL_dG = b.y * dL_dopa;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * a.z - gdy * a.w;
                        const float dG_ddely = -gdy * b.x - gdx * a.w;
                        v3 = dL_dG * dG_ddelx * ddelx_dx;
                        v4 = dL_dG * dG_ddely * ddely_dy;
                        v5 = -0.5f * gdx * dx * dL_dG;
                        v6 = -0.5f * gdx * dy * dL_dG;
                        v7 = -0.5f * gdy * dy * dL_dG;
                        v8 = G * dL_dopa;
                    }
                }
                if (__ballot(contrib) != 0ull) {
                    v0 = wave_sum_to_lane63(v0); v1 = wave_sum_to_lane63(v1); v2 = wave_sum_to_lane63(v2);
                    v3 = wave_sum_to_lane63(v3); v4 = wave_sum_to_lane63(v4); v5 = wave_sum_to_lane63(v5);
                    v6 = wave_sum_to_lane63(v6); v7 = wave_sum_to_lane63(v7); v8 = wave_sum_to_lane63(v8);
                    if (lane == 63) {
                        float4* dst = reinterpret_cast<float4*>(&s_part[wv][e][0]);
                        dst[0] = make_float4(v0, v1, v2, v3);
                        dst[1] = make_float4(v4, v5, v6, v7);
                        s_part[wv][e][8] = v8;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < cnt) {
            // combine the four quadrant sums in a fixed order and emit this instance's record
            float r[9];
#pragma unroll
            for (int k = 0; k < 9; k++)
                r[k] = ((s_part[0][tid][k] + s_part[1][tid][k]) + s_part[2][tid][k]) + s_part[3][tid][k];
            const int pos = n_eff - 1 - (base + tid);
            float4* dst = inst_grad + 3 * ((size_t)rg.x + pos);
            dst[0] = make_float4(r[0], r[1], r[2], r[3]);
            dst[1] = make_float4(r[4], r[5], r[6], r[7]);
            dst[2] = make_float4(r[8], 0.f, 0.f, 0.f);
        }
    }
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // anonymous namespace

void launch_blend_forward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                          float* out_color, float* out_depth, float* out_alpha, hipStream_t s)
{
    static const int use_cull = env_int("MOSS_BLEND_CULL", 1);
    hipLaunchKernelGGL(blend_forward_kernel, dim3(fp.gx * fp.gy), dim3(256), 0, s, fp.W, fp.H, fp.gx, im.ranges, b.point_list,
                       g.geo_a, g.geo_b, g.geo_c, fp.bg_dev, out_color, out_depth, out_alpha, im.final_T, im.n_contrib, use_cull);
}

void launch_blend_backward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                           const float* dL_dpix, const float* dL_ddepth, const float* dL_dalpha, hipStream_t s)
{
    static const int use_cull = env_int("MOSS_BLEND_CULL", 1);
    hipLaunchKernelGGL(blend_backward_kernel, dim3(fp.gx * fp.gy), dim3(256), 0, s, fp.W, fp.H, fp.gx, im.ranges, b.point_list,
                       g.geo_a, g.geo_b, g.geo_c, fp.bg_dev, im.final_T, im.n_contrib, dL_dpix, dL_ddepth, dL_dalpha,
                       b.inst_grad, use_cull);
}

}  // namespace moss
