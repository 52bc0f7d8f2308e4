// blend.hip -- the 16x16-tile alpha blend, forward (DGR/cuda_rasterizer/forward.cu:261-383) and backward
// (DGR/cuda_rasterizer/backward.cu:399-587), designed for wave64 / CDNA4.
//
// WHY NOT "one thread per pixel, 256 threads per tile" (the reference's shape): a training view of a human puts
// ~100k Gaussians into ~180 of 1024 tiles, 1-5k entries per tile.  The blend is a serial recurrence over a tile's
// entry list, so with 4 waves per tile the kernel time is (longest list) x (instructions per entry) at ONE wave per
// SIMD (one instruction per ~4-5 cycles, every LDS/exp latency exposed) while 3/4 of the chip idles.
//
// Shape used here: a wavefront owns a 4x4-pixel block; lane l = (pixel l>>2, slot l&3): the FOUR lanes of a pixel
// evaluate FOUR consecutive list entries at once, and the order-dependent parts (transmittance T, the backward's
// suffix blend) are carried across the four slots with quad-permute DPP moves in exactly the sequential order
// (forward T is bit-identical to a one-entry-at-a-time loop).  A workgroup is NW waves (NW=16: a whole tile, NW=4: an
// 8x8 quadrant, so that one long tile spreads over four CUs); it stages each batch of entries ONCE into LDS (three
// 16-byte records per entry with coalescable dwordx4 gathers -- colour and depth included, which the reference re-reads
// from global memory per pixel, forward.cu:360,362 -- plus a packed {x, y, hx, hy} record for the cull test).
//   * per-wave culling: each wave tests 64 staged entries at a time, one entry per lane, against its 4x4 block
//     (conservative bounding box of the alpha >= 1/255 ellipse, from the preprocess kernel), __ballot()s the hits and
//     compacts their indices into a per-wave LDS list (v_mbcnt prefix); the blend loop then walks that list four
//     entries per iteration with the next iteration's records prefetched.  A 4x4 block is touched by roughly half as
//     many entries as an 8x8 block and a quarter as many as the tile;
//   * the inner loops are written to stay on the VECTOR unit: per-lane conditions are float selects, not lane-mask
//     algebra -- the scalar unit is shared by every wave of a CU and was the measured bottleneck of an earlier version
//     (SQ_INSTS_SALU > SQ_INSTS_VALU, profiles/r01_notes.md);
//   * early out: a wave stops when its 16 pixels are done, the workgroup when all its waves are (forward).
//
// Backward: the reference issues 9 global float atomics per (pixel, Gaussian) pair (backward.cu:538,574-584).  Here the
// 9 partial gradients of FOUR entries are summed over a wave's 16 pixels by a reduce-scatter: v_permlane32_swap and
// v_permlane16_swap each fold TWO values one level (5+3 swaps), two DPP row rotations finish (22 instructions per four
// entries); the wave sums are combined through LDS in a fixed order and each (tile, entry) instance stores ONE 48-byte
// record with plain coalesced stores.  The per-Gaussian kernel (preprocess.hip) gathers a Gaussian's records in a
// fixed order.  No float atomics => gradients are bitwise reproducible and no accumulator needs zero-filling.
// The five suffix blends of backward.cu:529-549 (colour x3, depth, alpha) enter dL/dalpha only through
// sum_k (x_k - accum_k) * g_k with per-pixel constants g, so ONE running scalar Q = sum_k accum_k g_k is carried
// instead of five.
//
// Arithmetic: power/alpha of a (pixel, entry) pair is ONE inline function shared by both kernels (explicit fmaf chain),
// so forward and backward take bit-identical skip decisions.
#include "common.h"

namespace moss {

namespace {

constexpr int FWD_BATCH = 256;
constexpr int BWD_BATCH_QUADRANT = 128, BWD_BATCH_TILE = 64;   // entries staged per round (4-wave / 16-wave workgroups)
constexpr int NPART = 12;           // 9 partial gradients padded to 12 floats (48 B) per (wave, entry)
constexpr int LIST_PAD = 32;           // sentinel entries behind a hit list: two groups per trip + one trip of prefetch

struct PairEval { float power, G, alpha; };

// alpha of one (pixel, entry) pair, 0 if the pair fails either skip test of forward.cu:340-350 / backward.cu:507-514
__device__ __forceinline__ PairEval eval_pair(float dx, float dy, float A, float B, float C, float opacity)
{
    PairEval r;
    const float q = __fmaf_rn(A * dx, dx, (C * dy) * dy);      // power = -0.5*(A dx^2 + C dy^2) - B dx dy
    r.power = __fmaf_rn(-0.5f, q, -(B * dx) * dy);
    r.G = __expf(r.power);
    float al = fminf(0.99f, opacity * r.G);
    al = (r.power <= 0.0f) ? al : 0.0f;
    r.alpha = (al >= 1.0f / 255.0f) ? al : 0.0f;
    return r;
}

// bounding box of an entry's alpha >= 1/255 region {x, y, hx, hy} vs. the wave's 4x4 pixel block [bx0,bx0+3] x [by0,by0+3]
__device__ __forceinline__ bool block_hit(float4 d, float bx0, float by0)
{
    return (d.x + d.z >= bx0) && (d.x - d.z <= bx0 + 3.0f) && (d.y + d.w >= by0) && (d.y - d.w <= by0 + 3.0f);
}

// quad-permute DPP moves (lane l = 4*pixel + slot)
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true))
#define QUAD_PREV(v)   DPP_F(v, 0x90)     // quad_perm:[0,0,1,2]  slot s reads slot s-1 (slot 0 reads itself)
#define QUAD_BCAST3(v) DPP_F(v, 0xFF)     // quad_perm:[3,3,3,3]
#define QUAD_XOR1(v)   DPP_F(v, 0xB1)     // quad_perm:[1,0,3,2]
#define QUAD_XOR2(v)   DPP_F(v, 0x4E)     // quad_perm:[2,3,0,1]

// which 4x4 block of the tile a wave owns
template <int NW>
__device__ __forceinline__ void block_of_wave(int sub, int wv, int& bx, int& by)
{
    if (NW == 16) { bx = wv & 3; by = wv >> 2; }
    else { bx = 2 * (sub & 1) + (wv & 1); by = 2 * (sub >> 1) + (wv >> 1); }      // NW == 4: quadrant `sub`
}

// Cooperative staging of a batch of list entries first, first+dir, ... : thread e < cnt owns entry e.  The gather is
// split in two so that the loads of batch k+1 are in flight (in registers) while the workgroup processes batch k.
struct Staged { float4 a, b, c; };

__device__ __forceinline__ Staged load_entry(int cnt, const float4* __restrict__ inst_rec, int first, int dir)
{
    Staged s;
    const int e = threadIdx.x;
    if (e < cnt) {
        const float4* rec = inst_rec + 3 * (size_t)(first + dir * e);      // contiguous per tile: coalesced, single level
        s.a = rec[0]; s.b = rec[1]; s.c = rec[2];
    } else {
        s.a = s.b = s.c = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return s;
}

__device__ __forceinline__ void store_entry(const Staged& s, int cnt, float4* s_a, float4* s_b, float4* s_c)
{
    const int e = threadIdx.x;
    if (e < cnt) { s_a[e] = s.a; s_b[e] = s.b; s_c[e] = s.c; }     // s_a = {x, y, hx, hy} doubles as the cull record
}

// Append the indices (c0 + lane) of the lanes with `hit` to this wave's list; returns the new length.
__device__ __forceinline__ int append_hits(uint16_t* list, int len, bool hit, int value)
{
    const unsigned long long m = __ballot(hit);
    const int pos = len + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (hit) list[pos] = (uint16_t)value;
    return len + __popcll(m);
}

// ---------------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NW * 64)
blend_forward_kernel(int W, int H, int gx, int T_tiles, int T_pad, const uint2* __restrict__ ranges, const float4* __restrict__ inst_rec,
                     const float* __restrict__ bg_color, float* __restrict__ out_color, float* __restrict__ out_depth,
                     float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int use_cull)
{
    constexpr int NT = NW * 64;
    __shared__ float4 s_a[FWD_BATCH + 1];
    __shared__ float4 s_b[FWD_BATCH + 1];
    __shared__ float4 s_c[FWD_BATCH + 1];
    __shared__ uint16_t s_list[NW][FWD_BATCH + LIST_PAD];

    // workgroups of one tile differ by a multiple of T_pad (a multiple of 8): same XCD under round-robin placement
    const int tile = blockIdx.x % T_pad, sub = blockIdx.x / T_pad;
    if (tile >= T_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = lane & 3, pl = lane >> 2;
    int bx, by;
    block_of_wave<NW>(sub, wv, bx, by);
    const int ox = tx * TILE + bx * 4, oy = ty * TILE + by * 4;
    const int px = ox + (pl & 3), py = oy + (pl >> 2);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float bx0 = (float)ox, by0 = (float)oy;
    const int qshift = lane & ~3;
    const uint32_t below_mask = (1u << slot) - 1u;

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);

    if (tid == 0) {                       // sentinel entry: opacity 0 -> alpha 0 -> contributes nothing
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        s_a[FWD_BATCH] = z4; s_b[FWD_BATCH] = z4; s_c[FWD_BATCH] = z4;
    }

    float T = 1.0f, T_stop = -1.0f;
    float Cr = 0.f, Cg = 0.f, Cb = 0.f, weight = 0.f, Dacc = 0.f;      // this slot's share of the pixel's sums
    uint32_t last_contributor = 0;
    float live = inside ? 1.0f : 0.0f;
    uint16_t* my_list = s_list[wv];

    static_assert(NT >= FWD_BATCH, "one staging thread per batch entry");
    Staged pre = load_entry(min(FWD_BATCH, n), inst_rec, (int)rg.x, 1);
    for (int base = 0; base < n; base += FWD_BATCH) {
        if (__syncthreads_and(live == 0.0f)) break;           // also: everybody is done reading the previous batch
        const int cnt = min(FWD_BATCH, n - base);
        store_entry(pre, cnt, s_a, s_b, s_c);
        if (base + FWD_BATCH < n)                             // next batch's gathers fly while this one is blended
            pre = load_entry(min(FWD_BATCH, n - base - FWD_BATCH), inst_rec, (int)rg.x + base + FWD_BATCH, 1);
        __syncthreads();
        if (__ballot(live > 0.0f) == 0ull) continue;          // this wave is finished; keep meeting the barriers

        int len = 0;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            bool hit = j < cnt;
            if (hit && (use_cull & 1)) hit = block_hit(s_a[j], bx0, by0);
            if (use_cull & 4) hit = false;
            len = append_hits(my_list, len, hit, j);
        }
        if (lane < LIST_PAD) my_list[len + lane] = (uint16_t)FWD_BATCH;        // pad with the sentinel
        const int niter = (use_cull & 2) ? 0 : (len + 3) >> 2;

        // Two groups of four entries per trip: the two alpha evaluations (sub/mul/exp, ~25 dependent instructions each) are
        // independent and interleave; only the short T chains run one after the other.  Records are prefetched one trip ahead.
        int lp = slot;
        uint32_t eA = my_list[lp], eB = my_list[lp + 4];
        float4 aA = s_a[eA], bA = s_b[eA], cA = s_c[eA];
        float4 aB = s_a[eB], bB = s_b[eB], cB = s_c[eB];
        uint32_t eA1 = my_list[lp + 8], eB1 = my_list[lp + 12];
        for (int it = 0; it < niter; it += 2) {
            const float4 aAn = s_a[eA1], bAn = s_b[eA1], cAn = s_c[eA1];
            const float4 aBn = s_a[eB1], bBn = s_b[eB1], cBn = s_c[eB1];
            const uint32_t eA2 = my_list[lp + 16], eB2 = my_list[lp + 20];

            const PairEval peA = eval_pair(aA.x - pixx, aA.y - pixy, bA.x, bA.y, bA.z, bA.w);
            const PairEval peB = eval_pair(aB.x - pixx, aB.y - pixy, bB.x, bB.y, bB.z, bB.w);
#define FWD_GROUP(pe, b, c, e)                                                                                              \
            {                                                                                                               \
                const float al = pe.alpha * live;                       /* 0 for finished / outside pixels */               \
                const float f = 1.0f - al;                                                                                  \
                /* X_s = T after the entries of slots 0..s, multiplied in list order (bit-identical to the serial loop) */ \
                float X = T * f, Y;                                                                                         \
                Y = QUAD_PREV(X); X = slot >= 1 ? Y * f : X;                                                                \
                Y = QUAD_PREV(X); X = slot >= 2 ? Y * f : X;                                                                \
                Y = QUAD_PREV(X); X = slot >= 3 ? Y * f : X;                                                                \
                Y = QUAD_PREV(X);                                                                                           \
                const float Tb = slot == 0 ? T : Y;                     /* T in front of this slot's entry */               \
                const float st = (X < 0.0001f) ? al : 0.0f;             /* > 0: this entry ends the pixel (forward.cu:351-356) */ \
                const unsigned long long sb = __ballot(st > 0.0f);                                                          \
                const uint32_t q = (uint32_t)(sb >> qshift) & 0xFu;     /* stop flags of this pixel's 4 slots */            \
                const uint32_t below = q & below_mask;                  /* an earlier slot already stopped the pixel */     \
                float wgt = al * Tb;                                                                                        \
                wgt = (st > 0.0f) ? 0.0f : wgt;                                                                             \
                wgt = (below != 0u) ? 0.0f : wgt;                                                                           \
                const float ts = (st > 0.0f) ? Tb : T_stop;                                                                 \
                T_stop = (below != 0u) ? T_stop : ts;                   /* the first stopping slot records the final T */   \
                Cr = __fmaf_rn(c.x, wgt, Cr); Cg = __fmaf_rn(c.y, wgt, Cg); Cb = __fmaf_rn(c.z, wgt, Cb);                   \
                weight += wgt;                                                                                              \
                Dacc = __fmaf_rn(c.w, wgt, Dacc);                                                                           \
                last_contributor = (wgt > 0.0f) ? (uint32_t)base + e + 1u : last_contributor;                               \
                T = QUAD_BCAST3(X);                                                                                         \
                live = (q != 0u) ? 0.0f : live;                                                                             \
            }
            FWD_GROUP(peA, bA, cA, eA)
            FWD_GROUP(peB, bB, cB, eB)
#undef FWD_GROUP
            if (__ballot(live > 0.0f) == 0ull) break;

            aA = aAn; bA = bAn; cA = cAn; aB = aBn; bB = bBn; cB = cBn;
            eA = eA1; eB = eB1; eA1 = eA2; eB1 = eB2; lp += 8;
        }
    }

    // combine the four slots of each pixel
    Cr += QUAD_XOR1(Cr); Cr += QUAD_XOR2(Cr);
    Cg += QUAD_XOR1(Cg); Cg += QUAD_XOR2(Cg);
    Cb += QUAD_XOR1(Cb); Cb += QUAD_XOR2(Cb);
    weight += QUAD_XOR1(weight); weight += QUAD_XOR2(weight);
    Dacc += QUAD_XOR1(Dacc); Dacc += QUAD_XOR2(Dacc);
    T_stop = fmaxf(T_stop, QUAD_XOR1(T_stop)); T_stop = fmaxf(T_stop, QUAD_XOR2(T_stop));
    uint32_t lc = last_contributor;
    lc = max(lc, (uint32_t)__builtin_amdgcn_mov_dpp((int)lc, 0xB1, 0xf, 0xf, true));
    lc = max(lc, (uint32_t)__builtin_amdgcn_mov_dpp((int)lc, 0x4E, 0xf, 0xf, true));
    const float Tf = T_stop >= 0.0f ? T_stop : T;

    if (inside && slot == 0) {
        const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;
        final_T[pix_id] = Tf;
        n_contrib[pix_id] = lc;
        out_color[pix_id] = __fmaf_rn(Tf, bg_color[0], Cr);
        out_color[plane + pix_id] = __fmaf_rn(Tf, bg_color[1], Cg);
        out_color[2 * plane + pix_id] = __fmaf_rn(Tf, bg_color[2], Cb);
        out_alpha[pix_id] = weight;
        out_depth[pix_id] = Dacc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// two values, one reduction level each: lanes 0-31 get a's sum over (l, l+32), lanes 32-63 get b's
__device__ __forceinline__ float fold32(float a, float b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// even rows get a's sum over row pairs (0,1) / (2,3), odd rows get b's
__device__ __forceinline__ float fold16(float a, float b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 4 lanes of a row that share l&3
__device__ __forceinline__ float row_slot_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));   // row_ror:4
    return v;
}

template <int NW>
__global__ void __launch_bounds__(NW * 64)
blend_backward_kernel(int W, int H, int gx, int T_tiles, int T_pad, const uint2* __restrict__ ranges, const float4* __restrict__ inst_rec,
                      const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpixels, const float* __restrict__ dL_ddepths, const float* __restrict__ dL_dalphas,
                      float* __restrict__ inst_grad /* [16/NW][R][12] */, size_t slab_stride, int use_cull)
{
    constexpr int NT = NW * 64;
    constexpr int BB = (NW == 4) ? BWD_BATCH_QUADRANT : BWD_BATCH_TILE;      // entries staged per round
    __shared__ float4 s_a[BB + 1];
    __shared__ float4 s_b[BB + 1];
    __shared__ float4 s_c[BB + 1];
    __shared__ uint16_t s_list[NW][BB + LIST_PAD];
    __shared__ __attribute__((aligned(16))) float s_part[NW][BB + 1][NPART];
    __shared__ uint32_t s_nmax;

    const int tile = blockIdx.x % T_pad, sub = blockIdx.x / T_pad;
    if (tile >= T_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = lane & 3, pl = lane >> 2;
    int bx, by;
    block_of_wave<NW>(sub, wv, bx, by);
    const int ox = tx * TILE + bx * 4, oy = ty * TILE + by * 4;
    const int px = ox + (pl & 3), py = oy + (pl >> 2);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float bx0 = (float)ox, by0 = (float)oy;
    const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n == 0) return;
    float* my_grad = inst_grad + (size_t)sub * slab_stride + (size_t)rg.x * NPART;      // this workgroup's slab of the tile's records

    const float T_final = inside ? final_Ts[pix_id] : 0.0f;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    float gpr = 0.f, gpg = 0.f, gpb = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        gpr = dL_dpixels[pix_id]; gpg = dL_dpixels[plane + pix_id]; gpb = dL_dpixels[2 * plane + pix_id];
        gpd = dL_ddepths[pix_id]; gpa = dL_dalphas[pix_id];
    }
    const float bg_dot = bg_color[0] * gpr + bg_color[1] * gpg + bg_color[2] * gpb;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    if (tid == 0) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        s_a[BB] = z4; s_b[BB] = z4; s_c[BB] = z4;
        s_nmax = 0;
    }
    __syncthreads();
    // Pixel state, replicated in the pixel's four lanes: T and Q = sum_k accum_k * g_k, where accum_k are the reference's
    // accum_rec[3] / accum_depth_rec / accum_alpha_rec at the moment they are used (backward.cu:529,543,548).
    float T = T_final, Q = 0.0f;

    // entries at list positions >= n_eff are behind every pixel's last contributor: nobody visits them
    int wave_max = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_max = max(wave_max, __shfl_xor(wave_max, d));
    if (lane == 0) atomicMax(&s_nmax, (uint32_t)wave_max);
    __syncthreads();
    const int n_eff = (int)s_nmax;

    for (int i = n_eff * NPART + tid; i < n * NPART; i += NT) my_grad[i] = 0.0f;

    uint16_t* my_list = s_list[wv];
    // which of the reduce-scatter's outputs this lane ends up holding (see the reduction below)
    const int row = lane >> 4, rh = row >> 1, rp = row & 1;
    const int m0 = 2 * rp + rh, m1 = 4 + m0;
    const bool writer = (lane & 12) == 0;                  // lanes 16*row + slot

    static_assert(NT >= BB, "one staging thread per batch entry");
    Staged pre = load_entry(min(BB, n_eff), inst_rec, (int)rg.x + n_eff - 1, -1);
    for (int base = 0; base < n_eff; base += BB) {
        const int cnt = min(BB, n_eff - base);
        __syncthreads();                                   // previous batch fully flushed
        store_entry(pre, cnt, s_a, s_b, s_c);
        if (base + BB < n_eff)                      // next batch's gathers fly while this one is processed
            pre = load_entry(min(BB, n_eff - base - BB), inst_rec, (int)rg.x + n_eff - 1 - base - BB, -1);
        {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4* zp = reinterpret_cast<float4*>(&s_part[0][0][0]);
            for (int i = tid; i < NW * (BB + 1) * NPART / 4; i += NT) zp[i] = z4;
        }
        __syncthreads();

        int len = 0;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            bool hit = (j < cnt) && (n_eff - 1 - (base + j) < wave_max);
            if (hit && (use_cull & 1)) hit = block_hit(s_a[j], bx0, by0);
            if (use_cull & 4) hit = false;
            len = append_hits(my_list, len, hit, j);
        }
        if (lane < LIST_PAD) my_list[len + lane] = (uint16_t)BB;
        const int niter = (use_cull & 2) ? 0 : (len + 3) >> 2;

        // Two groups of four entries per trip (see the forward kernel): evaluations and gradient formulas of the two groups
        // are independent and interleave; only the (T, Q) chains are sequential.  Records are prefetched one trip ahead.
        int lp = slot;
        uint32_t eA = my_list[lp], eB = my_list[lp + 4];
        float4 aA = s_a[eA], bA = s_b[eA], cA = s_c[eA];
        float4 aB = s_a[eB], bB = s_b[eB], cB = s_c[eB];
        uint32_t eA1 = my_list[lp + 8], eB1 = my_list[lp + 12];
        for (int it = 0; it < niter; it += 2) {
            const float4 aAn = s_a[eA1], bAn = s_b[eA1], cAn = s_c[eA1];
            const float4 aBn = s_a[eB1], bBn = s_b[eB1], cBn = s_c[eB1];
            const uint32_t eA2 = my_list[lp + 16], eB2 = my_list[lp + 20];

            const float dxA = aA.x - pixx, dyA = aA.y - pixy, dxB = aB.x - pixx, dyB = aB.y - pixy;
            const PairEval peA = eval_pair(dxA, dyA, bA.x, bA.y, bA.z, bA.w);
            const PairEval peB = eval_pair(dxB, dyB, bB.x, bB.y, bB.z, bB.w);
#define BWD_GROUP(pe, dx, dy, a, b, c, e)                                                                                   \
            {                                                                                                               \
                const int pos = n_eff - 1 - (base + (int)e);                 /* back to front */                            \
                const float al = (pos < last_contributor) ? pe.alpha : 0.0f; /* backward.cu:499-514; 0 = pair skipped */    \
                const float G = (al > 0.0f) ? pe.G : 0.0f;                                                                  \
                /* this entry's state transform:  T' = T / (1-alpha),  Q' = alpha*u + (1-alpha)*Q   (identity if skipped) */ \
                const float mm = 1.0f - al;                                                                                 \
                const float rinv = __builtin_amdgcn_rcpf(mm);                                                               \
                const float u = __fmaf_rn(c.x, gpr, __fmaf_rn(c.y, gpg, __fmaf_rn(c.z, gpb, __fmaf_rn(c.w, gpd, gpa))));    \
                const float kq = al * u;                                                                                    \
                /* run the four slots' transforms in visiting order: slot s starts from the output of slot s-1 */          \
                float Ti = T, Qi = Q;                                                                                       \
                float To = Ti * rinv, Qo = __fmaf_rn(mm, Qi, kq);                                                           \
                _Pragma("unroll") for (int k = 1; k <= 3; k++) {                                                            \
                    const float yT = QUAD_PREV(To), yQ = QUAD_PREV(Qo);                                                     \
                    Ti = slot >= k ? yT : Ti; Qi = slot >= k ? yQ : Qi;                                                     \
                    To = Ti * rinv; Qo = __fmaf_rn(mm, Qi, kq);                                                             \
                }                                                                                                           \
                T = QUAD_BCAST3(To); Q = QUAD_BCAST3(Qo);                    /* the pixel's state after these four entries */ \
                /* To = T after the division (backward.cu:516); (u - Qi) = sum_k (x_k - accum_k) g_k */                     \
                float dL_dopa = __fmaf_rn(u - Qi, To, (-T_final * rinv) * bg_dot);                                          \
                dL_dopa = (al > 0.0f) ? dL_dopa : 0.0f;                                                                     \
                const float dchannel_dcolor = al * To;                                                                      \
                const float dL_dG = b.w * dL_dopa;                                                                          \
                const float gdx = G * dx, gdy = G * dy;                                                                     \
                const float dG_ddelx = -gdx * b.x - gdy * b.y;                                                              \
                const float dG_ddely = -gdy * b.z - gdx * b.y;                                                              \
                const float v0 = dchannel_dcolor * gpr, v1 = dchannel_dcolor * gpg, v2 = dchannel_dcolor * gpb;             \
                const float v3 = dL_dG * dG_ddelx * ddelx_dx;                                                               \
                const float v4 = dL_dG * dG_ddely * ddely_dy;                                                               \
                const float hdG = -0.5f * dL_dG;                                                                            \
                const float v5 = hdG * gdx * dx, v6 = hdG * gdx * dy, v7 = hdG * gdy * dy;                                  \
                const float v8 = G * dL_dopa;                                                                               \
                if (__ballot(al > 0.0f) != 0ull) {                                                                          \
                    /* reduce-scatter over the 16 pixels, separately per slot: after fold32 the lower/upper half-waves hold \
                       different values, after fold16 even/odd rows do; lane (row r, slot s) ends with m0, m1 (and 8 in row 0) */ \
                    const float r0 = fold32(v0, v1), r1 = fold32(v2, v3), r2 = fold32(v4, v5), r3 = fold32(v6, v7), r4 = fold32(v8, 0.0f); \
                    const float s0 = row_slot_sum(fold16(r0, r1)), s1 = row_slot_sum(fold16(r2, r3)), s2 = row_slot_sum(fold16(r4, 0.0f)); \
                    if (writer) {                                                                                           \
                        float* dst = &s_part[wv][e][0];                                                                     \
                        dst[m0] = s0; dst[m1] = s1;                                                                         \
                        if (row == 0) dst[8] = s2;                                                                          \
                    }                                                                                                       \
                }                                                                                                           \
            }
            BWD_GROUP(peA, dxA, dyA, aA, bA, cA, eA)
            BWD_GROUP(peB, dxB, dyB, aB, bB, cB, eB)
#undef BWD_GROUP
            aA = aAn; bA = bAn; cA = cAn; aB = aBn; bB = bBn; cB = cBn;
            eA = eA1; eB = eB1; eA1 = eA2; eB1 = eB2; lp += 8;
        }
        __syncthreads();
        // combine the wave sums in a fixed order; 16 lanes per entry write its 48-byte record contiguously
        for (int i = tid; i < BB * 16; i += NT) {
            const int ee = i >> 4, k = i & 15;
            if (ee < cnt && k < NPART) {
                float r = 0.f;
                if (k < 9) {
#pragma unroll
                    for (int w = 0; w < NW; w++) r += s_part[w][ee][k];
                }
                const int pos = n_eff - 1 - (base + ee);
                my_grad[(size_t)pos * NPART + k] = r;
            }
        }
    }
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // anonymous namespace

int blend_subgroups()      // workgroups per tile (1: 16 waves own a tile; 4: one 4-wave workgroup per 8x8 quadrant)
{
    static const int nw = env_int("MOSS_BLEND_WAVES", 4);
    return nw == 4 ? 4 : 1;
}

void launch_blend_forward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                          float* out_color, float* out_depth, float* out_alpha, hipStream_t s)
{
    static const int use_cull = env_int("MOSS_BLEND_CULL", 1);
    const int T = fp.gx * fp.gy, T_pad = (T + 7) / 8 * 8;
    if (blend_subgroups() == 4)
        hipLaunchKernelGGL(blend_forward_kernel<4>, dim3(T_pad * 4), dim3(256), 0, s, fp.W, fp.H, fp.gx, T, T_pad, im.ranges, b.inst_rec,
                           fp.bg_dev, out_color, out_depth, out_alpha, im.final_T, im.n_contrib, use_cull);
    else
        hipLaunchKernelGGL(blend_forward_kernel<16>, dim3(T_pad), dim3(1024), 0, s, fp.W, fp.H, fp.gx, T, T_pad, im.ranges, b.inst_rec,
                           fp.bg_dev, out_color, out_depth, out_alpha, im.final_T, im.n_contrib, use_cull);
}

void launch_blend_backward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                           const float* dL_dpix, const float* dL_ddepth, const float* dL_dalpha, hipStream_t s)
{
    static const int use_cull = env_int("MOSS_BLEND_CULL", 1);
    const int T = fp.gx * fp.gy, T_pad = (T + 7) / 8 * 8;
    float* ig = reinterpret_cast<float*>(b.inst_grad);
    if (blend_subgroups() == 4)
        hipLaunchKernelGGL(blend_backward_kernel<4>, dim3(T_pad * 4), dim3(256), 0, s, fp.W, fp.H, fp.gx, T, T_pad, im.ranges, b.inst_rec,
                           fp.bg_dev, im.final_T, im.n_contrib, dL_dpix, dL_ddepth, dL_dalpha,
                           ig, b.slab_stride_floats, use_cull);
    else
        hipLaunchKernelGGL(blend_backward_kernel<16>, dim3(T_pad), dim3(1024), 0, s, fp.W, fp.H, fp.gx, T, T_pad, im.ranges, b.inst_rec,
                           fp.bg_dev, im.final_T, im.n_contrib, dL_dpix, dL_ddepth, dL_dalpha,
                           ig, b.slab_stride_floats, use_cull);
}

}  // namespace moss
