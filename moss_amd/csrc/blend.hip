// blend.hip -- the 16x16-tile alpha blend, forward (DGR/cuda_rasterizer/forward.cu:261-383) and backward
// (DGR/cuda_rasterizer/backward.cu:399-587), designed for wave64 / CDNA4.
//
// WHY NOT "one thread per pixel, 256 threads per tile" (the reference's shape): a training view of a human puts ~100k Gaussians
// into ~250 of 1024 tiles, 1-5k entries per tile.  The blend is a serial recurrence over a tile's entry list and a single
// wavefront issues at most one instruction per ~4 cycles whatever its ILP, so the kernel time is (instructions executed by the
// busiest wave) x 4 cycles while most SIMDs idle (measured: profiles/r01_notes.md).  The design therefore minimises the work
// of the busiest WAVE, not the total work:
//
//   * a work item is ONE WAVE -- in the forward kernel a wave PAIR for heavy items: a scanner (block masks, hit list, record DMAs)
//     and a blender (trips only), see PairCtl -- and there is no workgroup barrier anywhere (the first versions, one workgroup per
//     tile quadrant with per-batch barriers, spent 38 % of their critical path waiting at those barriers; they are in the history
//     of this file and in profiles/r01_notes.md);
//   * heavy tiles (>= 2^LIGHT_TILE_LOG2 = 32 entries): item = one 4x4 pixel block, lane = (pixel, slot), 4 consecutive list entries per trip; the
//     order-dependent parts (transmittance T; the backward's suffix blend) are carried across the 4 slots of a pixel by a 3-step
//     DPP chain that is bit-identical to the serial loop;
//   * light tiles: item = one 8x8 quadrant, lane = pixel, entries one after the other;
//   * the inner loops stay on the VECTOR unit: per-lane conditions are float selects, not lane-mask algebra -- the scalar unit
//     is shared by every wave of a CU and was the measured bottleneck of the first version;
//   * persistent workgroups of 4 waves (one per SIMD; four workgroups per CU in both kernels) pull items from per-XCD queues in LPT
//     order; the backward's long lists arrive cut into 64-hit depth segments (the forward leaves the state at every cut).
//
// Backward: the reference issues 9 global float atomics per (pixel, Gaussian) pair (backward.cu:538,574-584).  Here the 9
// partial gradients of a trip's entries are summed over the wave's pixels by a reduce-scatter (v_permlane32_swap and
// v_permlane16_swap each fold TWO values one level, DPP row rotations finish) and each (entry, block) pair that was really
// blended stores ONE 48-byte record with plain stores and sets its block's bit in the instance's mask.  The per-Gaussian kernel
// (preprocess.hip) gathers the flagged records in a fixed order.  No float atomics => gradients are bitwise reproducible and no
// accumulator needs zero-filling.  The five suffix blends of backward.cu:529-549 (colour x3, depth, alpha) enter dL/dalpha only
// through sum_k (x_k - accum_k) * g_k with per-pixel constants g, so ONE running scalar Q = sum_k accum_k g_k is carried instead
// of five; (T, Q) advance through a trip's slots as a prefix composition of the affine maps  T -> T/(1-a),  Q -> (1-a) Q + a u.
//
// Arithmetic: power/alpha of a (pixel, entry) pair is ONE inline function shared by both kernels (explicit fmaf chain), so
// forward and backward take bit-identical skip decisions.
#include "common.h"

namespace moss {

namespace {

constexpr int NPART = GRAD_REC_FLOATS;   // 9 partial gradients per (instance, block) record, padded (common.h)

struct PairEval { float power, G, alpha; };
typedef float v2f __attribute__((ext_vector_type(2)));        // an aligned register pair: v_pk_add / v_pk_mul / v_pk_fma_f32 do both halves at once

// The exponent of one (pixel, entry) pair in the heavy paths' trips, where vector-instruction issue is what the kernels wait for: the
// operations of eval_pair below, one for one (same roundings, same order), but the two products by dx / dy that do not feed an FMA
// are done as PAIRS -- (B dx, C dy), then ((B dx) dy, (C dy) dy) -- which is why the record's second word is {B, C, A, opacity}
// (merge_gather_kernel): (B, C) and (x, y) are aligned register pairs as they come out of the LDS read.  Six instructions for nine.
struct PairGeom { v2f d; float power; };                      // d = (dx, dy)

// ---- MOSS_DEBUG_EXACT_MATH (include/moss_raster.h): the arithmetic that DECIDES a pixel's list exactly as the reference's source reads
// (forward.cu:336-356, backward.cu:504-516), i.e. as the CPU oracle evaluates it (gcc -ffp-contract=off): bit-identical n_contrib and
// final_T on every pixel.  Template parameter EXACT of everything below; the product path (EXACT = false) is unchanged.
// power = -0.5f * (A dx dx + C dy dy) - B dx dy, one rounding per operation, C's precedence (no contraction inside this function)
__device__ __forceinline__ float power_exact(float dx, float dy, float A, float B, float C)
{
#pragma clang fp contract(off)
    const float t1 = (A * dx) * dx, t2 = (C * dy) * dy;
    const float sum = t1 + t2;
    const float half = -0.5f * sum;
    const float t3 = (B * dx) * dy;
    return half - t3;
}
// exp(x), x <= 0: glibc's expf algorithm (sysdeps/ieee754/flt-32/e_expf.c: 2^(k/32) table + cubic, in double) with every double
// operation one rounded operation -- oracle/moss_oracle.c: moss_expf_det is the SAME function, bit for bit (checked there against the
// C library's expf on a grid of 11.5 M arguments)
__device__ const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
__device__ __forceinline__ float expf_det(float x)
{
#pragma clang fp contract(off)
    if (x != x) return x;                                     // expf(NaN) = NaN (a degenerate conic): the reference's min(0.99f, NaN * opacity) then blends at 0.99
    if (!(x > -104.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;                                   // (never used for a decision: power > 0 is skipped; keeps the table index sane)
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = z - kd;
    const uint64_t t = EXP2F_TAB[ki % 32u] + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double zz = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * sc;
    return (float)y;
}
template <bool EXACT> __device__ __forceinline__ float blend_exp(float power) { if constexpr (EXACT) return expf_det(power); else return __expf(power); }
// 1 / (1 - alpha): v_rcp_f32 (~1 ulp) on the product path, the IEEE quotient in the exact mode
template <bool EXACT> __device__ __forceinline__ float blend_rcp(float v) { if constexpr (EXACT) return 1.0f / v; else return __builtin_amdgcn_rcpf(v); }

template <bool EXACT = false>
__device__ __forceinline__ PairGeom pair_power(const float4& a, const float4& b, v2f pix)
{
    PairGeom r;
    r.d = v2f{a.x, a.y} - pix;
    if constexpr (EXACT) {
        r.power = power_exact(r.d.x, r.d.y, b.z, b.x, b.y);      // (record word b = {B, C, A, opacity})
    } else {
        const v2f bc = v2f{b.x, b.y} * r.d;                       // (B dx, C dy)
        const v2f t = bc * v2f{r.d.y, r.d.y};                     // ((B dx) dy, (C dy) dy)
        const float q = __fmaf_rn(b.z * r.d.x, r.d.x, t.y);       // power = -0.5*(A dx^2 + C dy^2) - B dx dy
        r.power = __fmaf_rn(-0.5f, q, -t.x);
    }
    return r;
}

// alpha of one (pixel, entry) pair, 0 if the pair fails either skip test of forward.cu:340-350 / backward.cu:507-514
template <bool EXACT = false>
__device__ __forceinline__ PairEval eval_pair(float dx, float dy, float A, float B, float C, float opacity)
{
    PairEval r;
    if constexpr (EXACT) {
        r.power = power_exact(dx, dy, A, B, C);
    } else {
        const float q = __fmaf_rn(A * dx, dx, (C * dy) * dy);      // power = -0.5*(A dx^2 + C dy^2) - B dx dy
        r.power = __fmaf_rn(-0.5f, q, -(B * dx) * dy);
    }
    r.G = blend_exp<EXACT>(r.power);
    float al = fminf(0.99f, opacity * r.G);
    al = (r.power <= 0.0f) ? al : 0.0f;
    r.alpha = (al >= 1.0f / 255.0f) ? al : 0.0f;
    return r;
}

// bounding box of an entry's alpha >= 1/255 region {x, y, hx, hy} vs. a pixel block [bx0,bx0+w] x [by0,by0+h]
// ---- cross-lane helpers (lane = pixel * SLOTS + slot; the SLOTS lanes of a pixel are contiguous inside a 16-lane DPP row)
#define DPP_MOV(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true))
// butterfly all-reduce over the slots of a pixel: every lane ends with bit-identical results (commutative ops, symmetric tree)
#define GROUP_ALLREDUCE(SLOTS, v, OP)                                          \
    {                                                                          \
        v = OP(v, DPP_MOV(v, 0xB1));                      /* quad_perm:[1,0,3,2] */ \
        v = OP(v, DPP_MOV(v, 0x4E));                      /* quad_perm:[2,3,0,1] */ \
        if (SLOTS > 4) v = OP(v, DPP_MOV(v, 0x141));      /* row_half_mirror     */ \
        if (SLOTS > 8) v = OP(v, DPP_MOV(v, 0x140));      /* row_mirror          */ \
    }
// predicates of __builtin_amdgcn_fcmpf / uicmp (the compare's 64-bit lane mask lands in a scalar register pair)
constexpr int FCMP_OGE = 3, FCMP_OLT = 4, FCMP_OLE = 5, ICMP_NE = 33;
// lane (pixel, slot) takes `keep` where its bit in keep_mask is set, else the value `prev` holds in the lane of the previous slot of
// the same pixel (quad_perm:[0,0,1,2]): ONE v_cndmask_b32 with the DPP read on its first source (the compiler's own select takes its
// condition from an arbitrary scalar pair, an encoding without DPP, and spends a v_mov_b32_dpp in front of it).  The two wait
// states a DPP read needs after the write of its source are inside the statement (the scalar move and the s_nop): the compiler does
// not look into it.
__device__ __forceinline__ float take_prev_slot_unless(float prev, float keep, unsigned long long keep_mask)
{
    float r;
    asm("s_mov_b64 vcc, %3\n\ts_nop 0\n\tv_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf"
        : "=v"(r) : "v"(prev), "v"(keep), "s"(keep_mask) : "vcc");
    return r;
}
#define OP_MUL(a, b) ((a) * (b))
#define OP_ADD(a, b) ((a) + (b))
#define OP_MAX(a, b) fmaxf((a), (b))

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
// sum over the lanes of a row that share the slot (lane % SLOTS)
template <int SLOTS>
__device__ __forceinline__ float row_slot_sum(float v)
{
    if (SLOTS < 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8
    if (SLOTS < 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));    // row_ror:4
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Work items.  100k Gaussians of a person fall into ~250 tiles: with one workgroup per tile quadrant nothing was ever balanced and
// the kernel's duration equalled its longest item.  Here a work item is ONE WAVE (16 block items per heavy tile, 4 quadrant items
// per light tile, pulled from LPT-ordered per-XCD queues); a wave that is done leaves at once; waves of the same tile hit in L2.
// Backward: each wave stores the partial-gradient record of an (entry, block) pair it really blended into that block's slab
// and sets the block's bit in the instance's mask; the per-Gaussian gather reads only flagged records (a fixed order -> still
// bitwise reproducible).
// ---------------------------------------------------------------------------------------------------------

constexpr int WAVE_BLOCKS = 16;           // 4x4-pixel blocks per tile = items per tile

struct Rec { float4 a, b, c; };

// Lane j reads entry first + dir*min(j, n_left-1) of the record stream (48 contiguous bytes per entry).  UNCONDITIONAL on purpose:
// lanes (and whole steps) past the end of the list re-read its last entry and are masked by the caller.  A predicated load puts
// the load in a conditional block, and at the join the compiler can no longer count outstanding loads: it then waits with
// vmcnt(0) -- i.e. for the loads it has just issued -- which serialises the whole rolling window (seen in the ISA).
__device__ __forceinline__ Rec load_rec(const float4* __restrict__ inst_rec, int first, int n_left, int lane, int dir)
{
    const int k = max(min(lane, n_left - 1), 0);
    const float4* p = inst_rec + 3 * (size_t)(first + dir * k);
    Rec r;
    r.a = p[0]; r.b = p[1]; r.c = p[2];
    return r;
}


// ---------------------------------------------------------------------------------------------------------
// LIGHT tiles (fewer than 2^LIGHT_TILE_LOG2 entries): one wave per 8x8 QUADRANT, lane = pixel, entries visited one after
// the other exactly like the reference's loop.  Per (pixel, entry) pair this costs ~0.4 wave-instructions against ~1.3 for the
// 4-slot layout of the heavy path (no cross-slot prefix, no stop-flag exchange), and a tile is 4 work items instead of 16 -- but
// the quadrant's wave visits every entry whose box touches any of its four blocks with all 64 lanes, where the heavy path's
// block masks send an entry only to the blocks it reaches.  Measured (scripts/stage_times.py, MOSS_LIGHT_LOG2 = 3 ... 9): the
// threshold was 128 entries until the backward blend got depth segments and dynamic block items; with those, 32 is where the
// backward kernel stops gaining (bench scene 37.0 -> 33.6 us, configs[1] 38.4 -> 26.0 us, configs[4] 88 -> 80 us; 8-16 the
// same, 256 and up twice as slow), the forward kernel does not care.
// ---------------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ void light_forward_item(int W, int H, int gx, int tile, int q, int lane, const uint2 rg,
                                                   const float4* __restrict__ inst_rec, const uint16_t* __restrict__ inst_bmask,
                                                   float4 (*ring)[3], const float* __restrict__ bg_color, float* __restrict__ out_color,
                                                   float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int flags)
{
    const int ox = (tile % gx) * TILE + (q & 1) * 8, oy = (tile / gx) * TILE + (q >> 1) * 8;
    const int px = ox + (lane & 7), py = oy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const int n = (int)(rg.y - rg.x);
    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, weight = 0.f, Dacc = 0.f;
    uint32_t last_contributor = 0;
    bool live = inside;
    // the quadrant's blocks in the per-instance block mask: q = (qy, qx) covers blocks (2qy..2qy+1, 2qx..2qx+1)
    const uint32_t qmask = 0x0033u << (2 * (q & 1) + 8 * (q >> 1));
    const uint16_t* const bm = inst_bmask + rg.x;
    Rec cur = load_rec(inst_rec, (int)rg.x, n, lane, 1);
    uint32_t bmk = bm[min(lane, n - 1)];
    for (int base = 0; base < n; base += 64) {
        const Rec nxt = load_rec(inst_rec, (int)rg.x + min(base + 64, n - 1), n - base - 64, lane, 1);
        const uint32_t bmk_nxt = bm[min(base + 64 + lane, n - 1)];
        bool hit = base + lane < n;
        if (hit && (flags & 1)) hit = (bmk & qmask) != 0u;                   // one of the quadrant's four 4x4 blocks can be reached
        unsigned long long m = __ballot(hit);
        bmk = bmk_nxt;
        ring[lane][0] = cur.a; ring[lane][1] = cur.b; ring[lane][2] = cur.c;
        __builtin_amdgcn_wave_barrier();
        while (m != 0ull) {
            const int e = __ffsll((long long)m) - 1;         // wave-uniform: the records are LDS broadcasts
            m &= m - 1ull;
            const float4 a = ring[e][0], b = ring[e][1], c = ring[e][2];
            const PairEval pe = eval_pair<EXACT>(a.x - pixx, a.y - pixy, b.z, b.x, b.y, b.w);   // (record word b = {B, C, A, opacity})
            const float al = live ? pe.alpha : 0.0f;
            const float test_T = T * (1.0f - al);
            const bool stop = al > 0.0f && test_T < 0.0001f;                  // forward.cu:351-356: this entry is NOT blended
            const float wgt = stop ? 0.0f : al * T;
            Cr = __fmaf_rn(c.x, wgt, Cr); Cg = __fmaf_rn(c.y, wgt, Cg); Cb = __fmaf_rn(c.z, wgt, Cb);
            weight += wgt;
            Dacc = __fmaf_rn(c.w, wgt, Dacc);
            last_contributor = (wgt > 0.0f) ? (uint32_t)(base + e + 1) : last_contributor;
            T = stop ? T : test_T;
            live = live && !stop;
            if (__ballot(live) == 0ull) { m = 0ull; base = n; }
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
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

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row_sum16(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true));   // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true));   // row_ror:1
    return v;
}

template <bool EXACT>
__device__ __forceinline__ void light_backward_item(int W, int H, int gx, int tile, int q, int lane, const uint2 rg,
                                                    const float4* __restrict__ inst_rec, const uint16_t* __restrict__ inst_bmask,
                                                    float4 (*ring)[3], const float* __restrict__ bg_color, const float* __restrict__ final_Ts,
                                                    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                                                    const float* __restrict__ dL_ddepths, const float* __restrict__ dL_dalphas,
                                                    float* __restrict__ inst_grad /* the record pool */, uint32_t* __restrict__ cell_valid,
                                                    int flags)
{
    const int ox = (tile % gx) * TILE + (q & 1) * 8, oy = (tile / gx) * TILE + (q >> 1) * 8;
    const int px = ox + (lane & 7), py = oy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;
    const float T_final = inside ? final_Ts[pix_id] : 0.0f;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    int n_eff = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_eff = max(n_eff, __shfl_xor(n_eff, d));
    if (n_eff == 0) return;
    float gpr = 0.f, gpg = 0.f, gpb = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        if (dL_dpixels) { gpr = dL_dpixels[pix_id]; gpg = dL_dpixels[plane + pix_id]; gpb = dL_dpixels[2 * plane + pix_id]; }
        if (dL_ddepths) gpd = dL_ddepths[pix_id];
        if (dL_dalphas) gpa = dL_dalphas[pix_id];
    }
    const float bg_dot = bg_color[0] * gpr + bg_color[1] * gpg + bg_color[2] * gpb;
    const float nTb = -T_final * bg_dot;
    // A record goes to a CELL of the Gaussian's run in the record pool (common.h: box_cells; the record's third word says where the
    // instance's cells start): here, one record per (entry, QUADRANT), to the cell of the first of the quadrant's four blocks that the
    // entry's block mask flags (that block is inside the box: the mask is the box test, refined).  The per-Gaussian gather sums the
    // cells whose validity bit is set, whatever tile or path left them.
    float* const my_grad = inst_grad;
    const int row = lane >> 4;
    float T = T_final, Q = 0.0f;

    const int first = (int)rg.x + n_eff - 1;                 // back to front
    const uint32_t qmask = 0x0033u << (2 * (q & 1) + 8 * (q >> 1));
    const uint16_t* const bm = inst_bmask + rg.x;
    Rec cur = load_rec(inst_rec, first, n_eff, lane, -1);
    uint32_t bmk = bm[max(n_eff - 1 - lane, 0)];
    for (int base = 0; base < n_eff; base += 64) {
        const Rec nxt = load_rec(inst_rec, first - min(base + 64, n_eff - 1), n_eff - base - 64, lane, -1);
        const uint32_t bmk_nxt = bm[max(n_eff - 1 - (base + 64 + lane), 0)];
        bool hit = base + lane < n_eff;
        if (hit && (flags & 1)) hit = (bmk & qmask) != 0u;
        unsigned long long m = __ballot(hit);
        const uint32_t bmk_cur = bmk;                        // (lane e: the block mask of this batch's entry e)
        bmk = bmk_nxt;
        ring[lane][0] = cur.a; ring[lane][1] = cur.b; ring[lane][2] = cur.c;
        __builtin_amdgcn_wave_barrier();
        while (m != 0ull) {
            // up to four entries per round; the 4 x 9 partial gradients are summed over the 64 pixels by ONE reduce-scatter:
            // fold32 pairs entries (0,2) and (1,3), fold16 pairs those, after which row r of the wave holds entry r's values
            float v[4][9];
            int pos4[4]; uint32_t slot4[4];
            bool any4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = m != 0ull;
                const int e = valid ? __ffsll((long long)m) - 1 : 0;
                if (valid) m &= m - 1ull;
                const int pos = n_eff - 1 - (base + e);
                const float4 a = ring[e][0], b = ring[e][1], c = ring[e][2];
                const float dx = a.x - pixx, dy = a.y - pixy;
                const PairEval pe = eval_pair<EXACT>(dx, dy, b.z, b.x, b.y, b.w);            // (record word b = {B, C, A, opacity})
                const float al = (valid && pos < last_contributor) ? pe.alpha : 0.0f;   // backward.cu:499-514
                const float G = (al > 0.0f) ? pe.G : 0.0f;
                const float mm = 1.0f - al;
                const float rinv = blend_rcp<EXACT>(mm);
                const float u = __fmaf_rn(c.x, gpr, __fmaf_rn(c.y, gpg, __fmaf_rn(c.z, gpb, __fmaf_rn(c.w, gpd, gpa))));
                const float To = EXACT ? T / mm : T * rinv;                          // T after the division (backward.cu:516)
                const float dL_dopa = __fmaf_rn(u - Q, To, nTb * rinv);              // (u - Q) = sum_k (x_k - accum_k) g_k
                Q = __fmaf_rn(mm, Q, al * u);
                T = To;
                // (the record's geometry sums go without their constant factors: see the heavy path's trip)
                const float dchannel_dcolor = al * To;
                const float v8 = G * dL_dopa;                                       // G = 0 for a skipped pair
                const float w = b.w * v8;
                const float wx = w * dx, wy = w * dy;
                v[k][0] = dchannel_dcolor * gpr; v[k][1] = dchannel_dcolor * gpg; v[k][2] = dchannel_dcolor * gpb;
                v[k][3] = __fmaf_rn(wx, b.z, wy * b.x);
                v[k][4] = __fmaf_rn(wy, b.y, wx * b.x);
                v[k][5] = wx * dx; v[k][6] = wx * dy; v[k][7] = wy * dy;
                v[k][8] = v8;
                pos4[k] = pos;
                uint32_t be;
                {   // the entry's cell for this quadrant: first flagged block b of the quadrant -> (w >> 2) - 16 + (b >> 2) (nbx) + (b & 3)
                    const uint32_t w = __float_as_uint(a.z);
                    be = (uint32_t)__builtin_amdgcn_readlane((int)bmk_cur, e) & qmask;
                    const int b = be ? __ffs((int)be) - 1 : 0;
                    slot4[k] = (w >> 2) - 16u + (uint32_t)((b >> 2) * (int)((w & 3u) + 1u) + (b & 3));
                }
                // (an entry whose block mask has no bit in this quadrant owns no cell here: with culling on it is never visited; with
                // MOSS_DEBUG_NO_BLOCK_CULL it is, and a record for it would land in front of the instance's cells -- dropped instead)
                any4[k] = be != 0u && __ballot(al > 0.0f) != 0ull;
            }
            if (any4[0] || any4[1] || any4[2] || any4[3]) {
                float tot[9];
#pragma unroll
                for (int j = 0; j < 9; j++)
                    tot[j] = row_sum16(fold16(fold32(v[0][j], v[2][j]), fold32(v[1][j], v[3][j])));   // row r: entry r
                const uint32_t my_pos = row == 0 ? slot4[0] : row == 1 ? slot4[1] : row == 2 ? slot4[2] : slot4[3];
                const bool my_any = row == 0 ? any4[0] : row == 1 ? any4[1] : row == 2 ? any4[2] : any4[3];
                if ((lane & 15) == 0 && my_any) {
                    float4* dst = reinterpret_cast<float4*>(my_grad + (size_t)my_pos * NPART);
                    dst[0] = make_float4(tot[0], tot[1], tot[2], tot[3]);
                    dst[1] = make_float4(tot[4], tot[5], tot[6], tot[7]);
                    dst[2] = make_float4(tot[8], 0.f, 0.f, 0.f);
                    atomicOr(&cell_valid[my_pos >> 5], 1u << (my_pos & 31u));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------
// HEAVY tiles: one wave per 4x4 block, lane = (pixel, slot).
//
// Scan: the wave does not read the tile's 48-byte records to find its hits -- stamps showed that scan costing as many cycles as
// the blending (1000-1400 cycles per 64 entries: per-CU load-path throughput, 24 cache lines per step whether 16 or 48 bytes of a
// record are used).  It reads the 2-byte BLOCK MASK merge_gather_kernel wrote for every instance (bit b set <=> the bounding
// box of the entry's alpha >= 1/255 region touches 4x4 block b of its tile): 64 entries = ONE cache line.  Hit positions go to a
// private LDS list.
// Fetch: the records of the hits (only) are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: per-lane source address,
// lane-linear destination, no VGPRs held), one round AHEAD of their use: a round scans until it has found >= 64 new hits (16+
// trips = 8k+ cycles of blending, which covers the DMA latency), issues their DMAs, then blends the previous round's records.  Fetching them into registers
// inside the trip loop (first version of this scheme) made every trip wait on L2: 640 instead of 520 cycles per trip.
// ---------------------------------------------------------------------------------------------------------
// Per-wave LDS of a heavy item: the list of hit positions (LCAP) and the record ring (RCAP slots of 48 bytes; a multiple of the
// 64-slot DMA batch).  A scan round ends once it has found ROUND_HITS new hits (<= ROUND_HITS + 63 with its last group); the ring
// holds the hits of two rounds + 3 carried + one batch of slack: 2 x 87 + 3 + 64 = 241 <= 256.
// Round 1 ran the forward with 1024 / 512 / 64 (28 KB per wave, ONE workgroup per CU: "co-resident waves only slow the heaviest
// item down").  Measured in round 2 with the work queues fixed (common.h): 13 KB per wave and TWO workgroups per CU take the
// forward kernel from 57 to 48 us -- the second wave fills the issue slots the first leaves between its dependent instructions, and
// with twice the waves every heavy block of the frame starts at once.  Same sizes for the backward kernel (whose items are short
// since the forward cuts the lists into depth segments: up to three workgroups per CU).
constexpr int LCAP = 256, RCAP = 256;
constexpr int CHAPTER = 8;                       // 64-entry groups of block masks turned into hit masks at a time (512 entries)
// Backward: 128 ring slots = two 64-slot DMA batches, 6.6 KB per wave -- a wave collects the hits of up to two batches, has their
// records copied, waits, and runs their trips (no scan / DMA / blend pipeline inside an item: with four or five waves per SIMD the
// other waves cover the DMA round trip).  The small footprint is what lets EVERY depth segment of the frame start in the first round
// (round 2 of this kernel: 3072 resident waves for 3498 segments -- 426 waves ran a second full segment while the rest idled).
constexpr int LCAP_BWD = 128, RCAP_BWD = 128;
template <int RC, int LC>
struct HeavyLdsT { float4 a[RC], b[RC], c[RC]; uint32_t lst[LC]; static constexpr int RMASK = RC - 1, LMASK = LC - 1; };
typedef HeavyLdsT<RCAP, LCAP> HeavyLds;
typedef HeavyLdsT<RCAP_BWD, LCAP_BWD> HeavyLdsBwd;

// Issue the DMA batches that cover list entries [from, to): batch q = entries 64q .. 64q+63 -> ring slots (64q & RMASK) + lane.
// Lanes whose entry is not in the list yet copy the tile's first record (overwritten when the batch is re-issued with that
// entry; never read before); lanes whose entry was fetched before re-copy the same record.
// FPOS: the list holds (position + 1) as a FLOAT (the backward kernel: its trips compare positions as floats and never need the integer).
template <bool FPOS = false, typename LDS>
__device__ __forceinline__ void dma_records(LDS* L, const float4* __restrict__ recs, int from, int to, int nlist, int lane)
{
    constexpr int LMASK = LDS::LMASK, RMASK = LDS::RMASK;
    for (int q = from >> 6; q <= (to - 1) >> 6; q++) {
        const int li = 64 * q + lane;
        const uint32_t v = L->lst[li & LMASK];
        const uint32_t p = li < nlist ? (FPOS ? (uint32_t)__uint_as_float(v) - 1u : v) : 0u;
        const float4* r = recs + 3 * (size_t)p;
        const int base = (64 * q) & RMASK;
        __builtin_amdgcn_global_load_lds(r, &L->a[base], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(r + 1, &L->b[base], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(r + 2, &L->c[base], 16, 0, 0);
    }
}

struct Fetched { float4 a, b, c; float pos1; };   // pos1 = 1-based list position (as a float; 0 / 3e38 for a padding slot of the forward / backward)

// ---- depth segments (see common.h): per-wave emission state of the forward kernel.  Every forward wave owns a private range of
// slots in its XCD's region and fills it front to back: NO atomics and NO global-memory traffic while a block is being blended (a
// returning atomic and three stores per cut, waited for by the next round's `s_waitcnt 0`, cost the forward kernel 6 us: measured).
// The cuts of a block wait in LDS; the end of the item writes them out.  A full range simply ends the cutting (the uncut remainder
// of a block's list is the block's own backward item, as before), so the range size is a tuning knob, not a correctness bound.
struct SegEmit {
    uint4* desc; float* state;   // this wave's slot range
    uint32_t cap;                // slots in the range
    uint32_t count;              // slots filled so far
    int seg_hits;                // hits per segment (a power of two, multiple of 4)
    bool store;                  // false (MOSS_FORWARD_ONLY): the cuts are made -- the sums are folded piece by piece, so that the image has
                                 // the bits of the training forward -- but nothing of them is kept: no LDS rows, no descriptors, no state
    float* cut_sums;             // LDS, per wave: [MAX_CUTS][16 pixels][6] -- T at the cut + the five sums the piece in front of it collected
    uint2* cut_pos;              // LDS, per wave: [MAX_CUTS] -- {first position, end position} of that piece
};
constexpr int MAX_CUTS = 16;     // cuts per block (a block with more than MAX_CUTS * SEG_HITS blended hits keeps the rest as its own item)

// ---- wave PAIRS (forward kernel).  Round 2's stamps: the forward kernel lasts exactly as long as its longest item, and 32k of that
// item's 102k cycles are NOT blending -- scanning block masks, issuing the record DMAs and waiting for them to land, all in the same
// wave that then blends.  So a heavy item is now worked on by TWO waves of a workgroup: the SCANNER scans the masks, keeps the hit
// list and has the hits' records copied into the pair's LDS ring; the BLENDER does nothing but trips.  They talk through a few LDS
// words (no s_barrier: a workgroup barrier would tie the other pair in):
//   scanner -> blender   post_seq / post_*   "item #seq is (tile, block, range)" or "no more heavy items"
//                        ready               list entries whose records have LANDED in the ring (bit 31: the list is complete)
//   blender -> scanner   consumed            list entries it is done with (ring slots below may be overwritten)
//                        stop_seq            "every pixel of item #seq is finished: stop scanning"
//                        fin_seq             "item #seq is written out: the ring is yours"
// Every spin loop is bounded by the other wave's progress: the blender needs >= 4 landed entries (or the complete list) to run a
// trip and the scanner always has room for them (RCAP >> 4 + one batch); the scanner needs ring room, which every trip frees.
struct PairCtl {
    uint32_t post_seq, post_kind, post_tile, post_blk, post_rank, rg_x, rg_y;
    uint32_t ready, consumed, stop_seq, fin_seq, pad;
};
constexpr int PAIR_RCAP = 256, PAIR_LCAP = 256;           // ring slots / list entries per pair (powers of two, multiples of the 64-slot DMA batch)
constexpr int PAIR_FIRST_HITS = 12, PAIR_ROUND_HITS = 24; // hits the scanner collects before it requests their records (first / later rounds)
typedef HeavyLdsT<PAIR_RCAP, PAIR_LCAP> PairRing;

// (explicit LDS address space: through a generic pointer a volatile access becomes a FLAT load with system-scope cache bits)
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p)
{
    const uint32_t v = *(const volatile lds_u32_t*)p;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_poke(uint32_t* p, uint32_t v) { *(volatile lds_u32_t*)p = v; }

// s_setprio takes an immediate
__device__ __forceinline__ void set_wave_prio(int p)
{
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
// issue priority of an item by the length of its tile's list: the longest items set the kernel's duration
__device__ __forceinline__ int prio_of_length(uint32_t n) { return n >= 2048u ? 3 : n >= 1024u ? 2 : n >= 512u ? 1 : 0; }

// SCANNER of a heavy item: block masks -> hit list -> record DMAs -> `ready`.  nx = the first chapter of masks (requested by the
// caller before it waited for the blender to release the ring).
__device__ __forceinline__ void heavy_forward_scan(int blk, int lane, const uint2 rg, const float4* __restrict__ inst_rec,
                                                   const uint16_t* __restrict__ inst_bmask, PairRing* L, PairCtl* ctl, uint32_t seq,
                                                   int flags, uint32_t (&nx)[CHAPTER])
{
    constexpr int LMASK = PairRing::LMASK;
    const int n = (int)(rg.y - rg.x);
    const uint16_t* const bm = inst_bmask + rg.x;
    const float4* const recs = inst_rec + 3 * (size_t)rg.x;
    const uint32_t all_hit = (flags & 1) ? 0u : 0xffffu;     // culling switched off (diagnostics): every entry is a hit
    // list entries: [0, F) requested by DMA (and landed once `ready` says so), [F, nlist) found but not requested yet
    int scan_pos = 0, nlist = 0, F = 0, grp = CHAPTER;
    bool scan_done = n <= 0, stop = false;
    int c_seen = 0;                                          // a lower bound of what the blender has consumed
    uint32_t vb_lo = 0u, vb_hi = 0u;                         // lane k: the hit mask of the current chapter's group k
    for (;;) {
        int new_hits = 0;
        const int target = F == 0 ? PAIR_FIRST_HITS : PAIR_ROUND_HITS;
        while (!scan_done && new_hits < target) {
            // Room: a group adds up to 64 entries and the DMA re-copies whole 64-slot batches, so ring slots up to entry nlist + 126
            // are written -- the entries that lived there (index - RCAP) must be consumed.
            if (nlist + 128 - PAIR_RCAP > c_seen) {
                c_seen = (int)lds_peek(&ctl->consumed);
                stop = lds_peek(&ctl->stop_seq) == seq;
                if (stop) break;
                if (nlist + 128 - PAIR_RCAP > c_seen) {
                    if (new_hits > 0) break;                 // hand over what there is first
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
            }
            if (grp == CHAPTER) {
                // A new chapter: the 8 hit masks of its 64-entry groups are formed at once from the masks that were in flight (8
                // independent ballots) and parked in lane k of a VGPR pair; the chapter after it is requested.
                unsigned long long bl = 0ull;
#pragma unroll
                for (int k = 0; k < CHAPTER; k++) {
                    const int idx = scan_pos + 64 * k + lane;
                    const unsigned long long b = __ballot(idx < n && (((nx[k] | all_hit) >> blk) & 1u) != 0u);
                    bl = lane == k ? b : bl;
                }
                vb_lo = (uint32_t)bl; vb_hi = (uint32_t)(bl >> 32);
#pragma unroll
                for (int k = 0; k < CHAPTER; k++) nx[k] = bm[min(scan_pos + 64 * (CHAPTER + k) + lane, n - 1)];
                grp = 0;
            }
            const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)vb_hi, grp) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)vb_lo, grp);
            grp++;
            if (m != 0ull) {
                const int r = nlist + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((m >> lane) & 1ull) L->lst[r & LMASK] = __float_as_uint((float)(scan_pos + lane + 1));     // position + 1, as a float (exact below 2^24)
                const int c = __popcll(m);
                nlist += c; new_hits += c;
            }
            scan_pos += 64;
            scan_done = scan_pos >= n;
        }
        if (!stop) stop = lds_peek(&ctl->stop_seq) == seq;
        if (stop) break;
        __builtin_amdgcn_wave_barrier();
        if (nlist > F) { dma_records<true>(L, recs, F, nlist, nlist, lane); F = nlist; }
        __builtin_amdgcn_s_waitcnt(0);                       // the records (and the list) are in LDS
        if (scan_done && (F & 3) != 0) {
            // The list is complete: pad it to whole trips with entries NOBODY blends -- an all-zero record has opacity 0, hence alpha 0
            // for every pixel -- so that the blender's trips need neither a validity flag per slot nor a test for it (three of its ~57
            // vector instructions; the kernel's critical SIMDs are bound by instruction issue).  The slots are free: the ring holds
            // at most RCAP - 64 live entries in front of them, and the DMA batches that could overwrite them have landed (wait above).
            const int pad = 4 - (F & 3);
            if (lane < pad) {
                const int li = F + lane;
                L->a[li & PairRing::RMASK] = make_float4(0.f, 0.f, 0.f, 0.f); L->b[li & PairRing::RMASK] = make_float4(0.f, 0.f, 0.f, 0.f);
                L->c[li & PairRing::RMASK] = make_float4(0.f, 0.f, 0.f, 0.f);
                L->lst[li & LMASK] = __float_as_uint(0.0f);
            }
            F += pad;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_poke(&ctl->ready, (uint32_t)F | (scan_done ? 0x80000000u : 0u));
        if (scan_done) break;
    }
    __builtin_amdgcn_s_waitcnt(0);                           // no DMA may still be writing the ring when the next item starts
}

// BLENDER of a heavy item: trips over the landed entries, cuts, and the block's outputs.
template <bool EXACT>
__device__ __forceinline__ void heavy_forward_blend(int W, int H, int gx, int tile, int blk, int lane, const uint2 rg,
                                                    PairRing* L, PairCtl* ctl, uint32_t seq, const float* __restrict__ bg_color,
                                                    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int flags,
                                                    unsigned long long* stamp_out, SegEmit& se, uint32_t* __restrict__ tail_start)
{
    constexpr int RMASK = PairRing::RMASK;
    const int slot = lane & 3, pl = lane >> 2, gbase = lane & ~3;
    const uint32_t below_mask = (1u << slot) - 1u;
    const int ox = (tile % gx) * TILE + (blk & 3) * 4, oy = (tile / gx) * TILE + (blk >> 2) * 4;
    const int px = ox + (pl & 3), py = oy + (pl >> 2);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const int n = (int)(rg.y - rg.x);
#define STAMP() (stamp_out ? __builtin_amdgcn_s_memtime() : 0ull)
    const unsigned long long t_begin = STAMP();
    const unsigned long long rt_begin = stamp_out ? __builtin_amdgcn_s_memrealtime() : 0ull;      // (100 MHz, one clock for the device)
    unsigned long long d_trip = 0, d_starve = 0, n_rounds = 0, n_trips = 0, t_first = 0;

    float T = 1.0f, T_stop = -1.0f;
    v2f Crg = v2f{0.f, 0.f}, CbD = v2f{0.f, 0.f};                       // this slot's share of the pixel's sums: (r, g), (b, depth), weight
    float weight = 0.f;
#define Cr Crg.x
#define Cg Crg.y
#define Cb CbD.x
#define Dacc CbD.y
    const v2f pix2 = v2f{pixx, pixy};
    float last_contributor = 0.0f;                                       // list positions < 2^24: exact in fp32
    float live = inside ? 1.0f : 0.0f;
    bool finished = __ballot(live > 0.0f) == 0ull;

    // one trip: 4 consecutive hits x 16 pixels; returns true when every pixel of the block is finished
    // (Round 3 software-pipelined this -- the next trip's alpha evaluated inside the current trip, compiled with the max-ilp scheduling
    // strategy so that the two chains interleave: 40.4 -> 42.8 us.  The SIMDs that set the kernel's length host two blenders and are
    // bound by instruction ISSUE, not by one wave's dependent latencies; the extra copies cost more than the overlap gave.)
    // Per-lane CONDITIONS of the trip are kept as 64-bit lane masks in scalar registers (v_cmp writes one; s_and / s_andn2 combine them;
    // __builtin_amdgcn_inverse_ballot_w64 turns one back into the select condition of a v_cndmask): "this pair blends", "this entry
    // stops its pixel", "an earlier slot stopped it", "the pixel is alive".  As float flags they cost a multiply, a compare and a select
    // each (round 2); the SIMDs that set this kernel's length are bound by vector-instruction issue (profiles/r03_notes.md, finding 2)
    // and the scalar unit has room: 51 -> 44 vector instructions per trip.  The arithmetic is eval_pair's, operation for operation.
    unsigned long long live_m = __ballot(live > 0.0f);
    auto trip = [&](const Fetched& f) -> bool {
        const float power = pair_power<EXACT>(f.a, f.b, pix2).power;
        const float ao = fminf(0.99f, f.b.w * blend_exp<EXACT>(power));
        // pairs that blend: power <= 0, alpha >= 1/255 (forward.cu:340-350), pixel alive (padding slots hold zero records: alpha 0)
        const unsigned long long m = __builtin_amdgcn_fcmpf(power, 0.0f, FCMP_OLE) & __builtin_amdgcn_fcmpf(ao, 1.0f / 255.0f, FCMP_OGE) & live_m;
        // none of the four entries reaches any live pixel (the block masks are conservative): nothing changes -- T, the sums, the
        // stop flags -- so the rest of the trip is skipped (-1.6 us on the kernel, same-box A/B)
        if (m == 0ull) return false;
        const float al = __builtin_amdgcn_inverse_ballot_w64(m) ? ao : 0.0f;
        const float fm = 1.0f - al;
        // multiplied in list order: bit-identical to the serial loop
        float X = T * fm, Y;
        Y = DPP_MOV(X, 0x90); X = slot >= 1 ? Y * fm : X;              // quad_perm:[0,0,1,2]
        Y = DPP_MOV(X, 0x90); X = slot >= 2 ? Y * fm : X;
        Y = DPP_MOV(X, 0x90); X = slot >= 3 ? Y * fm : X;
        Y = DPP_MOV(X, 0x90);
        const float Tb = slot == 0 ? T : Y;                            // T in front of this slot's entry
        const unsigned long long sb = __builtin_amdgcn_fcmpf(X, 0.0001f, FCMP_OLT) & m;     // this entry ends its pixel (forward.cu:351-356)
        const uint32_t q = (uint32_t)(sb >> gbase) & 15u;              // stop flags of this pixel's slots
        const unsigned long long mb = __builtin_amdgcn_uicmp(q & below_mask, 0u, ICMP_NE);  // an earlier slot already stopped the pixel
        const unsigned long long dead = sb | mb;                       // pairs that are NOT blended after all
        const float wgt = __builtin_amdgcn_inverse_ballot_w64(dead) ? 0.0f : al * Tb;
        T_stop = __builtin_amdgcn_inverse_ballot_w64(sb & ~mb) ? Tb : T_stop;               // the first stopping slot records the final T
        Crg = __builtin_elementwise_fma(v2f{f.c.x, f.c.y}, v2f{wgt, wgt}, Crg);          // (two packed FMAs for the four sums)
        CbD = __builtin_elementwise_fma(v2f{f.c.z, f.c.w}, v2f{wgt, wgt}, CbD);
        weight += wgt;
        last_contributor = __builtin_amdgcn_inverse_ballot_w64(m & ~dead) ? f.pos1 : last_contributor;
        T = DPP_MOV(X, 0xFF);
        live_m &= ~__builtin_amdgcn_uicmp(q, 0u, ICMP_NE);
        return live_m == 0ull;
    };

    // A cut after the hit at list position p_cur - 1: everything this block blended from p_prev up to here becomes a backward work
    // item of its own.  What that item needs is the pixel state at ITS FAR END: T (known now) and the suffix blends -- the sums of
    // everything BEHIND the cut, which only exist once the block is finished.  So every piece accumulates its five sums FROM ZERO
    // (the running totals are carried separately: tC*), the pieces' sums wait in LDS, and the end of the item adds them up from the back
    // (far pieces are small numbers: summed among themselves they keep their relative precision, where final_total - prefix_total
    // would lose it to the rounding of the large near pieces -- measured: 6x the float32 oracle's error against float64).
    int p_prev = 0, n_cuts = 0;
    float tCr = 0.f, tCg = 0.f, tCb = 0.f, tD = 0.f, tW = 0.f;          // totals of the pieces already cut off (per pixel, all four lanes)
    auto emit_cut = [&](int p_cur) {
        if (n_cuts == MAX_CUTS || se.count + (uint32_t)n_cuts >= se.cap) return;
        GROUP_ALLREDUCE(4, Cr, OP_ADD) GROUP_ALLREDUCE(4, Cg, OP_ADD) GROUP_ALLREDUCE(4, Cb, OP_ADD)
        GROUP_ALLREDUCE(4, Dacc, OP_ADD) GROUP_ALLREDUCE(4, weight, OP_ADD)
        if (se.store) {                                      // (wave-uniform)
            if (slot == 0) {
                float* cs = se.cut_sums + (n_cuts * 16 + pl) * 6;
                cs[0] = T; cs[1] = Cr; cs[2] = Cg; cs[3] = Cb; cs[4] = Dacc; cs[5] = weight;
            }
            if (lane == 0) se.cut_pos[n_cuts] = make_uint2((uint32_t)p_prev, (uint32_t)p_cur);
        }
        tCr += Cr; tCg += Cg; tCb += Cb; tD += Dacc; tW += weight;
        Cr = 0.f; Cg = 0.f; Cb = 0.f; Dacc = 0.f; weight = 0.f;
        p_prev = p_cur;
        n_cuts++;
    };
    // After a trip that left `hits_done` hits behind it, a multiple of the segment length: a cut is DUE there -- and made when the
    // NEXT trip starts, i.e. only if the list goes on (the state at a trip's start is the state at the previous trip's end).  Round 3
    // cut right after the trip unless it knew the trip to be the item's last, which it learned from the scanner's "list complete"
    // flag: whether that flag had arrived when a list of exactly k x 64 hits ran its last trip was a matter of timing, so the same
    // frame was now and then partitioned differently (one more piece, an empty remainder) -- the same gradients up to rounding, but not
    // the same bits (found in round 4: two identical trainings, a handful of last-bit differences every ~20 steps).
    int pend_pos = -1;
    auto note_cut_due = [&](const Fetched& f, int hits_done) {
        if ((hits_done & (se.seg_hits - 1)) != 0) return;
        // (only the item's LAST trip can hold padding slots, and no trip follows it: its cut is never made)
        pend_pos = (int)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.pos1), 3));   // slot 3 of pixel 0: the trip's last hit
    };
    auto cut_if_due = [&]() {
        if (pend_pos >= 0) { emit_cut(pend_pos); pend_pos = -1; }
    };

    int C = 0;                                               // list entries [0, C) are blended
    if (finished) lds_poke(&ctl->stop_seq, seq);
    while (!finished) {
        const unsigned long long t_poll = STAMP();
        const uint32_t r = lds_peek(&ctl->ready);
        const bool final_round = (r >> 31) != 0u;            // the list is complete: blend everything that is left
        const int ready = (int)(r & 0x7fffffffu);            // records of [C, ready) are in the ring
        const int avail = ready - C;
        const int ntrip = (avail >> 2) + ((final_round && (avail & 3) != 0) ? 1 : 0);
        if (ntrip == 0) {
            if (final_round) break;
            __builtin_amdgcn_s_sleep(2);
            d_starve += STAMP() - t_poll;
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        n_rounds++;
        if (!(flags & 2)) {
            const unsigned long long t4 = STAMP();
            if (t_first == 0) t_first = t4 - t_begin;
            n_trips += ntrip;
            auto get = [&](int t) -> Fetched {
                // (C and 4t are multiples of four and wave-uniform: the ring index of the trip is a SCALAR, the slot is added without a
                // wrap test -- one vector add for the address where `(C + 4t + slot) & RMASK` per lane was five instructions)
                const int li = ((C + 4 * t) & RMASK) + slot;
                Fetched f;
                f.a = L->a[li]; f.b = L->b[li]; f.c = L->c[li];
                f.pos1 = f.a.w;                                                             // the entry's 1-based list position (merge_gather)
                return f;
            };
            Fetched f0 = get(0);
            for (int t = 0; t < ntrip; t += 2) {
                const Fetched f1 = get(t + 1);                                             // next trip's LDS reads under this trip
                cut_if_due();
                if (trip(f0)) { finished = true; break; }
                note_cut_due(f0, C + 4 * (t + 1));
                if (t + 1 >= ntrip) break;
                f0 = get(t + 2);
                cut_if_due();
                if (trip(f1)) { finished = true; break; }
                note_cut_due(f1, C + 4 * (t + 2));
            }
            d_trip += STAMP() - t4;
        }
        C += 4 * ntrip;
        if (finished) { lds_poke(&ctl->stop_seq, seq); break; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // (the ring reads above are complete before the slots are given back)
        lds_poke(&ctl->consumed, (uint32_t)min(C, ready));
        if (final_round) break;
    }
    if (stamp_out && lane == 0) {
        stamp_out[0] = STAMP() - t_begin; stamp_out[1] = (unsigned long long)n; stamp_out[2] = d_starve; stamp_out[3] = t_first; stamp_out[4] = rt_begin;
        stamp_out[5] = d_trip; stamp_out[6] = n_rounds | (__builtin_amdgcn_s_memrealtime() << 16); stamp_out[7] = n_trips;     // (scripts/fwd_timeline.py)
    }
#undef STAMP

    // combine the slots of each pixel
    GROUP_ALLREDUCE(4, Cr, OP_ADD) GROUP_ALLREDUCE(4, Cg, OP_ADD) GROUP_ALLREDUCE(4, Cb, OP_ADD)
    GROUP_ALLREDUCE(4, weight, OP_ADD) GROUP_ALLREDUCE(4, Dacc, OP_ADD)
    GROUP_ALLREDUCE(4, T_stop, OP_MAX) GROUP_ALLREDUCE(4, last_contributor, OP_MAX)
    if (n_cuts > 0 && !se.store) {                            // forward only: the totals, nothing else
        Cr += tCr; Cg += tCg; Cb += tCb; Dacc += tD; weight += tW;
    } else if (n_cuts > 0) {
        // Write the block's pieces out: descriptor, T at the piece's far end, and the suffix blends there -- the sums of everything
        // behind it, added up from the back: behind the last cut lies what was collected since (Cr ...).
        __builtin_amdgcn_wave_barrier();
        float sr = Cr, sg = Cg, sb = Cb, sd = Dacc, sw = weight;
        for (int k = n_cuts - 1; k >= 0; k--) {
            const uint32_t idx = se.count + (uint32_t)k;
            const float* cs = se.cut_sums + (k * 16 + pl) * 6;
            if (slot == 0) {
                float2* st = reinterpret_cast<float2*>(se.state + (size_t)idx * SEG_STATE_FLOATS + 6 * pl);
                st[0] = make_float2(cs[0], sr); st[1] = make_float2(sg, sb); st[2] = make_float2(sd, sw);
            }
            if (lane == 0) {
                // {tile | block << 28, first instance of the tile, first position, end position}: all a backward item needs, in ONE load
                const uint2 pp = se.cut_pos[k];
                se.desc[idx] = make_uint4((uint32_t)tile | ((uint32_t)blk << 28), rg.x, pp.x, pp.y);
            }
            sr += cs[1]; sg += cs[2]; sb += cs[3]; sd += cs[4]; sw += cs[5];
        }
        __builtin_amdgcn_wave_barrier();
        se.count += (uint32_t)n_cuts;
        Cr += tCr; Cg += tCg; Cb += tCb; Dacc += tD; weight += tW;     // the pixel's totals
    }
    const float Tf = T_stop >= 0.0f ? T_stop : T;
    if (lane == 0 && se.store) tail_start[(size_t)tile * WAVE_BLOCKS + blk] = (uint32_t)p_prev;   // the block's own backward item starts here
    if (inside && slot == 0) {
        const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;
        final_T[pix_id] = Tf;
        n_contrib[pix_id] = (uint32_t)last_contributor;
        out_color[pix_id] = __fmaf_rn(Tf, bg_color[0], Cr);
        out_color[plane + pix_id] = __fmaf_rn(Tf, bg_color[1], Cg);
        out_color[2 * plane + pix_id] = __fmaf_rn(Tf, bg_color[2], Cb);
        out_alpha[pix_id] = weight;
        out_depth[pix_id] = Dacc;
    }
}

#undef Cr
#undef Cg
#undef Cb
#undef Dacc

template <bool EXACT>
__device__ __forceinline__ void heavy_backward_item(int W, int H, int gx, int tile, int blk, int lane, const uint2 rg,
                                                    const float4* __restrict__ inst_rec, const uint16_t* __restrict__ inst_bmask,
                                                    HeavyLdsBwd* L, const float* __restrict__ bg_color,
                                                    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                                    const float* __restrict__ dL_dpixels, const float* __restrict__ dL_ddepths,
                                                    const float* __restrict__ dL_dalphas, float* __restrict__ inst_grad,
                                                    uint32_t* __restrict__ cell_valid, int flags,
                                                    int lo, int hi_limit, const float* __restrict__ seg_state)
{
    // Walks the list positions [lo, end) of the block back to front.  seg_state == NULL: the block's own item -- its range ends where
    // the list ends for these pixels (end = the largest n_contrib), so every pixel starts from (T_final, Q = 0), exactly the
    // reference's start (backward.cu:440-447).  seg_state != NULL: a depth segment cut by the forward kernel (end = hi_limit) -- the
    // pixels still alive at its far end start from the state the forward left there (see below).
    constexpr int LMASK = HeavyLdsBwd::LMASK, RMASK = HeavyLdsBwd::RMASK;
    const int slot = lane & 3, pl = lane >> 2;
    // which of the reduce-scatter's outputs this lane ends up holding (see the reduction below)
    const int row = lane >> 4, rh = row >> 1, rp = row & 1;
    const uint32_t m0_bytes = 4u * (uint32_t)(2 * rp + rh);  // value m0 of the record, m1 = m0 + 4
    const int ox = (tile % gx) * TILE + (blk & 3) * 4, oy = (tile / gx) * TILE + (blk >> 2) * 4;
    const int px = ox + (pl & 3), py = oy + (pl >> 2);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;

    const float T_final = inside ? final_Ts[pix_id] : 0.0f;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    // entries at list positions >= n_eff are behind every pixel's last contributor: nobody visits them
    int n_eff = hi_limit;                                    // a segment ends where the forward cut it (entries behind a pixel's last
    if (seg_state == nullptr) {                              // contributor are masked per pixel): its mask loads need not wait for n_contrib
        n_eff = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n_eff = max(n_eff, __shfl_xor(n_eff, d));
        n_eff = __builtin_amdgcn_readfirstlane(n_eff);
    }
    if (n_eff <= lo) return;                                 // nothing was blended into this range: no record, no mask bit

    float gpr = 0.f, gpg = 0.f, gpb = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        // a null incoming gradient = that output did not take part in the loss (zeros, without a zero-filled image)
        if (dL_dpixels) { gpr = dL_dpixels[pix_id]; gpg = dL_dpixels[plane + pix_id]; gpb = dL_dpixels[2 * plane + pix_id]; }
        if (dL_ddepths) gpd = dL_ddepths[pix_id];
        if (dL_dalphas) gpa = dL_dalphas[pix_id];
    }
    const float bg_dot = bg_color[0] * gpr + bg_color[1] * gpg + bg_color[2] * gpb;
    float* const my_grad = inst_grad;                        // the record pool: an (entry, block) pair's cell from the record's third word (see the light path)
    const uint32_t cell_row = (uint32_t)(blk >> 2), cell_off = (uint32_t)(blk & 3) - 16u;     // (wave-uniform)
    const uint16_t* const bm = inst_bmask + rg.x;
    const float4* const recs = inst_rec + 3 * (size_t)rg.x;
    // MOSS_DEBUG_NO_BLOCK_CULL is NOT followed here: a gradient record's cell exists only for the blocks of the entry's alpha >= 1/255 box
    // (common.h: pack_cell_word), so a pair outside the box has nowhere to leave a record -- its cell index would be another instance's.
    // The backward of a heavy tile always walks the masks (the light path above visits everything and drops such records).  Whether a
    // pair outside its box could have reached 1/255 at all is what the no-cull FORWARD shows (the image differs from the culled one).
    const uint32_t all_hit = 0u;

    // Pixel state, replicated in the pixel's lanes: T and Q = sum_k accum_k * g_k, where accum_k are the reference's
    // accum_rec[3] / accum_depth_rec / accum_alpha_rec at the moment they are used (backward.cu:529,543,548).
    float T = T_final, Q = 0.0f;
    if (seg_state != nullptr && inside && last_contributor > hi_limit) {
        // This pixel blends entries BEHIND the segment's far end, so it does not start from the end of its list.  There, T is what
        // the forward had, and the reference's suffix blends accum_x (backward.cu:529,543,548 unrolled: sum over the entries k behind
        // of x_k alpha_k T_k, seen from T = 1) are the sums the forward collected behind the cut, divided by T:  Q = sum_x accum_x g_x.
        const float* st = seg_state + 6 * pl;
        const float2 s0 = *reinterpret_cast<const float2*>(st), s1 = *reinterpret_cast<const float2*>(st + 2),
                     s2 = *reinterpret_cast<const float2*>(st + 4);
        T = s0.x;
        Q = __fmaf_rn(s0.y, gpr, __fmaf_rn(s1.x, gpg, __fmaf_rn(s1.y, gpb, __fmaf_rn(s2.x, gpd, s2.y * gpa)))) / T;
    }

    // (positions are compared as floats -- exact below 2^24 -- and a padding slot of the last trip carries a position behind every
    // contributor: no integer conversion and no validity flag in the trip; this kernel is bound by vector-instruction issue)
    const float last_contributor_f = (float)last_contributor;
    const float nTb = -T_final * bg_dot;
    const v2f pix2 = v2f{pixx, pixy}, gp_rg = v2f{gpr, gpg};
    // wave-uniform address of the block's slab as a buffer resource: a record store is one 32-bit offset + an immediate
    const __amdgpu_buffer_rsrc_t rs_grad = __builtin_amdgcn_make_buffer_rsrc((void*)my_grad, 0, 0xffffff00u, 0x00020000u);
    auto trip = [&](const Fetched& f) {
        // the pair's alpha exactly as the forward kernel decides it (pair_power / eval_pair); the three skip tests (power > 0, alpha <
        // 1/255: backward.cu:507-514; position behind the pixel's last contributor: backward.cu:499) as lane masks in scalar registers
        const PairGeom pg = pair_power<EXACT>(f.a, f.b, pix2);
        const float power = pg.power;
        const float G0 = blend_exp<EXACT>(power);
        const float ao = fminf(0.99f, f.b.w * G0);
        const unsigned long long contrib = __builtin_amdgcn_fcmpf(power, 0.0f, FCMP_OLE) & __builtin_amdgcn_fcmpf(ao, 1.0f / 255.0f, FCMP_OGE) &
                                           __builtin_amdgcn_fcmpf(f.pos1, last_contributor_f, FCMP_OLE);
        const bool on = __builtin_amdgcn_inverse_ballot_w64(contrib);
        const float al = on ? ao : 0.0f;                     // 0 = pair skipped
        const float G = on ? G0 : 0.0f;
        // this entry's state transform:  T' = T / (1-alpha),  Q' = alpha*u + (1-alpha)*Q   (identity if skipped)
        const float mm = 1.0f - al;
        const float rinv = blend_rcp<EXACT>(mm);
        const float u = __fmaf_rn(f.c.x, gpr, __fmaf_rn(f.c.y, gpg, __fmaf_rn(f.c.z, gpb, __fmaf_rn(f.c.w, gpd, gpa))));
        const float kq = al * u;
        // The four slots' transforms in visiting order.  T: slot s ends with T times the product of the slots' 1/(1-alpha) up to its
        // own -- a prefix product over the pixel's four lanes in two steps (the multiply takes its DPP operand directly).  Q: slot s
        // starts from the output of slot s-1, three dependent steps of one fused select and one FMA.
        float To;
        if constexpr (EXACT) {
            // T = T / (1 - alpha), entry after entry, IEEE division (backward.cu:516): slot s divides what slot s-1 left
            float X = T / mm, Y;
            Y = DPP_MOV(X, 0x90); X = slot >= 1 ? Y / mm : X;                // quad_perm:[0,0,1,2]
            Y = DPP_MOV(X, 0x90); X = slot >= 2 ? Y / mm : X;
            Y = DPP_MOV(X, 0x90); X = slot >= 3 ? Y / mm : X;
            To = X;
        } else {
            float P = rinv;
            { const float y = DPP_MOV(P, 0x90) * P; P = slot >= 1 ? y : P; }     // quad_perm:[0,0,1,2]: r0, r0 r1, r1 r2, r2 r3
            { const float y = DPP_MOV(P, 0x44) * P; P = slot >= 2 ? y : P; }     // quad_perm:[0,1,0,1]: r0, r0 r1, r0 r1 r2, r0 r1 r2 r3
            To = T * P;
        }
        float Qi = Q;
        float Qo = __fmaf_rn(mm, Qi, kq);
        Qi = take_prev_slot_unless(Qo, Qi, 0x1111111111111111ull); Qo = __fmaf_rn(mm, Qi, kq);
        Qi = take_prev_slot_unless(Qo, Qi, 0x3333333333333333ull); Qo = __fmaf_rn(mm, Qi, kq);
        Qi = take_prev_slot_unless(Qo, Qi, 0x7777777777777777ull); Qo = __fmaf_rn(mm, Qi, kq);
        T = DPP_MOV(To, 0xFF); Q = DPP_MOV(Qo, 0xFF);               // the pixel's state after these four entries

        if (contrib != 0ull) {
            // To = T after the division (backward.cu:516); (u - Qi) = sum_k (x_k - accum_k) g_k.  G = 0 for a skipped pair zeroes
            // everything below but the colour terms, which carry alpha = 0.
            // The record holds the sums WITHOUT their constant factors (the per-Gaussian kernel applies them once per Gaussian):
            //   [3] sum w (A dx + B dy)   x -0.5 W = dL/dmean2D.x      [5] sum w dx dx   x -0.5 = dL/dconic.x
            //   [4] sum w (C dy + B dx)   x -0.5 H = dL/dmean2D.y      [6] sum w dx dy   x -0.5 = dL/dconic.y   [7] sum w dy dy: .w
            // with w = dL/dG * G (backward.cu:566-580 factored; the record's second word is {B, C, A, opacity}).  (Leaving the conic
            // to the per-Gaussian kernel as well saved four instructions here and cost that kernel a cache line per Gaussian: +0.5 us.)
            const float dL_dopa = __fmaf_rn(u - Qi, To, nTb * rinv);
            const float v8 = G * dL_dopa;
            const float w = f.b.w * v8;
            const v2f wd = v2f{w, w} * pg.d;                                      // (w dx, w dy); packed: two products per instruction
            const v2f v56 = v2f{wd.x, wd.x} * pg.d;
            const float v3 = __fmaf_rn(wd.x, f.b.z, wd.y * f.b.x);
            const float v4 = __fmaf_rn(wd.y, f.b.y, wd.x * f.b.x);
            const float v5 = v56.x, v6 = v56.y, v7 = wd.y * pg.d.y;
            const float dchannel_dcolor = al * To;
            const v2f v01 = v2f{dchannel_dcolor, dchannel_dcolor} * gp_rg;
            const float v0 = v01.x, v1 = v01.y, v2 = dchannel_dcolor * gpb;
            // reduce-scatter over the wave's pixels, separately per slot: after fold32 the lower/upper half-waves hold
            // different values, after fold16 even/odd rows do; lane (row r, slot s) ends with values m0, m1 (and 8 in row 0)
            const float r0 = fold32(v0, v1), r1 = fold32(v2, v3), r2 = fold32(v4, v5), r3 = fold32(v6, v7), r4 = fold32(v8, 0.0f);
            const float s0 = row_slot_sum<4>(fold16(r0, r1)), s1 = row_slot_sum<4>(fold16(r2, r3)),
                        s2 = row_slot_sum<4>(fold16(r4, 0.0f));
            // an entry leaves a record (and its block bit) only if one of the block's pixels blended it: OR of the pixels' bits per
            // slot (scalar unit), spread over the writer lanes (lane & 15 = slot: bits s, 16 + s, 32 + s, 48 + s)
            unsigned long long any = contrib;
            any |= any >> 32; any |= any >> 16; any |= any >> 8; any |= any >> 4;
            const uint32_t any_rows = ((uint32_t)any & 15u) * 0x00010001u;
            if (__builtin_amdgcn_inverse_ballot_w64(((unsigned long long)any_rows << 32) | any_rows)) {
                // the cell of (this entry, this block): (w >> 2) - 16 + block row x box width in the tile + block column (common.h: pack_cell_word)
                const uint32_t w = __float_as_uint(f.a.z);
                const uint32_t cell = (w >> 2) + __umul24(cell_row, (w & 3u) + 1u) + cell_off;
                const uint32_t o = (cell << 5) + (cell << 4) + m0_bytes;                 // x 48 bytes
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s0), rs_grad, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s1), rs_grad, o + 16u, 0, 0);
                if (row == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s2), rs_grad, o + 32u, 0, 0);
                    atomicOr(&cell_valid[cell >> 5], 1u << (cell & 31u));
                }
            }
        }
    };

    // back to front: scan offset o <-> list position n_eff-1-o.  Rounds of (scan -> record DMAs -> wait -> trips); list entries
    // [0, C) are blended, [C, nlist) found.  At most two DMA batches (entries 64q .. 64q+63 -> ring slots (64q & RMASK) + lane) are
    // alive at a time: a round scans while nlist + 64 <= 64 * ((C >> 6) + 2).
    int scan_off = 0, nlist = 0, C = 0, F = 0, grp = CHAPTER;
    bool scan_done = n_eff - lo <= 0;
    uint32_t nx[CHAPTER];
    uint32_t vb_lo = 0u, vb_hi = 0u;
#pragma unroll
    for (int k = 0; k < CHAPTER; k++) nx[k] = bm[max(n_eff - 1 - (64 * k + lane), 0)];
    for (;;) {
        const int limit = 64 * ((C >> 6) + 2);
        while (!scan_done && nlist + 64 <= limit) {
            if (grp == CHAPTER) {                            // (see the forward kernel's scanner)
                unsigned long long bl = 0ull;
#pragma unroll
                for (int k = 0; k < CHAPTER; k++) {
                    const int idx = n_eff - 1 - (scan_off + 64 * k + lane);
                    const unsigned long long b = __ballot(idx >= lo && (((nx[k] | all_hit) >> blk) & 1u) != 0u);
                    bl = lane == k ? b : bl;
                }
                vb_lo = (uint32_t)bl; vb_hi = (uint32_t)(bl >> 32);
#pragma unroll
                for (int k = 0; k < CHAPTER; k++) nx[k] = bm[max(n_eff - 1 - (scan_off + 64 * (CHAPTER + k) + lane), 0)];
                grp = 0;
            }
            const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)vb_hi, grp) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)vb_lo, grp);
            grp++;
            if (m != 0ull) {
                const int r = nlist + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((m >> lane) & 1ull) L->lst[r & LMASK] = __float_as_uint((float)(n_eff - (scan_off + lane)));      // position + 1, as a float
                nlist += __popcll(m);
            }
            scan_off += 64;
            scan_done = scan_off >= n_eff - lo;
        }
        __builtin_amdgcn_wave_barrier();
        if (nlist > F) { dma_records<true>(L, recs, F, nlist, nlist, lane); F = nlist; }
        // the last trip's padding slots (list entries nlist .. next multiple of four: free, the list holds at most LCAP entries from C
        // on): a position no pixel reaches; their ring slots hold the tile's first record (see dma_records)
        __builtin_amdgcn_s_waitcnt(0);
        // (the position a trip compares is the fourth word of the entry's record: for the padding slots it is overwritten in the ring,
        // AFTER the batch that holds them has landed)
        if (scan_done && lane < ((C - nlist) & 3)) L->a[(nlist + lane) & RMASK].w = 3.0e38f;
        __builtin_amdgcn_wave_barrier();
        // (both counters are wave-uniform, but their uses in lane arithmetic above make the compiler keep them in vector registers
        // and derive the trip loop's ring index from those: two vector instructions per trip)
        C = __builtin_amdgcn_readfirstlane(C); nlist = __builtin_amdgcn_readfirstlane(nlist);
        const int avail = nlist - C;
        const int ntrip = (avail >> 2) + ((scan_done && (avail & 3) != 0) ? 1 : 0);
        if (ntrip > 0 && !(flags & 2)) {
            auto get = [&](int ring_slot) -> Fetched {                                      // (scalar ring index + slot: see the forward kernel)
                const int li = ring_slot + slot;
                Fetched f;
                f.a = L->a[li]; f.b = L->b[li]; f.c = L->c[li];
                f.pos1 = f.a.w;
                return f;
            };
            int rs = C & RMASK;
            Fetched f0 = get(rs);
            for (int t = 0; t < ntrip; t += 2) {
                const Fetched f1 = get((rs + 4) & RMASK);
                trip(f0);
                if (t + 1 >= ntrip) break;
                rs = (rs + 8) & RMASK;
                f0 = get(rs);
                trip(f1);
            }
        }
        C += 4 * ntrip;
        if (scan_done) break;
    }
}

// Work-item decode shared by both kernels.  One queue per XCD (workgroups are dealt round-robin to the 8 XCDs, so blockIdx % 8
// names the XCD): the 16 blocks of a tile are pulled by waves that share an L2 (PMC: the forward kernel's HBM traffic equals its
// algorithmic bytes).  Tiles are dealt to the queues in LPT order, rank r -> queue r % nq; of a queue's tiles the first hx are
// heavy (16 block items each), the rest light (4 quadrant items each).
struct WaveItem { int tile, sub, rank; bool heavy, valid; uint2 rg; };
// The FIRST item of a wave is its own rank among the waves of its queue -- no atomic: with one returning atomic per wave at kernel
// start the last of 1024 waves waited 2-12 us for its first item (same-line atomics serialise, see common.h).  Later items:
// (waves of the queue) + the value of the queue head.
__device__ __forceinline__ WaveItem pull_item(uint32_t* my_head, int lane, int nq, int qx, int hx, int n_work,
                                              const uint4* __restrict__ work_table, int first_rank, int q_waves)
{
    int qi = first_rank;
    if (first_rank < 0) {
        if (lane == 0) qi = (int)atomicAdd(my_head, 1u) + q_waves;
        qi = __builtin_amdgcn_readfirstlane(qi);
    }
    WaveItem it;
    it.heavy = qi < WAVE_BLOCKS * hx;
    const int k = it.heavy ? (qi >> 4) : hx + ((qi - WAVE_BLOCKS * hx) >> 2);
    it.sub = it.heavy ? (qi & 15) : ((qi - WAVE_BLOCKS * hx) & 3);
    it.rank = k * nq + qx;
    it.valid = it.rank < n_work;
    // {tile, list start, list end} of the rank in ONE load (the scan block's work table; rank -> tile -> range were two round trips)
    const uint4 wt = it.valid ? work_table[it.rank] : make_uint4(0u, 0u, 0u, 0u);
    it.tile = __builtin_amdgcn_readfirstlane((int)wt.x);
    it.rg = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)wt.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)wt.z));
    return it;
}

// PAIRS = wave pairs per workgroup: 2 (256 threads, four workgroups per CU: the product configuration) or 8 (ONE 1024-thread workgroup
// per CU, all its pairs sharing LDS: the experimental form, diagnostic builds only, MOSS_FWD_PAIRS=8).
template <int PAIRS, bool EXACT>
__global__ void __launch_bounds__(128 * PAIRS)
blend_forward_wave_kernel(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ header,
                          uint32_t* __restrict__ queue_head, const uint4* __restrict__ work_table, const float4* __restrict__ inst_rec,
                          const uint16_t* __restrict__ inst_bmask,
                          const float* __restrict__ bg_color, float* __restrict__ out_color, float* __restrict__ out_depth,
                          float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int flags,
                          unsigned long long* __restrict__ stamps /* optional diagnostics: 8 words per item, else NULL */,
                          uint4* __restrict__ seg_desc, float* __restrict__ seg_state, uint32_t seg_cap, int seg_hits,
                          uint32_t* __restrict__ tail_start, uint32_t* __restrict__ seg_counts, int role_swap, int prio_mode,
                          int seg_store /* 0: MOSS_FORWARD_ONLY -- cut (same image bits as the training forward) but keep nothing */)
{
    // Two wave PAIRS per workgroup (see PairCtl): per pair a record ring + hit list, the cuts of the block being blended, and the
    // control words.  A light item uses 3 KB of the pair's ring per wave as its record ring.
    // (dynamic LDS, FwdLds<PAIRS>::bytes: eight pairs are 157 KB, beyond what a static allocation may hold)
    extern __shared__ __attribute__((aligned(16))) char s_fwd_lds[];
    PairRing* const s_ring = reinterpret_cast<PairRing*>(s_fwd_lds);
    float (*const s_cut_sums)[MAX_CUTS * 16 * 6] = reinterpret_cast<float (*)[MAX_CUTS * 16 * 6]>(s_ring + PAIRS);
    uint2 (*const s_cut_pos)[MAX_CUTS] = reinterpret_cast<uint2 (*)[MAX_CUTS]>(s_cut_sums + PAIRS);
    PairCtl* const s_ctl = reinterpret_cast<PairCtl*>(s_cut_pos + PAIRS);
    static_assert(sizeof(PairRing) >= 2 * 64 * 3 * sizeof(float4), "the light path's rings live inside the pair's ring");
    static_assert(sizeof(PairRing) % 16 == 0, "the arrays behind the rings stay 16-byte aligned");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pair = wv >> 1;   // (the wave's index is wave-uniform: say so)
    // A pair's FIRST item is its rank in its queue, and whether that is a heavy or a light item only the header says: the work-table
    // entry it would be as a heavy item (the common case) is requested here, together with the header words, so that the start-up
    // chain of the kernel is (header | table entry) -> block masks -> records -> first trip: three round trips where round 2 had
    // five (header -> tile order -> range -> masks -> records).
    const int nq_ = min(NUM_XCD_QUEUES, (int)gridDim.x), qx_ = (int)blockIdx.x % nq_;
    const int first_qi = ((int)blockIdx.x / nq_) * PAIRS + pair;
    const uint4 first_wt = work_table[min((first_qi >> 4) * nq_ + qx_, T_tiles - 1)];
    // Which wave of the pair blends: the waves of a workgroup sit on SIMD 0..3 in order, and a CU hosts several workgroups -- the roles
    // are swapped between them so that a SIMD gets blenders (busy) and scanners (mostly waiting) in equal numbers.
    // (eight pairs in one workgroup: pairs p and p + 2 sit on the same two SIMDs -- the roles swap between them)
    const bool is_scanner = PAIRS == 2 ? (((wv ^ role_swap ^ (int)((blockIdx.x >> 3) / 32u)) & 1) == 0)
                                       : (((wv ^ role_swap ^ (pair >> 1)) & 1) == 0);
    if (threadIdx.x < PAIRS * (int)(sizeof(PairCtl) / 4)) reinterpret_cast<uint32_t*>(s_ctl)[threadIdx.x] = 0u;
    // header[8] = this kernel's grid: the backward kernel numbers the cuts by it (how many pairs share a segment region), whatever
    // instantiation -- product or MOSS_DEBUG_EXACT_MATH, with its own residency -- the forward call ran
    if (blockIdx.x == 0 && threadIdx.x == 0) header[8] = gridDim.x;
    __syncthreads();                                         // (the only workgroup barrier of the kernel)
    PairRing* const L = &s_ring[pair];
    PairCtl* const ctl = &s_ctl[pair];
    float4 (*const ring)[3] = reinterpret_cast<float4 (*)[3]>(reinterpret_cast<char*>(L) + (size_t)(wv & 1) * 64 * 3 * sizeof(float4));
    const int n_work = (int)header[5];                       // tile_order lists the tiles that own instances first
    const int nq = min(NUM_XCD_QUEUES, (int)gridDim.x), qx = (int)blockIdx.x % nq;
    const int n_heavy = (int)header[7];
    const int hx = n_heavy > qx ? (n_heavy - qx + nq - 1) / nq : 0;
    const int q_pairs = PAIRS * (((int)gridDim.x - qx + nq - 1) / nq);                // pairs that pull from this queue
    const int my_rank = ((int)blockIdx.x / nq) * PAIRS + pair;                        // this pair's rank among them
    uint32_t* const my_head = queue_head + (size_t)qx * QLINE_WORDS;
    // queue index -> item.  Of a queue's tiles the first hx are heavy (16 block items each), the rest light (4 quadrant items each).
    auto decode = [&](int qi) -> WaveItem {
        WaveItem it;
        it.heavy = qi < WAVE_BLOCKS * hx;
        const int k = it.heavy ? (qi >> 4) : hx + ((qi - WAVE_BLOCKS * hx) >> 2);
        it.sub = it.heavy ? (qi & 15) : ((qi - WAVE_BLOCKS * hx) & 3);
        it.rank = k * nq + qx;
        it.valid = it.rank < n_work;
        const uint4 wt = !it.valid ? make_uint4(0u, 0u, 0u, 0u) : (qi == first_qi && it.heavy) ? first_wt : work_table[it.rank];
        it.tile = (int)wt.x; it.rg = make_uint2(wt.y, wt.z);
        return it;
    };
    auto pop = [&]() -> int {
        int qi = 0;
        if (lane == 0) qi = (int)atomicAdd(my_head, 1u) + q_pairs;
        return __builtin_amdgcn_readfirstlane(qi);
    };
    int light_qi = -1;                                       // scanner: the first light item it drew (the blender draws its own)
    if (is_scanner) {
        // ---- heavy phase, scanner: draws the pair's items.  The FIRST item is the pair's rank in its queue (no atomic: with one
        // returning atomic per wave at kernel start the last wave waited 2-12 us for its first item, see common.h).
        uint32_t seq = 0u;
        int qi = my_rank;
        for (;;) {
            const WaveItem it = decode(qi);
            const uint2 rg = it.rg;
            const bool heavy = it.valid && it.heavy;
            uint32_t nx[CHAPTER];
            if (heavy) {                                     // the first chapter of block masks: requested before the wait below
                const int n = (int)(rg.y - rg.x);
                const uint16_t* const bm = inst_bmask + rg.x;
#pragma unroll
                for (int k = 0; k < CHAPTER; k++) nx[k] = bm[max(min(64 * k + lane, n - 1), 0)];
            }
            while (lds_peek(&ctl->fin_seq) != seq) __builtin_amdgcn_s_sleep(2);       // the blender is done with the previous item
            seq++;
            if (lane == 0) {
                lds_poke(&ctl->ready, 0u); lds_poke(&ctl->consumed, 0u);
                lds_poke(&ctl->post_kind, heavy ? 1u : 0u); lds_poke(&ctl->post_tile, (uint32_t)it.tile);
                lds_poke(&ctl->post_blk, (uint32_t)it.sub); lds_poke(&ctl->post_rank, (uint32_t)it.rank);
                lds_poke(&ctl->rg_x, rg.x); lds_poke(&ctl->rg_y, rg.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_poke(&ctl->post_seq, seq);
            if (!heavy) { light_qi = it.valid ? qi : -1; break; }
            if (prio_mode == 2) set_wave_prio(prio_of_length(rg.y - rg.x));
            else if (prio_mode == 3) set_wave_prio(min(3, prio_of_length(rg.y - rg.x) + 1));
            heavy_forward_scan(it.sub, lane, rg, inst_rec, inst_bmask, L, ctl, seq, flags, nx);
            qi = pop();
        }
    } else {
        // ---- heavy phase, blender: this pair's slot range for depth segments (region of its XCD, q_pairs equal shares)
        SegEmit se;
        {
            const uint32_t share = seg_cap / (uint32_t)max(q_pairs, 1);
            const size_t first = (size_t)qx * seg_cap + (size_t)my_rank * share;
            se.desc = seg_desc + first; se.state = seg_state + first * SEG_STATE_FLOATS;
            se.cap = (seg_hits > 0 && q_pairs <= MAX_FWD_QUEUE_WAVES && seg_cap <= 65535u) ? share : 0u;
            se.store = seg_store != 0;
            if (!se.store && se.cap != 0u) se.cap = 0x7fffffffu;            // (cuts where the training forward would cut; no slot is filled, the range never runs full)
            se.count = 0u; se.seg_hits = seg_hits > 0 ? seg_hits : (1 << 30);
            se.cut_sums = s_cut_sums[pair]; se.cut_pos = s_cut_pos[pair];
        }
        uint32_t seq = 0u;
        for (;;) {
            seq++;
            while (lds_peek(&ctl->post_seq) != seq) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lds_peek(&ctl->post_kind) == 0u) break;
            const int tile = (int)lds_peek(&ctl->post_tile), blk = (int)lds_peek(&ctl->post_blk), rank = (int)lds_peek(&ctl->post_rank);
            const uint2 rg = make_uint2(lds_peek(&ctl->rg_x), lds_peek(&ctl->rg_y));
            // (4: every blender above every scanner; 5: additionally the blenders of the LATER dispatch rounds above those of the
            // earlier ones -- priority outranks age, so this turns the age order of a SIMD's two blenders around; 6: all blenders 3)
            if (prio_mode == 4) set_wave_prio(1);
            else if (prio_mode == 5) set_wave_prio(((int)blockIdx.x * 4 / (int)gridDim.x) >= 2 ? 2 : 1);
            else if (prio_mode == 6) set_wave_prio(3);
            else if (prio_mode == 7) set_wave_prio(1 + ((int)blockIdx.x * 4 / (int)gridDim.x) / 2);
            else if (prio_mode != 0) set_wave_prio(prio_of_length(rg.y - rg.x));
            heavy_forward_blend<EXACT>(W, H, gx, tile, blk, lane, rg, L, ctl, seq, bg_color, out_color, out_depth, out_alpha, final_T, n_contrib,
                                flags, stamps ? stamps + (size_t)(rank * WAVE_BLOCKS + blk) * 8 : nullptr, se, tail_start);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_poke(&ctl->fin_seq, seq);
        }
        // how many of its slots this pair filled (EVERY pair writes its count: the backward kernel sums them per region)
        if (lane == 0 && my_rank < MAX_FWD_QUEUE_WAVES && se.store) seg_counts[(size_t)qx * MAX_FWD_QUEUE_WAVES + my_rank] = se.count;
    }
    if (prio_mode != 0) set_wave_prio(0);
    // ---- light phase: every wave on its own (queue indices only grow: once a pair has drawn a light item, heavy ones are gone)
    for (;;) {
        const int qi = light_qi >= 0 ? light_qi : pop();
        light_qi = -1;
        const WaveItem it = decode(qi);
        if (!it.valid) break;
        light_forward_item<EXACT>(W, H, gx, it.tile, it.sub, lane, it.rg, inst_rec, inst_bmask, ring, bg_color, out_color, out_depth,
                           out_alpha, final_T, n_contrib, flags);
    }

    // Tiles without instances get the background only (forward.cu:374-382 with an empty range); done after the queue so that
    // the heavy items start immediately.
    const int pl = lane >> 2;
    const int wave_id = (int)blockIdx.x * (2 * PAIRS) + wv, n_waves = (int)gridDim.x * (2 * PAIRS);
    for (int i = WAVE_BLOCKS * n_work + wave_id; i < WAVE_BLOCKS * T_tiles; i += n_waves) {
        const int tile = (int)tile_order[i >> 4], blk = i & 15;
        const int px = (tile % gx) * TILE + (blk & 3) * 4 + (pl & 3), py = (tile / gx) * TILE + (blk >> 2) * 4 + (pl >> 2);
        if (px < W && py < H && (lane & 3) == 0) {
            const size_t pix_id = (size_t)W * py + px, plane = (size_t)W * H;
            final_T[pix_id] = 1.0f; n_contrib[pix_id] = 0u;
            out_color[pix_id] = bg_color[0]; out_color[plane + pix_id] = bg_color[1]; out_color[2 * plane + pix_id] = bg_color[2];
            out_alpha[pix_id] = 0.0f; out_depth[pix_id] = 0.0f;
        }
    }
}

template <bool EXACT>
__global__ void __launch_bounds__(256, 4)      // (four waves per SIMD: at most 128 VGPRs)
blend_backward_wave_kernel(int W, int H, int gx, const uint4* __restrict__ work_table, const uint32_t* __restrict__ header,
                           uint32_t* __restrict__ queue_head, const float4* __restrict__ inst_rec,
                           const uint16_t* __restrict__ inst_bmask,
                           const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                           const float* __restrict__ dL_dpixels, const float* __restrict__ dL_ddepths, const float* __restrict__ dL_dalphas,
                           float* __restrict__ inst_grad /* the record pool: [cells][12] */, uint32_t* __restrict__ cell_valid,
                           int flags, unsigned long long* __restrict__ wstamps /* diagnostics: 16 words per wave, else NULL */,
                           const uint4* __restrict__ seg_desc, const float* __restrict__ seg_state, uint32_t seg_cap,
                           const uint32_t* __restrict__ tail_start, const uint32_t* __restrict__ seg_counts, int fwd_grid, int fwd_pairs)
{
    __shared__ HeavyLdsBwd s_heavy[4];                       // per wave: 6.6 KB; a light item uses its first 3 KB as the record ring
    __shared__ uint16_t s_prefix[MAX_FWD_QUEUE_WAVES];       // inclusive prefix sums of the forward waves' segment counts (this XCD's region;
                                                             // 16 bits: the forward cuts nothing when a region has more than 65535 slots)
    static_assert(sizeof(HeavyLdsBwd) >= 64 * 3 * sizeof(float4), "the light path's ring lives inside the heavy path's LDS");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: say so)
    float4 (*const ring)[3] = reinterpret_cast<float4 (*)[3]>(&s_heavy[wv]);
    // a forward that was told MOSS_FORWARD_ONLY left no backward state (and a binning buffer without a record pool): nothing to do --
    // the per-Gaussian backward then writes zero gradients, as after a capacity overflow
    if (header[2] & ERRFLAG_FORWARD_ONLY) return;
    const int n_work = (int)header[5];
    const int nq = min(NUM_XCD_QUEUES, (int)gridDim.x), qx = (int)blockIdx.x % nq;
    const int n_heavy = (int)header[7];
    const int hx = n_heavy > qx ? (n_heavy - qx + nq - 1) / nq : 0;
    // queue_head = the line Q_BWD of the queue area; the segment pop heads are the lines Q_SEG_HEAD
    uint32_t* const seg_heads = queue_head + (size_t)(Q_SEG_HEAD - Q_BWD) * QLINE_WORDS;
    const int q_waves = 4 * (((int)gridDim.x - qx + nq - 1) / nq);
    int first_rank = ((int)blockIdx.x / nq) * 4 + wv;
#define WSTAMP() (wstamps ? __builtin_amdgcn_s_memtime() : 0ull)
#define RSTAMP() (wstamps ? __builtin_amdgcn_s_memrealtime() : 0ull)     /* 100 MHz, the same clock on every CU (s_memtime is not) */
    const unsigned long long t_start = RSTAMP();
    unsigned long long n_seg = 0, c_seg = 0, c_pop = 0, n_tail = 0, c_tail = 0, c_tailpop = 0;
    // ---- 1. the depth segments the forward kernel cut on this XCD (pieces of SEG_HITS hits of the heavy tiles' lists; the records of
    // those tiles are in this L2).  A wave's first segment is its rank among the queue's waves, later ones come from the region's pop head
    // (one returning atomic on the region's own cache line, AFTER the item: requesting the next one while the current item runs --
    // tried -- lets every wave sit on two items, and with fewer than two items per wave that decided the kernel's length).
    if (!(flags & 16)) {
        // The forward waves of this XCD filled private slot ranges; their counts give every segment of the region a flat index.  Each
        // wave builds the prefix table itself (all four waves of the workgroup write the same values: no barrier needed).
        { const int g8 = (int)header[8]; if (g8 > 0) fwd_grid = g8; }     // (the forward kernel's own record of its grid)
        const int fq = min(NUM_XCD_QUEUES, fwd_grid);
        const int f_waves = qx < fq ? fwd_pairs * ((fwd_grid - qx + fq - 1) / fq) : 0;          // forward pairs (blender waves) that fed this region
        uint32_t total = 0u;
        if (f_waves > 0 && f_waves <= MAX_FWD_QUEUE_WAVES && seg_cap <= 65535u) {
            const uint32_t* cnt = seg_counts + (size_t)qx * MAX_FWD_QUEUE_WAVES;
            uint32_t carry = 0u;
            for (int b0 = 0; b0 < f_waves; b0 += 64) {
                uint32_t v = b0 + lane < f_waves ? cnt[b0 + lane] : 0u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += y; }
                v += carry;
                if (b0 + lane < f_waves) s_prefix[b0 + lane] = (uint16_t)v;
                carry = (uint32_t)__shfl((int)v, 63);
            }
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)carry);
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t share = seg_cap / (uint32_t)max(f_waves, 1);
        uint32_t* const head = seg_heads + (size_t)qx * QLINE_WORDS;
        uint32_t i = (uint32_t)first_rank;
        while (i < total) {
            const unsigned long long tp0 = WSTAMP();
            // flat index -> (forward wave, slot in its range): first wave whose inclusive prefix exceeds i
            int lo_w = 0, hi_w = f_waves - 1;
            while (lo_w < hi_w) { const int mid = (lo_w + hi_w) >> 1; if ((uint32_t)s_prefix[mid] <= i) lo_w = mid + 1; else hi_w = mid; }
            const uint32_t before = lo_w > 0 ? (uint32_t)s_prefix[lo_w - 1] : 0u;
            const size_t slot_idx = (size_t)qx * seg_cap + (size_t)lo_w * share + (i - before);
            uint4 d = seg_desc[slot_idx];
            // (wave-uniform by construction; saying so keeps the item's loop control, ring indices and slab address in scalar registers)
            d.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.x); d.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.y);
            d.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.z); d.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.w);
            const unsigned long long tp1 = WSTAMP();
            c_pop += tp1 - tp0;
            n_seg++;
            // {tile | block << 28, first instance of the tile, first position, end position}: everything the item needs in ONE load
            heavy_backward_item<EXACT>(W, H, gx, (int)(d.x & 0x0fffffffu), (int)(d.x >> 28), lane, make_uint2(d.y, d.y + d.w), inst_rec, inst_bmask,
                                &s_heavy[wv], bg_color, final_Ts, n_contrib, dL_dpixels, dL_ddepths, dL_dalphas, inst_grad,
                                cell_valid, flags, (int)d.z, (int)d.w, (flags & 64) ? nullptr : seg_state + slot_idx * SEG_STATE_FLOATS);
            c_seg += WSTAMP() - tp1;
            uint32_t nxt = 0u;
            if (lane == 0) nxt = atomicAdd(head, 1u) + (uint32_t)q_waves;
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
        }
    }
    const unsigned long long t_phase1 = RSTAMP();
    // ---- 2. every block's own item: the part of its list behind the last cut (the whole list if it was never cut).  ALL of them
    // are drawn from the queue (LPT order) when a wave gets here, none by rank: with 4096 waves for ~3500 segments and ~2400 block
    // items a static deal gave wave r segment r AND block item r -- the waves whose segment ended last (35 us) then still had a block
    // item of their own to do; drawn dynamically, the waves without a segment and the early finishers take the long ones first.
    (void)first_rank;
    for (;;) {
        const unsigned long long tq0 = WSTAMP();
        const WaveItem it = pull_item(queue_head + (size_t)qx * QLINE_WORDS, lane, nq, qx, hx, n_work, work_table, -1, 0);
        const unsigned long long tq1 = WSTAMP();
        c_tailpop += tq1 - tq0;
        if (!it.valid) break;
        n_tail++;
        if (it.heavy)
            heavy_backward_item<EXACT>(W, H, gx, it.tile, it.sub, lane, it.rg, inst_rec, inst_bmask, &s_heavy[wv], bg_color, final_Ts,
                                n_contrib, dL_dpixels, dL_ddepths, dL_dalphas, inst_grad, cell_valid, flags,
                                (flags & 32) ? 0 : __builtin_amdgcn_readfirstlane((int)tail_start[(size_t)it.tile * WAVE_BLOCKS + it.sub]),
                                0x7fffffff, nullptr);
        else
            light_backward_item<EXACT>(W, H, gx, it.tile, it.sub, lane, it.rg, inst_rec, inst_bmask, ring, bg_color, final_Ts, n_contrib,
                                dL_dpixels, dL_ddepths, dL_dalphas, inst_grad, cell_valid, flags);
        c_tail += WSTAMP() - tq1;
    }
    if (wstamps && lane == 0) {
        unsigned long long* w = wstamps + ((size_t)blockIdx.x * 4 + wv) * 16;
        w[0] = t_start; w[1] = t_phase1; w[2] = RSTAMP(); w[3] = n_seg; w[4] = c_seg; w[5] = c_pop; w[6] = 0; w[7] = n_tail;
        w[8] = c_tail; w[9] = c_tailpop; w[10] = (unsigned long long)qx;
    }
#undef WSTAMP
#undef RSTAMP
    // (the queue heads of this kernel are rewound by the per-Gaussian backward kernel that follows it on the stream -- a counter that
    // every wave increments on its way out was 1024 serialised atomics at the very end of the kernel)
}

int device_cus()
{
    static const int n = [] {
        int dev = 0; hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        return cus;
    }();
    return n;
}

// Resident workgroups of the persistent blend kernels: a fixed number per CU (dynamic balancing does the rest), never more than fit
// together -- a wave's FIRST work item is its rank in its queue (no atomic), so a workgroup that only started once another one had
// left would sit on its items until then (measured: +10 us when 4 were launched where 3 fit).
template <typename K>
int resident_wgs_per_cu(K kernel, const char* env, int dflt, int cap, int threads = 256, size_t lds = 0)
{
    int occ = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || occ < 1) occ = 1;
    return std::max(1, std::min(std::min(occ, cap), knob(env, dflt)));
}
template <int PAIRS> struct FwdLds { static constexpr size_t bytes = (size_t)PAIRS * (sizeof(PairRing) + sizeof(float) * MAX_CUTS * 16 * 6 + sizeof(uint2) * MAX_CUTS + sizeof(PairCtl)); };
// wave pairs per workgroup of the forward kernel: 2 (product), or 8 = one workgroup per CU (diagnostic builds, MOSS_FWD_PAIRS=8)
// (Measured, round 3: eight pairs in ONE workgroup per CU -- the form in which pairs could hand work to each other through LDS -- is
// 47.1 us against 40.4: its sixteen waves are of one age, the SIMD arbiter then shares issue slots evenly, and the longest item runs
// at ~760 cycles per trip for as long as its SIMD-mate lives instead of at 520-610 as the older wave of two workgroups.)
int forward_pairs()
{
#ifdef MOSS_DIAG
    static const int p = knob("MOSS_FWD_PAIRS", FWD_PAIRS_PER_WG) == 8 ? 8 : FWD_PAIRS_PER_WG;
    return p;
#else
    return FWD_PAIRS_PER_WG;
#endif
}
// (exact: the MOSS_DEBUG_EXACT_MATH instantiation has its own register count, hence its own residency)
int forward_grid(int T, bool exact = false)
{
    static const int per_cu = [] {
#ifdef MOSS_DIAG
        if (forward_pairs() == 8) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blend_forward_wave_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FwdLds<8>::bytes);
            return resident_wgs_per_cu(blend_forward_wave_kernel<8, false>, "MOSS_BLEND_WGS_PER_CU", 1, 1, 1024, FwdLds<8>::bytes);
        }
#endif
        return resident_wgs_per_cu(blend_forward_wave_kernel<2, false>, "MOSS_BLEND_WGS_PER_CU", 4, 4, 256, FwdLds<2>::bytes);
    }();
    static const int per_cu_exact = resident_wgs_per_cu(blend_forward_wave_kernel<2, true>, "MOSS_BLEND_WGS_PER_CU", 4, 4, 256, FwdLds<2>::bytes);
    const bool ex = exact && forward_pairs() == FWD_PAIRS_PER_WG;
    return min((8 / forward_pairs()) * T, device_cus() * (ex ? per_cu_exact : per_cu));
}

}  // anonymous namespace

#ifdef MOSS_DIAG
unsigned long long* g_stamps = nullptr;      // diagnostics buffer registered by moss_raster_debug_set_stamps (NULL = off)
unsigned long long* g_bwd_stamps = nullptr;  // ... by moss_raster_debug_set_bwd_stamps: 16 words per wave of the backward blend kernel
#endif


void launch_blend_forward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                          float* out_color, float* out_depth, float* out_alpha, hipStream_t s)
{
    (void)g;
    static const int cull_knob = knob("MOSS_BLEND_CULL", 1);
    const int flags = fp.no_block_cull ? 0 : cull_knob;             // bit 0: use the per-instance block masks
    const int T = fp.gx * fp.gy;
    const bool exact = fp.exact_math != 0 && forward_pairs() == FWD_PAIRS_PER_WG;   // (MOSS_DEBUG_EXACT_MATH)
    const int wgs = forward_grid(T, exact);                            // 4 independent waves per workgroup, 16 items per tile
    // hits per depth segment of the backward (0 = never cut: every block is ONE backward item, the round-1 behaviour)
    static const int seg_hits_env = [] { const int v = knob("MOSS_SEG_HITS", 64); return (v > 0 && (v & (v - 1)) == 0 && v >= 4) ? v : 0; }();
    const int seg_hits = T < (1 << 28) ? seg_hits_env : 0;             // (a descriptor packs the tile index into 28 bits)
    static const int role_swap = knob("MOSS_FWD_ROLE_SWAP", 0) & 1, prio_mode = knob("MOSS_FWD_PRIO", 0);
#ifdef MOSS_DIAG
    if (forward_pairs() == 8)
        MOSS_LAUNCH_TIMED((blend_forward_wave_kernel<8, false>), dim3(wgs), dim3(1024), FwdLds<8>::bytes, s, fp.W, fp.H, fp.gx, T, im.tile_order, im.header,
                          im.queues + (size_t)Q_FWD * QLINE_WORDS, im.work_table, b.inst_rec, b.inst_bmask, fp.bg_dev, out_color, out_depth, out_alpha,
                          im.final_T, im.n_contrib, flags, g_stamps, b.seg_desc, b.seg_state, b.seg_cap, seg_hits, im.tail_start, im.seg_counts,
                          role_swap, prio_mode, fp.forward_only ? 0 : 1);
    else
#endif
    if (exact)
        MOSS_LAUNCH_TIMED((blend_forward_wave_kernel<2, true>), dim3(wgs), dim3(256), FwdLds<2>::bytes, s, fp.W, fp.H, fp.gx, T, im.tile_order, im.header,
                          im.queues + (size_t)Q_FWD * QLINE_WORDS, im.work_table, b.inst_rec, b.inst_bmask, fp.bg_dev, out_color, out_depth, out_alpha,
                          im.final_T, im.n_contrib, flags, g_stamps, b.seg_desc, b.seg_state, b.seg_cap, seg_hits, im.tail_start, im.seg_counts,
                          role_swap, prio_mode, fp.forward_only ? 0 : 1);
    else
        MOSS_LAUNCH_TIMED((blend_forward_wave_kernel<2, false>), dim3(wgs), dim3(256), FwdLds<2>::bytes, s, fp.W, fp.H, fp.gx, T, im.tile_order, im.header,
                          im.queues + (size_t)Q_FWD * QLINE_WORDS, im.work_table, b.inst_rec, b.inst_bmask, fp.bg_dev, out_color, out_depth, out_alpha,
                          im.final_T, im.n_contrib, flags, g_stamps, b.seg_desc, b.seg_state, b.seg_cap, seg_hits, im.tail_start, im.seg_counts,
                          role_swap, prio_mode, fp.forward_only ? 0 : 1);
}

void launch_blend_backward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                           const float* dL_dpix, const float* dL_ddepth, const float* dL_dalpha, hipStream_t s)
{
    (void)g;
    static const int cull_knob = knob("MOSS_BLEND_CULL", 1);
    // diagnostics, MOSS_DIAG builds only (results are wrong with any of them): 16 = skip the segment items, 32 = block items ignore
    // the cuts, 64 = segment items start from (T_final, 0)
    static const int dbg = knob("MOSS_BWD_DEBUG", 0) & (16 | 32 | 64);
    const int flags = (fp.no_block_cull ? 0 : cull_knob) | dbg;
    const int T = fp.gx * fp.gy;
    static const int bwd_wgs_per_cu = resident_wgs_per_cu(blend_backward_wave_kernel<false>, "MOSS_BWD_WGS_PER_CU", 4, 5);   // 27 KB of LDS each
    static const int bwd_wgs_per_cu_exact = resident_wgs_per_cu(blend_backward_wave_kernel<true>, "MOSS_BWD_WGS_PER_CU", 4, 5);
    // (MOSS_DEBUG_EXACT_MATH must be given to the forward AND the backward call: the forward's grid -- how its cuts are numbered -- follows it)
    const bool exact = fp.exact_math != 0 && forward_pairs() == FWD_PAIRS_PER_WG;
    const int wgs = min(4 * T, device_cus() * (exact ? bwd_wgs_per_cu_exact : bwd_wgs_per_cu));
    // the queue heads are zero here: cleared by the forward, rewound after each backward (preprocess_backward_kernel)
    if (exact)
        MOSS_LAUNCH_TIMED(blend_backward_wave_kernel<true>, dim3(wgs), dim3(256), 0, s, fp.W, fp.H, fp.gx, im.work_table, im.header,
                           im.queues + (size_t)Q_BWD * QLINE_WORDS, b.inst_rec, b.inst_bmask, fp.bg_dev, im.final_T, im.n_contrib, dL_dpix,
                           dL_ddepth, dL_dalpha, reinterpret_cast<float*>(b.inst_grad), b.cell_valid,
                           flags, g_bwd_stamps, b.seg_desc, b.seg_state, b.seg_cap, im.tail_start, im.seg_counts,
                           forward_grid(T, true) /* the forward kernel's grid */, forward_pairs());
    else
        MOSS_LAUNCH_TIMED(blend_backward_wave_kernel<false>, dim3(wgs), dim3(256), 0, s, fp.W, fp.H, fp.gx, im.work_table, im.header,
                           im.queues + (size_t)Q_BWD * QLINE_WORDS, b.inst_rec, b.inst_bmask, fp.bg_dev, im.final_T, im.n_contrib, dL_dpix,
                           dL_ddepth, dL_dalpha, reinterpret_cast<float*>(b.inst_grad), b.cell_valid,
                           flags, g_bwd_stamps, b.seg_desc, b.seg_state, b.seg_cap, im.tail_start, im.seg_counts,
                           forward_grid(T, false) /* the forward kernel's grid */, forward_pairs());
}

}  // namespace moss
