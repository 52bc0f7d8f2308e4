// preprocess.hip -- per-Gaussian kernels: forward preprocess (+ per-tile histogram), fused backward
// (instance-gradient gather + conic/cov2D backward + projection/SH/cov3D backward), markVisible.
//
// COMPILED WITH -ffp-contract=off (see moss_amd/build.py): everything that decides an integer result
// (radius, tile rectangle, depth bits = sort key) is evaluated with one rounding per source operation, in the
// operation order of the reference's source text, so those results are bit-identical to the CPU oracle:
//   frustum test      DGR/cuda_rasterizer/auxiliary.h:139-164
//   projection        DGR/cuda_rasterizer/forward.cu:196-200, auxiliary.h:41-44 (ndc2Pix in double), :46-56 (getRect)
//   cov3D             forward.cu:118-152         cov2D (EWA)  forward.cu:74-113
//   conic / radius    forward.cu:218-237         SH -> RGB    forward.cu:20-71
// These kernels stream ~100-300 B per Gaussian and are HBM-bound; contraction would buy nothing.
#include "common.h"
#include "adamw.h"

namespace moss {

namespace {

struct M3 { float m[3][3]; };   // m[c][r], the column-major convention of the reference's glm matrices

// ---- the GaussianModel getters, for the raw-parameter mode (FrameParams::raw; same expressions as csrc/activations.hip) ----
__device__ __forceinline__ float sigmoid_act(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ void activate_scale_rot(int raw, float* sc, float* q)
{
    if (raw & RAW_SCALE) { sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]); }
    if (raw & RAW_ROTATION) {                                // torch.nn.functional.normalize: x / max(|x|, 1e-12)
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float inv = 1.0f / fmaxf(n, 1e-12f);
        q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
    }
}
// y = q / d, d = max(|q|, eps):  |q| >= eps: dq = (g - y (y.g)) / d;  |q| < eps: d is a constant, dq = g / d.  In place on g.
__device__ __forceinline__ void normalize_backward(const float* q, const float* y, float* g)
{
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    const float yg = n >= 1e-12f ? (y[0] * g[0] + y[1] * g[1] + y[2] * g[2] + y[3] * g[3]) : 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = (g[k] - y[k] * yg) * inv;
}

__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[r][c];
    return R;
}
__device__ __forceinline__ M3 m3_cols(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
{
    M3 R;
    R.m[0][0] = x0; R.m[0][1] = y0; R.m[0][2] = z0;
    R.m[1][0] = x1; R.m[1][1] = y1; R.m[1][2] = z1;
    R.m[2][0] = x2; R.m[2][1] = y2; R.m[2][2] = z2;
    return R;
}

__device__ __forceinline__ int sat_int(float v)   // float->int with saturation, NaN -> 0 (v_cvt_i32_f32 semantics, spelled out)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ float3 xform4x3(float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

__device__ __forceinline__ M3 quat_to_R(float r, float x, float y, float z)   // quaternion used un-normalised (forward.cu:127)
{
    return m3_cols(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* sc, float mod, const float* q, float* cov)
{
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * sc[0]; S.m[1][1] = mod * sc[1]; S.m[2][2] = mod * sc[2];
    M3 R = quat_to_R(q[0], q[1], q[2], q[3]);
    M3 M = m3_mul(S, R);
    M3 Sigma = m3_mul(m3_t(M), M);
    cov[0] = Sigma.m[0][0]; cov[1] = Sigma.m[0][1]; cov[2] = Sigma.m[0][2];
    cov[3] = Sigma.m[1][1]; cov[4] = Sigma.m[1][2]; cov[5] = Sigma.m[2][2];
}

// n2 extension: covariance with a per-Gaussian 3x3 transform INSIDE the op, Sigma' = T Sigma T^T, as
// scene/gaussian_model.py:37-44 builds it in Python.  T row-major, Sigma = 6-float upper triangle.  Same summation order as the
// oracle's transformCov3D / transformCov3D_bw (this file is compiled with -ffp-contract=off: bit-identical).
__device__ __forceinline__ void sym6_to_full(const float* c, float S[3][3])
{
    S[0][0] = c[0]; S[0][1] = c[1]; S[0][2] = c[2];
    S[1][0] = c[1]; S[1][1] = c[3]; S[1][2] = c[4];
    S[2][0] = c[2]; S[2][1] = c[4]; S[2][2] = c[5];
}

__device__ __forceinline__ void transform_cov3d(const float* T, const float* cov6, float* out6)
{
    float S[3][3], tmp[3][3];
    sym6_to_full(cov6, S);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            tmp[a][j] = T[3 * a + 0] * S[0][j] + T[3 * a + 1] * S[1][j] + T[3 * a + 2] * S[2][j];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a; b < 3; b++)
            out6[k++] = tmp[a][0] * T[3 * b + 0] + tmp[a][1] * T[3 * b + 1] + tmp[a][2] * T[3 * b + 2];
}

__device__ __forceinline__ void transform_cov3d_bw(const float* T, const float* cov6_pre, const float* d6, float* d6_pre, float* dT)
{
    float S[3][3], G[3][3], U[3][3];
    sym6_to_full(cov6_pre, S);
    G[0][0] = d6[0];        G[0][1] = 0.5f * d6[1]; G[0][2] = 0.5f * d6[2];
    G[1][0] = 0.5f * d6[1]; G[1][1] = d6[3];        G[1][2] = 0.5f * d6[4];
    G[2][0] = 0.5f * d6[2]; G[2][1] = 0.5f * d6[4]; G[2][2] = d6[5];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            U[a][j] = G[a][0] * T[0 + j] + G[a][1] * T[3 + j] + G[a][2] * T[6 + j];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            dT[3 * a + b] = 2.0f * (U[a][0] * S[0][b] + U[a][1] * S[1][b] + U[a][2] * S[2][b]);
    float Gp[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int jj = i; jj < 3; jj++)
            Gp[i][jj] = T[0 + i] * U[0][jj] + T[3 + i] * U[1][jj] + T[6 + i] * U[2][jj];
    d6_pre[0] = Gp[0][0]; d6_pre[1] = 2.0f * Gp[0][1]; d6_pre[2] = 2.0f * Gp[0][2];
    d6_pre[3] = Gp[1][1]; d6_pre[4] = 2.0f * Gp[1][2]; d6_pre[5] = Gp[2][2];
}

// p = T x + t, T row-major (one rounding per operation, left to right: -ffp-contract=off); the oracle-side tests pose their means
// with the same expression
__device__ __forceinline__ float3 pose_point(const float* T, float3 x, float3 t)
{
    return make_float3(T[0] * x.x + T[1] * x.y + T[2] * x.z + t.x, T[3] * x.x + T[4] * x.y + T[5] * x.z + t.y,
                       T[6] * x.x + T[7] * x.y + T[8] * x.z + t.z);
}

struct Cov2DSetup { float3 t; float txtz, tytz; M3 W, T, Vrk; };

__device__ __forceinline__ Cov2DSetup cov2d_setup(float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                                  const float* cov3D, const float* view)
{
    Cov2DSetup s;
    float3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    s.txtz = t.x / t.z; s.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, s.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, s.tytz)) * t.z;
    M3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0, 0, 0);
    s.W = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    s.T = m3_mul(s.W, J);
    s.Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    s.t = t;
    return s;
}

__constant__ const float SH_C0 = 0.28209479177387814f;
__constant__ const float SH_C1 = 0.4886025119029199f;
__constant__ const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                      -1.0925484305920792f, 0.5462742152960396f };
__constant__ const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                      -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

// ---------------------------------------------------------------------------------------------------------
// K1: forward preprocess, one thread per Gaussian (grid-stride).  Also counts instances per tile: lanes add into
// an LDS-private histogram and the block flushes its non-zero bins with one global atomic each.
// ---------------------------------------------------------------------------------------------------------
constexpr int SH_ROW_F = 49;           // LDS row stride (floats) of a staged 48-float SH record
__global__ void __launch_bounds__(256)
preprocess_forward_kernel(int P, int D, int M, int W, int H, int gx, int gy,
                          float tan_fovx, float tan_fovy, float focal_x, float focal_y, float scale_modifier, int prefiltered,
                          const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                          const float* __restrict__ opacities, const float* __restrict__ scales, const float* __restrict__ rotations,
                          const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
                          const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
                          GeomView g, uint32_t* __restrict__ tile_count, uint32_t* __restrict__ header /* the error-flag word */,
                          int* __restrict__ radii_out, int lds_hist, int stage_sh, const float* __restrict__ transforms, int raw,
                          unsigned long long* __restrict__ stamps /* diagnostics: 8 words per block, else NULL */,
                          const float* __restrict__ translation /* RAW_POSE only, may be NULL */,
                          uint64_t* __restrict__ keys /* scatter mode (below): the tiles' key buckets, else NULL */, uint32_t key_stride)
{
#define FSTAMP(i) if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
    FSTAMP(0);
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_wsum[4];
    const int T = gx * gy;
    // SCATTER MODE (keys != NULL; the asynchronous forward with the histogram in LDS): this kernel also does duplicateWithKeys
    // (rasterizer_impl.cu:70-111).  The reference -- and rounds 1-4 here -- scan the tile counts first, so that every tile's keys land
    // in its final, compact range: a device-wide dependency between "all Gaussians projected" and "first key written", i.e. a kernel
    // boundary and a second pass over the Gaussians (scatter_kernel: 13.6 us on the bench frame, 14.9 us for configs[1]'s 6.9k Gaussians).
    // Here every tile owns a BUCKET of key_stride slots in the key area (the binning buffer exists before this kernel in asynchronous
    // mode: it is sized for the caller's capacity; key_stride = what that area holds / tiles, ~40x the average list): a block reserves
    // its run in a tile's bucket with ONE returning atomic on the tile's counter -- the same atomic that builds the histogram -- and
    // writes its keys there at once.  Nothing needs the scan before the sort, whose workgroups turn the counts into their own chunk
    // tables (binning.hip, chunk_sort_kernel).  A tile that outgrows its bucket drops the frame like a capacity overflow (flagged by
    // the scan block, `needed` scaled so that the caller's next capacity fits it).
    const bool scatter = keys != nullptr;
    if (lds_hist) {
        for (int i = threadIdx.x; i < T; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
    }
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { view[i] = viewmatrix[i]; proj[i] = projmatrix[i]; }
    const float3 campos = make_float3(cam_pos[0], cam_pos[1], cam_pos[2]);

    // every thread runs the same number of iterations (the slot reservation below uses block barriers)
    const int iters = (P + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    // SH staging (M == 16 only): per thread a record is 48 floats at a 192-byte stride, i.e. each of the 48 scalar loads of a wave
    // touches 64 cache lines.  The block's 256 records of this iteration are instead read with coalesced 16-byte loads into LDS
    // (row stride 49 words: conflict-free) and evaluated from there -- same values, same arithmetic, so still bit-exact.
    float* const s_shf = reinterpret_cast<float*>(s_hist + (lds_hist ? ((T + 3) & ~3) : 0));
    for (int it = 0; it < iters; it++) {
        const int idx = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        if (stage_sh) {
            const size_t base4 = (size_t)(it * gridDim.x + blockIdx.x) * blockDim.x * 12, total4 = (size_t)P * 12;
            // DEGREE-AWARE: only the float4 parts of a record that hold a coefficient of the ACTIVE degree are fetched -- 1 / 3 / 7 / 12 of
            // 12 at degree 0 / 1 / 2 / 3 (MOSS trains below degree 3 for iterations 1-2999, train_ZJU.py:85-86); the others are never read
            // by the evaluation below.  Buffer loads: an out-of-range offset returns 0 WITHOUT a memory request, and no branch surrounds
            // the loads (the twelve stay in flight together).  (the launcher stages only while the tensor is < 4 GB: 32-bit offsets)
            const int na = (3 * (D + 1) * (D + 1) + 3) >> 2;
            const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc((void*)shs, 0, 0xffffff00u, 0x00020000u);
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f v[12];
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int f = threadIdx.x + j * blockDim.x;  // (base4 is a multiple of 12: f % 12 is the float4's part of its record)
                const size_t a = min(base4 + (size_t)f, total4 - 1);
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_sh, (f % 12) < na ? (uint32_t)a * 16u : 0xffffffffu, 0, 0);
            }
            __syncthreads();                                 // the previous iteration's readers are done
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int f = threadIdx.x + j * blockDim.x;
                float* d = &s_shf[(f / 12) * SH_ROW_F + (f % 12) * 4];
                d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
            }
            __syncthreads();
        }
        // Every per-Gaussian input is requested HERE, unconditionally (clamped index), before anything is waited for: read where
        // they are used -- means, then (behind the frustum test) scales / rotations, then (behind the covariance and colour math) the
        // opacity -- each was a memory round trip of its own on this kernel's critical path (~2 us each at 1.5 waves per SIMD).
        const size_t ic = (size_t)min(idx, P - 1);
        const float3 p_ld = make_float3(means3D[3 * ic], means3D[3 * ic + 1], means3D[3 * ic + 2]);
        const float opa_ld = opacities[ic];
        float cov_ld[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, sc_ld[3] = { 0.f, 0.f, 0.f }, q_ld[4] = { 0.f, 0.f, 0.f, 0.f };
        float tm_ld[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (cov3D_precomp != nullptr) {                      // (kernel-uniform branches)
#pragma unroll
            for (int i = 0; i < 6; i++) cov_ld[i] = cov3D_precomp[6 * ic + i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++) sc_ld[i] = scales[3 * ic + i];
#pragma unroll
            for (int i = 0; i < 4; i++) q_ld[i] = rotations[4 * ic + i];
            if (transforms != nullptr) {
#pragma unroll
                for (int i = 0; i < 9; i++) tm_ld[i] = transforms[9 * ic + i];
            }
        }
        float3 col_ld = make_float3(0.f, 0.f, 0.f);
        if (colors_precomp != nullptr) col_ld = make_float3(colors_precomp[3 * ic], colors_precomp[3 * ic + 1], colors_precomp[3 * ic + 2]);
        float3 tr_ld = make_float3(0.f, 0.f, 0.f);
        if ((raw & RAW_POSE) && translation != nullptr) tr_ld = make_float3(translation[3 * ic], translation[3 * ic + 1], translation[3 * ic + 2]);
        FSTAMP(1);
        int out_radius = 0; uint32_t out_tiles = 0; uint2 out_rect = make_uint2(0u, 0u);
        float out_depth = 0.0f;
        uint32_t out_cells = 0u;                             // gradient-record cells of this Gaussian (common.h: box_cells)
        if (idx < P) do {
            // RAW_POSE: the canonical position is posed here, p = T x + t (the reference's caller does it with torch ops,
            // gaussian_renderer/__init__.py:74-77); rows of T times x summed left to right, then the translation
            const float3 p_orig = (raw & RAW_POSE) ? pose_point(tm_ld, p_ld, tr_ld) : p_ld;
            const float3 p_view = xform4x3(p_orig, view);
            if (p_view.z <= 0.2f) {                                   // in_frustum, auxiliary.h:154
                if (prefiltered) atomicOr(header, ERRFLAG_PREFILTERED);
                break;
            }
            const float4 p_hom = xform4x4(p_orig, proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);

            float cov3D[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int i = 0; i < 6; i++) cov3D[i] = cov_ld[i];
            } else {
                float sc[3] = { sc_ld[0], sc_ld[1], sc_ld[2] };
                float q[4] = { q_ld[0], q_ld[1], q_ld[2], q_ld[3] };
                activate_scale_rot(raw, sc, q);
                cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
                if (transforms != nullptr) {
                    float Tm[9], pre[6];
#pragma unroll
                    for (int i = 0; i < 9; i++) Tm[i] = tm_ld[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) pre[i] = cov3D[i];
                    transform_cov3d(Tm, pre, cov3D);
                }
#pragma unroll
                for (int i = 0; i < 6; i++) g.cov3D[6 * (size_t)idx + i] = cov3D[i];
            }

            const Cov2DSetup cs = cov2d_setup(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view);
            M3 cov = m3_mul(m3_mul(m3_t(cs.T), m3_t(cs.Vrk)), cs.T);
            cov.m[0][0] += 0.3f;
            cov.m[1][1] += 0.3f;
            const float cx = cov.m[0][0], cy = cov.m[0][1], cz = cov.m[1][1];

            const float det = (cx * cz - cy * cy);
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float3 conic = make_float3(cz * det_inv, -cy * det_inv, cx * det_inv);

            const float mid = 0.5f * (cx + cz);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float2 pix = make_float2(ndc2pix(p_proj.x, W), ndc2pix(p_proj.y, H));
            const int rad = sat_int(my_radius);
            const int x0 = min(gx, max(0, sat_int((pix.x - rad) / TILE)));
            const int y0 = min(gy, max(0, sat_int((pix.y - rad) / TILE)));
            const int x1 = min(gx, max(0, sat_int((pix.x + rad + TILE - 1) / TILE)));
            const int y1 = min(gy, max(0, sat_int((pix.y + rad + TILE - 1) / TILE)));
            if ((uint32_t)(x1 - x0) * (uint32_t)(y1 - y0) == 0u) break;

            float3 rgb;
            uint8_t clamp_bits = 0;
            if (colors_precomp == nullptr) {
                const float* sh = stage_sh ? &s_shf[threadIdx.x * SH_ROW_F] : shs + (size_t)idx * M * 3;
                float3 dir = make_float3(p_orig.x - campos.x, p_orig.y - campos.y, p_orig.z - campos.z);
                const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
                dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
                const float x = dir.x, y = dir.y, z = dir.z;
                float res[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
#define SH(k) sh[3 * (k) + c]
                    float result = SH_C0 * SH(0);
                    if (D > 0) {
                        result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
                        if (D > 1) {
                            const float xx = x * x, yy = y * y, zz = z * z;
                            const float xy = x * y, yz = y * z, xz = x * z;
                            result = result +
                                SH_C2[0] * xy * SH(4) +
                                SH_C2[1] * yz * SH(5) +
                                SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                                SH_C2[3] * xz * SH(7) +
                                SH_C2[4] * (xx - yy) * SH(8);
                            if (D > 2) {
                                result = result +
                                    SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                                    SH_C3[1] * xy * z * SH(10) +
                                    SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                                    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                                    SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                                    SH_C3[5] * z * (xx - yy) * SH(14) +
                                    SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                            }
                        }
                    }
#undef SH
                    result += 0.5f;
                    if (result < 0) clamp_bits |= (uint8_t)(1u << c);
                    res[c] = fmaxf(result, 0.0f);
                }
                rgb = make_float3(res[0], res[1], res[2]);
            } else {
                rgb = col_ld;
            }

            // Conservative extent of the region where this Gaussian can reach alpha >= 1/255 (power >= -tau,
            // tau = ln(255*opacity)): half-widths sqrt(2*tau*cov_xx), sqrt(2*tau*cov_yy) of the ellipse's bounding box,
            // widened by 1e-4 relative + 0.01 px against rounding.  Since round 5 the box does more than let the blend kernels SKIP
            // work: it SIZES and ADDRESSES memory -- the Gaussian's run of gradient-record cells is the box's 4x4 blocks inside its
            // tile rectangle (common.h: box_cells), and the backward blend files a record under the (entry, block) pair's cell.  The block
            // masks are a refinement of this box, so with culling on no pair outside it is ever blended; MOSS_DEBUG_NO_BLOCK_CULL relies
            // on the same containment (a pixel outside the box never reaches alpha >= 1/255: the margins above are four orders of
            // magnitude wider than the arithmetic's error; tests/test_gpu_ops.py::test_block_mask_culling_never_changes_a_result).
            const float opa = (raw & RAW_OPACITY) ? sigmoid_act(opa_ld) : opa_ld;
            float hx = __builtin_huge_valf(), hy = __builtin_huge_valf();
            if (opa == opa) {
                if (!(opa > 0.0f)) { hx = -1.0f; hy = -1.0f; }
                else if (det > 0.0f && cx > 0.0f && cz > 0.0f) {
                    const float tau = logf(255.0f * opa);
                    if (tau != tau || tau == __builtin_huge_valf()) { /* keep inf */ }
                    else if (tau <= 0.0f) { hx = -1.0f; hy = -1.0f; }
                    else {
                        hx = sqrtf(2.0f * tau * cx) * 1.0001f + 0.01f;
                        hy = sqrtf(2.0f * tau * cz) * 1.0001f + 0.01f;
                    }
                }
            }

            g.geo[4 * (size_t)idx + 0] = make_float4(pix.x, pix.y, hx, hy);  // position + cull extents (the cull record)
            g.geo[4 * (size_t)idx + 1] = make_float4(conic.x, conic.y, conic.z, opa);
            g.geo[4 * (size_t)idx + 2] = make_float4(rgb.x, rgb.y, rgb.z, p_view.z);
            g.clamped[idx] = clamp_bits;
            out_radius = rad;
            out_depth = p_view.z;
            out_tiles = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
            out_rect = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
            { const BoxCells bc = box_cells(pix.x, pix.y, hx, hy, out_rect); out_cells = (uint32_t)(max(bc.nbx, 0) * max(bc.nby, 0)); }

        } while (0);
        FSTAMP(2);
        // per-tile histogram of the (Gaussian, tile) instances; large rectangles are walked by the whole wave
        wave_for_each_tile(out_rect, gx, 0ull, [&](int t, uint64_t) {
            if (lds_hist) atomicAdd(&s_hist[t], 1u);
            else atomicAdd(&tile_count[t], 1u);
        });
        // ---- scatter mode, first half: one returning atomic per (block, non-empty tile) reserves the block's run in the tile's bucket and
        // adds to the tile's count.  Four tiles per thread at a time, as BUFFER atomics whose offset is out of range for an empty tile
        // (dropped without a memory request): no branch around the atomic, so the four are in flight together (a device-scope
        // returning atomic is ~2 us; see scatter_kernel, which this replaces on the asynchronous path).  The LAST batch's answers --
        // for up to 1024 tiles the only one -- are waited for BEHIND the slot-run prefix sum and its stores below, not here.
        uint32_t pend_c[4] = { 0u, 0u, 0u, 0u }, pend_old[4] = { 0u, 0u, 0u, 0u };
        int pend_b0 = 0;
        if (scatter) {
            __syncthreads();                                 // this iteration's counts are complete
            const __amdgpu_buffer_rsrc_t rs_cnt = __builtin_amdgcn_make_buffer_rsrc((void*)tile_count, 0, 0xffffff00u, 0x00020000u);
            for (int b0 = 0; b0 < T; b0 += 4 * (int)blockDim.x) {
                uint32_t c[4], old[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                    c[u] = i < T ? s_hist[i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                    old[u] = (uint32_t)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)c[u], rs_cnt, c[u] ? (uint32_t)i * 4u : 0xfffffffcu /* out of range AND dword-aligned */, 0, 0);
                }
                if (b0 + 4 * (int)blockDim.x < T) {          // (kernel-uniform) not the last batch: its answers are needed now
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                        if (c[u]) s_hist[i] = old[u];        // (the count word becomes the CURSOR of the pass below: it starts at the run's start)
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) { pend_c[u] = c[u]; pend_old[u] = old[u]; }
                    pend_b0 = b0;
                }
            }
        }
        // rec_offsets: each Gaussian needs a private run of `out_cells` records in the record pool (the reference's counterpart: the
        // device-wide inclusive scan of tiles_touched, rasterizer_impl.cu:279).  Here: the run's start RELATIVE to the block's group of 256
        // Gaussians (a block prefix sum) and the group's total; the scan block that rides along with the sort kernel turns the totals
        // into group bases.  (Round 1-2 reserved the group's run with ONE returning atomic per block on a header word: 391 atomics on
        // one address, 11 ns each -- the last block got its answer 4.5 us after the first: the tail of this kernel, scripts/sort_stamps.py.)
        uint32_t rincl = out_cells;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ry = __shfl_up(rincl, d);
            if (lane >= d) rincl += ry;
        }
        if (lane == 63) s_wsum[wv] = rincl;
        __syncthreads();
        if (threadIdx.x == 0 && (it * gridDim.x + blockIdx.x) * blockDim.x < (unsigned)P)
            g.group_rtot[it * gridDim.x + blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        uint32_t rwbase = 0u;
        for (int w = 0; w < wv; w++) rwbase += s_wsum[w];
        if (idx < P) {
            g.radius[idx] = out_radius;
            g.tiles_touched[idx] = out_tiles;
            g.rec_offsets[idx] = rwbase + rincl - out_cells; g.rec_count[idx] = out_cells;
            // (third word: where the Gaussian's cell run starts, relative to its group: merge_gather reads it with the rectangle)
            g.geo[4 * (size_t)idx + 3] = make_float4(__uint_as_float(out_rect.x), __uint_as_float(out_rect.y),
                                                     __uint_as_float(rwbase + rincl - out_cells), __uint_as_float(out_cells));
            if (radii_out) radii_out[idx] = out_radius;
        }
        __syncthreads();
        if (scatter) {
            // ---- scatter mode, second half: the reserved runs' starts, then the keys -- (depth bits << 32 | Gaussian id) -- into them (slot
            // order inside a bucket is arbitrary: the sort orders by the whole key)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = pend_b0 + u * (int)blockDim.x + (int)threadIdx.x;
                if (pend_c[u]) s_hist[i] = pend_old[u];       // (the count word becomes the cursor of the pass below)
            }
            __syncthreads();
            const uint64_t key = out_tiles ? (((uint64_t)__float_as_uint(out_depth) << 32) | (uint32_t)idx) : 0ull;
            wave_for_each_tile(out_rect, gx, key, [&](int t, uint64_t k) {
                const uint32_t pos = atomicAdd(&s_hist[t], 1u);
                if (pos < key_stride) keys[(size_t)t * key_stride + pos] = k;      // (a bucket that overflows drops the frame: scan block)
            });
            if (it + 1 < iters) {                            // the next iteration counts from zero again
                __syncthreads();
                for (int i = threadIdx.x; i < T; i += blockDim.x) s_hist[i] = 0;
                __syncthreads();
            }
        }
        FSTAMP(4);
    }
    if (lds_hist && !scatter) {                              // (scatter mode added its counts with the reserving atomics)
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += blockDim.x) {
            const uint32_t v = s_hist[i];
            if (v && !(raw & 0x100)) atomicAdd(&tile_count[i], v);        // (0x100: timing experiment, MOSS_EXPERIMENT=1 -- wrong counts)
        }
    }
    FSTAMP(5);
#undef FSTAMP
}

// ---------------------------------------------------------------------------------------------------------
// K8+K9 fused: per-Gaussian backward.  Sums this Gaussian's per-instance partial gradients (written by the
// blend-backward kernel, one 48-byte record per (Gaussian, tile) instance, found through inst_pos) in a FIXED
// order -> bitwise reproducible; then conic->cov2D->cov3D/mean (backward.cu:144-274), projection of the 2-D mean
// gradient (:370-387), SH (:20-139) and scale/rotation (:278-341).  Writes every output element exactly once.
// ---------------------------------------------------------------------------------------------------------
// ---- steps (1)-(4) of the per-Gaussian backward: conic gradient -> dL/dcov3D (six-vector) and dL/dmean (covariance and
// screen-position paths).  gca/gcb/gcc = dL/dconic (A, B per off-diagonal entry, C), gmx/gmy = dL/dmean2D in NDC units.
__device__ __forceinline__ void cov_proj_backward(int idx, float gca, float gcb, float gcc, float gmx, float gmy,
                                                  float tan_fovx, float tan_fovy, float h_x, float h_y,
                                                  const float (&c6)[6] /* the 3-D covariance the forward used (given or stored by it) */,
                                                  const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
                                                  const float3 mean /* the (posed) mean the forward projected */, float* dmean, float* dcov, float (&A_out)[2][3], float (&d2_out)[3])
{
    // ================= per-Gaussian backward, in matrix form =================================================================
    // What is differentiated is the reference's forward (forward.cu:74-113, 118-152, 20-71, 196-237) with the conventions its
    // backward fixes (backward.cu:144-396): 1/(det^2 + 1e-7) in the inverse, no gradient through a clamped t.x/t.z, the SH clamp
    // flags, the quaternion used as given.  The expressions are this file's own: every step is a small matrix identity.
    //
    // (1) conic K = inverse(S2), S2 = [[a b][b c]] the dilated 2-D covariance.  The blend kernels deliver G = dL/dK as a full
    //     symmetric matrix [[gA gB][gB gC]] (gB per off-diagonal entry).  d(inverse):  dL/dS2 = -K G K = -(adj G adj) / det^2 with
    //     adj = [[c -b][-b a]]; the reference regularises 1/det^2 as 1/(det^2 + 1e-7) (backward.cu:203) -- kept.
    // (2) S2 = A S3 A^T + 0.3 I with A = J Rv (2x3): dL/dS3 = A^T dS2 A (the six-vector doubles the off-diagonals, each
    //     appearing twice in the matrix), dL/dA = 2 dS2 A S3.
    // (3) A = J Rv: dL/dJ = dL/dA Rv^T; J = [[fx/tz 0 -fx tx/tz^2][0 fy/tz -fy ty/tz^2]] so, written with J's own entries,
    //     dJ00/dtz = -J00/tz, dJ02/dtx = -J00/tz, dJ02/dtz = -2 J02/tz (same for the y row); dL/dmean = Rv^T dL/dt.
    const float Rv[3][3] = { { viewmatrix[0], viewmatrix[4], viewmatrix[8] }, { viewmatrix[1], viewmatrix[5], viewmatrix[9] },
                             { viewmatrix[2], viewmatrix[6], viewmatrix[10] } };            // t = Rv p + (view[12..14])
    (void)idx;
    const float S3[3][3] = { { c6[0], c6[1], c6[2] }, { c6[1], c6[3], c6[4] }, { c6[2], c6[4], c6[5] } };
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; r++) t[r] = Rv[r][0] * mean.x + Rv[r][1] * mean.y + Rv[r][2] * mean.z + viewmatrix[12 + r];
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float rx = t[0] / t[2], ry = t[1] / t[2];
    const bool x_free = !(rx < -limx || rx > limx), y_free = !(ry < -limy || ry > limy);     // forward.cu:82-87 clamp inactive
    const float tx = fminf(limx, fmaxf(-limx, rx)) * t[2], ty = fminf(limy, fmaxf(-limy, ry)) * t[2];
    const float itz = 1.0f / t[2];
    const float J00 = h_x * itz, J11 = h_y * itz, J02 = -(h_x * tx) * (itz * itz), J12 = -(h_y * ty) * (itz * itz);
    float A[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) { A[0][k] = J00 * Rv[0][k] + J02 * Rv[2][k]; A[1][k] = J11 * Rv[1][k] + J12 * Rv[2][k]; }
    float AS[2][3];                                                                          // A S3
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) AS[i][k] = A[i][0] * S3[0][k] + A[i][1] * S3[1][k] + A[i][2] * S3[2][k];
    const float a2 = AS[0][0] * A[0][0] + AS[0][1] * A[0][1] + AS[0][2] * A[0][2] + 0.3f;
    const float b2 = AS[0][0] * A[1][0] + AS[0][1] * A[1][1] + AS[0][2] * A[1][2];
    const float c2 = AS[1][0] * A[1][0] + AS[1][1] * A[1][1] + AS[1][2] * A[1][2] + 0.3f;
    const float det = a2 * c2 - b2 * b2;
    const float w = 1.0f / (det * det + 0.0000001f);
    // (1)  dS2 = -w adj G adj
    const float u0 = c2 * gca - b2 * gcb, u1 = c2 * gcb - b2 * gcc;           // rows of adj G
    const float v0 = a2 * gcb - b2 * gca, v1 = a2 * gcc - b2 * gcb;
    float d2[2][2];
    d2[0][0] = -w * (u0 * c2 - u1 * b2);
    d2[0][1] = -w * (u1 * a2 - u0 * b2);
    d2[1][1] = -w * (v1 * a2 - v0 * b2);
    d2[1][0] = d2[0][1];
    if (w == 0.0f) { d2[0][0] = 0.f; d2[0][1] = 0.f; d2[1][0] = 0.f; d2[1][1] = 0.f; }   // (denom2inv == 0 case of backward.cu:205)
#pragma unroll
    for (int k = 0; k < 3; k++) { A_out[0][k] = A[0][k]; A_out[1][k] = A[1][k]; }             // (for the scale / rotation gradient: step (6))
    d2_out[0] = d2[0][0]; d2_out[1] = d2[0][1]; d2_out[2] = d2[1][1];
    // (2)  dS3 = A^T dS2 A,  dA = 2 dS2 (A S3)
    float DA[2][3];                                                                          // dS2 A
#pragma unroll
    for (int k = 0; k < 3; k++) { DA[0][k] = d2[0][0] * A[0][k] + d2[0][1] * A[1][k]; DA[1][k] = d2[1][0] * A[0][k] + d2[1][1] * A[1][k]; }
    {
        const float f00 = A[0][0] * DA[0][0] + A[1][0] * DA[1][0], f11 = A[0][1] * DA[0][1] + A[1][1] * DA[1][1],
                    f22 = A[0][2] * DA[0][2] + A[1][2] * DA[1][2];
        const float f01 = A[0][0] * DA[0][1] + A[1][0] * DA[1][1], f02 = A[0][0] * DA[0][2] + A[1][0] * DA[1][2],
                    f12 = A[0][1] * DA[0][2] + A[1][1] * DA[1][2];
        dcov[0] = f00; dcov[3] = f11; dcov[5] = f22; dcov[1] = 2.0f * f01; dcov[2] = 2.0f * f02; dcov[4] = 2.0f * f12;
    }
    float gA[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) { gA[0][k] = 2.0f * (d2[0][0] * AS[0][k] + d2[0][1] * AS[1][k]); gA[1][k] = 2.0f * (d2[1][0] * AS[0][k] + d2[1][1] * AS[1][k]); }
    // (3)  dJ = dA Rv^T (only the four non-zero entries of J), then t, then the mean
    const float dJ00 = gA[0][0] * Rv[0][0] + gA[0][1] * Rv[0][1] + gA[0][2] * Rv[0][2];
    const float dJ02 = gA[0][0] * Rv[2][0] + gA[0][1] * Rv[2][1] + gA[0][2] * Rv[2][2];
    const float dJ11 = gA[1][0] * Rv[1][0] + gA[1][1] * Rv[1][1] + gA[1][2] * Rv[1][2];
    const float dJ12 = gA[1][0] * Rv[2][0] + gA[1][1] * Rv[2][1] + gA[1][2] * Rv[2][2];
    float dt[3];
    dt[0] = x_free ? -(J00 * itz) * dJ02 : 0.0f;
    dt[1] = y_free ? -(J11 * itz) * dJ12 : 0.0f;
    dt[2] = -itz * (J00 * dJ00 + J11 * dJ11 + 2.0f * (J02 * dJ02 + J12 * dJ12));
#pragma unroll
    for (int k = 0; k < 3; k++) dmean[k] = Rv[0][k] * dt[0] + Rv[1][k] * dt[1] + Rv[2][k] * dt[2];

    // (4) screen position: ndc = (P p)_{xy} / ((P p)_w + 1e-7); the gradient the blend delivers (gmx, gmy) is w.r.t. ndc
    //     (backward.cu:370-387): d ndc_x / dp = (row_x - ndc_x row_w) / w'.   P's row k is projmatrix[4 c + k].
    {
        const float hx = projmatrix[0] * mean.x + projmatrix[4] * mean.y + projmatrix[8] * mean.z + projmatrix[12];
        const float hy = projmatrix[1] * mean.x + projmatrix[5] * mean.y + projmatrix[9] * mean.z + projmatrix[13];
        const float hw = projmatrix[3] * mean.x + projmatrix[7] * mean.y + projmatrix[11] * mean.z + projmatrix[15];
        const float iw = 1.0f / (hw + 0.0000001f);
        const float nx = hx * iw, ny = hy * iw;
#pragma unroll
        for (int k = 0; k < 3; k++)
            dmean[k] += iw * ((projmatrix[4 * k] - nx * projmatrix[4 * k + 3]) * gmx + (projmatrix[4 * k + 1] - ny * projmatrix[4 * k + 3]) * gmy);
    }

}

// ---- step (6): dL/dscale, dL/drot (and dL/dtransforms), chained through the raw-parameter getters if asked
__device__ __forceinline__ void scale_rot_backward(int idx, const float* dcov, const float (&A)[2][3], const float (&d2)[3],
                                                   float scale_modifier, int raw,
                                                   bool has_scales, const float (&sc_in)[3], const float (&q_in)[4],
                                                   bool has_transforms, const float (&Tm)[9], float* dscale, float* drot, float* dtf)
{
    // (6) S3 = L L^T, L = R(q) diag(mod s) with the quaternion as given (forward.cu:118-152; with a transform: L = T R diag(mod s)).
    //     dL/dL = 2 dS3 L with dS3 = A^T dS2 A (step 2).  The reference forms the six-vector dS3 in float32 and multiplies it by L
    //     (backward.cu:278-341); for a needle -- scales 300 : 1 : 1 -- whose thin axes point nearly along the viewing ray that is a
    //     difference of terms 10^3 times its size TWICE over (dS3 r_k, then r_k . (dS3 r_k)): the gradient of the thin scales comes out
    //     with 10^-7 x 10^3..10^4 of relative error, in the reference's arithmetic and in ours alike (the fuzz outliers of
    //     tests/golden/fuzz_outlier_seeds.json).  Evaluated from the right instead,
    //         u_k = A' r_k (A' = A T),   w_k = dS2 u_k,   column k of dL/dL = 2 m_k A'^T w_k,   dL/dm_k = 2 m_k u_k . w_k   (m = mod s),
    //     the small vector u_k is formed ONCE and the quadratic form is taken of it: the cancellation enters once, not squared.
    //     That is how dL/dm_k is evaluated here (seed 3163 of the fuzz: 5e-3 of the Gaussian's contribution mass from float64 -> 1.3e-6;
    //     the float32 restatement of the reference: 4.5e-5).
    //     G = dL/dR, G_ik = (dL/dL)_ik m_k;  and for R(q) = I + 2 [..] with q = (r, v):
    //       dq_r = 2 v . a,   dq_v = 2 (Soff v + r a) - 4 v * (tr G - diag G),   a = (G21-G12, G02-G20, G10-G01), Soff = offdiag(G + G^T).
    (void)idx;
    if (has_scales) {
        float scr[3] = { sc_in[0], sc_in[1], sc_in[2] };
        float qr[4] = { q_in[0], q_in[1], q_in[2], q_in[3] };
        const float q_raw[4] = { qr[0], qr[1], qr[2], qr[3] };
        activate_scale_rot(raw, scr, qr);                    // raw mode: exp / normalize as in the forward
        float Ap[2][3] = { { A[0][0], A[0][1], A[0][2] }, { A[1][0], A[1][1], A[1][2] } };   // A' = A T
        float d6[6] = { dcov[0], dcov[1], dcov[2], dcov[3], dcov[4], dcov[5] };
        if (has_transforms) {
            float pre[6], d6_pre[6];
            cov3d_from_scale_rot(scr, scale_modifier, qr, pre);
            transform_cov3d_bw(Tm, pre, d6, d6_pre, dtf);
#pragma unroll
            for (int i = 0; i < 6; i++) d6[i] = d6_pre[i];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) Ap[j][i] = A[j][0] * Tm[i] + A[j][1] * Tm[3 + i] + A[j][2] * Tm[6 + i];
        }
        const float qw = qr[0], qx = qr[1], qy = qr[2], qz = qr[3];
        const float R[3][3] = { { 1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy) },
                                { 2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx) },
                                { 2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy) } };
        const float sm[3] = { scale_modifier * scr[0], scale_modifier * scr[1], scale_modifier * scr[2] };
        const float D3[3][3] = { { d6[0], 0.5f * d6[1], 0.5f * d6[2] }, { 0.5f * d6[1], d6[3], 0.5f * d6[4] }, { 0.5f * d6[2], 0.5f * d6[4], d6[5] } };
        float G[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // the scale gradient from the right (see above) ...
            const float u0 = Ap[0][0] * R[0][k] + Ap[0][1] * R[1][k] + Ap[0][2] * R[2][k];
            const float u1 = Ap[1][0] * R[0][k] + Ap[1][1] * R[1][k] + Ap[1][2] * R[2][k];
            const float w0 = d2[0] * u0 + d2[1] * u1, w1 = d2[1] * u0 + d2[2] * u1;
            dscale[k] = 2.0f * sm[k] * (u0 * w0 + u1 * w1);  // w.r.t. mod * s, reported as is (backward.cu:322-325: no factor mod)
            // ... dL/dR from the symmetric dS3, as the reference does (its cancellation is single; and an exactly symmetric dS3 gives
            // the quaternion of an isotropic, unrotated Gaussian -- MOSS's initialisation -- a gradient of exactly zero, like the
            // reference: noise of 1e-7 there would still move it by a full learning rate per step under Adam)
            float col[3];                                    // column k of dL/dL = 2 dS3 L
#pragma unroll
            for (int i = 0; i < 3; i++) col[i] = 2.0f * sm[k] * (D3[i][0] * R[0][k] + D3[i][1] * R[1][k] + D3[i][2] * R[2][k]);
#pragma unroll
            for (int i = 0; i < 3; i++) G[i][k] = col[i] * sm[k];
        }
        const float a0 = G[2][1] - G[1][2], a1 = G[0][2] - G[2][0], a2q = G[1][0] - G[0][1];
        const float s01 = G[0][1] + G[1][0], s02 = G[0][2] + G[2][0], s12 = G[1][2] + G[2][1];
        const float trG = G[0][0] + G[1][1] + G[2][2];
        drot[0] = 2.0f * (qx * a0 + qy * a1 + qz * a2q);
        drot[1] = 2.0f * (qy * s01 + qz * s02 + qw * a0) - 4.0f * qx * (trG - G[0][0]);
        drot[2] = 2.0f * (qx * s01 + qz * s12 + qw * a1) - 4.0f * qy * (trG - G[1][1]);
        drot[3] = 2.0f * (qx * s02 + qy * s12 + qw * a2q) - 4.0f * qz * (trG - G[2][2]);
        // raw mode: chain through the getters (same expressions as csrc/activations.hip)
        if (raw & RAW_SCALE) {
#pragma unroll
            for (int k = 0; k < 3; k++) dscale[k] *= scr[k];                                  // d exp(v) = exp(v)
        }
        if (raw & RAW_ROTATION) normalize_backward(q_raw, qr, drot);
    }
}

// A Gaussian's gradient records are the CELLS of its run in the record pool whose validity bit is set (common.h: box_cells; the bits
// are set by the backward blend).  One that covers much of the image owns thousands of cells -- a serial sum of thousands of 48-byte
// gathers for one lane -- so those (`is_big`: more than COOP_WORDS words of validity bits) are summed by the 64 lanes of the wave
// together, one Gaussian at a time: lane l takes cells c0 + l, c0 + l + 64, ... (consecutive lanes read consecutive records: a wide
// Gaussian's cells are mostly flagged, and its run is contiguous -- rounds 1-4 walked instance slots x slabs with dependent loads),
// eight cells per lane in flight; then a fixed butterfly.  A fixed order, hence bitwise reproducible.  The owner's lane receives
// sums[0..8] = {colour r, g, b, mean2D x, y, conic A, B, C, opacity}.  Must be called with the whole wave converged.
__device__ __forceinline__ uint32_t run_word_mask(uint32_t word, uint32_t c0, uint32_t c1)   // bits of validity word `word` that belong to cells [c0, c1)
{
    const uint32_t base = word << 5;
    const uint32_t lo = c0 > base ? c0 - base : 0u, hi = c1 > base ? min(c1 - base, 32u) : 0u;
    if (hi <= lo) return 0u;
    return (hi >= 32u ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
}
// NG Gaussians per round, 64 / NG lanes each: 1 -- the whole wave on one Gaussian -- for runs of more than COOP_WORDS validity words;
// 4 -- sixteen lanes each, 128 cells per Gaussian and round -- for the MID-SIZE runs (more than MID_CELLS cells): at MOSS's own sizes
// (45 695 Gaussians at 1024 x 1024: ~50 cells and ~25 records per Gaussian, a tail of several hundred) the per-lane forms below file a
// lane's hundreds of records one by one while 63 lanes wait, and eight of 714 blocks took 190k cycles against a median of 55k
// (profiles/r05_notes.md section 13).
template <int NG>
__device__ __forceinline__ void coop_gather(bool is_big, uint32_t c0, uint32_t c1,
                                            const uint32_t* __restrict__ cell_valid, const float4* __restrict__ inst_grad, float* sums)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr uint32_t OOB = 0xffffffffu, RSRC3 = 0x00020000u;
        constexpr uint32_t LPR = 64u / (uint32_t)NG;                             // lanes per Gaussian of a round
        // (buffer loads: an out-of-range offset returns 0 without a memory request -- an absent or unflagged cell costs nothing and adds 0)
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)cell_valid, 0, 0xffffff00u, RSRC3);
        const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc((void*)inst_grad, 0, 0xffffff00u, RSRC3);
        unsigned long long big = __ballot(is_big);
        const uint32_t lane = threadIdx.x & 63u, grp = lane / LPR, j = lane % LPR;
        while (big) {
            // this round's Gaussians: the next NG flagged lanes (wave-uniform), group g of lanes takes src[g]
            int src[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) { src[g] = big ? __ffsll(big) - 1 : -1; big &= big - 1ull; }
            int my_src = src[0];
#pragma unroll
            for (int g = 1; g < NG; g++) my_src = grp == (uint32_t)g ? src[g] : my_src;
            // (both shuffles by EVERY lane: a lane that sat out a conditional shuffle would hand its value to nobody -- ds_bpermute returns 0
            // for an inactive source lane -- and the owner of a run is usually not among the lanes that sum it)
            const uint32_t bc0 = __shfl(c0, max(my_src, 0)), bc1_any = __shfl(c1, max(my_src, 0));
            const uint32_t bc1 = my_src >= 0 ? bc1_any : bc0;
            uint32_t nmax = 0u;                                                  // the longest run of the round: the loop below is wave-uniform
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const uint32_t n = src[g] >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)(c1 - c0), max(src[g], 0)) : 0u;
                nmax = max(nmax, n);
            }
            float acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            constexpr int CPL = 8;                                               // cells per lane per round
            for (uint32_t r0 = 0u; r0 < nmax; r0 += LPR * CPL) {
                uint32_t vw[CPL];
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    const uint32_t cell = bc0 + r0 + LPR * (uint32_t)i + j;
                    vw[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_msk, cell < bc1 ? (cell >> 5) * 4u : OOB, 0, 0);
                }
                v4f ra[CPL], rb[CPL]; float rc[CPL];
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    const uint32_t cell = bc0 + r0 + LPR * (uint32_t)i + j;
                    const bool on = ((vw[i] >> (cell & 31u)) & 1u) != 0u;       // (vw = 0 beyond the run)
                    const uint32_t o = on ? cell * (uint32_t)(GRAD_REC_FLOATS * 4) : OOB;
                    ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, o, 0, 0);
                    rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, on ? o + 16u : OOB, 0, 0);
                    rc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rec, on ? o + 32u : OOB, 0, 0));
                }
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    acc[0] += ra[i].x; acc[1] += ra[i].y; acc[2] += ra[i].z; acc[3] += ra[i].w;
                    acc[4] += rb[i].x; acc[5] += rb[i].y; acc[6] += rb[i].z; acc[7] += rb[i].w;
                    acc[8] += rc[i];
                }
            }
            // a fixed butterfly over the Gaussian's lanes, then every owner picks up its group's sums
#pragma unroll
            for (int q = 0; q < 9; q++) {
#pragma unroll
                for (int d = (int)LPR / 2; d >= 1; d >>= 1) acc[q] += __shfl_xor(acc[q], d);
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (src[g] >= 0) {                                               // (wave-uniform)
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[q]), g * (int)LPR));
                        if ((int)lane == src[g]) sums[q] = v;
                    }
                }
            }
        }
    }

// STAGE_SH (requires M == 16, shs and dL_dsh given): the block's SH records are read from HBM with coalesced loads into LDS and
// the dL_dsh records leave the same way.  Per thread a record is 48 floats at a 192-byte stride, i.e. every one of the 48 loads
// and 48 stores of a wave would touch 64 different cache lines; through LDS (row stride 49 words: conflict-free) the global side
// is 12 fully coalesced 16-byte accesses per lane.
// Gradient records are sparse: cell c of the Gaussian's run holds a record only if bit c of cell_valid is set (blend.hip).
constexpr int SH_ROW = 49;
constexpr int GATHER_CAP = 320;                          // records of a wave gathered per pass (5 rows of 64; 384: no gain, round 5)
constexpr int GATHER_WORDS = GATHER_CAP + 64 * 9;        // per wave: the descriptor list + one row of scanned values
// FUSED: the kernel also takes the AdamW step of the parameters named in fa.tensors (moss_raster_backward_raw_adamw) -- an
// instantiation of its own, so that the plain backward keeps its code and registers and a kernel trace tells the two apart.
// (amdgpu_waves_per_eu(2): TWO waves per SIMD = at most 256 VGPRs.  The fused instantiation needs ~254 and the register allocator's own
// choice flips between 254 and 280-330 -- one wave per SIMD: 38 -> 54 us -- with unrelated edits of this file; stated, it is a bound)
template <bool STAGE_SH, bool FUSED, int LPG_L2 /* log2 of the lanes per Gaussian: 0, or 4 for small P (see below) */>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
preprocess_backward_kernel(int P, int D, int M, float tan_fovx, float tan_fovy, float h_x, float h_y, float mean2d_sx, float mean2d_sy,
                           float scale_modifier,
                           const float* __restrict__ means3D, const float* __restrict__ shs,
                           const float* __restrict__ scales, const float* __restrict__ rotations,
                           const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
                           const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
                           GeomView g, const float4* __restrict__ inst_grad /* the record pool */,
                           const uint32_t* __restrict__ cell_valid /* one bit per cell: a record was left there */, const uint32_t* __restrict__ header,
                           float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                           float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
                           float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
                           const float* __restrict__ transforms, float* __restrict__ dL_dtransforms,
                           const float* __restrict__ opacities /* raw mode only */, int raw,
                           unsigned long long* __restrict__ g_stamps_dev /* diagnostics: 8 words per block, else NULL */,
                           uint32_t* __restrict__ queues,
                           const float* __restrict__ translation /* RAW_POSE only, may be NULL */, float* __restrict__ dL_dtranslation /* may be NULL */,
                           FusedAdam fa /* fa.tensors != 0: this kernel also applies the AdamW update of those parameters (adamw.h) */)
{
    // Rewind the work-queue heads of the blend-backward kernel that ran just before this one on the stream, so that another backward
    // over the same forward state (retain_graph) starts from zero again.  (The forward clears them per frame; doing it HERE instead
    // of by the last wave to leave the blend kernel keeps 1024 same-address atomics off that kernel's critical path.)
    if (queues != nullptr && blockIdx.x == 0 && threadIdx.x < 2 * NUM_XCD_QUEUES) {
        const int line = threadIdx.x < NUM_XCD_QUEUES ? Q_BWD + (int)threadIdx.x : Q_SEG_HEAD + (int)threadIdx.x - NUM_XCD_QUEUES;
        queues[(size_t)line * QLINE_WORDS] = 0u;
    }
    // Fused AdamW: the step count and its bias corrections (read below, used at the very end)
    float fa_bc1 = 1.0f, fa_bc2_sqrt = 1.0f;
    int fa_t = 0;
    extern __shared__ float s_sh[];                          // when STAGE_SH: [blockDim.x][SH_ROW] SH in, then dL_dsh out (in place)
#define PSTAMP(i) if (g_stamps_dev && threadIdx.x == 0) g_stamps_dev[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime()
#define PRSTAMP(i) if (g_stamps_dev && threadIdx.x == 0) g_stamps_dev[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime()
    PSTAMP(0); PRSTAMP(13);
    // dL_dsh leaves through the SAME rows the SH record came in by: a thread reads coefficient k of its row and then writes its
    // gradient there.  (A second set of rows made this kernel's LDS 25 KB per 64-thread block: SIX blocks per CU, 1536 resident for
    // cfg3's 1563 -- the 27 left over started when the first ones ended, 18 us in, and ran alone: scripts/prebwd_stamps.py.)
    float* const s_dsh = s_sh;
    // Which Gaussians a block takes: its rows r = 0 .. blockDim-1 are GROUPS of 16 consecutive Gaussians taken from blockDim/16 places
    // spread over the index range (group q of block b = Gaussians 16 (q gridDim + b) ...).  With neighbouring indices being spatial
    // neighbours (densify.spatial_order; MOSS's own order is partly so) a block of 64 CONSECUTIVE Gaussians is all-heavy or
    // all-light -- the kernel then lasted as long as its block in the densest part of the image (bench frame, Morton order: 28.8 us
    // against 25.0 in random order; interleaved: see profiles/r02_notes.md) -- while 16 consecutive ones still share their cache lines.
    const int gl2 = (raw >> 12) & 7;                       // log2 of the group size (launch_preprocess_backward)
    // SMALL P (lanes-per-Gaussian, LPG = 2^lpg_l2 > 1): MOSS trains 6 890 ... 45 695 Gaussians (scene/dataset_readers.py:720,
    // scene/gaussian_model.py:496).  One thread per Gaussian is then 108 waves for the whole device, each gathering dozens of records per
    // lane one round trip after the other (configs[1]: 33 us for 6.9k Gaussians, 70 % of the 100k frame's time).  With LPG lanes per
    // Gaussian only every LPG-th lane OWNS one; the others own nothing but take their share of the wave-balanced gather (which hands the
    // wave's records to all 64 lanes whoever owns them), so a wave holds 64 / LPG Gaussians' records -- one pass -- and the frame has LPG
    // times the waves, all resident at once.  The per-Gaussian arithmetic runs on the owner lanes (the wave issues it once either way).
    // Chosen per launch (launch_preprocess_backward); the order of every sum is unchanged: bitwise the same results.
    // (a template parameter: with the default, 0, every expression below folds to the one-thread-per-Gaussian kernel of rounds 2-4 --
    // the fused instantiation sits at 254 of the 256 VGPRs two waves per SIMD allow, and a run-time parameter cost it 24 more: 38 -> 54 us)
    // the block's SH rows: staged record q (0 .. rows_used - 1) belongs to thread-row q << lpg_l2.  (A MACRO, evaluated where it is used
    // and only in the LPG instantiations: as a variable at the top of the kernel -- dead code when LPG_L2 = 0 -- the early read of
    // blockDim moved the scheduler of the fused one-thread-per-Gaussian instantiation to 280 bytes more scratch per lane.)
#define rows_used ((int)blockDim.x >> LPG_L2)
    constexpr int lpg_l2 = LPG_L2;
    auto gaussian_of_row = [&](int r) -> int {
        if constexpr (LPG_L2 == 0) { return (((r >> gl2) * (int)gridDim.x + (int)blockIdx.x) << gl2) + (r & ((1 << gl2) - 1)); }
        else { return (r & ((1 << LPG_L2) - 1)) ? 0x7fffffff : (int)blockIdx.x * rows_used + (r >> LPG_L2); }
    };
    const int idx = gaussian_of_row((int)threadIdx.x);
    const bool in_range = idx < P;                           // no early return: the wave gathers large Gaussians together
    // All first-level loads are issued together and unconditionally (clamped indices): the 12 SH float4 of this thread's share of
    // the block's records, tiles_touched, point_offsets, the header flags.  Stamps showed this phase -- 12 SH loads each waited
    // for in turn, then (round 2) a five-deep chain tiles_touched -> point_offsets -> inst_pos -> inst_mask -> record -- taking 48k of a
    // block's 62k cycles.
    const int idc = min(idx, P - 1);
    // float4 of the block's SH rows per lane: 12 with a Gaussian per lane (64 rows x 12 float4 over 64 lanes); with sixteen lanes per
    // Gaussian the block's rows_used x 12 float4 are fewer than its lanes: ONE (round 5's first version ran all twelve iterations of
    // the staging, hoisting and update loops there, eleven of them on clamped dummy rows: 48 AdamW elements per lane for one used)
    constexpr int SH_J = LPG_L2 == 0 ? 12 : 1;
    // DEGREE-AWARE SH traffic (MOSS trains at degree 0 / 1 / 2 for iterations 1-2999, train_ZJU.py:85-86): of a record's 12 float4 only
    // the first na_D hold a coefficient of the active degree -- the rest is neither read here nor (sinks that ask for it, the fused
    // update) written.  Buffer loads throughout: an out-of-range offset returns 0 without a memory request and needs no branch.
    // (staged only while the tensors are < 4 GB: 32-bit byte offsets, launch_preprocess_backward)
    typedef float v4f_sh __attribute__((ext_vector_type(4)));
    constexpr uint32_t SH_OOB = 0xffffffffu;
    const int na_D = (3 * (D + 1) * (D + 1) + 3) >> 2;
    const __amdgpu_buffer_rsrc_t rs_shs = __builtin_amdgcn_make_buffer_rsrc((void*)shs, 0, 0xffffff00u, 0x00020000u);
    v4f_sh shv[12];
    if (STAGE_SH) {
        const size_t total4 = (size_t)P * 12;               // in float4 units (48 floats = 12)
#pragma unroll
        for (int j = 0; j < SH_J; j++) {
            const int f = (int)threadIdx.x + j * (int)blockDim.x;         // float4 f of the block's rows: staged row f / 12, part f % 12
            size_t a;
            if constexpr (LPG_L2 == 0) a = min((size_t)gaussian_of_row(f / 12) * 12 + (size_t)(f % 12), total4 - 1);
            else a = f / 12 < rows_used ? min((size_t)min(gaussian_of_row((f / 12) << lpg_l2), P - 1) * 12 + (size_t)(f % 12), total4 - 1) : total4 - 1;
            shv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_shs, (f % 12) < na_D ? (uint32_t)a * 16u : SH_OOB, 0, 0);
        }
    }
    // (the Gaussian's run of cells in the record pool: where it starts -- group base + start in the group -- and how long it is)
    const uint32_t tt_raw = g.tiles_touched[idc], off_raw = g.rec_offsets[idc] + g.group_rbase[idc >> 8], cells_raw = g.rec_count[idc], hdr_flags = header[2];
    // Every per-Gaussian input of the arithmetic at the end is requested HERE too, unconditionally (clamped index): the position, the
    // transform, the covariance, scales / rotation, the raw opacity, the clamp flags -- none depends on the gather.  Loads complete
    // in order: read where they are used -- behind the fused update's moment loads, which are requested before the arithmetic so that
    // they are in flight during it -- each would wait for all of those.
    float in_x[3], in_T[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, in_t[3] = { 0.f, 0.f, 0.f }, in_c6[6], in_sc[3] = { 0.f, 0.f, 0.f }, in_q[4] = { 0.f, 0.f, 0.f, 0.f };
    float in_opa = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) in_x[i] = means3D[3 * (size_t)idc + i];
    if (transforms != nullptr) {
#pragma unroll
        for (int i = 0; i < 9; i++) in_T[i] = transforms[9 * (size_t)idc + i];
    }
    if ((raw & RAW_POSE) && translation != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; i++) in_t[i] = translation[3 * (size_t)idc + i];
    }
    {
        const float* c6p = (cov3D_precomp != nullptr) ? cov3D_precomp + 6 * (size_t)idc : g.cov3D + 6 * (size_t)idc;
#pragma unroll
        for (int i = 0; i < 6; i++) in_c6[i] = c6p[i];
    }
    if (scales != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; i++) in_sc[i] = scales[3 * (size_t)idc + i];
#pragma unroll
        for (int i = 0; i < 4; i++) in_q[i] = rotations[4 * (size_t)idc + i];
    }
    if (raw & RAW_OPACITY) in_opa = opacities[idc];
    const uint8_t in_clamped = g.clamped[idc];
    if (STAGE_SH) {
        // into LDS right away (row stride 49: conflict-free rows): the SH loads are the oldest outstanding ones, so this waits for
        // them only, and their 48 registers are free during the gather (holding them across it spilled to scratch)
#pragma unroll
        for (int j = 0; j < SH_J; j++) {
            const int f = threadIdx.x + j * blockDim.x;
            if (LPG_L2 == 0 || f / 12 < rows_used) {
                float* d = &s_sh[((f / 12) << lpg_l2) * SH_ROW + (f % 12) * 4];
                d[0] = shv[j].x; d[1] = shv[j].y; d[2] = shv[j].z; d[3] = shv[j].w;
            }
        }
    }
    const uint32_t n_inst = in_range ? tt_raw : 0u;
    // (pinned: the values must be in registers HERE -- where the wave waits for its first-level loads anyway -- or the compiler sinks
    // these loads of read-only data down to their uses, behind the moment loads)
#pragma unroll
    for (int i = 0; i < 3; i++) { asm volatile("" : "+v"(in_x[i])); asm volatile("" : "+v"(in_t[i])); asm volatile("" : "+v"(in_sc[i])); }
#pragma unroll
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(in_T[i]));
#pragma unroll
    for (int i = 0; i < 6; i++) asm volatile("" : "+v"(in_c6[i]));
#pragma unroll
    for (int i = 0; i < 4; i++) asm volatile("" : "+v"(in_q[i]));
    asm volatile("" : "+v"(in_opa));
    float3 gcol = make_float3(0, 0, 0); float gmx = 0, gmy = 0, gca = 0, gcb = 0, gcc = 0, gop = 0;
    float dmean[3] = { 0, 0, 0 }, dcov[6] = { 0, 0, 0, 0, 0, 0 }, dscale[3] = { 0, 0, 0 }, drot[4] = { 0, 0, 0, 0 };
    float dtf[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, dpose_t[3] = { 0, 0, 0 };
    float A_cov[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } }, d2_cov[3] = { 0, 0, 0 };   // the 2x3 projection A = J Rv and dL/dS2 of step (1): kept for step (6)
    // visible <=> radii > 0 (backward.cu:156,367).  After a capacity overflow of the asynchronous forward nothing was
    // rendered and the instance tables are unwritten: every Gaussian then gets zero gradients.
    // (likewise after a forward that was told MOSS_FORWARD_ONLY: no backward state exists, the binning buffer has no record pool)
    const bool visible = n_inst > 0 && !(hdr_flags & (ERRFLAG_OVERFLOW | ERRFLAG_FORWARD_ONLY));
    // a frame that overflowed its capacity rendered nothing: its optimizer step is a no-op (parameters, moments and the step count
    // stay bit for bit), like moss_adamw_flat_guarded on the frame's status word
    const bool fa_on = FUSED && !(hdr_flags & (ERRFLAG_OVERFLOW | ERRFLAG_FORWARD_ONLY));

    // Sum the Gaussian's gradient records: the cells of its run [c0, c1) in the record pool whose validity bit is set (common.h:
    // box_cells; the backward blend sets the bits).  A run of <= COOP_WORDS words of validity bits (the norm: 10-20 cells, one or two
    // words) is summed by the Gaussian's own lane -- or its group of lanes; one that covers much of the image owns thousands of cells
    // and is summed by the 64 lanes of the wave together (coop_gather: lane-strided partial sums, then a fixed butterfly): still a
    // fixed order, hence bitwise reproducible.
    constexpr uint32_t COOP_WORDS = 16, MID_CELLS = 96;
    const uint32_t c0 = visible ? off_raw : 0u, c1 = c0 + (visible ? cells_raw : 0u);
    const uint32_t word0 = c0 >> 5, n_words = c1 > c0 ? ((c1 + 31u) >> 5) - word0 : 0u;
    // (mid-size runs -- MID_CELLS < cells, <= COOP_WORDS words -- go four at a time through the lane-per-cell form, coop_gather<4>; with
    // sixteen lanes per Gaussian, the small-P instantiation, the group's lanes take a word each already)
    const bool mid_size = LPG_L2 == 0 && (c1 - c0) > MID_CELLS && n_words <= COOP_WORDS;
    {
        // Sparse records (a Gaussian owns 2-3 instances of 4-5 cells each, about half of them flagged): walked one after the other, every
        // record costs two DEPENDENT loads (validity word -> record) and the wave waits for its lane with the most: stamps showed this
        // phase taking 48k of a block's 62k cycles in round 2.  Here all validity words, then -- four words at a time -- the first
        // two flagged records of each word are fetched as batches of independent loads.  The summation order is fixed (batch of 4 words,
        // round of 2 records, word, bit), so gradients stay bitwise reproducible.
        const bool mine = n_words != 0u && n_words <= COOP_WORDS && !mid_size;
        PSTAMP(8);
        // LPG = 16 (LPG_L2 == 4): the 16 lanes of a Gaussian's group take ONE validity word each -- lane (g, j) the j-th of Gaussian g's
        // <= 16 words: one word load per lane instead of sixteen per owner, and the filing loop below (serial over a lane's words and
        // their flagged cells, the whole wave in step) runs over one word per lane: with four owners doing the filing of four Gaussians
        // alone it was 16k of a block's 34k cycles (scripts/prebwd_fused_stamps.py, configs[1]).
        const int lane_g = (int)(threadIdx.x & 63u);
        uint32_t g_c0 = c0, g_c1 = c1, g_n = mine ? n_words : 0u;
        if constexpr (LPG_L2 == 4) {
            g_c0 = (uint32_t)__shfl((int)g_c0, lane_g & ~15); g_c1 = (uint32_t)__shfl((int)g_c1, lane_g & ~15); g_n = (uint32_t)__shfl((int)g_n, lane_g & ~15);
        }
        const bool has_word = (uint32_t)(lane_g & 15) < g_n;         // (LPG = 16 only)
        uint32_t wmax = LPG_L2 == 4 ? (has_word ? 1u : 0u) : (mine ? n_words : 0u);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d));
        // BUFFER loads: a lane without a k-th word / j-th record passes an out-of-range offset, which returns 0 WITHOUT a memory
        // request.  That keeps every load unconditional (no branch, so the compiler counts its waits exactly and a batch is in
        // flight together) and free for absent lanes -- these random 4- and 48-byte reads are bound by request count (stamps:
        // 104 unconditional global loads per lane cost 51k cycles per wave; predicated ones were each waited for at their join).
        // (32-bit byte offsets: the scan refuses a frame of more than POOL_MAX_CELLS cells -- 48 B x 80M < 4 GB)
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr uint32_t OOB = 0xffffffffu, RSRC3 = 0x00020000u;
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)cell_valid, 0, 0xffffff00u, RSRC3);
        const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc((void*)inst_grad, 0, 0xffffff00u, RSRC3);
        // (no table between the Gaussian and its records: the run's start and length are per-Gaussian values of the forward pass)
        uint32_t pp[COOP_WORDS], mm[COOP_WORDS];             // validity word index, its bits inside the run
#pragma unroll
        for (int k = 0; k < (int)COOP_WORDS; k++) pp[k] = LPG_L2 == 4 ? (k == 0 ? (g_c0 >> 5) + (uint32_t)(lane_g & 15) : 0u) : word0 + (uint32_t)k;
        PSTAMP(9);
#pragma unroll
        for (int k = 0; k < (int)COOP_WORDS; k++) {
            if constexpr (LPG_L2 == 4) mm[k] = k == 0 ? (__builtin_amdgcn_raw_buffer_load_b32(rs_msk, has_word ? pp[0] * 4u : OOB, 0, 0) & run_word_mask(pp[0], g_c0, g_c1)) : 0u;
            else mm[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_msk, (mine && (uint32_t)k < n_words) ? pp[k] * 4u : OOB, 0, 0) & run_word_mask(pp[k], c0, c1);
        }
        PSTAMP(10);
        // ---- two ways to fetch the flagged records, chosen per wave.  How many records each lane owns:
        uint32_t cnt = 0u;
#pragma unroll
        for (int k = 0; k < (int)COOP_WORDS; k++) cnt += (uint32_t)__popc(mm[k]);
        uint32_t incl = cnt;
        uint32_t bmax = 0u;                                   // byte b: the most records one word of batch b (words 4b .. 4b+3) has
#pragma unroll
        for (int k = 0; k < (int)COOP_WORDS; k++) {
            const uint32_t pc = (uint32_t)__popc(mm[k]), sh = 8u * (uint32_t)(k >> 2);
            bmax = max(bmax & (0xffu << sh), pc << sh) | (bmax & ~(0xffu << sh));
        }
        {
            const int lane_ = (int)(threadIdx.x & 63u);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if (lane_ >= d) incl += y; }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {               // bytewise maximum over the wave
                const uint32_t o = (uint32_t)__shfl_xor((int)bmax, d);
                uint32_t r = 0u;
#pragma unroll
                for (int b = 0; b < 4; b++) r |= max(bmax & (0xffu << (8 * b)), o & (0xffu << (8 * b)));
                bmax = r;
            }
        }
        const uint32_t W_all = (uint32_t)__shfl((int)incl, 63);
        // DIRECT (lane by lane: four words x two records per round trip) makes sum over the batches of ceil(most records of one
        // word / 2) round trips of 2-4k cycles (few waves ... a busy memory system); BALANCED (below) a pass of ~20k cycles -- filing,
        // one round of requests, the scans --
        // per GATHER_CAP records of the wave, whoever owns them.  Balanced pays when the records are unevenly spread (cfg3: 5 per lane
        // on average, 30 for the busiest: 24 vs 28 us); when every lane owns many (cfg2's wide Gaussians) it would be passes on end
        // (61 vs 24 us).
        const uint32_t passes = (W_all + (uint32_t)GATHER_CAP - 1u) / (uint32_t)GATHER_CAP;
        const uint32_t direct_trips = ((bmax & 0xffu) + 1u) / 2u + (((bmax >> 8) & 0xffu) + 1u) / 2u + (((bmax >> 16) & 0xffu) + 1u) / 2u + ((bmax >> 24) + 1u) / 2u;
        const bool balanced = LPG_L2 == 4 ? true :           // (the group's lanes hold one word each: only the list form sums a Gaussian's records)
                              (raw & 0x200) ? false : (raw & 0x400) ? true : (passes == 1u ? direct_trips >= 5u : passes * 8u < direct_trips);     // (0x200 / 0x400: diagnostics, MOSS_GATHER=1 / 2)
        if (!balanced) {
#pragma unroll
        for (int kb = 0; kb < (int)COOP_WORDS; kb += 4) {
            if ((uint32_t)kb < wmax) {                       // wave-uniform
                uint32_t bits[4] = { mm[kb], mm[kb + 1], mm[kb + 2], mm[kb + 3] };
                // rounds of (4 words x their next 2 flagged records); a word whose cells are all flagged has 32 records = 16 rounds.
                // The order (round, word, bit) is fixed.
                do {
                    v4f rr[4][2][3];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const bool any = bits[k] != 0u;
                            const uint32_t sl = any ? (uint32_t)(__ffs((int)bits[k]) - 1) : 0u;
                            bits[k] &= bits[k] - 1u;                               // (0 stays 0)
                            const uint32_t o = any ? ((pp[kb + k] << 5) + sl) * (uint32_t)(GRAD_REC_FLOATS * 4) : OOB;
                            rr[k][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, o, 0, 0);
                            rr[k][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, any ? o + 16u : OOB, 0, 0);
                            rr[k][j][2] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, any ? o + 32u : OOB, 0, 0);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            // (slots not loaded are exact zeros: adding them changes nothing)
                            gcol.x += rr[k][j][0].x; gcol.y += rr[k][j][0].y; gcol.z += rr[k][j][0].z; gmx += rr[k][j][0].w;
                            gmy += rr[k][j][1].x; gca += rr[k][j][1].y; gcb += rr[k][j][1].z; gcc += rr[k][j][1].w;
                            gop += rr[k][j][2].x;
                        }
                    }
                } while (__ballot((bits[0] | bits[1] | bits[2] | bits[3]) != 0u) != 0ull);
            }
        }
        } else
        // ---- wave-balanced gather.  A lane owns 3-5 flagged records on average but some own 30+: fetched lane by lane (round 2: four
        // instances x two records per round trip) the wave made 4-6 dependent round trips for its busiest lane -- 17k of a block's 41k
        // cycles.  Here the wave's records are put in ONE list (lane i's records at list positions base_i .. base_i + c_i - 1, in the
        // fixed order of their cells), every lane fetches the records at positions lane, lane + 64, ... whoever owns them -- all
        // requested before any is waited for -- and a segmented scan over each row of 64 hands every owner its sum.  The order of the
        // additions is fixed by the list: bitwise reproducible.
        {
            uint32_t* const w_desc = reinterpret_cast<uint32_t*>(s_sh + (STAGE_SH ? blockDim.x * SH_ROW : 0)) + (threadIdx.x >> 6) * GATHER_WORDS;
            float* const w_row = reinterpret_cast<float*>(w_desc + GATHER_CAP);          // [64][9]: a row's scanned values
            const int lane = (int)(threadIdx.x & 63u);
            const uint32_t base = incl - cnt, W = W_all;
            // whose run a lane picks up: its own records -- or, LPG = 16, the owner lane its whole group's (the sixteen lanes' runs are
            // consecutive in the list, in cell order: the same order of additions as one lane filing all of them)
            uint32_t run_base = base, run_cnt = cnt;
            bool head_ok = true;                                                          // the lane's first record starts a Gaussian's run
            if constexpr (LPG_L2 == 4) {
                const uint32_t gbase = (uint32_t)__shfl((int)base, lane & ~15), gend = (uint32_t)__shfl((int)incl, lane | 15);
                head_ok = base == gbase;                                                  // no earlier lane of the group has a record
                run_base = gbase; run_cnt = (lane & 15) == 0 ? gend - gbase : 0u;
            }
            for (uint32_t w0 = 0u; w0 < W; w0 += (uint32_t)GATHER_CAP) {                 // (one pass unless the wave owns > GATHER_CAP records)
                const uint32_t w1 = min(W, w0 + (uint32_t)GATHER_CAP);
                // 1. every lane files its records' descriptors {head of the lane's run << 31 | cell}
                __builtin_amdgcn_wave_barrier();
                {
                    uint32_t j = base;
#pragma unroll
                    for (int k = 0; k < (int)COOP_WORDS; k++) {
                        if ((uint32_t)k < wmax) {                                        // wave-uniform
                            uint32_t bits = mm[k];
                            while (__ballot(bits != 0u) != 0ull) {
                                if (bits != 0u) {
                                    const uint32_t sl = (uint32_t)(__ffs((int)bits) - 1);
                                    bits &= bits - 1u;
                                    if (j >= w0 && j < w1) w_desc[j - w0] = ((j == base && head_ok) ? 0x80000000u : 0u) | ((pp[k] << 5) + sl);
                                    j++;
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // 2. row r of the list = positions w0 + 64 r + lane: all rows requested together
                constexpr int ROWS = GATHER_CAP / 64;
                const int nrows = (int)((w1 - w0 + 63u) / 64u);
                v4f ra[ROWS], rb[ROWS]; float rc[ROWS]; uint32_t dsc[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const uint32_t w = w0 + 64u * (uint32_t)r + (uint32_t)lane;
                    const bool on = w < w1;
                    dsc[r] = on ? w_desc[w - w0] : 0x80000000u;
                    const uint32_t o = on ? (dsc[r] & 0x7fffffffu) * (uint32_t)(GRAD_REC_FLOATS * 4) : OOB;
                    ra[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, o, 0, 0);
                    rb[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, on ? o + 16u : OOB, 0, 0);
                    rc[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rec, on ? o + 32u : OOB, 0, 0));
                }
                // 3. per row: segmented inclusive scan (a lane's run starts at its head flag; a row starts a segment of its own), then
                //    every owner whose run reaches into the row takes the value at the run's last position in it
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    if (r < nrows) {                                                     // wave-uniform
                        float v[9] = { ra[r].x, ra[r].y, ra[r].z, ra[r].w, rb[r].x, rb[r].y, rb[r].z, rb[r].w, rc[r] };
                        uint32_t f = (dsc[r] >> 31) | (lane == 0 ? 1u : 0u);
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const uint32_t fu = (uint32_t)__shfl_up((int)f, d);
                            float t[9];
#pragma unroll
                            for (int q = 0; q < 9; q++) t[q] = __shfl_up(v[q], d);
                            const bool take = lane >= d && f == 0u;
#pragma unroll
                            for (int q = 0; q < 9; q++) v[q] += take ? t[q] : 0.0f;
                            f |= lane >= d ? fu : 0u;
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int q = 0; q < 9; q++) w_row[lane * 9 + q] = v[q];
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t row0 = w0 + 64u * (uint32_t)r, row1 = min(w1, row0 + 64u);
                        const uint32_t lo = max(run_base, row0), hi = min(run_base + run_cnt, row1);
                        if (lo < hi) {                                                   // this lane's run reaches into the row
                            const float* e = w_row + (hi - 1u - row0) * 9u;
                            gcol.x += e[0]; gcol.y += e[1]; gcol.z += e[2]; gmx += e[3];
                            gmy += e[4]; gca += e[5]; gcb += e[6]; gcc += e[7]; gop += e[8];
                        }
                    }
                }
            }
        }
        PSTAMP(11);
    }
    {
        float sums[9] = { gcol.x, gcol.y, gcol.z, gmx, gmy, gca, gcb, gcc, gop };
        if constexpr (LPG_L2 == 0) coop_gather<4>(mid_size, c0, c1, cell_valid, inst_grad, sums);
        coop_gather<1>(n_words > COOP_WORDS, c0, c1, cell_valid, inst_grad, sums);
        gcol.x = sums[0]; gcol.y = sums[1]; gcol.z = sums[2]; gmx = sums[3]; gmy = sums[4]; gca = sums[5]; gcb = sums[6]; gcc = sums[7]; gop = sums[8];
    }
    PSTAMP(12);
    PSTAMP(1);
    if (STAGE_SH) __syncthreads();                           // SH records are in LDS
    PSTAMP(2);
    // the blend kernel's records carry the geometry sums without their constant factors (blend.hip, the backward trip):
    // d pixel / d ndc = W/2, H/2 (backward.cu:472-473, 574-575) and the -1/2 of the exponent (backward.cu:578-580)
    gmx *= mean2d_sx; gmy *= mean2d_sy; gca *= -0.5f; gcb *= -0.5f; gcc *= -0.5f;
    // (the step count, its bias corrections and the learning rates: chains of dependent SCALAR loads, requested HERE -- behind the
    // gather, in front of the arithmetic -- and used after it.  At the top of the kernel they delayed every block's first vector load
    // by three to four scalar round trips (2.3 us of the kernel, measured); between the first requests and the first wait their
    // registers, live across the gather, cost the second wave per SIMD (260 VGPRs: 51 us))
    if (FUSED) {
        const int t_prev = reinterpret_cast<const int*>(fa.step_state)[0], lr_flag = reinterpret_cast<const int*>(fa.step_state)[ADAMW_LR_VALID_WORD];
        fa_t = adamw_step_begin(fa.step_state, fa.betas, false, fa_bc1, fa_bc2_sqrt, t_prev);
        // learning rates kept in the step-state block (a schedule without re-capturing the step): scalar loads, like the step count
        if (lr_flag != 0) {
#pragma unroll
            for (int i = 0; i < 5; i++) if (fa.lr_segment[i] >= 0) fa.lr[i] = fa.step_state[ADAMW_LR_WORD0 + fa.lr_segment[i]];
            if (fa.lr_segment[1] >= 0) fa.lr_sh_rest = fa.step_state[ADAMW_LR2_WORD0 + fa.lr_segment[1]];
        }
    }
    if (fa_on && blockIdx.x == 0 && threadIdx.x == 0) adamw_cache_next(fa.step_state, fa.betas, fa_t);   // (the next step's bias corrections)
    // Fused AdamW: the eleven per-Gaussian scalars' parameters and moments are requested HERE, before the backward arithmetic, and
    // used after it (stamps: loaded where they are used, the 33 strided loads were a round trip of 10k cycles on every block's path)
    const bool fa_scalars = fa_on && in_range && (fa.tensors & (OPT_MEANS | OPT_OPACITY | OPT_SCALES | OPT_ROTATIONS));
    float fa_p[11], fa_m[11], fa_v[11];
#pragma unroll
    for (int i = 0; i < 11; i++) { fa_p[i] = 0.f; fa_m[i] = 0.f; fa_v[i] = 0.f; }
    if (fa_scalars) {
        const size_t i3 = 3 * (size_t)idx;
        if (fa.tensors & OPT_MEANS) {
#pragma unroll
            for (int i = 0; i < 3; i++) { fa_p[i] = fa.p[0][i3 + i]; fa_m[i] = fa.m[0][i3 + i]; fa_v[i] = fa.v[0][i3 + i]; }
        }
        if (fa.tensors & OPT_OPACITY) { fa_p[3] = fa.p[2][idx]; fa_m[3] = fa.m[2][idx]; fa_v[3] = fa.v[2][idx]; }
        if (fa.tensors & OPT_SCALES) {
#pragma unroll
            for (int i = 0; i < 3; i++) { fa_p[4 + i] = fa.p[3][i3 + i]; fa_m[4 + i] = fa.m[3][i3 + i]; fa_v[4 + i] = fa.v[3][i3 + i]; }
        }
        if (fa.tensors & OPT_ROTATIONS) {
            const float4 a = reinterpret_cast<const float4*>(fa.p[4])[idx], b = reinterpret_cast<const float4*>(fa.m[4])[idx], c = reinterpret_cast<const float4*>(fa.v[4])[idx];
            fa_p[7] = a.x; fa_p[8] = a.y; fa_p[9] = a.z; fa_p[10] = a.w; fa_m[7] = b.x; fa_m[8] = b.y; fa_m[9] = b.z; fa_m[10] = b.w;
            fa_v[7] = c.x; fa_v[8] = c.y; fa_v[9] = c.z; fa_v[10] = c.w;
        }
    }
    // ... and so are the moments of the block's SH rows (24 float4 per lane): in flight during the arithmetic below
    // (FA_HOIST of the 12 slots: all 12 pairs are 96 registers held across the arithmetic -- 260 in all, one wave per SIMD)
    constexpr int FA_HOIST = SH_J < 10 ? SH_J : 10;
    float4 fa_m4[12], fa_v4[12];
    const bool fa_sh = STAGE_SH && fa_on && (fa.tensors & OPT_SH);
    const __amdgpu_buffer_rsrc_t rs_fm = __builtin_amdgcn_make_buffer_rsrc((void*)fa.m[1], 0, 0xffffff00u, 0x00020000u);
    const __amdgpu_buffer_rsrc_t rs_fv = __builtin_amdgcn_make_buffer_rsrc((void*)fa.v[1], 0, 0xffffff00u, 0x00020000u);
    if (fa_sh) {
#pragma unroll
        for (int j = 0; j < FA_HOIST; j++) {
            const int f = (int)threadIdx.x + j * (int)blockDim.x;
            const size_t a = (size_t)min(gaussian_of_row(LPG_L2 == 0 ? f / 12 : (min(f / 12, rows_used - 1) << lpg_l2)), P - 1) * 12 + (size_t)(f % 12);
            // (moments of the float4 behind the ever-active part are exactly zero: not read -- the load returns the zeros for free)
            const uint32_t o = (f % 12) < fa.sh_active_parts ? (uint32_t)a * 16u : SH_OOB;
            const v4f_sh mq = __builtin_amdgcn_raw_buffer_load_b128(rs_fm, o, 0, 0), vq = __builtin_amdgcn_raw_buffer_load_b128(rs_fv, o, 0, 0);
            fa_m4[j] = make_float4(mq.x, mq.y, mq.z, mq.w); fa_v4[j] = make_float4(vq.x, vq.y, vq.z, vq.w);
        }
    }
    if (in_range) {
    dL_dmean2D[3 * (size_t)idx] = gmx; dL_dmean2D[3 * (size_t)idx + 1] = gmy; dL_dmean2D[3 * (size_t)idx + 2] = 0.0f;
    if (dL_dconic != nullptr) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(gca, gcb, 0.0f, gcc);   // (NULL = not wanted)
    if (raw & RAW_OPACITY) {                                 // d sigmoid(v) = y (1 - y)
        const float sg = sigmoid_act(in_opa);
        gop = gop * ((1.0f - sg) * sg);
    }
    if (dL_dopacity != nullptr) dL_dopacity[idx] = gop;
    if (dL_dcolor != nullptr) { dL_dcolor[3 * (size_t)idx] = gcol.x; dL_dcolor[3 * (size_t)idx + 1] = gcol.y; dL_dcolor[3 * (size_t)idx + 2] = gcol.z; }

    float* dsh = STAGE_SH ? &s_dsh[threadIdx.x * SH_ROW]
                          : ((M > 0 && dL_dsh != nullptr) ? dL_dsh + (size_t)idx * M * 3 : nullptr);

    if (visible) {
        const float3 xc = make_float3(in_x[0], in_x[1], in_x[2]);
        float3 mean = xc;
        if (raw & RAW_POSE) mean = pose_point(in_T, xc, make_float3(in_t[0], in_t[1], in_t[2]));   // the forward's posed mean, same expression
        cov_proj_backward(idx, gca, gcb, gcc, gmx, gmy, tan_fovx, tan_fovy, h_x, h_y, in_c6, viewmatrix, projmatrix, mean, dmean, dcov, A_cov, d2_cov);

        PSTAMP(3);
        // (5) colour = max(0, 0.5 + sum_k b_k(n) sh_k), n = (mean - campos)/|.| (forward.cu:20-71).  dL/dsh_k = b_k(n) g (g = colour
        //     gradient, zero in a clamped channel), and with s_k = sh_k . g the direction gets  sum_k s_k grad b_k(n), pushed through
        //     the normalisation: (I - n n^T) / |v|.
        if (shs != nullptr) {
            const float* sh = STAGE_SH ? &s_sh[threadIdx.x * SH_ROW] : shs + (size_t)idx * M * 3;
            const float vx = mean.x - cam_pos[0], vy = mean.y - cam_pos[1], vz = mean.z - cam_pos[2];
            const float vlen = sqrtf(vx * vx + vy * vy + vz * vz);
            const float x = vx / vlen, y = vy / vlen, z = vz / vlen;
            const uint8_t cl = in_clamped;
            const float gc[3] = { (cl & 1) ? 0.0f : gcol.x, (cl & 2) ? 0.0f : gcol.y, (cl & 4) ? 0.0f : gcol.z };
            float basis[16], gbx[16], gby[16], gbz[16];
            const int used = (D + 1) * (D + 1);
            basis[0] = SH_C0; gbx[0] = 0.f; gby[0] = 0.f; gbz[0] = 0.f;
            if (D > 0) {
                basis[1] = -SH_C1 * y; gbx[1] = 0.f;    gby[1] = -SH_C1; gbz[1] = 0.f;
                basis[2] = SH_C1 * z;  gbx[2] = 0.f;    gby[2] = 0.f;    gbz[2] = SH_C1;
                basis[3] = -SH_C1 * x; gbx[3] = -SH_C1; gby[3] = 0.f;    gbz[3] = 0.f;
            }
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                basis[4] = SH_C2[0] * (x * y);            gbx[4] = SH_C2[0] * y;          gby[4] = SH_C2[0] * x;          gbz[4] = 0.f;
                basis[5] = SH_C2[1] * (y * z);            gbx[5] = 0.f;                   gby[5] = SH_C2[1] * z;          gbz[5] = SH_C2[1] * y;
                basis[6] = SH_C2[2] * (2.f * zz - xx - yy); gbx[6] = SH_C2[2] * (-2.f * x); gby[6] = SH_C2[2] * (-2.f * y); gbz[6] = SH_C2[2] * (4.f * z);
                basis[7] = SH_C2[3] * (x * z);            gbx[7] = SH_C2[3] * z;          gby[7] = 0.f;                   gbz[7] = SH_C2[3] * x;
                basis[8] = SH_C2[4] * (xx - yy);          gbx[8] = SH_C2[4] * (2.f * x);  gby[8] = SH_C2[4] * (-2.f * y); gbz[8] = 0.f;
                if (D > 2) {
                    const float xy = x * y, yz = y * z, xz = x * z;
                    basis[9] = SH_C3[0] * (y * (3.f * xx - yy));            gbx[9] = SH_C3[0] * (6.f * xy);              gby[9] = SH_C3[0] * (3.f * (xx - yy));           gbz[9] = 0.f;
                    basis[10] = SH_C3[1] * (xy * z);                        gbx[10] = SH_C3[1] * yz;                     gby[10] = SH_C3[1] * xz;                          gbz[10] = SH_C3[1] * xy;
                    basis[11] = SH_C3[2] * (y * (4.f * zz - xx - yy));      gbx[11] = SH_C3[2] * (-2.f * xy);            gby[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy);   gbz[11] = SH_C3[2] * (8.f * yz);
                    basis[12] = SH_C3[3] * (z * (2.f * zz - 3.f * xx - 3.f * yy)); gbx[12] = SH_C3[3] * (-6.f * xz);     gby[12] = SH_C3[3] * (-6.f * yz);                 gbz[12] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                    basis[13] = SH_C3[4] * (x * (4.f * zz - xx - yy));      gbx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); gby[13] = SH_C3[4] * (-2.f * xy);             gbz[13] = SH_C3[4] * (8.f * xz);
                    basis[14] = SH_C3[5] * (z * (xx - yy));                 gbx[14] = SH_C3[5] * (2.f * xz);             gby[14] = SH_C3[5] * (-2.f * yz);                 gbz[14] = SH_C3[5] * (xx - yy);
                    basis[15] = SH_C3[6] * (x * (xx - 3.f * yy));           gbx[15] = SH_C3[6] * (3.f * (xx - yy));      gby[15] = SH_C3[6] * (-6.f * xy);                 gbz[15] = 0.f;
                }
            }
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < used) {
                    const float sk = sh[3 * k] * gc[0] + sh[3 * k + 1] * gc[1] + sh[3 * k + 2] * gc[2];
                    dsh[3 * k] = basis[k] * gc[0]; dsh[3 * k + 1] = basis[k] * gc[1]; dsh[3 * k + 2] = basis[k] * gc[2];
                    ddx += gbx[k] * sk; ddy += gby[k] * sk; ddz += gbz[k] * sk;
                }
            }
            for (int k = used; k < M; k++) { dsh[3 * k] = 0.0f; dsh[3 * k + 1] = 0.0f; dsh[3 * k + 2] = 0.0f; }   // above the active degree
            const float nd = x * ddx + y * ddy + z * ddz, il = 1.0f / vlen;
            dmean[0] += (ddx - x * nd) * il; dmean[1] += (ddy - y * nd) * il; dmean[2] += (ddz - z * nd) * il;
        }


        PSTAMP(4);
        scale_rot_backward(idx, dcov, A_cov, d2_cov, scale_modifier, raw, scales != nullptr, in_sc, in_q, transforms != nullptr, in_T, dscale, drot, dtf);
        if (raw & RAW_POSE) {
            // p = T x + t:  dL/dt = g,  dL/dT += g x^T,  dL/dx = T^T g   (g = dL/dp collected above; reported in place of it)
            const float gp[3] = { dmean[0], dmean[1], dmean[2] }, xv[3] = { xc.x, xc.y, xc.z };
            const float (&Tm)[9] = in_T;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) dtf[3 * a + c] += gp[a] * xv[c];
#pragma unroll
            for (int c = 0; c < 3; c++) dmean[c] = Tm[c] * gp[0] + Tm[3 + c] * gp[1] + Tm[6 + c] * gp[2];
            dpose_t[0] = gp[0]; dpose_t[1] = gp[1]; dpose_t[2] = gp[2];
        }
    } else if (dsh != nullptr) {
        for (int k = 0; k < 3 * M; k++) dsh[k] = 0.0f;
    }
    if (visible && dsh != nullptr && shs == nullptr) {
        for (int k = 0; k < 3 * M; k++) dsh[k] = 0.0f;
    }

    // (with the fused update the gradients of the updated tensors need not leave the kernel: NULL = not wanted)
    if (dL_dmean3D != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dmean3D[3 * (size_t)idx + i] = dmean[i];
    }
    if (dL_dcov3D != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (dL_dscale != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dscale[3 * (size_t)idx + i] = dscale[i];
    }
    if (dL_drot != nullptr) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    if (dL_dtransforms != nullptr) {
#pragma unroll
        for (int i = 0; i < 9; i++) dL_dtransforms[9 * (size_t)idx + i] = dtf[i];
    }
    if (dL_dtranslation != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dtranslation[3 * (size_t)idx + i] = dpose_t[i];
    }
    // ---- fused AdamW, the eleven per-Gaussian scalars (position, opacity, scale, rotation): this thread's own parameters, read
    //      above by nobody else; every Gaussian takes the step, with zero gradients if it was not rendered (torch.optim.AdamW does
    //      the same: the moments decay).  (Loads: requested before the backward arithmetic, see above.)
    if (fa_scalars) {
        const float ib1 = 1.0f / fa_bc1, ib2 = 1.0f / fa_bc2_sqrt;
        const size_t i3 = 3 * (size_t)idx;
        const float gv[11] = { dmean[0], dmean[1], dmean[2], gop, dscale[0], dscale[1], dscale[2], drot[0], drot[1], drot[2], drot[3] };
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float lr = i < 3 ? fa.lr[0] : i == 3 ? fa.lr[2] : i < 7 ? fa.lr[3] : fa.lr[4];
            adamw_element(fa_p[i], gv[i], fa_m[i], fa_v[i], lr, fa.betas, fa.eps, fa.weight_decay, ib1, ib2);
        }
        if (fa.tensors & OPT_MEANS) {
#pragma unroll
            for (int i = 0; i < 3; i++) { fa.p[0][i3 + i] = fa_p[i]; fa.m[0][i3 + i] = fa_m[i]; fa.v[0][i3 + i] = fa_v[i]; }
        }
        if (fa.tensors & OPT_OPACITY) { fa.p[2][idx] = fa_p[3]; fa.m[2][idx] = fa_m[3]; fa.v[2][idx] = fa_v[3]; }
        if (fa.tensors & OPT_SCALES) {
#pragma unroll
            for (int i = 0; i < 3; i++) { fa.p[3][i3 + i] = fa_p[4 + i]; fa.m[3][i3 + i] = fa_m[4 + i]; fa.v[3][i3 + i] = fa_v[4 + i]; }
        }
        if (fa.tensors & OPT_ROTATIONS) {
            reinterpret_cast<float4*>(fa.p[4])[idx] = make_float4(fa_p[7], fa_p[8], fa_p[9], fa_p[10]);
            reinterpret_cast<float4*>(fa.m[4])[idx] = make_float4(fa_m[7], fa_m[8], fa_m[9], fa_m[10]);
            reinterpret_cast<float4*>(fa.v[4])[idx] = make_float4(fa_v[7], fa_v[8], fa_v[9], fa_v[10]);
        }
    }
    }   // in_range
    PSTAMP(5);
    if (STAGE_SH) {
        __syncthreads();                                     // every row now holds dL_dsh
        if (dL_dsh != nullptr) {
            float4* dst = reinterpret_cast<float4*>(dL_dsh);
            // (SH_GRAD_ACTIVE_ONLY, raw bit 0x20: the caller's destination holds zeros above the active degree already -- a gradient
            // sink into a zero-initialised bucket whose consumers never look there -- so those float4 are not written: at degree 0
            // eleven twelfths of this kernel's largest output)
            const int n_write = (raw & SH_GRAD_ACTIVE_ONLY) ? na_D : 12;
            for (int f = threadIdx.x; f < (LPG_L2 == 0 ? (int)blockDim.x : rows_used) * 12; f += blockDim.x) {
                const int gi = gaussian_of_row((f / 12) << lpg_l2);
                if (gi < P && (f % 12) < n_write) {
                    const float* r = &s_dsh[((f / 12) << lpg_l2) * SH_ROW + (f % 12) * 4];
                    dst[(size_t)gi * 12 + (size_t)(f % 12)] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
        // ---- fused AdamW, the SH records (48 of a Gaussian's 59 parameters): the gradient rows are in LDS; parameters and both
        //      moments come and go as the block's 12 coalesced float4 per lane (the parameters were read by this block at its start:
        //      cache hits mostly).  All 36 loads are requested before the first is used.  Learning rate: lr[1] for a record's first
        //      three floats (MOSS's features_dc group), lr_sh_rest for the other 45 (features_rest, scene/gaussian_model.py:218-219).
        if (fa_on && (fa.tensors & OPT_SH)) {
            const float ib1 = 1.0f / fa_bc1, ib2 = 1.0f / fa_bc2_sqrt;
            float4* const pw = reinterpret_cast<float4*>(fa.p[1]);
            float4* const mw = reinterpret_cast<float4*>(fa.m[1]);
            float4* const vw = reinterpret_cast<float4*>(fa.v[1]);
            float4 p4[12];
            float4 (&m4)[12] = fa_m4, (&v4)[12] = fa_v4;
            const __amdgpu_buffer_rsrc_t rs_fp = __builtin_amdgcn_make_buffer_rsrc((void*)fa.p[1], 0, 0xffffff00u, 0x00020000u);
            const int n_act = fa.sh_active_parts;            // float4 parts with an ever-active coefficient: the full update
            const bool skip_rest = fa.sh_inactive_zero != 0; // the others: weight decay alone -- or, known to be zero, nothing at all
#pragma unroll
            for (int j = 0; j < SH_J; j++) {
                const int f = (int)threadIdx.x + j * (int)blockDim.x, part = f % 12;
                const size_t a = (size_t)min(gaussian_of_row(LPG_L2 == 0 ? f / 12 : (min(f / 12, rows_used - 1) << lpg_l2)), P - 1) * 12 + (size_t)part;
                const uint32_t o = (uint32_t)a * 16u;
                const v4f_sh pq = __builtin_amdgcn_raw_buffer_load_b128(rs_fp, (part < n_act || !skip_rest) ? o : SH_OOB, 0, 0);
                p4[j] = make_float4(pq.x, pq.y, pq.z, pq.w);
                if (j >= FA_HOIST) {
                    const v4f_sh mq = __builtin_amdgcn_raw_buffer_load_b128(rs_fm, part < n_act ? o : SH_OOB, 0, 0);
                    const v4f_sh vq = __builtin_amdgcn_raw_buffer_load_b128(rs_fv, part < n_act ? o : SH_OOB, 0, 0);
                    m4[j] = make_float4(mq.x, mq.y, mq.z, mq.w); v4[j] = make_float4(vq.x, vq.y, vq.z, vq.w);
                }
            }
#pragma unroll
            for (int j = 0; j < SH_J; j++) {
                const int f = (int)threadIdx.x + j * (int)blockDim.x, part = f % 12;
                const int q = LPG_L2 == 0 ? f / 12 : min(f / 12, rows_used - 1);
                const int gi = (LPG_L2 == 0 || f / 12 < rows_used) ? gaussian_of_row(q << lpg_l2) : 0x7fffffff;
                const float* r = &s_dsh[(q << lpg_l2) * SH_ROW + part * 4];
                float pe[4] = { p4[j].x, p4[j].y, p4[j].z, p4[j].w }, me[4] = { m4[j].x, m4[j].y, m4[j].z, m4[j].w };
                float ve[4] = { v4[j].x, v4[j].y, v4[j].z, v4[j].w };
                if (part < n_act) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        adamw_element(pe[k], r[k], me[k], ve[k], (part == 0 && k < 3) ? fa.lr[1] : fa.lr_sh_rest, fa.betas, fa.eps,
                                      fa.weight_decay, ib1, ib2);
                    if (gi < P) {
                        const size_t a = (size_t)gi * 12 + (size_t)part;
                        pw[a] = make_float4(pe[0], pe[1], pe[2], pe[3]);
                        mw[a] = make_float4(me[0], me[1], me[2], me[3]);
                        vw[a] = make_float4(ve[0], ve[1], ve[2], ve[3]);
                    }
                } else if (!skip_rest) {
                    // never a gradient, zero moments: adamw_element's result for g = m = v = 0 is p (1 - lr wd) exactly (its last FMA adds
                    // -(lr / bc1) x 0 x rcp(eps)), and the moments stay zero -- the same bits without touching them
                    float pn[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) pn[k] = __fmul_rn(pe[k], __fsub_rn(1.0f, __fmul_rn(fa.lr_sh_rest, fa.weight_decay)));
                    if (gi < P && (pn[0] != pe[0] || pn[1] != pe[1] || pn[2] != pe[2] || pn[3] != pe[3]))     // (a zero stays a zero: no write)
                        pw[(size_t)gi * 12 + (size_t)part] = make_float4(pn[0], pn[1], pn[2], pn[3]);
                }
            }
        }
    }
    // the step word was read at this block's start: the last block to get here stores the new count (adamw.h)
    if (fa_on && threadIdx.x == 0) adamw_step_end(fa.step_state, fa_t);
    PSTAMP(6); PRSTAMP(14);
#undef PSTAMP
#undef PRSTAMP
#undef rows_used
}

__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float view[16];
#pragma unroll
    for (int i = 0; i < 16; i++) view[i] = viewmatrix[i];
    const float3 p = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
    const float3 pv = xform4x3(p, view);
    present[idx] = (pv.z <= 0.2f) ? 0 : 1;      // rasterizer_impl.cu:54-66, auxiliary.h:154
}

__global__ void __launch_bounds__(256)
export_geometry_kernel(int P, GeomView g, float* depths, float* means2D, float* conic_opacity, float* rgb,
                       uint32_t* tiles_touched, uint8_t* clamped, float* cov3D)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const bool vis = g.tiles_touched[idx] > 0;
    const float4 z4 = make_float4(0, 0, 0, 0);
    const float4 a = vis ? g.geo[4 * (size_t)idx] : z4, b = vis ? g.geo[4 * (size_t)idx + 1] : z4, c = vis ? g.geo[4 * (size_t)idx + 2] : z4;
    if (depths) depths[idx] = c.w;
    if (means2D) { means2D[2 * (size_t)idx] = a.x; means2D[2 * (size_t)idx + 1] = a.y; }
    if (conic_opacity) reinterpret_cast<float4*>(conic_opacity)[idx] = b;
    if (rgb) { rgb[3 * (size_t)idx] = c.x; rgb[3 * (size_t)idx + 1] = c.y; rgb[3 * (size_t)idx + 2] = c.z; }
    if (tiles_touched) tiles_touched[idx] = g.tiles_touched[idx];
    if (clamped) {
        const uint8_t cl = vis ? g.clamped[idx] : 0;
        clamped[3 * (size_t)idx] = cl & 1; clamped[3 * (size_t)idx + 1] = (cl >> 1) & 1; clamped[3 * (size_t)idx + 2] = (cl >> 2) & 1;
    }
    if (cov3D) for (int i = 0; i < 6; i++) cov3D[6 * (size_t)idx + i] = g.cov3D[6 * (size_t)idx + i];
}

}  // anonymous namespace

static int device_cus()
{
    static const int n = [] {
        int dev = 0; hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return n;
}


void launch_preprocess_forward(const FrameParams& fp, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                               const float* transforms, const float* translation, GeomView g, ImageView im, int* radii_out, hipStream_t s,
                               uint64_t* scatter_keys, uint32_t key_stride)
{
    const int T = fp.gx * fp.gy;
    const int lds_hist = (T <= MAX_LDS_TILES) ? 1 : 0;
    if (!lds_hist) scatter_keys = nullptr;                   // (scatter mode works on the LDS histogram: raster_api.hip asks for it only then)
    // Gaussians per thread: more -> better aggregation of the tile-histogram atomics, fewer -> more waves in flight.
    static const int per_thread = knob("MOSS_PREPROCESS_ITEMS", 1);
    int blocks = (fp.P + 256 * per_thread - 1) / (256 * per_thread);
    if (blocks < 1) blocks = 1;
    const size_t lds_h = lds_hist ? (size_t)((T + 3) & ~3) * sizeof(uint32_t) : 0, lds_s = (size_t)256 * SH_ROW_F * sizeof(float);
    // (dynamic LDS beyond the default 64 KB of a launch -- 1024 x 1024: 16 KB of histogram + 49 KB of SH rows -- is asked for once; two
    // such blocks still fit a CU's 160 KB.  Round 5's first scatter mode kept a second array of run starts beside the histogram: 82 KB
    // there, no SH staging and ONE block per CU)
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    const int stage_sh = (fp.M == 16 && shs != nullptr && colors_precomp == nullptr && (reinterpret_cast<uintptr_t>(shs) & 15u) == 0 &&
                          (size_t)fp.P * 192u < 0xffffff00u /* the staging loads address the tensor with 32-bit byte offsets */ &&
                          lds_h + lds_s <= (big_lds ? 80u * 1024u : 65536u) && knob("MOSS_PREFWD_STAGE", 1)) ? 1 : 0;
    const size_t lds = lds_h + (stage_sh ? lds_s : 0);
    MOSS_LAUNCH_TIMED(preprocess_forward_kernel, dim3(blocks), dim3(256), lds, s,
                       fp.P, fp.D, fp.M, fp.W, fp.H, fp.gx, fp.gy, fp.tan_fovx, fp.tan_fovy, fp.focal_x, fp.focal_y,
                       fp.scale_modifier, fp.prefiltered, means3D, shs, colors_precomp, opacities, scales, rotations,
                       cov3D_precomp, fp.view_dev, fp.proj_dev, fp.campos_dev, g, im.tile_count, im.flags_acc, radii_out, lds_hist, stage_sh,
                       transforms, fp.raw | ((knob("MOSS_EXPERIMENT", 0) & 1) ? 0x100 : 0),
                       (g_stamps && knob("MOSS_SORT_STAMPS", 0)) ? g_stamps + 131072 + 32768 : nullptr, translation, scatter_keys, key_stride);
}

// resident blocks per CU of the per-Gaussian backward (the fused instantiation: the larger one)
static int prebwd_resident_per_cu(int threads)
{
    static const int occ = [&] {
        const size_t lds_bytes_staged = (size_t)threads * SH_ROW * sizeof(float) + (size_t)((threads + 63) / 64) * GATHER_WORDS * 4;
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, preprocess_backward_kernel<true, true, 4>, threads, lds_bytes_staged) != hipSuccess || o < 1) o = 1;
        return o;
    }();
    return occ;
}

void launch_preprocess_backward(const FrameParams& fp, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                                GeomView g, BinView b, const uint32_t* header, uint32_t* queues,
                                float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                                const float* transforms, float* dL_dtransforms, const float* translation, float* dL_dtranslation, hipStream_t s,
                                const FusedAdam* fused)
{
    (void)colors_precomp;
    const FusedAdam fa = fused ? *fused : FusedAdam();
    static const int threads = std::max(64, knob("MOSS_PREBWD_THREADS", 64) & ~63);
    // Rows of a block = groups of 2^gl2 consecutive Gaussians (gaussian_of_row); 6 = the block's 64 rows are consecutive: the default.
    // When the caller says that index neighbours are spatial neighbours (MOSS_HINT_SPATIAL_ORDER) 64 consecutive Gaussians are all-heavy
    // or all-light, and while every block of the grid is resident at once the kernel lasts as long as its heaviest block: then groups of
    // 16 from four places (bench frame in Morton order 29.3 -> 25.0 us).  Not otherwise: in an order without locality the groups cost
    // coalescing (stress case with image-covering Gaussians 91.5 -> 99.3 us), and once the grid runs in several rounds the dispatcher
    // does the balancing while the shared cache lines of 64 neighbours count for more (configs[4]: 81 us consecutive, 87-95 in groups).
    static const int gl2_env = knob("MOSS_PREBWD_GROUP_LOG2", 0);
    static const int gather_knob = knob("MOSS_GATHER", 0) == 1 ? 0x200 : knob("MOSS_GATHER", 0) == 2 ? 0x400 : 0;
    // (the fused update of the SH records works on the staged rows: raster_api.hip refuses it unless M == 16 and the arrays are aligned)
    const bool stage = fp.M == 16 && shs != nullptr && (dL_dsh != nullptr || (fa.tensors & OPT_SH)) && (size_t)fp.P * 192u < 0xffffff00u &&
                       (knob("MOSS_PREBWD_STAGE", 1) || (fa.tensors & OPT_SH)) &&
                       (reinterpret_cast<uintptr_t>(shs) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dL_dsh) & 15u) == 0;
    // lanes per Gaussian (see the kernel): 16 while the whole grid is then still resident at once -- P <= 8192 with two waves per SIMD
    // (instantiated for the staged-SH kernels -- MOSS's case, M = 16 -- only)
    static const int lpg_env = knob("MOSS_PREBWD_LPG_LOG2", -1);
    int lpg_l2 = 0;
    if (stage) {
        const long long resident = (long long)prebwd_resident_per_cu(threads) * device_cus();
        if (((long long)fp.P + (threads >> 4) - 1) / (threads >> 4) <= resident) lpg_l2 = 4;
        if (lpg_env >= 0) lpg_l2 = lpg_env >= 4 ? 4 : 0;
    }
    const int blocks = (fp.P + (threads >> lpg_l2) - 1) / (threads >> lpg_l2);
    const int resident_per_cu = prebwd_resident_per_cu(threads);
    const int gl2 = gl2_env ? std::max(4, std::min(6, gl2_env))
                            : (((fp.raw & HINT_SPATIAL_ORDER) && blocks <= resident_per_cu * device_cus()) ? 4 : 6);
#define LAUNCH_PB(STAGE, FUSE, LPG)                                                                                             \
    MOSS_LAUNCH_TIMED((preprocess_backward_kernel<STAGE, FUSE, LPG>), dim3(blocks), dim3(threads),                             \
                       ((STAGE) ? (size_t)threads * SH_ROW * sizeof(float) : 0) + (size_t)((threads + 63) / 64) * GATHER_WORDS * 4, s,                                               \
                       fp.P, fp.D, fp.M, fp.tan_fovx, fp.tan_fovy, fp.focal_x, fp.focal_y, -0.5f * (float)fp.W, -0.5f * (float)fp.H, fp.scale_modifier, \
                       means3D, shs, scales, rotations, cov3D_precomp, fp.view_dev, fp.proj_dev, fp.campos_dev,                 \
                       g, b.inst_grad, b.cell_valid, header, dL_dmean2D, dL_dconic, \
                       dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, transforms, dL_dtransforms, opacities, fp.raw | gather_knob | (gl2 << 12), g_stamps, queues, translation, dL_dtranslation, fa)
    // (lanes per Gaussian: instantiated for the staged-SH kernels -- MOSS's case, M = 16 -- only)
    if (stage && lpg_l2 == 4) { if (fa.tensors != 0u) LAUNCH_PB(true, true, 4); else LAUNCH_PB(true, false, 4); }
    else if (fa.tensors != 0u) { if (stage) LAUNCH_PB(true, true, 0); else LAUNCH_PB(false, true, 0); }
    else if (stage) LAUNCH_PB(true, false, 0); else LAUNCH_PB(false, false, 0);
#undef LAUNCH_PB
}

void launch_mark_visible(int P, const float* means3D, const float* view16_dev, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view16_dev, present);
}

void launch_export_geometry(int P, GeomView g, float* depths, float* means2D, float* conic_opacity, float* rgb,
                            uint32_t* tiles_touched, uint8_t* clamped, float* cov3D, hipStream_t s)
{
    hipLaunchKernelGGL(export_geometry_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g, depths, means2D, conic_opacity, rgb,
                       tiles_touched, clamped, cov3D);
}

}  // namespace moss
