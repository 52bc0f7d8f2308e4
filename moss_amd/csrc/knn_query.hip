// knn_query.hip -- k nearest reference points of every query point.
//
// MOSS calls the third-party `knn_cuda.KNN` (a CUDA-only binary wheel that is not part of the repository) in its per-step LBS
// path -- for every Gaussian the nearest SMPL vertex, k = 1, 6 890 references (scene/gaussian_model.py:85,657,827) -- and with
// k = 2 on the Gaussians themselves when densifying (:86,586,759).  Without a replacement the real MOSS step cannot run on an
// MI355X at all.  This is a brute-force, exact k <= 4 query for 3-D points: one thread per query, the references staged through
// LDS in tiles of 1024 points (every lane of a wave reads the same reference at a time: an LDS broadcast), k best kept in
// registers.  Ties keep the lower reference index (references are visited in ascending order and a candidate must be strictly
// closer to displace a kept one).  Distances are Euclidean (sqrt of the fp32 sum of squares), ascending.
// Cost: Nq x Nr pairs at ~10 lane-instructions each: 100k x 6 890 = 0.2 ms; 100k x 100k = ~3 ms (densification only).
#include "common.h"

namespace moss {
namespace {

constexpr int KNN_TILE = 1024;

template <int K>
__global__ void __launch_bounds__(256)
knn_query_kernel(int Nr, int Nq, const float* __restrict__ ref, const float* __restrict__ query, float* __restrict__ dist_out,
                 long long* __restrict__ idx_out)
{
    __shared__ float4 s_ref[KNN_TILE];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int qc = min(q, Nq - 1);
    const float qx = query[3 * (size_t)qc], qy = query[3 * (size_t)qc + 1], qz = query[3 * (size_t)qc + 2];
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = __builtin_huge_valf(); bi[k] = -1; }
    for (int base = 0; base < Nr; base += KNN_TILE) {
        const int cnt = min(KNN_TILE, Nr - base);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
            const float* r = ref + 3 * (size_t)(base + j);
            s_ref[j] = make_float4(r[0], r[1], r[2], 0.0f);
        }
        __syncthreads();
        for (int j = 0; j < cnt; j++) {
            const float4 r = s_ref[j];
            const float dx = r.x - qx, dy = r.y - qy, dz = r.z - qz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < bd[K - 1]) {                                           // strictly closer than the worst kept one
                float cd = d; int ci = base + j;
                bool shifting = false;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    // insertion: on equal distance the kept (lower) index stays in front; once the newcomer is placed, everything
                    // behind it moves down one slot unconditionally (a displaced entry must not stop at an equal distance)
                    if (shifting || cd < bd[k]) { const float td = bd[k]; const int ti = bi[k]; bd[k] = cd; bi[k] = ci; cd = td; ci = ti; shifting = true; }
                }
            }
        }
    }
    if (q < Nq) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            dist_out[(size_t)q * K + k] = sqrtf(bd[k]);
            idx_out[(size_t)q * K + k] = (long long)bi[k];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Cell-grid variant (exact, same results as the brute-force kernel above bit for bit, including the order of ties).
//
// The self-query MOSS runs when densifying (k = 2 over all Gaussians, scene/gaussian_model.py:586,759) is 10^10 pairs by brute
// force (7.7 ms measured at 100k).  Here the references are binned ONCE into a uniform grid of roughly cubic cells laid over their
// bounding box (about two cells per reference; a counting sort: count with atomics that also hand out each reference's rank inside
// its cell, scan the counts, scatter), and a query walks Chebyshev shells of cells around its own cell until its k-th best
// distance is closer than the nearest face of the cube of cells it has covered -- nothing outside that cube can be closer.  Cells
// that are consecutive along x are consecutive in the sorted array, so a shell costs one range lookup per (y, z) row.
// Everything (bounding box, grid dimensions, prefix sums) stays on the device: no host round trip, one stream.
// Equal distances are ordered by reference index explicitly (the brute-force kernel gets the same order from its ascending visit),
// which also makes the result independent of the order in which atomics placed the references inside a cell.
// build + query are separate entry points: MOSS's per-step query is against the SAME 6 890 template vertices every step
// (scene/gaussian_model.py:827 with t_vertices of the big pose), whose grid can be built once.

constexpr int GRID_HDR = 32;                // words: [0..2] min, [3..5] max (ordered uints); [6] Nr; [8..10] Gx,Gy,Gz; floats [12..14] lo, [15..17] h, [18..20] 1/h, [21] slack
constexpr int SCAN_BLOCK = 1024;            // cells per block of the first scan level

__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct GridView {
    uint32_t* hdr; uint32_t* start; uint32_t* bsum; uint32_t* cell; uint32_t* rank; float4* sorted;
    int ncells_max, nblocks;
    static int cells_for(int Nr) { long long c = 2ll * Nr; if (c < 64) c = 64; if (c > (1ll << 21)) c = 1ll << 21; return (int)c; }
    static GridView at(char* base, int Nr)
    {
        GridView v; char* p = base;
        v.ncells_max = cells_for(Nr);
        v.nblocks = (v.ncells_max + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
        v.hdr = carve<uint32_t>(p, GRID_HDR);
        v.start = carve<uint32_t>(p, (size_t)v.nblocks * SCAN_BLOCK);
        v.bsum = carve<uint32_t>(p, (size_t)v.nblocks);
        v.cell = carve<uint32_t>(p, (size_t)Nr);
        v.rank = carve<uint32_t>(p, (size_t)Nr);
        v.sorted = carve<float4>(p, (size_t)Nr);
        return v;
    }
    static size_t bytes(int Nr) { char* z = nullptr; GridView v = at(z, Nr); return (size_t)((char*)v.sorted - z) + align_up((size_t)Nr * sizeof(float4)); }
};

__global__ void __launch_bounds__(256)
grid_init_kernel(uint32_t* __restrict__ hdr, uint32_t* __restrict__ start, int words, int Nr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < GRID_HDR) hdr[i] = i < 3 ? 0xffffffffu : (i == 6 ? (uint32_t)Nr : 0u);
    for (int j = i; j < words; j += gridDim.x * blockDim.x) start[j] = 0u;
}

__global__ void __launch_bounds__(256)
grid_bounds_kernel(int Nr, const float* __restrict__ pts, uint32_t* __restrict__ hdr)
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    float mn[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float mx[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Nr; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], d)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d)); }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][k] = mn[k]; s_mx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    // one set of atomics per workgroup (and few workgroups): thousands of ordered atomics on the same six words serialise
    // (24k of them took 108 us at 100k points)
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        atomicMin(&hdr[k], f2ord(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]))));
        atomicMax(&hdr[3 + k], f2ord(fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]))));
    }
}

// one thread: cell size and grid dimensions from the bounding box (roughly cubic cells, at most ncells_max of them)
__global__ void grid_dims_kernel(uint32_t* __restrict__ hdr, int ncells_max)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3], ext[3], emax = 0.0f, amax = 0.0f;
    for (int k = 0; k < 3; k++) {
        lo[k] = ord2f(hdr[k]); const float hi = ord2f(hdr[3 + k]);
        ext[k] = hi - lo[k]; emax = fmaxf(emax, ext[k]); amax = fmaxf(amax, fmaxf(fabsf(lo[k]), fabsf(hi)));
    }
    int G[3] = { 1, 1, 1 };
    if (emax > 0.0f) {
        float e[3], vol = 1.0f;
        for (int k = 0; k < 3; k++) { e[k] = fmaxf(ext[k], 1e-3f * emax); vol *= e[k]; }
        float h = cbrtf(vol / (float)ncells_max);
        for (int it = 0; it < 64; it++) {
            long long total = 1;
            for (int k = 0; k < 3; k++) {
                G[k] = ext[k] > 0.0f ? (int)fminf(fmaxf(ceilf(e[k] / h), 1.0f), 1024.0f) : 1;
                total *= G[k];
            }
            if (total <= ncells_max) break;
            h *= 1.1f;
            if (it == 63) { G[0] = G[1] = G[2] = 1; }
        }
    }
    float* hf = reinterpret_cast<float*>(hdr);
    for (int k = 0; k < 3; k++) {
        hdr[8 + k] = (uint32_t)G[k];
        hf[12 + k] = lo[k];
        hf[15 + k] = ext[k] > 0.0f ? ext[k] / (float)G[k] : 0.0f;
        hf[18 + k] = ext[k] > 0.0f ? (float)G[k] / ext[k] : 0.0f;
    }
    // a reference may sit a few ulps on the wrong side of a computed cell face: the covered-cube bound is shrunk by this much
    hf[21] = 1e-5f * fmaxf(emax, amax);
}

struct GridParams { int G[3]; float lo[3], h[3], inv_h[3], slack; };

__device__ __forceinline__ GridParams load_grid(const uint32_t* __restrict__ hdr)
{
    GridParams g; const float* hf = reinterpret_cast<const float*>(hdr);
#pragma unroll
    for (int k = 0; k < 3; k++) { g.G[k] = (int)hdr[8 + k]; g.lo[k] = hf[12 + k]; g.h[k] = hf[15 + k]; g.inv_h[k] = hf[18 + k]; }
    g.slack = hf[21];
    return g;
}

__device__ __forceinline__ int cell_coord(const GridParams& g, int k, float x)
{
    const float v = floorf((x - g.lo[k]) * g.inv_h[k]);
    return (int)fminf(fmaxf(v, 0.0f), (float)(g.G[k] - 1));        // NaN -> 0
}

__global__ void __launch_bounds__(256)
grid_count_kernel(int Nr, const float* __restrict__ pts, const uint32_t* __restrict__ hdr, uint32_t* __restrict__ counts,
                  uint32_t* __restrict__ cell, uint32_t* __restrict__ rank)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nr) return;
    const GridParams g = load_grid(hdr);
    const int cx = cell_coord(g, 0, pts[3 * (size_t)i]), cy = cell_coord(g, 1, pts[3 * (size_t)i + 1]), cz = cell_coord(g, 2, pts[3 * (size_t)i + 2]);
    const uint32_t c = (uint32_t)((cz * g.G[1] + cy) * g.G[0] + cx);
    cell[i] = c;
    rank[i] = atomicAdd(&counts[c], 1u);
}

// exclusive scan, level 1: each block turns its 1024 counts into block-local exclusive prefixes and reports its total
__global__ void __launch_bounds__(256)
grid_scan_blocks_kernel(uint32_t* __restrict__ start, uint32_t* __restrict__ bsum)
{
    __shared__ uint32_t s_wave[4];
    uint4* row = reinterpret_cast<uint4*>(start + (size_t)blockIdx.x * SCAN_BLOCK);
    const uint4 v = row[threadIdx.x];
    const uint32_t mine = v.x + v.y + v.z + v.w;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if ((int)(threadIdx.x & 63) >= d) inc += t; }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += s_wave[w];
    const uint32_t ex = base + inc - mine;
    row[threadIdx.x] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
    if (threadIdx.x == 255) bsum[blockIdx.x] = base + inc;
}

// level 2: one workgroup scans the (<= 4096) block totals in place (exclusive)
__global__ void __launch_bounds__(1024)
grid_scan_sums_kernel(uint32_t* __restrict__ bsum, int nblocks)
{
    __shared__ uint32_t s_wave[16];
    uint32_t v[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = 4 * (int)threadIdx.x + k; v[k] = i < nblocks ? bsum[i] : 0u; mine += v[k]; }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if ((int)(threadIdx.x & 63) >= d) inc += t; }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += s_wave[w];
    uint32_t run = base + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = 4 * (int)threadIdx.x + k; if (i < nblocks) bsum[i] = run; run += v[k]; }
}

// level 3: make the prefixes absolute
__global__ void __launch_bounds__(256)
grid_scan_add_kernel(uint32_t* __restrict__ start, const uint32_t* __restrict__ bsum)
{
    const uint32_t add = bsum[blockIdx.x];
    uint4* row = reinterpret_cast<uint4*>(start + (size_t)blockIdx.x * SCAN_BLOCK);
    uint4 v = row[threadIdx.x];
    v.x += add; v.y += add; v.z += add; v.w += add;
    row[threadIdx.x] = v;
}

__global__ void __launch_bounds__(256)
grid_scatter_kernel(int Nr, const float* __restrict__ pts, const uint32_t* __restrict__ start, const uint32_t* __restrict__ cell,
                    const uint32_t* __restrict__ rank, float4* __restrict__ sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nr) return;
    const uint32_t pos = start[cell[i]] + rank[i];
    if (pos < (uint32_t)Nr)
        sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

template <int K>
__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, uint32_t s, uint32_t e, float qx, float qy, float qz,
                                           float* bd, int* bi)
{
    for (uint32_t j = s; j < e; j++) {
        const float4 r = sorted[j];
        const float dx = r.x - qx, dy = r.y - qy, dz = r.z - qz;
        const float d = dx * dx + dy * dy + dz * dz;                         // the brute-force kernel's expression
        const int id = __float_as_int(r.w);
        if (d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1])) {
            float cd = d; int ci = id;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (cd < bd[k] || (cd == bd[k] && ci < bi[k])) { const float td = bd[k]; const int ti = bi[k]; bd[k] = cd; bi[k] = ci; cd = td; ci = ti; }
            }
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256)
grid_query_kernel(int Nr, int Nq, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ start, const float4* __restrict__ sorted,
                  const float* __restrict__ query, float* __restrict__ dist_out, long long* __restrict__ idx_out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Nq) return;
    const GridParams g = load_grid(hdr);
    const float qx = query[3 * (size_t)q], qy = query[3 * (size_t)q + 1], qz = query[3 * (size_t)q + 2];
    const int cx = cell_coord(g, 0, qx), cy = cell_coord(g, 1, qy), cz = cell_coord(g, 2, qz);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = __builtin_huge_valf(); bi[k] = 0x7fffffff; }
    const uint32_t cap = (uint32_t)Nr;                                       // never read past the array whatever the table holds
    const int rmax = max(max(g.G[0], g.G[1]), g.G[2]);
    for (int r = 0; r < rmax; r++) {
        const int xlo = max(cx - r, 0), xhi = min(cx + r, g.G[0] - 1);
        const int ylo = max(cy - r, 0), yhi = min(cy + r, g.G[1] - 1);
        const int zlo = max(cz - r, 0), zhi = min(cz + r, g.G[2] - 1);
        for (int z = zlo; z <= zhi; z++)
            for (int y = ylo; y <= yhi; y++) {
                const int row = (z * g.G[1] + y) * g.G[0];
                const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                if (face) {                                                 // the whole run of cells along x belongs to the shell
                    scan_range<K>(sorted, min(start[row + xlo], cap), min(start[row + xhi + 1], cap), qx, qy, qz, bd, bi);
                } else {                                                    // only the two end cells do
                    if (cx - r >= 0) scan_range<K>(sorted, min(start[row + cx - r], cap), min(start[row + cx - r + 1], cap), qx, qy, qz, bd, bi);
                    if (cx + r < g.G[0]) scan_range<K>(sorted, min(start[row + cx + r], cap), min(start[row + cx + r + 1], cap), qx, qy, qz, bd, bi);
                }
            }
        // distance from the query to the nearest face of the covered cube that has cells beyond it
        float b = __builtin_huge_valf();
        if (cx - r > 0) b = fminf(b, qx - (g.lo[0] + (float)(cx - r) * g.h[0]));
        if (cx + r < g.G[0] - 1) b = fminf(b, (g.lo[0] + (float)(cx + r + 1) * g.h[0]) - qx);
        if (cy - r > 0) b = fminf(b, qy - (g.lo[1] + (float)(cy - r) * g.h[1]));
        if (cy + r < g.G[1] - 1) b = fminf(b, (g.lo[1] + (float)(cy + r + 1) * g.h[1]) - qy);
        if (cz - r > 0) b = fminf(b, qz - (g.lo[2] + (float)(cz - r) * g.h[2]));
        if (cz + r < g.G[2] - 1) b = fminf(b, (g.lo[2] + (float)(cz + r + 1) * g.h[2]) - qz);
        if (b == __builtin_huge_valf()) break;                               // the cube covers the whole grid
        b -= g.slack;
        if (b > 0.0f && bd[K - 1] < b * b * 0.9999f) break;                  // nothing outside the cube can be as close
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        dist_out[(size_t)q * K + k] = sqrtf(bd[k]);
        idx_out[(size_t)q * K + k] = (long long)bi[k];
    }
}

}  // namespace
}  // namespace moss

extern "C" int moss_knn_query(int Nr, int Nq, int k, const float* ref, const float* query, float* dist_out, long long* idx_out,
                              void* stream)
{
    if (Nr < 0 || Nq < 0 || k < 1 || k > 4) return MOSS_ERR_INVALID_ARG;
    if (Nq == 0) return 0;
    if (Nr < k || !ref || !query || !dist_out || !idx_out) return MOSS_ERR_INVALID_ARG;
    const dim3 grid((Nq + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
    case 1: hipLaunchKernelGGL(moss::knn_query_kernel<1>, grid, block, 0, s, Nr, Nq, ref, query, dist_out, idx_out); break;
    case 2: hipLaunchKernelGGL(moss::knn_query_kernel<2>, grid, block, 0, s, Nr, Nq, ref, query, dist_out, idx_out); break;
    case 3: hipLaunchKernelGGL(moss::knn_query_kernel<3>, grid, block, 0, s, Nr, Nq, ref, query, dist_out, idx_out); break;
    default: hipLaunchKernelGGL(moss::knn_query_kernel<4>, grid, block, 0, s, Nr, Nq, ref, query, dist_out, idx_out); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

extern "C" size_t moss_knn_grid_workspace_bytes(int Nr) { return moss::GridView::bytes(Nr > 0 ? Nr : 1); }

extern "C" int moss_knn_grid_build(int Nr, const float* ref, char* workspace, size_t workspace_bytes, void* stream)
{
    using namespace moss;
    if (Nr < 1 || !ref || !workspace || workspace_bytes < GridView::bytes(Nr)) return MOSS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    GridView v = GridView::at(workspace, Nr);
    const int blocks = (Nr + 255) / 256;
    const int words = v.nblocks * SCAN_BLOCK;
    hipLaunchKernelGGL(grid_init_kernel, dim3(min((words + 255) / 256, 2048)), dim3(256), 0, s, v.hdr, v.start, words, Nr);
    hipLaunchKernelGGL(grid_bounds_kernel, dim3(min(blocks, 128)), dim3(256), 0, s, Nr, ref, v.hdr);
    hipLaunchKernelGGL(grid_dims_kernel, dim3(1), dim3(64), 0, s, v.hdr, v.ncells_max);
    hipLaunchKernelGGL(grid_count_kernel, dim3(blocks), dim3(256), 0, s, Nr, ref, v.hdr, v.start, v.cell, v.rank);
    hipLaunchKernelGGL(grid_scan_blocks_kernel, dim3(v.nblocks), dim3(256), 0, s, v.start, v.bsum);
    hipLaunchKernelGGL(grid_scan_sums_kernel, dim3(1), dim3(1024), 0, s, v.bsum, v.nblocks);
    hipLaunchKernelGGL(grid_scan_add_kernel, dim3(v.nblocks), dim3(256), 0, s, v.start, v.bsum);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(blocks), dim3(256), 0, s, Nr, ref, v.start, v.cell, v.rank, v.sorted);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

extern "C" int moss_knn_grid_query(int Nr, int Nq, int k, const char* workspace, size_t workspace_bytes, const float* query,
                                   float* dist_out, long long* idx_out, void* stream)
{
    using namespace moss;
    if (Nr < 1 || Nq < 0 || k < 1 || k > 4) return MOSS_ERR_INVALID_ARG;
    if (Nq == 0) return 0;
    if (Nr < k || !workspace || workspace_bytes < GridView::bytes(Nr) || !query || !dist_out || !idx_out) return MOSS_ERR_INVALID_ARG;
    GridView v = GridView::at(const_cast<char*>(workspace), Nr);
    const dim3 grid((Nq + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
    case 1: hipLaunchKernelGGL(grid_query_kernel<1>, grid, block, 0, s, Nr, Nq, v.hdr, v.start, v.sorted, query, dist_out, idx_out); break;
    case 2: hipLaunchKernelGGL(grid_query_kernel<2>, grid, block, 0, s, Nr, Nq, v.hdr, v.start, v.sorted, query, dist_out, idx_out); break;
    case 3: hipLaunchKernelGGL(grid_query_kernel<3>, grid, block, 0, s, Nr, Nq, v.hdr, v.start, v.sorted, query, dist_out, idx_out); break;
    default: hipLaunchKernelGGL(grid_query_kernel<4>, grid, block, 0, s, Nr, Nq, v.hdr, v.start, v.sorted, query, dist_out, idx_out); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
