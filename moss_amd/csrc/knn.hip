// knn.hip -- distCUDA2 replacement: mean squared distance to the 3 nearest other points, exact.
// Reference: SKNN/simple_knn.cu:185-221 (SimpleKNN::knn) and its kernels :45-183.
//
// Same algorithmic skeleton as the reference (Morton order -> boxes of 1024 consecutive points -> per point, scan
// only boxes that can still hold a closer neighbour), re-shaped for CDNA4:
//   * the Morton keys are sorted by a device-wide bitonic network (LDS tiles + streaming steps, see launch_sort_keys), not by one
//     workgroup;
//   * the scene bounding box is reduced ON THE DEVICE with ordered-integer atomics and consumed from device memory,
//     so there is no host round trip at all (the reference blocks twice, simple_knn.cu:197,200);
//   * points are gathered once into Morton order as float4 {x,y,z,original index}; a 256-thread workgroup then owns
//     256 Morton-consecutive (= spatially close) queries, whose candidate boxes largely coincide: a box needed by ANY
//     lane is staged once into LDS (12 KB... 16 KB as float4) with coalesced loads and scanned from there by the
//     lanes that need it, instead of every thread walking global memory on its own (simple_knn.cu:175-180).
// Compiled with -ffp-contract=off: squared distances are evaluated exactly as the source expression
// d.x*d.x + d.y*d.y + d.z*d.z (:134-135), and the three smallest values of a multiset do not depend on visiting order,
// so the result is bit-identical to the brute-force CPU oracle.
#include "common.h"
#include <cfloat>

namespace moss {

namespace {

constexpr int BOX = 1024;      // simple_knn.cu:12

__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void __launch_bounds__(256)
bounds_kernel(int P, const float* __restrict__ pts, uint32_t* __restrict__ mm /* [0..2]=min, [3..5]=max, ordered ints */)
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], d)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d)); }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][k] = mn[k]; s_mx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {                          // one set of atomics per workgroup: same-address atomics serialise
        const int k = threadIdx.x;
        atomicMin(&mm[k], f2ord(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]))));
        atomicMax(&mm[3 + k], f2ord(fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]))));
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)     // spread 10 bits, simple_knn.cu:45-52
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256)
morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm, uint64_t* __restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(mm[k]), hi = ord2f(mm[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (pts[3 * (size_t)i + k] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        code |= prep_morton((uint32_t)(u * 1023.0f)) << k;
    }
    keys[i] = ((uint64_t)code << 32) | (uint32_t)i;
}

// ---- sort of the n 64-bit keys (Morton code << 32 | index), in place, ascending.  A bitonic network whose phases start with a
// MIRRORED step (element i of a 2^lk block against element 2^lk - 1 - i), so that every later step of the phase is a plain ascending
// compare-exchange at distance j and elements past n behave as +infinity without being stored (`hi < n`).
//   * everything at distance < SORT_TILE stays inside a tile of SORT_TILE keys: done in LDS, many steps per launch
//     (sort_tiles_kernel: all phases up to the tile size; sort_tail_kernel: the last log2(SORT_TILE) steps of a larger phase);
//   * the steps at distance >= SORT_TILE are one streaming launch each (sort_step_kernel).
// Round 2 ran the whole network in ONE 1024-thread workgroup (fine for the 6 890 SMPL vertices of MOSS's initialisation,
// scene/gaussian_model.py:185; a cliff at the 100k - 1M points the rest of this library is sized for: 210 steps x 1M keys on one CU).
// Now 1 launch up to 2048 points and 1 + sum_{lk = 12}^{ceil(log2 n)} (lk - 10) launches beyond (55 for a million points), every
// one of them on the whole device.  Cold path either way: distCUDA2 runs when a model is created.
constexpr uint32_t SORT_TILE_LOG2 = 11, SORT_TILE = 1u << SORT_TILE_LOG2;         // keys per workgroup of the LDS kernels (16 KB)

__device__ __forceinline__ void cmpx(uint64_t& x, uint64_t& y) { if (x > y) { const uint64_t t = x; x = y; y = t; } }

// phases lk = 1 .. min(SORT_TILE_LOG2, lpad) entirely inside each tile: afterwards every tile of SORT_TILE keys is sorted
__global__ void __launch_bounds__(1024)
sort_tiles_kernel(uint64_t* __restrict__ a, uint32_t n, uint32_t lpad)
{
    __shared__ uint64_t s[SORT_TILE];
    const uint32_t base = blockIdx.x * SORT_TILE, t = threadIdx.x;
    for (uint32_t i = t; i < SORT_TILE; i += 1024u) s[i] = base + i < n ? a[base + i] : ~0ull;
    __syncthreads();
    const uint32_t last = min(lpad, SORT_TILE_LOG2);
    for (uint32_t lk = 1; lk <= last; lk++) {
        const uint32_t lhk = lk - 1, off = t & ((1u << lhk) - 1u), blk = t >> lhk;
        { const uint32_t lo = (blk << lk) + off, hi = (blk << lk) + ((1u << lk) - 1u) - off; cmpx(s[lo], s[hi]); }
        __syncthreads();
        for (int lj = (int)lhk - 1; lj >= 0; lj--) {
            const uint32_t j = 1u << lj, lo = ((t >> lj) << (lj + 1)) + (t & (j - 1u));
            cmpx(s[lo], s[lo + j]);
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < SORT_TILE; i += 1024u) if (base + i < n) a[base + i] = s[i];
}

// one step of a phase lk > SORT_TILE_LOG2 at a distance that crosses tiles: the mirrored first step (lj < 0) or a plain step lj
__global__ void __launch_bounds__(256)
sort_step_kernel(uint64_t* __restrict__ a, uint32_t n, uint32_t half, uint32_t lk, int lj)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= half) return;
    uint32_t lo, hi;
    if (lj < 0) { const uint32_t lhk = lk - 1, off = t & ((1u << lhk) - 1u), blk = t >> lhk; lo = (blk << lk) + off; hi = (blk << lk) + ((1u << lk) - 1u) - off; }
    else { const uint32_t j = 1u << lj; lo = ((t >> lj) << (lj + 1)) + (t & (j - 1u)); hi = lo + j; }
    if (hi < n) { uint64_t x = a[lo], y = a[hi]; if (x > y) { a[lo] = y; a[hi] = x; } }
}

// the last SORT_TILE_LOG2 steps (distances SORT_TILE / 2 ... 1) of a larger phase, inside each tile
__global__ void __launch_bounds__(1024)
sort_tail_kernel(uint64_t* __restrict__ a, uint32_t n)
{
    __shared__ uint64_t s[SORT_TILE];
    const uint32_t base = blockIdx.x * SORT_TILE, t = threadIdx.x;
    for (uint32_t i = t; i < SORT_TILE; i += 1024u) s[i] = base + i < n ? a[base + i] : ~0ull;
    __syncthreads();
    for (int lj = (int)SORT_TILE_LOG2 - 1; lj >= 0; lj--) {
        const uint32_t j = 1u << lj, lo = ((t >> lj) << (lj + 1)) + (t & (j - 1u));
        cmpx(s[lo], s[lo + j]);
        __syncthreads();
    }
    for (uint32_t i = t; i < SORT_TILE; i += 1024u) if (base + i < n) a[base + i] = s[i];
}

void launch_sort_keys(uint64_t* keys, uint32_t n, hipStream_t s)
{
    uint32_t lpad = 0;
    while ((1u << lpad) < n) lpad++;
    const uint32_t tiles = (n + SORT_TILE - 1) / SORT_TILE, half = (1u << lpad) >> 1;
    hipLaunchKernelGGL(sort_tiles_kernel, dim3(tiles), dim3(1024), 0, s, keys, n, lpad);
    for (uint32_t lk = SORT_TILE_LOG2 + 1; lk <= lpad; lk++) {
        hipLaunchKernelGGL(sort_step_kernel, dim3((half + 255) / 256), dim3(256), 0, s, keys, n, half, lk, -1);
        for (int lj = (int)lk - 2; lj >= (int)SORT_TILE_LOG2; lj--)
            hipLaunchKernelGGL(sort_step_kernel, dim3((half + 255) / 256), dim3(256), 0, s, keys, n, half, lk, lj);
        hipLaunchKernelGGL(sort_tail_kernel, dim3(tiles), dim3(1024), 0, s, keys, n);
    }
}

__global__ void __launch_bounds__(256)
gather_kernel(int P, const float* __restrict__ pts, const uint64_t* __restrict__ keys, float4* __restrict__ sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t id = (uint32_t)keys[i];
    sorted[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
}

struct Box { float mn[3], mx[3]; };

__global__ void __launch_bounds__(256)
box_kernel(int P, const float4* __restrict__ sorted, Box* __restrict__ boxes)      // simple_knn.cu:78-117
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int b = blockIdx.x;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = b * BOX + threadIdx.x; i < min(P, (b + 1) * BOX); i += blockDim.x) {
        const float4 p = sorted[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], d)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d)); }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][k] = mn[k]; s_mx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        boxes[b].mn[k] = fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]));
        boxes[b].mx[k] = fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]));
    }
}

__device__ __forceinline__ float dist_box_point(const Box& box, float3 p)   // simple_knn.cu:119-129
{
    float dx = 0, dy = 0, dz = 0;
    if (p.x < box.mn[0] || p.x > box.mx[0]) dx = fminf(fabsf(p.x - box.mn[0]), fabsf(p.x - box.mx[0]));
    if (p.y < box.mn[1] || p.y > box.mx[1]) dy = fminf(fabsf(p.y - box.mn[1]), fabsf(p.y - box.mx[1]));
    if (p.z < box.mn[2] || p.z > box.mx[2]) dz = fminf(fabsf(p.z - box.mn[2]), fabsf(p.z - box.mx[2]));
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void update3(float3 ref, float3 pt, float* best)   // updateKBest<3>, simple_knn.cu:131-145
{
    const float dx = pt.x - ref.x, dy = pt.y - ref.y, dz = pt.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

__global__ void __launch_bounds__(256)
mean_dist_kernel(int P, const float4* __restrict__ sorted, const Box* __restrict__ boxes, int num_boxes, float* __restrict__ dists)
{
    __shared__ float4 s_pts[BOX];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < P;
    const float4 me = live ? sorted[idx] : make_float4(0, 0, 0, 0);
    const float3 point = make_float3(me.x, me.y, me.z);
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    if (live) {                                   // seed the rejection radius from the +-3 Morton neighbours (:156-163)
        for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
            if (i == idx) continue;
            const float4 q = sorted[i];
            update3(point, make_float3(q.x, q.y, q.z), best);
        }
    }
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;

    for (int b = 0; b < num_boxes; b++) {
        bool need = false;
        if (live) {
            const float d = dist_box_point(boxes[b], point);
            need = !(d > reject || d > best[2]);                      // :172
        }
        if (!__syncthreads_or(need)) continue;
        const int lo = b * BOX, hi = min(P, (b + 1) * BOX);
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s_pts[i - lo] = sorted[i];
        __syncthreads();
        if (need) {
            for (int i = lo; i < hi; i++) {
                if (i == idx) continue;
                const float4 q = s_pts[i - lo];
                update3(point, make_float3(q.x, q.y, q.z), best);
            }
        }
        __syncthreads();
    }
    if (live) dists[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;       // :182
}

struct KnnView {
    uint32_t* mm; uint64_t* keys; float4* sorted; Box* boxes;
    static KnnView at(char* base, int P)
    {
        KnnView v; char* p = base; size_t n = (size_t)P;
        v.mm = carve<uint32_t>(p, 8);
        v.keys = carve<uint64_t>(p, n);
        v.sorted = carve<float4>(p, n);
        v.boxes = carve<Box>(p, (n + BOX - 1) / BOX);
        return v;
    }
    static size_t bytes(int P) { char* z = nullptr; KnnView v = at(z, P); return (size_t)((char*)v.boxes - z) + align_up(((size_t)P + BOX - 1) / BOX * sizeof(Box)); }
};

}  // anonymous namespace
}  // namespace moss

using namespace moss;

extern "C" size_t moss_knn_workspace_bytes(int P) { return KnnView::bytes(P > 0 ? P : 1); }

extern "C" int moss_knn_dist2(int P, const float* points, float* mean_dists, char* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0) return MOSS_ERR_INVALID_ARG;
    if (P == 0) return 0;
    if (!points || !mean_dists || !workspace || workspace_bytes < KnnView::bytes(P)) return MOSS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    KnnView v = KnnView::at(workspace, P);
    const int blocks = (P + 255) / 256;
    const int num_boxes = (P + BOX - 1) / BOX;
    if (hipMemsetAsync(v.mm, 0xff, 3 * sizeof(uint32_t), s) != hipSuccess) return MOSS_ERR_HIP;
    if (hipMemsetAsync(v.mm + 3, 0x00, 3 * sizeof(uint32_t), s) != hipSuccess) return MOSS_ERR_HIP;
    hipLaunchKernelGGL(bounds_kernel, dim3(blocks < 128 ? blocks : 128), dim3(256), 0, s, P, points, v.mm);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(256), 0, s, P, points, v.mm, v.keys);
    launch_sort_keys(v.keys, (uint32_t)P, s);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, s, P, points, v.keys, v.sorted);
    hipLaunchKernelGGL(box_kernel, dim3(num_boxes), dim3(256), 0, s, P, v.sorted, v.boxes);
    hipLaunchKernelGGL(mean_dist_kernel, dim3(blocks), dim3(256), 0, s, P, v.sorted, v.boxes, num_boxes, mean_dists);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
