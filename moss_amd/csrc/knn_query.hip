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
#pragma unroll
                for (int k = 0; k < K; k++) {                              // insertion; on equal distance the kept (lower) index stays in front
                    if (cd < bd[k]) { const float td = bd[k]; const int ti = bi[k]; bd[k] = cd; bi[k] = ci; cd = td; ci = ti; }
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
