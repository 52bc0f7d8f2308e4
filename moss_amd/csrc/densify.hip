// densify.hip -- the per-step densification bookkeeping and the KL test of MOSS's KL-guided densify.
//
// 1. Per-step statistics, one launch.  The reference does, every step until densify_until_iter (train_ZJU.py:171-174,
//    scene/gaussian_model.py:815-817):
//        max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);  vis = radii > 0
//        xyz_gradient_accum[vis] += norm(viewspace_points.grad[vis, :2], dim=-1, keepdim=True);  denom[vis] += 1
//    i.e. three boolean-mask gathers / scatters and a norm: ~12 launches and a mask->index conversion that synchronises with the
//    host.  Here: 24 B read + 12 B read-modify-write per Gaussian, no host involvement, graph-capturable.
// 2. kl_div(mu_0, q_0, s_0, mu_1, q_1, s_1) (scene/gaussian_model.py:773-813) between every Gaussian and its nearest neighbour,
//    with the neighbour gather (:586-597 / :759-770) fused in.  The reference builds (P,3,3) matrices with torch ops and takes the
//    trace in a PYTHON loop over P (:803-804); per Gaussian this is ~150 flops on 2 x 40 B of input.
//        KL = 0.5 (tr(S1^-1 S0) + d^T S1^-1 d + ln prod((s1/s0)^2) - 3),  S = R diag(s^2) R^T,  R = build_rotation(q / |q|)
//    computed as  tr = sum_ij (M_ij s0_j / s1_i)^2  with  M = R1^T R0,  and  d^T S1^-1 d = sum_i ((R1^T d)_i / s1_i)^2,
//    which needs no 3x3 products beyond M and never forms a covariance.
#include "common.h"

namespace moss {
namespace {

__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ grad2d, int grad_stride,
                     float* __restrict__ grad_accum, float* __restrict__ denom, float* __restrict__ max_radii)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = grad2d[(size_t)i * grad_stride], gy = grad2d[(size_t)i * grad_stride + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
    if (max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
}

__device__ __forceinline__ void rotation_of(const float* __restrict__ q4, float R[3][3])     // utils/general_utils.py:79-100
{
    const float a = q4[0], b = q4[1], c = q4[2], d = q4[3];
    const float inv = 1.0f / sqrtf(a * a + b * b + c * c + d * d);
    const float r = a * inv, x = b * inv, y = c * inv, z = d * inv;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}

__global__ void __launch_bounds__(256)
neighbour_kl_kernel(int P, int Nsrc, const float* __restrict__ xyz, const float* __restrict__ rotation, const float* __restrict__ scaling,
                    const long long* __restrict__ pair_idx, float* __restrict__ kl_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const long long i0 = pair_idx[2 * (size_t)i], i1 = pair_idx[2 * (size_t)i + 1];
    if (i0 < 0 || i0 >= Nsrc || i1 < 0 || i1 >= Nsrc) { kl_out[i] = __builtin_nanf(""); return; }     // never read out of range
    float R0[3][3], R1[3][3];
    rotation_of(rotation + 4 * i0, R0);
    rotation_of(rotation + 4 * i1, R1);
    float s0[3], s1i[3], dm[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        s0[k] = scaling[3 * i0 + k];
        s1i[k] = 1.0f / scaling[3 * i1 + k];
        dm[k] = xyz[3 * i1 + k] - xyz[3 * i0 + k];                           // mu_1 - mu_0 (:789)
    }
    float tr = 0.0f, maha = 0.0f, logdet = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float proj = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) proj += R1[k][a] * dm[k];                // (R1^T d)_a
        proj *= s1i[a];
        maha += proj * proj;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) m += R1[k][a] * R0[k][b];            // (R1^T R0)_ab
            m *= s0[b] * s1i[a];
            tr += m * m;
        }
        const float ratio = s1i[a] * s0[a];                                  // s0/s1
        logdet -= logf(ratio * ratio);                                       // ln prod (s1/s0)^2 (:811), term by term
    }
    kl_out[i] = 0.5f * (tr + maha + logdet - 3.0f);
}

}  // namespace
}  // namespace moss

extern "C" int moss_densify_stats(int P, const int* radii, const float* viewspace_grad, int grad_stride,
                                  float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream)
{
    if (P < 0 || grad_stride < 2) return MOSS_ERR_INVALID_ARG;
    if (P == 0) return 0;
    if (!radii || !viewspace_grad || !xyz_gradient_accum || !denom) return MOSS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(moss::densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, radii, viewspace_grad,
                       grad_stride, xyz_gradient_accum, denom, max_radii2D);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

extern "C" int moss_neighbour_kl(int P, int Nsrc, const float* xyz, const float* rotation, const float* scaling,
                                 const long long* pair_idx, float* kl_out, void* stream)
{
    if (P < 0 || Nsrc < 0) return MOSS_ERR_INVALID_ARG;
    if (P == 0) return 0;
    if (Nsrc < 1 || !xyz || !rotation || !scaling || !pair_idx || !kl_out) return MOSS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(moss::neighbour_kl_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, Nsrc, xyz, rotation,
                       scaling, pair_idx, kl_out);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
