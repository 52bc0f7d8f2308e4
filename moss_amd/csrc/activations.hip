// activations.hip -- the Gaussian parameter activations of MOSS's GaussianModel getters as ONE kernel each way
// (the caller side of the rasterizer boundary).
//
// The reference builds the rasterizer's inputs with five torch ops per render (scene/gaussian_model.py:46-53,134-166):
//     get_xyz      = _xyz                                      identity
//     get_features = cat(_features_dc, _features_rest, dim=1)   (P,1,3)+(P,K-1,3) -> (P,K,3)
//     get_opacity  = sigmoid(_opacity)
//     get_scaling  = exp(_scaling)
//     get_rotation = normalize(_rotation)                      x / max(|x|_2, 1e-12)
// and autograd runs about a dozen more small kernels for their backward plus one accumulate per parameter.  On MI355X a
// minimal kernel occupies ~4-5 us of the stream even inside a hipGraph, so ~25 such launches were a quarter of the
// 0.64 ms training step.  Here the forward is one launch and the backward is one launch that writes the gradients of the
// raw parameters straight to their final destination (the flat gradient bucket).
#include "common.h"

namespace moss {
namespace {

constexpr float NORM_EPS = 1e-12f;          // torch.nn.functional.normalize default

// Work is laid out as consecutive float4 "cells": [features | xyz | opacity | scaling | rotation]; each thread walks cells with
// a grid stride, so all global accesses are 16-byte and coalesced.  Section sizes in floats are padded up to cells; the tail
// cell of a section is handled element-wise.
struct Sections { long long feat, xyz, opa, scl, rot; };     // cumulative END of each section, in cells

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// quaternions sit at a multiple of 4 floats inside their tensor, but a tensor that is a view into a flat bucket need not start
// on a 16-byte boundary: four scalar accesses (the compiler merges them when it can prove alignment)
__device__ __forceinline__ float4 load4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void store4(float* p, float4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }

__global__ void __launch_bounds__(256)
activate_forward_kernel(int P, int K, Sections sec,
                        const float* __restrict__ xyz, const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                        const float* __restrict__ opacity, const float* __restrict__ scaling, const float* __restrict__ rotation,
                        float* __restrict__ o_xyz, float* __restrict__ o_feat, float* __restrict__ o_opa, float* __restrict__ o_scl,
                        float* __restrict__ o_rot)
{
    const long long n_feat = (long long)P * K * 3, n3 = (long long)P * 3;
    const int row = K * 3;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < sec.rot; c += (long long)gridDim.x * blockDim.x) {
        if (c < sec.feat) {
            const long long e0 = c * 4;
            long long i = e0 / row; int j = (int)(e0 - i * row);             // one division per cell, then step
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long e = e0 + k;
                if (e < n_feat) o_feat[e] = j < 3 ? f_dc[i * 3 + j] : f_rest[i * (row - 3) + (j - 3)];
                if (++j == row) { j = 0; i++; }
            }
        } else if (c < sec.xyz) {
            const long long e0 = (c - sec.feat) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < n3) o_xyz[e0 + k] = xyz[e0 + k];
        } else if (c < sec.opa) {
            const long long e0 = (c - sec.xyz) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < P) o_opa[e0 + k] = sigmoidf_(opacity[e0 + k]);
        } else if (c < sec.scl) {
            const long long e0 = (c - sec.opa) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < n3) o_scl[e0 + k] = expf(scaling[e0 + k]);
        } else {
            const long long i = c - sec.scl;                 // one quaternion per cell
            const float4 q = load4(rotation + 4 * i);
            const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            const float inv = 1.0f / fmaxf(n, NORM_EPS);
            store4(o_rot + 4 * i, make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv));
        }
    }
}

// Backward.  A null incoming gradient means "that output was not used": its parameter gets zeros.  Every element of every
// parameter gradient is written exactly once, so the destination needs no zero fill.
__global__ void __launch_bounds__(256)
activate_backward_kernel(int P, int K, Sections sec,
                         const float* __restrict__ rotation, const float* __restrict__ o_opa, const float* __restrict__ o_scl,
                         const float* __restrict__ g_xyz, const float* __restrict__ g_feat, const float* __restrict__ g_opa,
                         const float* __restrict__ g_scl, const float* __restrict__ g_rot,
                         float* __restrict__ d_xyz, float* __restrict__ d_dc, float* __restrict__ d_rest, float* __restrict__ d_opacity,
                         float* __restrict__ d_scaling, float* __restrict__ d_rotation)
{
    const long long n_feat = (long long)P * K * 3, n3 = (long long)P * 3;
    const int row = K * 3;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < sec.rot; c += (long long)gridDim.x * blockDim.x) {
        if (c < sec.feat) {
            const long long e0 = c * 4;
            long long i = e0 / row; int j = (int)(e0 - i * row);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long e = e0 + k;
                if (e < n_feat) {
                    const float v = g_feat ? g_feat[e] : 0.0f;
                    if (j < 3) d_dc[i * 3 + j] = v; else d_rest[i * (row - 3) + (j - 3)] = v;
                }
                if (++j == row) { j = 0; i++; }
            }
        } else if (c < sec.xyz) {
            const long long e0 = (c - sec.feat) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < n3) d_xyz[e0 + k] = g_xyz ? g_xyz[e0 + k] : 0.0f;
        } else if (c < sec.opa) {
            const long long e0 = (c - sec.xyz) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < P) {
                const float s = o_opa[e0 + k];
                d_opacity[e0 + k] = g_opa ? g_opa[e0 + k] * ((1.0f - s) * s) : 0.0f;       // sigmoid_backward: g * (1 - y) * y
            }
        } else if (c < sec.scl) {
            const long long e0 = (c - sec.opa) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) if (e0 + k < n3) d_scaling[e0 + k] = g_scl ? g_scl[e0 + k] * o_scl[e0 + k] : 0.0f;
        } else {
            const long long i = c - sec.scl;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_rot) {
                // y = q / d with d = max(|q|, eps).  |q| >= eps: dq = (g - y (y.g)) / d;  |q| < eps: d is constant, dq = g / d.
                const float4 q = load4(rotation + 4 * i);
                const float4 g = load4(g_rot + 4 * i);
                const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
                const float inv = 1.0f / fmaxf(n, NORM_EPS);
                if (n >= NORM_EPS) {
                    const float4 y = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
                    const float yg = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
                    r = make_float4((g.x - y.x * yg) * inv, (g.y - y.y * yg) * inv, (g.z - y.z * yg) * inv, (g.w - y.w * yg) * inv);
                } else {
                    r = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
                }
            }
            store4(d_rotation + 4 * i, r);
        }
    }
}

Sections make_sections(int P, int K)
{
    auto cells = [](long long floats) { return (floats + 3) / 4; };
    Sections s;
    s.feat = cells((long long)P * K * 3);
    s.xyz = s.feat + cells((long long)P * 3);
    s.opa = s.xyz + cells(P);
    s.scl = s.opa + cells((long long)P * 3);
    s.rot = s.scl + P;
    return s;
}

unsigned grid_for(long long cells)
{
    long long b = (cells + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace
}  // namespace moss

extern "C" int moss_gaussian_activate_forward(int P, int K, const float* xyz, const float* features_dc, const float* features_rest,
                                              const float* opacity, const float* scaling, const float* rotation,
                                              float* out_xyz, float* out_features, float* out_opacity, float* out_scaling,
                                              float* out_rotation, void* stream)
{
    if (P < 0 || K < 0) return MOSS_ERR_INVALID_ARG;
    if (P == 0) return 0;
    if (!xyz || (K > 0 && !features_dc) || (K > 1 && !features_rest) || !opacity || !scaling || !rotation || !out_xyz || (K > 0 && !out_features) ||
        !out_opacity || !out_scaling || !out_rotation)
        return MOSS_ERR_INVALID_ARG;
    const moss::Sections sec = moss::make_sections(P, K);
    hipLaunchKernelGGL(moss::activate_forward_kernel, dim3(moss::grid_for(sec.rot)), dim3(256), 0, (hipStream_t)stream, P, K, sec,
                       xyz, features_dc, features_rest, opacity, scaling, rotation, out_xyz, out_features, out_opacity, out_scaling,
                       out_rotation);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

extern "C" int moss_gaussian_activate_backward(int P, int K, const float* rotation, const float* out_opacity, const float* out_scaling,
                                               const float* g_xyz, const float* g_features, const float* g_opacity,
                                               const float* g_scaling, const float* g_rotation,
                                               float* d_xyz, float* d_features_dc, float* d_features_rest, float* d_opacity,
                                               float* d_scaling, float* d_rotation, void* stream)
{
    if (P < 0 || K < 0) return MOSS_ERR_INVALID_ARG;
    if (P == 0) return 0;
    if (!rotation || !out_opacity || !out_scaling || !d_xyz || (K > 0 && !d_features_dc) || (K > 1 && !d_features_rest) || !d_opacity ||
        !d_scaling || !d_rotation)
        return MOSS_ERR_INVALID_ARG;
    const moss::Sections sec = moss::make_sections(P, K);
    hipLaunchKernelGGL(moss::activate_backward_kernel, dim3(moss::grid_for(sec.rot)), dim3(256), 0, (hipStream_t)stream, P, K, sec,
                       rotation, out_opacity, out_scaling, g_xyz, g_features, g_opacity, g_scaling, g_rotation,
                       d_xyz, d_features_dc, d_features_rest, d_opacity, d_scaling, d_rotation);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}
