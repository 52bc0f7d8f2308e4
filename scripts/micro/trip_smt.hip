// Micro-benchmark behind profiles/r03_notes.md finding 13: what a forward-blend trip costs a wave when (A) it runs alone on its SIMD,
// (B) two waves share the SIMD (the kernel's structure today), (C) ONE wave runs two independent trip chains interleaved (a "dual-item"
// blender).  The trip body is the one of blend.hip (lane = (pixel, slot), records from LDS), with synthetic records; no early-out.
// hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-sched-strategy=max-ilp] -o trip_smt.bin trip_smt.hip && ./trip_smt.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DPP_MOV(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true))
struct St { float T, T_stop, Cr, Cg, Cb, weight, Dacc, lc; unsigned long long live_m; };
struct Ring { float4 a[256], b[256], c[256]; };
__device__ __forceinline__ void trip(St& s, const float4 a, const float4 b, const float4 c, float pixx, float pixy, int slot, int gbase, uint32_t below_mask)
{
    constexpr int FCMP_OGE = 3, FCMP_OLT = 4, FCMP_OLE = 5, ICMP_NE = 33;
    const float dx = a.x - pixx, dy = a.y - pixy;
    const float qv = __fmaf_rn(b.x * dx, dx, (b.z * dy) * dy);
    const float power = __fmaf_rn(-0.5f, qv, -(b.y * dx) * dy);
    const float ao = fminf(0.99f, b.w * __expf(power));
    const unsigned long long m = __builtin_amdgcn_fcmpf(power, 0.0f, FCMP_OLE) & __builtin_amdgcn_fcmpf(ao, 1.0f / 255.0f, FCMP_OGE) & s.live_m;
    const float al = __builtin_amdgcn_inverse_ballot_w64(m) ? ao : 0.0f;
    const float fm = 1.0f - al;
    float X = s.T * fm, Y;
    Y = DPP_MOV(X, 0x90); X = slot >= 1 ? Y * fm : X;
    Y = DPP_MOV(X, 0x90); X = slot >= 2 ? Y * fm : X;
    Y = DPP_MOV(X, 0x90); X = slot >= 3 ? Y * fm : X;
    Y = DPP_MOV(X, 0x90);
    const float Tb = slot == 0 ? s.T : Y;
    const unsigned long long sb = __builtin_amdgcn_fcmpf(X, 0.0001f, FCMP_OLT) & m;
    const uint32_t q = (uint32_t)(sb >> gbase) & 15u;
    const unsigned long long mb = __builtin_amdgcn_uicmp(q & below_mask, 0u, ICMP_NE);
    const unsigned long long dead = sb | mb;
    const float wgt = __builtin_amdgcn_inverse_ballot_w64(dead) ? 0.0f : al * Tb;
    s.T_stop = __builtin_amdgcn_inverse_ballot_w64(sb & ~mb) ? Tb : s.T_stop;
    s.Cr = __fmaf_rn(c.x, wgt, s.Cr); s.Cg = __fmaf_rn(c.y, wgt, s.Cg); s.Cb = __fmaf_rn(c.z, wgt, s.Cb);
    s.weight += wgt; s.Dacc = __fmaf_rn(c.w, wgt, s.Dacc);
    s.lc = __builtin_amdgcn_inverse_ballot_w64(m & ~dead) ? a.w : s.lc;
    s.T = DPP_MOV(X, 0xFF);
    s.live_m &= ~__builtin_amdgcn_uicmp(q, 0u, ICMP_NE);
}
template <int DUAL>
__global__ void __launch_bounds__(1024) bench(unsigned long long* out, float* sink, int ntrips)
{
    extern __shared__ Ring rings[];                       // one ring per chain
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, slot = lane & 3, pl = lane >> 2, gbase = lane & ~3;
    const uint32_t below_mask = (1u << slot) - 1u;
    Ring* R0 = &rings[wv * (DUAL ? 2 : 1)];
    Ring* R1 = R0 + (DUAL ? 1 : 0);
    for (int i = lane; i < 256; i += 64) {
        // faint Gaussians around the block: alpha ~ 0.01-0.05, so no pixel ever finishes (the long items of the real kernel)
        R0->a[i] = make_float4(1.5f + 0.01f * (i & 7), 1.5f - 0.01f * (i & 3), 0.f, (float)(i + 1));
        R0->b[i] = make_float4(0.05f, 0.001f, 0.05f, 0.03f + 0.0001f * i);
        R0->c[i] = make_float4(0.3f, 0.5f, 0.7f, 2.0f);
        if (DUAL) { R1->a[i] = R0->a[i]; R1->b[i] = R0->b[i]; R1->c[i] = R0->c[i]; }
    }
    __syncthreads();
    const float pixx = (float)(pl & 3), pixy = (float)(pl >> 2);
    St s0 = {1.f, -1.f, 0, 0, 0, 0, 0, 0, ~0ull}, s1 = s0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntrips; t++) {
        const int li = ((4 * t) & 255) + slot;
        const float4 a0 = R0->a[li], b0 = R0->b[li], c0 = R0->c[li];
        if (DUAL) {
            const float4 a1 = R1->a[li], b1 = R1->b[li], c1 = R1->c[li];
            trip(s0, a0, b0, c0, pixx, pixy, slot, gbase, below_mask);
            trip(s1, a1, b1, c1, pixx, pixy, slot, gbase, below_mask);
        } else trip(s0, a0, b0, c0, pixx, pixy, slot, gbase, below_mask);
        if (s0.live_m == 0ull) s0.live_m = ~0ull;         // (never: keeps the exit test of the real loop in the chain)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wv] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s0.T + s0.Cr + s0.Cg + s0.Cb + s0.weight + s0.Dacc + s0.lc + s0.T_stop + s1.T + s1.Cr + s1.lc + s1.T_stop + s1.Dacc + s1.weight + s1.Cg + s1.Cb;
}
template <int DUAL>
double run(int waves_per_block, int ntrips)
{
    const int blocks = 256;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 16 * 8); hipMalloc(&sink, blocks * 1024 * 4);
    hipMemset(out, 0, blocks * 16 * 8);
    const size_t lds = sizeof(Ring) * waves_per_block * (DUAL ? 2 : 1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench<DUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(bench<DUAL>, dim3(blocks), dim3(64 * waves_per_block), lds, 0, out, sink, ntrips);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < waves_per_block; w++) { sum += (double)h[b * 16 + w]; n++; }
    hipFree(out); hipFree(sink);
    return sum / n / ntrips;
}
int main()
{
    const int nt = 2000;
    printf("s_memtime ticks per trip PER WAVE (one chain = one item's trips; 100 MHz... ticks are core-clock/?? -- compare ratios)\n");
    printf("A  one chain per wave, 1 wave  per SIMD : %.1f per trip\n", run<0>(4, nt));
    printf("B  one chain per wave, 2 waves per SIMD : %.1f per trip of each wave  (both progress: %.1f per trip aggregate)\n", run<0>(8, nt), run<0>(8, nt) / 2);
    printf("B4 one chain per wave, 4 waves per SIMD : %.1f per trip of each wave  (%.1f aggregate)\n", run<0>(16, nt), run<0>(16, nt) / 4);
    printf("C  TWO chains per wave, 1 wave per SIMD : %.1f per double trip (%.1f per trip aggregate)\n", run<1>(4, nt), run<1>(4, nt) / 2);
    printf("C2 TWO chains per wave, 2 waves per SIMD: %.1f per double trip of each wave (%.1f aggregate)\n", run<1>(8, nt), run<1>(8, nt) / 4);
    return 0;
}
