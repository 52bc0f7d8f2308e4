// Micro-benchmark: latency of the forward-blend trip for ONE wave alone on its SIMD (what the kernel's longest item pays per trip,
// profiles/r03_notes.md finding 27), in variants: V0 the trip as in blend.hip; V1 the next trip's alpha evaluated in front of the
// current trip's chain (software pipelining by hand: the evaluation does not depend on T); V2 as V1 with two trips per loop pass.
// hipcc --offload-arch=gfx950 -O3 -o trip_pipe.bin trip_pipe.hip && ./trip_pipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DPP_MOV(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true))
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int FCMP_OGE = 3, FCMP_OLT = 4, FCMP_OLE = 5, ICMP_NE = 33;
struct St { float T, T_stop; v2f Crg, CbD; float weight, lc; unsigned long long live_m; };
struct Ring { float4 a[256], b[256], c[256]; };
struct Ev { float ao; unsigned long long m; };
__device__ __forceinline__ Ev eval(const float4 a, const float4 b, v2f pix)
{
    const v2f d = v2f{a.x, a.y} - pix;
    const v2f bc = v2f{b.x, b.y} * d;
    const v2f t = bc * v2f{d.y, d.y};
    const float q = __fmaf_rn(b.z * d.x, d.x, t.y);
    const float power = __fmaf_rn(-0.5f, q, -t.x);
    Ev e;
    e.ao = fminf(0.99f, b.w * __expf(power));
    e.m = __builtin_amdgcn_fcmpf(power, 0.0f, FCMP_OLE) & __builtin_amdgcn_fcmpf(e.ao, 1.0f / 255.0f, FCMP_OGE);
    return e;
}
__device__ __forceinline__ void chain(St& s, const Ev e, const float4 a, const float4 c, int slot, int gbase, uint32_t below_mask)
{
    const unsigned long long m = e.m & s.live_m;
    const float al = __builtin_amdgcn_inverse_ballot_w64(m) ? e.ao : 0.0f;
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
    s.Crg = __builtin_elementwise_fma(v2f{c.x, c.y}, v2f{wgt, wgt}, s.Crg);
    s.CbD = __builtin_elementwise_fma(v2f{c.z, c.w}, v2f{wgt, wgt}, s.CbD);
    s.weight += wgt;
    s.lc = __builtin_amdgcn_inverse_ballot_w64(m & ~dead) ? a.w : s.lc;
    s.T = DPP_MOV(X, 0xFF);
    s.live_m &= ~__builtin_amdgcn_uicmp(q, 0u, ICMP_NE);
}
template <int V>
__global__ void __launch_bounds__(1024) bench(unsigned long long* out, float* sink, int ntrips)
{
    extern __shared__ Ring rings[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), slot = lane & 3, pl = lane >> 2, gbase = lane & ~3;
    const uint32_t below_mask = (1u << slot) - 1u;
    Ring* R = &rings[wv];
    for (int i = lane; i < 256; i += 64) {
        R->a[i] = make_float4(1.5f + 0.01f * (i & 7), 1.5f - 0.01f * (i & 3), 0.f, (float)(i + 1));
        R->b[i] = make_float4(0.001f, 0.05f, 0.05f, 0.03f + 0.0001f * i);       // {B, C, A, opacity}
        R->c[i] = make_float4(0.3f, 0.5f, 0.7f, 2.0f);
    }
    __syncthreads();
    const v2f pix = v2f{(float)(pl & 3), (float)(pl >> 2)};
    St s = {1.f, -1.f, v2f{0, 0}, v2f{0, 0}, 0, 0, ~0ull};
    auto ld = [&](int t, float4& a, float4& b, float4& c) { const int li = ((4 * t) & 255) + slot; a = R->a[li]; b = R->b[li]; c = R->c[li]; };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 0) {
        float4 a, b, c; ld(0, a, b, c);
        for (int t = 0; t < ntrips; t++) {
            float4 a1, b1, c1; ld(t + 1, a1, b1, c1);
            const Ev e = eval(a, b, pix);
            chain(s, e, a, c, slot, gbase, below_mask);
            if (s.live_m == 0ull) s.live_m = ~0ull;
            a = a1; b = b1; c = c1;
        }
    } else if (V == 1) {
        float4 a, b, c, a1, b1, c1; ld(0, a, b, c); ld(1, a1, b1, c1);
        Ev e = eval(a, b, pix);
        for (int t = 0; t < ntrips; t++) {
            float4 a2, b2, c2; ld(t + 2, a2, b2, c2);
            const Ev e1 = eval(a1, b1, pix);                 // the NEXT trip's alpha: independent of the chain below
            chain(s, e, a, c, slot, gbase, below_mask);
            if (s.live_m == 0ull) s.live_m = ~0ull;
            e = e1; a = a1; c = c1; a1 = a2; b1 = b2; c1 = c2;
        }
    } else {
        float4 a, b, c, a1, b1, c1; ld(0, a, b, c); ld(1, a1, b1, c1);
        Ev e = eval(a, b, pix), e1 = eval(a1, b1, pix);
        for (int t = 0; t < ntrips; t += 2) {
            float4 a2, b2, c2, a3, b3, c3; ld(t + 2, a2, b2, c2); ld(t + 3, a3, b3, c3);
            const Ev e2 = eval(a2, b2, pix);
            chain(s, e, a, c, slot, gbase, below_mask);
            const Ev e3 = eval(a3, b3, pix);
            chain(s, e1, a1, c1, slot, gbase, below_mask);
            if (s.live_m == 0ull) s.live_m = ~0ull;
            e = e2; e1 = e3; a = a2; c = c2; a1 = a3; c1 = c3;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wv] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s.T + s.Crg.x + s.Crg.y + s.CbD.x + s.CbD.y + s.weight + s.lc + s.T_stop;
}
template <int V>
double run(int waves_per_block, int ntrips)
{
    const int blocks = 256;
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, blocks * 16 * 8); (void)hipMalloc(&sink, blocks * 1024 * 4);
    (void)hipMemset(out, 0, blocks * 16 * 8);
    const size_t lds = sizeof(Ring) * waves_per_block;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bench<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(bench<V>, dim3(blocks), dim3(64 * waves_per_block), lds, 0, out, sink, ntrips);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < waves_per_block; w++) { sum += (double)h[b * 16 + w]; n++; }
    (void)hipFree(out); (void)hipFree(sink);
    return sum / n / ntrips;
}
int main()
{
    const int nt = 2000;
    printf("cycles per trip of a wave (s_memtime)            alone on its SIMD | two waves per SIMD\n");
    printf("V0 the trip of blend.hip                          %7.1f | %7.1f\n", run<0>(4, nt), run<0>(8, nt));
    printf("V1 next trip's alpha in front of the chain        %7.1f | %7.1f\n", run<1>(4, nt), run<1>(8, nt));
    printf("V2 as V1, two trips per loop pass                 %7.1f | %7.1f\n", run<2>(4, nt), run<2>(8, nt));
    return 0;
}
