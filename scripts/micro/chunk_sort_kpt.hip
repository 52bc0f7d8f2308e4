// chunk_sort_kpt.hip -- what a LARGER sort chunk costs: the bitonic network of csrc/binning.hip (one 64-bit key per thread in
// registers, partners inside a wave by DPP / permlane swaps, cross-wave steps through LDS) with KPT = 1, 2, 4 keys per thread, i.e.
// chunks of 1024, 2048, 4096 keys per 1024-thread workgroup.  The measurement behind round 4's review item 4 ("try 4 096-key chunks
// with 4 keys per thread in registers: the in-thread compare-exchanges add no LDS or permute steps, every tile of <= 4k entries is
// one chunk, and merge_gather loses its sibling-chunk loads and binary searches").
//
//   hipcc --offload-arch=gfx950 -O3 -o chunk_sort_kpt chunk_sort_kpt.hip && ./chunk_sort_kpt
//
// Element i of a chunk lives in thread i / KPT, register i % KPT: the steps with partner distance j < KPT are compare-exchanges
// between a thread's own registers, j / KPT < 64 goes through the wave, larger through LDS (double-buffered, one barrier per step).
// Reported per (KPT, number of chunks): kernel time (hipEvents around 20 launches), the workgroups' own cycles, sortedness against
// std::sort.  The bench frame (configs[2]) has 181 tiles with work: 102 of them hold 1 024 - 4 750 entries (192k of its 239k
// instances); its sort kernel today runs 340 chunks of <= 1 024 keys in 12.1 us, all resident at once.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define MOSS_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (ctrl), 0xf, 0xf, true))
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, uint32_t lane)
{
    if constexpr (J == 1) return MOSS_DPP(v, 0xB1);
    else if constexpr (J == 2) return MOSS_DPP(v, 0x4E);
    else if constexpr (J == 4) { const uint32_t m = MOSS_DPP(v, 0x141); return MOSS_DPP(m, 0x1B); }
    else if constexpr (J == 8) return MOSS_DPP(v, 0x128);
    else if constexpr (J == 16) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16u) ? r[0] : r[1]; }
    else { static_assert(J == 32, ""); auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32u) ? r[0] : r[1]; }
}

template <int KPT> struct Net {
    static constexpr int N = 1024 * KPT;
    // one step: partner distance J (in elements) inside the ascending/descending runs of length K
    template <int K, int J>
    static __device__ __forceinline__ void step(uint64_t (&key)[KPT], uint32_t tid, uint64_t (*s_buf)[N], int& p)
    {
        if constexpr (J < KPT) {                                   // both partners in this thread's registers
#pragma unroll
            for (int e = 0; e < KPT; e++) {
                if ((e & J) == 0) {
                    const uint32_t i = tid * KPT + e;
                    const bool asc = (i & (uint32_t)K) == 0u;
                    const uint64_t a = key[e], b = key[e | J];
                    const bool sw = (a > b) == asc;
                    key[e] = sw ? b : a; key[e | J] = sw ? a : b;
                }
            }
        } else if constexpr (J / KPT < 64) {                        // partner thread in the same wave
#pragma unroll
            for (int e = 0; e < KPT; e++) {
                const uint32_t lo = lane_xor<J / KPT>((uint32_t)key[e], tid), hi = lane_xor<J / KPT>((uint32_t)(key[e] >> 32), tid);
                const uint64_t other = ((uint64_t)hi << 32) | lo;
                const uint32_t i = tid * KPT + e;
                const bool take_min = ((i & (uint32_t)J) == 0u) == ((i & (uint32_t)K) == 0u);
                const bool lt = key[e] < other;
                key[e] = (lt == take_min) ? key[e] : other;
            }
        } else {                                                    // another wave: through LDS, one barrier per step
#pragma unroll
            for (int e = 0; e < KPT; e++) s_buf[p][tid * KPT + e] = key[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < KPT; e++) {
                const uint32_t i = tid * KPT + e;
                const uint64_t other = s_buf[p][i ^ (uint32_t)J];
                const bool take_min = ((i & (uint32_t)J) == 0u) == ((i & (uint32_t)K) == 0u);
                const bool lt = key[e] < other;
                key[e] = (lt == take_min) ? key[e] : other;
            }
            p ^= 1;
        }
        if constexpr (J > 1) step<K, J / 2>(key, tid, s_buf, p);
    }
    template <int K>
    static __device__ __forceinline__ void phases(uint64_t (&key)[KPT], uint32_t tid, uint32_t npad, uint64_t (*s_buf)[N], int& p)
    {
        if constexpr (K > 2) phases<K / 2>(key, tid, npad, s_buf, p);
        if ((uint32_t)K <= npad) step<K, K / 2>(key, tid, s_buf, p);
    }
};

template <int KPT>
__global__ void __launch_bounds__(1024) sort_chunks(uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, unsigned long long* __restrict__ cycles)
{
    constexpr int N = 1024 * KPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    uint64_t (*s_buf)[N] = reinterpret_cast<uint64_t (*)[N]>(s_raw);
    const uint32_t tid = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const uint32_t n = counts[blockIdx.x];
    uint64_t* gk = keys + (size_t)blockIdx.x * N;
    uint32_t npad = 64 * KPT;
    while (npad < n) npad <<= 1;
    uint64_t key[KPT];
#pragma unroll
    for (int e = 0; e < KPT; e++) { const uint32_t i = tid * KPT + e; key[e] = i < n ? gk[i] : ~0ull; }
    int p = 0;
    if (tid * KPT < npad) Net<KPT>::template phases<N>(key, tid, npad, s_buf, p);      // (whole waves: npad is a multiple of 64 KPT)
    else {
        for (uint32_t k = 128 * KPT; k <= npad; k <<= 1)
            for (uint32_t j = k >> 1; j >= 64u * KPT; j >>= 1) __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < KPT; e++) { const uint32_t i = tid * KPT + e; if (i < n) gk[i] = key[e]; }
    if (tid == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int KPT>
void run(int chunks, const std::vector<uint32_t>& sizes, const char* what)
{
    constexpr int N = 1024 * KPT;
    std::mt19937_64 rng(1234);
    std::vector<uint64_t> h((size_t)chunks * N);
    for (auto& v : h) v = ((rng() >> 34) << 32) | (uint32_t)(rng() % 100000);       // ~30 bits of depth, a Gaussian id: ties in the upper word happen
    std::vector<uint32_t> cnt(chunks);
    for (int c = 0; c < chunks; c++) cnt[c] = sizes[c % sizes.size()];
    uint64_t* d; uint32_t* dc; unsigned long long* dcy;
    hipMalloc(&d, h.size() * 8); hipMalloc(&dc, chunks * 4); hipMalloc(&dcy, chunks * 8);
    hipMemcpy(dc, cnt.data(), chunks * 4, hipMemcpyHostToDevice);
    const size_t lds = 2 * (size_t)N * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(sort_chunks<KPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(sort_chunks<KPT>, dim3(chunks), dim3(1024), lds, 0, d, dc, dcy);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 3) { best = std::min(best, ms); sum += ms; }
    }
    std::vector<uint64_t> out(h.size());
    std::vector<unsigned long long> cy(chunks);
    hipMemcpy(out.data(), d, out.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cy.data(), dcy, chunks * 8, hipMemcpyDeviceToHost);
    bool ok = true;
    size_t total = 0;
    for (int c = 0; c < chunks && ok; c++) {
        std::vector<uint64_t> ref(h.begin() + (size_t)c * N, h.begin() + (size_t)c * N + cnt[c]);
        std::sort(ref.begin(), ref.end());
        ok = std::equal(ref.begin(), ref.end(), out.begin() + (size_t)c * N);
        total += cnt[c];
    }
    std::sort(cy.begin(), cy.end());
    printf("KPT %d (chunks of <= %4d keys) %-34s %4d workgroups, %7zu keys: %6.2f us mean %6.2f us best; workgroup cycles median %6llu max %6llu; %s\n",
           KPT, N, what, chunks, total, 1e3f * sum / reps, 1e3f * best, cy[chunks / 2], cy[chunks - 1], ok ? "sorted = std::sort" : "WRONG");
    hipFree(d); hipFree(dc); hipFree(dcy);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    // (a) full chunks, the same 245 760 keys cut three ways
    run<1>(240, {1024}, "full chunks");
    run<2>(120, {2048}, "full chunks");
    run<4>(60, {4096}, "full chunks");
    // (b) the bench frame's tiles (list lengths by class: 26 x ~7, 14 x ~70, 18 x ~315, 22 x ~770, 56 x ~1630, 40 x ~2520, 5 x ~4540):
    //     today: ceil(n / 1024) chunks of <= 1024 per tile (340); with 4 096-key chunks: one chunk per tile (+ 5 second chunks)
    std::vector<uint32_t> tiles;
    for (int i = 0; i < 26; i++) tiles.push_back(7);
    for (int i = 0; i < 14; i++) tiles.push_back(70);
    for (int i = 0; i < 18; i++) tiles.push_back(315);
    for (int i = 0; i < 22; i++) tiles.push_back(770);
    for (int i = 0; i < 56; i++) tiles.push_back(1630);
    for (int i = 0; i < 40; i++) tiles.push_back(2520);
    for (int i = 0; i < 5; i++) tiles.push_back(4540);
    auto cut = [&](uint32_t cap) { std::vector<uint32_t> c; for (uint32_t n : tiles) for (uint32_t o = 0; o < n; o += cap) c.push_back(std::min(cap, n - o)); return c; };
    { auto c = cut(1024); run<1>((int)c.size(), c, "bench frame, chunks of 1024"); }
    { auto c = cut(2048); run<2>((int)c.size(), c, "bench frame, chunks of 2048"); }
    { auto c = cut(4096); run<4>((int)c.size(), c, "bench frame, chunks of 4096"); }
    return 0;
}
