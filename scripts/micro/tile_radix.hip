// tile_radix.hip -- micro-benchmark for round 3's review item 2: a per-tile LDS RADIX sort on the 32-bit depth keys (the tile is the
// bucket, ids ride as values, LSD => stable), one 1024-thread workgroup per tile, against what the product does today (1024-key
// bitonic chunks in registers + a rank merge: binning.hip).  Measures ONLY the sort of one tile per workgroup, every CU busy with a
// tile of the same size: the number a whole-tile design would have to beat per tile is the chunk path's
//   chunk_sort 11.6 us + the search half of merge_gather (10.4k of its 18k cycles per workgroup)   for the frame's ~340 chunks.
//
//   hipcc --offload-arch=gfx950 -O3 -o tile_radix tile_radix.hip && ./tile_radix
//
// The sort: keys (depth bits) and values (ids) in LDS, ping-pong; 8-bit digits, only the bytes on which the tile's keys differ (a
// body's depths sit in one binade: the top byte and usually half of the next are constant); per pass every wave ranks its contiguous
// slice of the keys with a ballot multi-split (8 ballots per 64 keys: stable inside the wave, waves in slice order => stable
// overall), a 16 x 256 table of per-wave digit counts is scanned by 256 threads, and the keys are scattered.  Equal depths are put
// in id order by a last pass over runs of equal keys (the reference's radix sort is stable on (tile | depth) with ids ascending on
// entry: rasterizer_impl.cu:305-310).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

constexpr int NT = 1024, NW = NT / 64, CAP = 6144;

struct Lds {
    uint32_t k[2][CAP], v[2][CAP];
    uint32_t hist[NW][256];
    uint32_t wsum[4], kor, kand, ties;
};

__global__ void __launch_bounds__(NT) tile_radix_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int n,
                                                        unsigned long long* __restrict__ cycles)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    Lds& S = *reinterpret_cast<Lds*>(s_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t* src = in + (size_t)blockIdx.x * n;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { S.kor = 0u; S.kand = 0xffffffffu; S.ties = 0u; }
    __syncthreads();
    uint32_t my_or = 0u, my_and = 0xffffffffu;
    for (int i = tid; i < n; i += NT) {
        const uint64_t kv = src[i];
        const uint32_t key = (uint32_t)(kv >> 32);
        S.k[0][i] = key; S.v[0][i] = (uint32_t)kv;
        my_or |= key; my_and &= key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { my_or |= __shfl_xor(my_or, d); my_and &= __shfl_xor(my_and, d); }
    if (lane == 0) { atomicOr(&S.kor, my_or); atomicAnd(&S.kand, my_and); }
    __syncthreads();
    const uint32_t varying = S.kor ^ S.kand;
    // each wave ranks a contiguous slice: ceil(n / NW) rounded up to whole rounds of 64
    const int slice = ((n + NW - 1) / NW + 63) & ~63;
    const int s0 = w * slice, s1 = min(n, s0 + slice);
    int cur = 0;
    for (int byte = 0; byte < 4; byte++) {
        if (((varying >> (8 * byte)) & 255u) == 0u) continue;            // every key agrees on this digit
        const int shift = 8 * byte;
        for (int j = lane; j < 256; j += 64) S.hist[w][j] = 0u;          // own row: no barrier needed (the scan's readers are behind one)
        uint32_t loc[CAP / NT / 1 + 1];
        uint32_t dig[CAP / NT / 1 + 1];
        int r = 0;
        for (int base = s0; base < s1; base += 64, r++) {
            const int i = base + lane;
            const bool valid = i < s1;
            const uint32_t d = valid ? (S.k[cur][i] >> shift) & 255u : 0u;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                same &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
            const uint32_t prev = S.hist[w][d];
            __builtin_amdgcn_wave_barrier();
            if (valid && below == 0u) S.hist[w][d] = prev + (uint32_t)__popcll(same);
            __builtin_amdgcn_wave_barrier();
            loc[r] = prev + below; dig[r] = d;
        }
        __syncthreads();
        // scan: thread d < 256 turns column d of the table into per-wave bases, then the digit totals are scanned over the digits
        uint32_t total = 0u;
        if (tid < 256) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { const uint32_t c = S.hist[ww][tid]; S.hist[ww][tid] = total; total += c; }
        }
        uint32_t incl = total;
        if (tid < 256) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (lane == 63) S.wsum[w] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t base = incl - total;
            for (int ww = 0; ww < w; ww++) base += S.wsum[ww];
#pragma unroll
            for (int ww = 0; ww < NW; ww++) S.hist[ww][tid] += base;
        }
        __syncthreads();
        r = 0;
        for (int base = s0; base < s1; base += 64, r++) {
            const int i = base + lane;
            if (i < s1) {
                const uint32_t dst = S.hist[w][dig[r]] + loc[r];
                S.k[cur ^ 1][dst] = S.k[cur][i]; S.v[cur ^ 1][dst] = S.v[cur][i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // equal depths in id order: an element of a run of equal keys goes to (run start) + (number of smaller ids in the run)
    bool tie = false;
    for (int i = tid; i < n; i += NT) {
        const uint32_t key = S.k[cur][i];
        tie = tie || (i > 0 && S.k[cur][i - 1] == key) || (i + 1 < n && S.k[cur][i + 1] == key);
    }
    if (__ballot(tie) != 0ull && lane == 0) S.ties = 1u;
    __syncthreads();
    if (S.ties) {
        for (int i = tid; i < n; i += NT) {
            const uint32_t key = S.k[cur][i], id = S.v[cur][i];
            int a = i, b = i + 1;
            while (a > 0 && S.k[cur][a - 1] == key) a--;
            while (b < n && S.k[cur][b] == key) b++;
            int smaller = 0;
            for (int j = a; j < b; j++) smaller += S.v[cur][j] < id ? 1 : 0;
            S.k[cur ^ 1][a + smaller] = key; S.v[cur ^ 1][a + smaller] = id;
        }
        __syncthreads();
        cur ^= 1;
    }
    uint64_t* dst = out + (size_t)blockIdx.x * n;
    for (int i = tid; i < n; i += NT) dst[i] = ((uint64_t)S.k[cur][i] << 32) | S.v[cur][i];
    if (tid == 0 && cycles) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

int main()
{
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int tiles = prop.multiProcessorCount;              // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_radix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    printf("device %s, %d CUs, LDS per workgroup %zu B\n", prop.name, tiles, sizeof(Lds));
    for (int n : {256, 1024, 2048, 3072, 4096, 5120, 6144}) {
        for (int ties : {0, 1}) {
            std::mt19937 rng(17 + n + ties);
            std::vector<uint64_t> h((size_t)tiles * n);
            for (int t = 0; t < tiles; t++) {
                std::vector<uint32_t> ids(n);
                for (int i = 0; i < n; i++) ids[i] = (uint32_t)(7 * t + i);        // unique inside a tile, like Gaussian ids
                std::shuffle(ids.begin(), ids.end(), rng);
                for (int i = 0; i < n; i++) {
                    // a body 3 m away, 0.5 m deep: depths in one binade; with `ties`, a tenth of the entries share their depth with another one
                    float depth = 2.75f + 0.5f * (float)(rng() % 1000003) / 1000003.0f;
                    if (ties && i > 0 && rng() % 10 == 0) { uint32_t prev; memcpy(&prev, reinterpret_cast<char*>(&h[(size_t)t * n + i - 1]) + 4, 4); memcpy(&depth, &prev, 4); }
                    uint32_t bits; memcpy(&bits, &depth, 4);
                    h[(size_t)t * n + i] = ((uint64_t)bits << 32) | ids[i];
                }
            }
            uint64_t *din, *dout; unsigned long long* dcyc;
            hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dcyc, tiles * 8);
            hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int it = 0; it < 3; it++) hipLaunchKernelGGL(tile_radix_kernel, dim3(tiles), dim3(NT), sizeof(Lds), 0, din, dout, n, dcyc);
            hipEventRecord(a);
            const int reps = 20;
            for (int it = 0; it < reps; it++) hipLaunchKernelGGL(tile_radix_kernel, dim3(tiles), dim3(NT), sizeof(Lds), 0, din, dout, n, dcyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            std::vector<uint64_t> got(h.size()); std::vector<unsigned long long> cyc(tiles);
            hipMemcpy(got.data(), dout, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), dcyc, tiles * 8, hipMemcpyDeviceToHost);
            bool ok = true;
            for (int t = 0; t < tiles && ok; t++) {
                std::vector<uint64_t> ref(h.begin() + (size_t)t * n, h.begin() + (size_t)(t + 1) * n);
                std::sort(ref.begin(), ref.end());               // (depth bits, id): the total order the product's 64-bit keys have
                ok = std::equal(ref.begin(), ref.end(), got.begin() + (size_t)t * n);
            }
            std::sort(cyc.begin(), cyc.end());
            printf("n = %5d ties %d: %7.2f us per launch (%d tiles at once), workgroup cycles median %llu max %llu, %s\n", n, ties, 1e3 * ms / reps, tiles,
                   cyc[tiles / 2], cyc[tiles - 1], ok ? "sorted = std::sort" : "MISMATCH");
            hipFree(din); hipFree(dout); hipFree(dcyc);
        }
    }
    return 0;
}
