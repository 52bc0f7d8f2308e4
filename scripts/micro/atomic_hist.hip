// Micro-benchmark: what the flush of per-block tile histograms costs (preprocess_forward_kernel: 391 blocks x ~230 non-zero counters,
// one global atomicAdd each, on ~250 busy tiles) with the counters DENSE (16 per 64-byte line, as tile_count is) or one per line, and
// the same for RETURNING atomics (scatter_kernel's run reservations on tile_cursor).
// hipcc --offload-arch=gfx950 -O3 -o atomic_hist.bin atomic_hist.hip && ./atomic_hist.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RET>
__global__ void __launch_bounds__(256) k(unsigned* cnt, int stride, int n_addr, unsigned* sink)
{
    const int t = threadIdx.x;
    if (t < n_addr) {
        unsigned* p = cnt + (size_t)((t * 7 + blockIdx.x) % n_addr) * stride;   // every block hits every counter once, in its own order
        if (RET) { const unsigned v = atomicAdd(p, 1u); if (v == 0xffffffffu) sink[0] = v; }
        else atomicAdd(p, 1u);
    }
}
template <int RET>
float run(int stride, int blocks, int n_addr)
{
    unsigned* cnt; unsigned* sink;
    (void)hipMalloc(&cnt, (size_t)n_addr * stride * 4 + 64); (void)hipMalloc(&sink, 64);
    (void)hipMemset(cnt, 0, (size_t)n_addr * stride * 4 + 64);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<RET>, dim3(blocks), dim3(256), 0, 0, cnt, stride, n_addr, sink);
    (void)hipEventRecord(a);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k<RET>, dim3(blocks), dim3(256), 0, 0, cnt, stride, n_addr, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(cnt); (void)hipFree(sink);
    return ms / 50 * 1000.f;
}
int main()
{
    printf("391 blocks x 230 atomics on 230 counters, us per launch (an empty launch of the same grid: %.1f)\n", run<0>(1, 391, 0));
    for (int stride : {1, 16, 32, 64})
        printf("stride %2d words: non-returning %.1f us   returning %.1f us\n", stride, run<0>(stride, 391, 230), run<1>(stride, 391, 230));
    return 0;
}
