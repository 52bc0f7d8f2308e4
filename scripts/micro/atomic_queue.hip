// Microbenchmark: cost of returning device-scope atomicAdd on few shared addresses (the work-queue pattern of the blend kernels).
// hipcc --offload-arch=gfx950 -O3 -o atomic_queue atomic_queue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) pop_kernel(unsigned* heads, int n_addr, int stride_words, int pops, unsigned* sink, int mode)
{
    const int lane = threadIdx.x & 63;
    const int a = ((int)blockIdx.x % n_addr) * stride_words;
    unsigned acc = 0;
    for (int i = 0; i < pops; i++) {
        unsigned v = 0;
        if (mode == 0) { if (lane == 0) v = atomicAdd(&heads[a], 1u); }                         // returning, device scope
        else if (mode == 1) { if (lane == 0) atomicAdd(&heads[a], 1u); }                       // non-returning
        else if (mode == 2) { if (lane == 0) v = __hip_atomic_fetch_add(&heads[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        acc += v;
    }
    if (lane == 0) sink[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

int main()
{
    unsigned *heads, *sink;
    hipMalloc(&heads, 1 << 20); hipMalloc(&sink, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256;
    for (int mode = 0; mode < 3; mode++)
    for (int n_addr : {1, 8, 64, 256})
    for (int stride : {1, 64})
    for (int pops : {8, 64}) {
        hipMemset(heads, 0, 1 << 20);
        pop_kernel<<<wgs, 256>>>(heads, n_addr, stride, pops, sink, mode);     // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        pop_kernel<<<wgs, 256>>>(heads, n_addr, stride, pops, sink, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double total = (double)wgs * 4 * pops;
        printf("mode %d addr %3d stride %2d pops/wave %2d: %8.1f us  -> %.1f ns per atomic overall, %.1f ns per atomic per address, %.2f us per pop as seen by a wave\n",
               mode, n_addr, stride, pops, ms * 1e3, ms * 1e6 / total, ms * 1e6 / (total / n_addr), ms * 1e3 / pops);
    }
    return 0;
}
