// handover.hip -- what it costs ONE workgroup to hand a few KB to ANOTHER workgroup inside a running kernel on MI355X (release store of
// a flag after the data, acquire load by the consumer, agent scope: the eight XCDs' L2s are not coherent with each other), against
// what a kernel boundary costs.  The measurement behind DESIGN.md section 8.3 (tile-granular chaining of merge_gather -> blend_forward,
// round 3's review item 5): a tile's blend could start when ITS records are written instead of when the whole sort kernel has ended --
// if the hand-over is cheaper than the ~5 us boundary it replaces.
//
//   hipcc --offload-arch=gfx950 -O3 -o handover handover.hip && ./handover
//
// Pairs of workgroups (producer 2i, consumer 2i+1; all resident at once) ping-pong `bytes` of payload: the producer writes the payload,
// releases a sequence number; the consumer acquires it, reads and checks the payload, releases its acknowledgement; `iters` rounds.
// One-way hand-over latency = wall time of the kernel / (2 * iters).  Every spin loop has a watchdog.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int LINE = 64;   // words between two flags: every flag on a 256-byte line of its own

__global__ void __launch_bounds__(256) pingpong(uint32_t* __restrict__ payload, uint32_t* flags, int words, int iters, uint32_t* err)
{
    const int pair = blockIdx.x >> 1, role = blockIdx.x & 1;
    uint32_t* data = payload + (size_t)pair * words;
    uint32_t* f_data = flags + (size_t)(2 * pair) * LINE;
    uint32_t* f_ack = flags + (size_t)(2 * pair + 1) * LINE;
    for (int it = 1; it <= iters; it++) {
        if (role == 0) {
            for (int i = threadIdx.x; i < words; i += blockDim.x) data[i] = (uint32_t)(it * 131 + i);
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(f_data, (uint32_t)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                long spins = 0;
                while (__hip_atomic_load(f_ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)it)
                    if (++spins > 50000000L) { atomicOr(err, 1u); break; }
            }
            __syncthreads();
        } else {
            if (threadIdx.x == 0) {
                long spins = 0;
                while (__hip_atomic_load(f_data, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)it)
                    if (++spins > 50000000L) { atomicOr(err, 2u); break; }
            }
            __syncthreads();
            uint32_t bad = 0;
            for (int i = threadIdx.x; i < words; i += blockDim.x)
                bad |= (__builtin_nontemporal_load(&data[i]) != (uint32_t)(it * 131 + i)) ? 1u : 0u;
            if (bad) atomicOr(err, 4u);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(f_ack, (uint32_t)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(256) write_k(uint32_t* __restrict__ payload, int words, uint32_t v)
{
    uint32_t* data = payload + (size_t)blockIdx.x * words;
    for (int i = threadIdx.x; i < words; i += blockDim.x) data[i] = v + i;
}
__global__ void __launch_bounds__(256) read_k(const uint32_t* __restrict__ payload, int words, uint32_t v, uint32_t* err)
{
    const uint32_t* data = payload + (size_t)blockIdx.x * words;
    uint32_t bad = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) bad |= data[i] != v + i;
    if (bad) atomicOr(err, 8u);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    uint32_t *payload, *flags, *err;
    const int max_pairs = cus * 4, max_words = 16384;
    hipMalloc(&payload, (size_t)max_pairs * max_words * 4); hipMalloc(&flags, (size_t)max_pairs * 2 * LINE * 4); hipMalloc(&err, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int per_cu : {1, 2, 4}) {
        for (int bytes : {256, 4096, 49152}) {
            const int pairs = cus * per_cu / 2, words = bytes / 4, iters = 200;
            hipMemset(flags, 0, (size_t)max_pairs * 2 * LINE * 4); hipMemset(err, 0, 4);
            hipLaunchKernelGGL(pingpong, dim3(2 * pairs), dim3(256), 0, 0, payload, flags, words, 20, err);      // warm-up
            hipDeviceSynchronize();
            hipMemset(flags, 0, (size_t)max_pairs * 2 * LINE * 4);
            hipEventRecord(a);
            hipLaunchKernelGGL(pingpong, dim3(2 * pairs), dim3(256), 0, 0, payload, flags, words, iters, err);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            uint32_t e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            printf("%d workgroups per CU (%4d pairs), %6d B per hand-over: %6.2f us one way%s\n", per_cu, pairs, bytes, 1e3 * ms / (2.0 * iters),
                   e ? "  ** ERROR flags set" : "");
        }
    }
    // the kernel boundary it would replace: write kernel -> read kernel on one stream, against the two kernels' own lengths
    for (int bytes : {4096, 49152}) {
        const int wgs = cus * 2, words = bytes / 4, reps = 200;
        hipMemset(err, 0, 4);
        for (int i = 0; i < 10; i++) { hipLaunchKernelGGL(write_k, dim3(wgs), dim3(256), 0, 0, payload, words, 7u); hipLaunchKernelGGL(read_k, dim3(wgs), dim3(256), 0, 0, payload, words, 7u, err); }
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(write_k, dim3(wgs), dim3(256), 0, 0, payload, words, (uint32_t)i); hipLaunchKernelGGL(read_k, dim3(wgs), dim3(256), 0, 0, payload, words, (uint32_t)i, err); }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        uint32_t e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("write kernel -> read kernel, %d workgroups x %6d B: %6.2f us per PAIR of back-to-back launches%s\n", wgs, bytes, 1e3 * ms / reps, e ? "  ** ERROR" : "");
    }
    return 0;
}
