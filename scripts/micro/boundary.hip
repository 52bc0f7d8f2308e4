// boundary.hip -- what ONE kernel costs inside a captured hipGraph on MI355X when it does (almost) nothing: the floor under every kernel
// of the training step (eight dependent kernels per step).  A chain of K dependent kernels on one stream is
// captured into a graph and replayed; per-kernel cost = replay time / K.  Varied: the grid (workgroups x threads), the bytes each kernel
// writes (left dirty in the eight per-XCD L2s: written back at the kernel's end so that the next kernel, whose workgroups run on other
// XCDs, sees them) and the bytes it reads of what its predecessor wrote.
//
//   hipcc --offload-arch=gfx950 -O3 -o boundary boundary.hip && ./boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// every workgroup reads `rd` floats per thread of `src` (written by the previous kernel) and writes `wr` floats per thread to `dst`
__global__ void __launch_bounds__(1024) link(const float* __restrict__ src, float* __restrict__ dst, int rd, int wr, size_t n)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (int i = 0; i < rd; i++) acc += src[(tid + (size_t)i * nt) % n];
    for (int i = 0; i < wr; i++) dst[(tid + (size_t)i * nt) % n] = acc + (float)i;
    if (rd == 0 && wr == 0 && tid == 0x7fffffffffffull) dst[0] = acc;      // (keeps the arguments alive)
}

static float run(int blocks, int threads, int rd, int wr, float* a, float* b, size_t n, int K = 40, int reps = 20)
{
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int k = 0; k < K; k++)
        hipLaunchKernelGGL(link, dim3(blocks), dim3(threads), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, rd, wr, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; i++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
    return 1e3f * ms / (float)(K * reps);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs; per-kernel time of a chain of 40 dependent kernels replayed as one hipGraph (us)\n", prop.name, prop.multiProcessorCount);
    const size_t n = (size_t)64 << 20;                    // 256 MB per buffer
    float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    struct G { int blocks, threads; const char* what; };
    const G grids[] = { {1, 64, "1 x 64"}, {256, 64, "256 x 64"}, {256, 256, "256 x 256"}, {768, 256, "768 x 256 (the loss kernels)"},
                        {1563, 64, "1563 x 64 (per-Gaussian backward)"}, {512, 1024, "512 x 1024 (sort)"}, {2048, 256, "2048 x 256"} };
    printf("%-36s %10s %10s %10s %10s %10s\n", "grid", "nothing", "wr 1 MB", "wr 4 MB", "wr 16 MB", "rd+wr 16MB");
    for (const G& g : grids) {
        const size_t nt = (size_t)g.blocks * g.threads;
        auto per = [&](double mb) { return (int)((mb * 1048576.0 / 4.0 + nt - 1) / nt); };
        const float t0 = run(g.blocks, g.threads, 0, 0, a, b, n);
        const float t1 = run(g.blocks, g.threads, 0, per(1), a, b, n);
        const float t4 = run(g.blocks, g.threads, 0, per(4), a, b, n);
        const float t16 = run(g.blocks, g.threads, 0, per(16), a, b, n);
        const float t16r = run(g.blocks, g.threads, per(16), per(16), a, b, n);
        printf("%-36s %10.2f %10.2f %10.2f %10.2f %10.2f\n", g.what, t0, t1, t4, t16, t16r);
    }
    return 0;
}
