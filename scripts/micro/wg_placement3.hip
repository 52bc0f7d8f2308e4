// Where do the workgroups of a resident-grid launch land?  Prints, per XCD-local workgroup index l = blockIdx / 8, the (SE, CU) from
// HW_REG_HW_ID and the SIMD of each of its 4 waves.  hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(192) probe(unsigned* out, int spin)
{
    extern __shared__ unsigned lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = hw;
    // stay resident for a while so that the whole grid is co-resident
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main(int argc, char** argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 4, lds = argc > 2 ? atoi(argv[2]) : 39000;
    const int wgs = 256 * per_cu;
    unsigned* d; hipMalloc(&d, wgs * 3 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(192), lds, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 6);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    printf("blockIdx  xcc  l=blk/8   se sh cu   simd of waves 0..3\n");
    for (int b = 0; b < wgs; b++) {
        if ((b % 8) != 0 && b >= 64) continue;     // XCD 0 only after the first 64
        unsigned hw = h[b * 6], xcc = h[b * 6 + 1] & 0xf;
        printf("%6d   %2u  %5d    %u  %u  %2u    ", b, xcc, b / 8, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
        for (int w = 0; w < 3; w++) printf("%u ", (h[(b * 3 + w) * 2] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
