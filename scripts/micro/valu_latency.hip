// Per-wave issue intervals on gfx950, one wave on an otherwise idle CU: cycles per instruction (s_memtime) of chains of dependent and of
// independent VALU instructions, of a DPP chain, of v_exp_f32, of a VALU compare feeding a scalar branch, of an LDS read -> use.
// hipcc --offload-arch=gfx950 -O3 -o valu_latency.bin valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void probe(unsigned long long* out, float* sink, int n)
{
    __shared__ float lds[256];
    lds[threadIdx.x] = (float)threadIdx.x;
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    unsigned long long t0, t1;
    // 1. dependent v_fma chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 2. four independent chains
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 3. dependent DPP multiply chain (quad_perm shift), as in the blend trip
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %1 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 4. dependent v_exp chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_exp_f32 %0, %0" : "+v"(c));) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 5. v_cmp -> s_cbranch_vccz (never taken) + one dependent VALU
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_lt_f32 vcc, 0, %0\n s_cbranch_vccz 1f\n v_add_f32 %0, %0, %1\n1:" : "+v"(a) : "v"(b) : "vcc");) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 6. v_cmp -> v_cndmask (VALU reads the mask it just wrote)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_lt_f32 vcc, 0, %0\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 7. ds_read_b128 -> use (dependent address)
    int addr = (threadIdx.x & 15) * 16;
    float v = 0.0f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        REP16(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xf0, %0" : "=&v"(v), "+v"(addr) :: "memory");)
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[6] = t1 - t0;
    // 8. ballot-style: v_cmp to SGPR pair, s_cmp on it, s_cbranch
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_lt_f32 vcc, 0, %0\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 2f\n v_add_f32 %0, %0, %1\n2:" : "+v"(a) : "v"(b) : "vcc", "scc");) }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[7] = t1 - t0;
    sink[threadIdx.x] = a + c + d + e + v + (float)addr;
}
int main()
{
    unsigned long long* d; float* s; const int n = 200;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 1024);
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 0, 0, d, s, n);
        (void)hipDeviceSynchronize();
        unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        const char* name[8] = { "dependent v_fma", "4 independent v_fma (per instr)", "s_nop 1 + dependent v_mul_dpp", "dependent v_exp", "v_cmp -> s_cbranch_vccz + v_add (per triple)",
                                "v_cmp -> s_nop 1 -> v_cndmask (per triple)", "ds_read_b32 -> wait -> dependent address", "v_cmp -> s_cmp_eq_u64 vcc -> s_cbranch_scc + v_add" };
        const double div[8] = { 16.0 * n, 64.0 * n, 16.0 * n, 16.0 * n, 16.0 * n, 16.0 * n, 16.0 * n, 16.0 * n };
        printf("---- %d wave(s) in the workgroup (one per SIMD up to 4): s_memtime ticks per step\n", waves);
        for (int i = 0; i < 8; i++) printf("  %-55s %7.2f\n", name[i], h[i] / div[i]);
    }
    return 0;
}
