// Issue cost of single VALU instructions on gfx950 (wave64): cycles per wave instruction per SIMD with 8 waves resident on every SIMD
// (inline asm, 64 back-to-back copies of ONE instruction per loop step over rotating registers).  Calibrates the "4 cycles per VALU
// instruction" model behind bench.py's valu_issue_frac and shows where the cycles of the BVH node test go.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/_build/valu_rate_probe && tools/_build/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

#define KERNEL(NAME, ASM)                                                                                             \
    __global__ void NAME(float* out, int iters)                                                                       \
    {                                                                                                                 \
        float a = (float)(threadIdx.x & 63) * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f;                        \
        for (int i = 0; i < iters; i++) { REP64(asm volatile(ASM : "+v"(a), "+v"(c) : "v"(b), "v"(d));) }            \
        if (a == 123.456f) out[0] = a + c;                                                                            \
    }

KERNEL(k_fma,      "v_fma_f32 %0, %0, %2, %1\n")
KERNEL(k_mul,      "v_mul_f32 %0, %0, %2\n")
KERNEL(k_max,      "v_max_f32 %0, %0, %1\n")
KERNEL(k_max3,     "v_max3_f32 %0, %0, %1, %3\n")
KERNEL(k_cvt_ub,   "v_cvt_f32_ubyte1 %0, %1\n")
KERNEL(k_cvt_u32,  "v_cvt_f32_u32 %0, %1\n")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL(k_cmp,      "v_cmp_le_f32 vcc, %0, %1\n")
KERNEL(k_and,      "v_and_b32 %0, %0, %1\n")
KERNEL(k_bfe,      "v_bfe_u32 %0, %0, 3, 17\n")
KERNEL(k_lshl_or,  "v_lshl_or_b32 %0, %0, 1, %1\n")
KERNEL(k_exp,      "v_exp_f32 %0, %1\n")
KERNEL(k_rcp,      "v_rcp_f32 %0, %1\n")
KERNEL(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1\n")
KERNEL(k_mad_u24,  "v_mad_u32_u24 %0, %0, %2, %1\n")
KERNEL(k_perm,     "v_perm_b32 %0, %0, %1, %3\n")
KERNEL(k_cvt_pk,   "v_cvt_pkrtz_f16_f32 %0, %0, %1\n")
KERNEL(k_fma_mix,  "v_fma_mix_f32 %0, %0, %2, %1\n")

template <typename K>
int run(const char* name, K kern, float* d)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<8192, 64>>>(d, 10);
    CK(hipEventRecord(e0));
    kern<<<8192, 64>>>(d, 2000);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = 8192.0 * 2000 * 64;
    printf("%-22s %7.3f ms  %5.2f cycles per wave instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / insts);
    return 0;
}

int main()
{
    float* d; CK(hipMalloc(&d, 64));
    run("v_fma_f32", k_fma, d); run("v_mul_f32", k_mul, d); run("v_max_f32", k_max, d); run("v_max3_f32", k_max3, d);
    run("v_cvt_f32_ubyte1", k_cvt_ub, d); run("v_cvt_f32_u32", k_cvt_u32, d); run("v_cndmask_b32", k_cndmask, d); run("v_cmp_le_f32", k_cmp, d);
    run("v_and_b32", k_and, d); run("v_bfe_u32", k_bfe, d); run("v_lshl_or_b32", k_lshl_or, d); run("v_exp_f32", k_exp, d); run("v_rcp_f32", k_rcp, d);
    run("v_mul_lo_u32", k_mul_lo, d); run("v_mad_u32_u24", k_mad_u24, d); run("v_perm_b32", k_perm, d); run("v_cvt_pkrtz_f16_f32", k_cvt_pk, d);
    run("v_fma_mix_f32", k_fma_mix, d);
    return 0;
}
