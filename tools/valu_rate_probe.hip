// Issue cost of single VALU instructions on gfx950 (wave64): cycles per wave instruction per SIMD with 8 waves resident on every SIMD
// (inline asm, 64 back-to-back copies of ONE instruction per loop step over rotating registers).  Calibrates the "4 cycles per VALU
// instruction" model behind bench.py's valu_issue_frac and shows where the cycles of the BVH node test go (docs/EXPERIMENTS.md §4.2).
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

KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %2, %1\n")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1\n")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %2\n")
KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %2\n")
KERNEL(k_max_f16,    "v_max_f16 %0, %0, %1\n")
KERNEL(k_fma_f16,    "v_fma_f16 %0, %0, %2, %1\n")
KERNEL(k_cvt_f16,    "v_cvt_f16_f32 %0, %1\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %2\n")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1\n")
KERNEL(k_pk_mad_i16, "v_pk_mad_i16 %0, %0, %2, %1\n")
KERNEL(k_pk_lshl,    "v_pk_lshlrev_b16 %0, 1, %0\n")
KERNEL(k_sad_u8,     "v_sad_u8 %0, %0, %2, %1\n")
KERNEL(k_dot4,       "v_dot4_i32_i8 %0, %0, %2, %1\n")
KERNEL(k_min3_i32,   "v_min3_i32 %0, %0, %1, %3\n")
KERNEL(k_mad_i32_i24,"v_mad_i32_i24 %0, %0, %2, %1\n")
KERNEL(k_add_u32,    "v_add_u32 %0, %0, %1\n")
KERNEL(k_sub_f32,    "v_sub_f32 %0, %0, %2\n")

KERNEL(k_mul_sdwa,  "v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_add_sdwa,  "v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n")
KERNEL(k_mov,       "v_mov_b32 %0, %1\n")
KERNEL(k_add_f32,   "v_add_f32 %0, %0, %2\n")
KERNEL(k_min_f32,   "v_min_f32 %0, %0, %1\n")
KERNEL(k_med3,      "v_med3_f32 %0, %0, %1, %3\n")
KERNEL(k_max_u32,   "v_max_u32 %0, %0, %1\n")
KERNEL(k_max_i32,   "v_max_i32 %0, %0, %1\n")
KERNEL(k_cmp_u32,   "v_cmp_le_u32 vcc, %0, %1\n")
KERNEL(k_sub_u32,   "v_sub_u32 %0, %0, %1\n")
KERNEL(k_or,        "v_or_b32 %0, %0, %1\n")
KERNEL(k_lshl,      "v_lshlrev_b32 %0, 1, %0\n")

// numerical check of the SDWA byte -> denormal trick: byte k of w read as an f32 bit pattern is b * 2^-149
__global__ void k_sdwa_check(const uint32_t* w, float A, float* out)
{
    const uint32_t v = w[threadIdx.x];
    float r0, r1, r2, r3;
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r0) : "v"(v), "v"(A));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r1) : "v"(v), "v"(A));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r2) : "v"(v), "v"(A));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r3) : "v"(v), "v"(A));
    out[threadIdx.x * 4 + 0] = r0; out[threadIdx.x * 4 + 1] = r1; out[threadIdx.x * 4 + 2] = r2; out[threadIdx.x * 4 + 3] = r3;
}

#define KERNEL64(NAME, ASM)                                                                                           \
    __global__ void NAME(float* out, int iters)                                                                       \
    {                                                                                                                 \
        double a = (double)(threadIdx.x & 63) * 1e-3 + 1.0, b = 1.0001, c = 0.5;                                      \
        for (int i = 0; i < iters; i++) { REP64(asm volatile(ASM : "+v"(a), "+v"(c) : "v"(b));) }                     \
        if (a == 123.456) out[0] = (float)(a + c);                                                                    \
    }
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %1\n")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %2\n")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2\n")

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
    run("v_pk_fma_f16", k_pk_fma_f16, d); run("v_pk_max_f16", k_pk_max_f16, d); run("v_pk_add_f16", k_pk_add_f16, d); run("v_pk_mul_f16", k_pk_mul_f16, d);
    run("v_max_f16", k_max_f16, d); run("v_fma_f16", k_fma_f16, d); run("v_cvt_f16_f32", k_cvt_f16, d);
    run("v_pk_add_u16", k_pk_add_u16, d); run("v_pk_max_i16", k_pk_max_i16, d); run("v_pk_mad_i16", k_pk_mad_i16, d); run("v_pk_lshlrev_b16", k_pk_lshl, d);
    run("v_sad_u8", k_sad_u8, d); run("v_dot4_i32_i8", k_dot4, d); run("v_min3_i32", k_min3_i32, d); run("v_mad_i32_i24", k_mad_i32_i24, d);
    run("v_add_u32", k_add_u32, d); run("v_sub_f32", k_sub_f32, d);
    run("v_mul_f32_sdwa BYTE", k_mul_sdwa, d); run("v_add_f32_sdwa BYTE", k_add_sdwa, d); run("v_mov_b32", k_mov, d); run("v_add_f32", k_add_f32, d);
    run("v_min_f32", k_min_f32, d); run("v_med3_f32", k_med3, d); run("v_max_u32", k_max_u32, d); run("v_max_i32", k_max_i32, d); run("v_cmp_le_u32", k_cmp_u32, d);
    run("v_sub_u32", k_sub_u32, d); run("v_or_b32", k_or, d); run("v_lshlrev_b32", k_lshl, d);
    {
        uint32_t hw[64]; for (int i = 0; i < 64; i++) hw[i] = 0x01000000u * (uint32_t)(255 - i) + 0x00010000u * (uint32_t)(i * 3 + 1) + 0x0100u * (uint32_t)(i + 100) + (uint32_t)i;
        uint32_t* dw; float* dout; CK(hipMalloc(&dw, sizeof(hw))); CK(hipMalloc(&dout, 256 * 4)); CK(hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice));
        const float A = 0.37f * 0x1p100f;
        k_sdwa_check<<<1, 64>>>(dw, A, dout);
        float ho[256]; CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 64; i++) for (int k = 0; k < 4; k++)
        {
            const float b = (float)((hw[i] >> (8 * k)) & 0xffu), want = (b * 0x1p-149f) * A;
            if (ho[i * 4 + k] != want) { if (bad < 5) printf("  lane %d byte %d: got %g want %g\n", i, k, ho[i * 4 + k], want); bad++; }
        }
        printf("SDWA byte-as-denormal multiply: %d of 256 values differ from (b * 2^-149) * A\n", bad);
    }
    run("v_pk_fma_f32", k_pk_fma_f32, d); run("v_pk_mul_f32", k_pk_mul_f32, d); run("v_pk_add_f32", k_pk_add_f32, d);
    return 0;
}
