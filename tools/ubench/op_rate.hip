// Microbenchmark: issue cost of single vector instructions on gfx950 (8 independent copies per loop iteration,
// 8 wavefronts per SIMD): cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define OP8(name, asmstr)                                                                                              \
    __global__ void k_##name(float *out, int iters) {                                                                 \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        int b = threadIdx.x * 7 + 3;                                                                                   \
        for (int it = 0; it < iters; ++it) {                                                                           \
            asm volatile(asmstr : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); \
        }                                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                   \
    }
#define R8(ins) ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7)
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_ANDL(i) "v_and_b32 %" #i ", 0x80808080, %" #i "\n"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define I_CMPS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define I_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define I_CVTU8(i) "v_cvt_pk_u8_f32 %" #i ", %8, 1, %" #i "\n"
#define I_UBYTE(i) "v_cvt_f32_ubyte1 %" #i ", %8\n"
#define I_SUBCL(i) "v_sub_f32 %" #i ", %" #i ", %8 clamp\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define I_SDWA(i) "v_cmp_eq_u32_sdwa vcc, %" #i ", %8 src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_READL(i) "v_readlane_b32 s20, %" #i ", 3\n"
OP8(fma, R8(I_FMA)) OP8(mul, R8(I_MUL)) OP8(addu, R8(I_ADDU)) OP8(and_, R8(I_AND)) OP8(andl, R8(I_ANDL)) OP8(bcnt, R8(I_BCNT))
OP8(lshlor, R8(I_LSHLOR)) OP8(cndmask, R8(I_CNDMASK)) OP8(cmp, R8(I_CMP)) OP8(cmps, R8(I_CMPS)) OP8(min_, R8(I_MIN)) OP8(cvtu8, R8(I_CVTU8))
OP8(ubyte, R8(I_UBYTE)) OP8(subcl, R8(I_SUBCL)) OP8(add3, R8(I_ADD3)) OP8(sdwa, R8(I_SDWA)) OP8(mov, R8(I_MOV))

template <typename K>
void run(const char *name, K kern, float *d) {
    const int iters = 20000, wps = 8, blocks = 256 * wps;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-10s %.2f cycles per wave-instruction per SIMD (8 waves/SIMD, 2.4 GHz nominal)\n", name, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8));
}
int main() {
    float *d; (void)hipMalloc(&d, 64 << 20);
    run("fma", k_fma, d); run("mul", k_mul, d); run("add_u32", k_addu, d); run("and", k_and_, d); run("and_lit", k_andl, d);
    run("bcnt", k_bcnt, d); run("lshl_or", k_lshlor, d); run("cndmask", k_cndmask, d); run("cmp_vcc", k_cmp, d); run("cmp_sgpr", k_cmps, d);
    run("min_f32", k_min_, d); run("cvt_pk_u8", k_cvtu8, d); run("cvt_ubyte", k_ubyte, d); run("sub_clamp", k_subcl, d); run("add3", k_add3, d);
    run("cmp_sdwa", k_sdwa, d); run("mov", k_mov, d);
    return 0;
}
