// Issue cost of single vector instructions on gfx950 in SHADER-CLOCK cycles (s_memtime around the loop: independent of where
// the clock happens to be), 8 wavefronts per SIMD, 8 independent copies per loop iteration:
//   cycles per wave-instruction per SIMD = (t1 - t0) / (8 wavefronts x iterations x 8 instructions)
// hipcc --offload-arch=gfx950 -O2 op_cycles.hip -o op_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define OP8(name, asmstr)                                                                                              \
    __global__ __launch_bounds__(256) void k_##name(unsigned long long *out, int iters) {                              \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        int b = threadIdx.x * 7 + 3;                                                                                   \
        asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n s_mov_b32 s20, 0x33333333\n s_mov_b32 s21, 0x33333333" ::: "vcc", "s20", "s21");                                                    \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < iters; ++it) {                                                                           \
            asm volatile(asmstr : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21"); \
        }                                                                                                              \
        asm volatile("s_nop 0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                             \
    }
#define R8(ins) ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7)
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define I_FMAC(i) "v_fma_f32 %" #i ", %" #i ", %8, 1.0 clamp\n"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define I_PKFMA(i) "v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[44:45]\n"
#define I_PKADD(i) "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_ANDL(i) "v_and_b32 %" #i ", 0x80808080, %" #i "\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CMPCNDS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_CND0(i) "v_cndmask_b32 %" #i ", 0, %8, vcc\n"
#define I_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %8\n"
#define I_SUBCL(i) "v_sub_f32 %" #i ", %" #i ", %8 clamp\n"
#define I_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define I_CMPI(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define I_CMPS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define I_CVTU8(i) "v_cvt_pk_u8_f32 %" #i ", %8, 1, %" #i "\n"
#define I_UBYTE(i) "v_cvt_f32_ubyte1 %" #i ", %8\n"
#define I_CVTI(i) "v_cvt_i32_f32 %" #i ", %8\n"
#define I_CVTF(i) "v_cvt_f32_i32 %" #i ", %8\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_DPP(i) "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %8\n"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %8\n"
#define I_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
OP8(fma, R8(I_FMA)) OP8(fma_clamp, R8(I_FMAC)) OP8(mul, R8(I_MUL)) OP8(addf, R8(I_ADDF)) OP8(minf, R8(I_MINF)) OP8(pkfma, R8(I_PKFMA)) OP8(pkadd, R8(I_PKADD))
OP8(addu, R8(I_ADDU)) OP8(subu, R8(I_SUBU)) OP8(and_, R8(I_AND)) OP8(andl, R8(I_ANDL)) OP8(xor_, R8(I_XOR)) OP8(bcnt, R8(I_BCNT)) OP8(lshl, R8(I_LSHL))
OP8(lshlor, R8(I_LSHLOR)) OP8(andor, R8(I_ANDOR)) OP8(bfe, R8(I_BFE)) OP8(perm, R8(I_PERM)) OP8(cndmask, R8(I_CNDMASK)) OP8(cmp, R8(I_CMP)) OP8(cmpi, R8(I_CMPI))
OP8(cmps, R8(I_CMPS)) OP8(cvtu8, R8(I_CVTU8)) OP8(ubyte, R8(I_UBYTE)) OP8(cvti, R8(I_CVTI)) OP8(cvtf, R8(I_CVTF)) OP8(add3, R8(I_ADD3)) OP8(mad24, R8(I_MAD24))
OP8(cnds, R8(I_CNDS)) OP8(cmpcnd, R8(I_CMPCND)) OP8(cmpcnds, R8(I_CMPCNDS)) OP8(cnd0, R8(I_CND0)) OP8(bfi, R8(I_BFI)) OP8(subcl, R8(I_SUBCL))
OP8(mul24, R8(I_MUL24)) OP8(mullo, R8(I_MULLO)) OP8(mov, R8(I_MOV)) OP8(dpp, R8(I_DPP)) OP8(rcp, R8(I_RCP)) OP8(sqrt_, R8(I_SQRT)) OP8(maxi, R8(I_MAXI)) OP8(med3, R8(I_MED3))

template <typename K>
void run(const char *name, K kern, unsigned long long *d) {
    const int iters = 4000, wps = 8, blocks = 256 * wps, nw = blocks * 4;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), d, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-12s %.2f cycles per wave-instruction per SIMD (median wavefront, 8 per SIMD)\n", name, (double)h[nw / 2] / ((double)wps * iters * 8));
}
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 1 << 20);
#define RUN(n) run(#n, k_##n, d)
    RUN(fma); RUN(fma_clamp); RUN(mul); RUN(addf); RUN(minf); RUN(pkfma); RUN(pkadd); RUN(addu); RUN(subu); RUN(and_); RUN(andl); RUN(xor_); RUN(bcnt);
    RUN(lshl); RUN(lshlor); RUN(andor); RUN(bfe); RUN(perm); RUN(cndmask); RUN(cmp); RUN(cmpi); RUN(cmps); RUN(cvtu8); RUN(ubyte); RUN(cvti); RUN(cvtf);
    RUN(cnds); RUN(cmpcnd); RUN(cmpcnds); RUN(cnd0); RUN(bfi); RUN(subcl);
    RUN(add3); RUN(mad24); RUN(mul24); RUN(mullo); RUN(mov); RUN(dpp); RUN(rcp); RUN(sqrt_); RUN(maxi); RUN(med3);
    return 0;
}
