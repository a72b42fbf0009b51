// Microbenchmark: sustained VALU issue rate per SIMD on gfx950 for wave64 code (fma / int / mixed).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    const float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 8 independent fp32 FMAs
            a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
            a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
        } else if (MODE == 1) {  // 4 fma + 4 int ops
            a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
            i0 = (i0 * 3) ^ it; i1 = (i1 + i0) & 0xffff; i2 = max(i2, i1) + 1; i3 = (i3 << 1) | (i2 & 1);
        } else if (MODE == 3) {  // 8 independent packed fp32 multiplies (2 x fp32 per lane per instruction)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b};
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 4) {  // 8 independent plain fp32 multiplies, same asm form (reference for MODE 3)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (MODE == 5) {  // SWAR count step with 32-bit literals: sub, and, bcnt-accumulate (x4 independent)
            asm volatile("v_sub_u32 %0, 0xbfbfbfbf, %4\n v_and_b32 %0, 0x80808080, %0\n v_bcnt_u32_b32 %4, %0, %4\n"
                         "v_sub_u32 %1, 0xbfbfbfbf, %5\n v_and_b32 %1, 0x80808080, %1\n v_bcnt_u32_b32 %5, %1, %5\n"
                         "v_sub_u32 %2, 0xbfbfbfbf, %6\n v_and_b32 %2, 0x80808080, %2\n v_bcnt_u32_b32 %6, %2, %6\n"
                         "v_sub_u32 %3, 0xbfbfbfbf, %7\n v_and_b32 %3, 0x80808080, %3\n v_bcnt_u32_b32 %7, %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
        } else if (MODE == 6) {  // the same with the constants in scalar registers
            int c1 = 0xbfbfbfbf, c2 = 0x80808080;
            asm volatile("v_sub_u32 %0, %8, %4\n v_and_b32 %0, %9, %0\n v_bcnt_u32_b32 %4, %0, %4\n"
                         "v_sub_u32 %1, %8, %5\n v_and_b32 %1, %9, %1\n v_bcnt_u32_b32 %5, %1, %5\n"
                         "v_sub_u32 %2, %8, %6\n v_and_b32 %2, %9, %2\n v_bcnt_u32_b32 %6, %2, %6\n"
                         "v_sub_u32 %3, %8, %7\n v_and_b32 %3, %9, %3\n v_bcnt_u32_b32 %7, %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "s"(c1), "s"(c2));
        } else {                 // compare + select chains
            a0 = a0 > a1 ? a0 * b : a1 + c; a1 = a1 > a2 ? a1 * b : a2 + c; a2 = a2 > a3 ? a2 * b : a3 + c; a3 = a3 > a0 ? a3 * b : a0 + c;
            a4 = a4 > a5 ? a4 * b : a5 + c; a5 = a5 > a6 ? a5 * b : a6 + c; a6 = a6 > a7 ? a6 * b : a7 + c; a7 = a7 > a4 ? a7 * b : a4 + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3;
}

template <int MODE>
void run(float *d, int waves_per_simd, int ops_per_iter) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;       // 256 threads = 4 waves per block = 1 wave per SIMD per block
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr_per_simd = (double)waves_per_simd * iters * ops_per_iter;
    printf("mode %d  waves/SIMD %d : %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", MODE, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / winstr_per_simd);
}

int main() {
    float *d; hipMalloc(&d, 64 << 20);
    for (int w : {1, 2, 4, 8}) run<0>(d, w, 8);
    for (int w : {1, 2, 4, 8}) run<1>(d, w, 12);
    for (int w : {1, 2, 4, 8}) run<2>(d, w, 24);
    for (int w : {1, 2, 4, 8}) run<3>(d, w, 8);
    for (int w : {1, 2, 4, 8}) run<4>(d, w, 8);
    for (int w : {2, 4, 8}) run<5>(d, w, 12);
    for (int w : {2, 4, 8}) run<6>(d, w, 12);
    return 0;
}
