// Microbenchmark: LDS atomic throughput on gfx950 (ds_add_u32 / ds_add_f32, private vs scattered).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    __shared__ unsigned s[32 * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32 * 256; i += 256) s[i] = 0;
    __syncthreads();
    unsigned x = tid * 2654435761u + blockIdx.x;
    float *sf = reinterpret_cast<float *>(s);
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const int bin = (x >> 20) & 31;
        if (MODE == 0) atomicAdd(&s[bin * 256 + tid], 1u);                 // private column, no-return int
        if (MODE == 1) atomicAdd(&sf[bin * 256 + tid], 1.0f);              // private column, float
        if (MODE == 2) atomicAdd(&s[(x >> 8) & 8191], 1u);                 // scattered int
        if (MODE == 3) atomicAdd(&sf[(x >> 8) & 8191], 1.0f);              // scattered float
        if (MODE == 4) s[bin * 256 + tid] += 1u;                           // private read-modify-write
        if (MODE == 5) { unsigned v = s[(x >> 8) & 8191]; x += v; }        // scattered read (dependent)
        if (MODE == 6) atomicAdd(&sf[((x >> 8) & 255) + bin * 256], 1.0f); // scattered within a row (like per-query acc)
        if (MODE == 7) atomicAdd(reinterpret_cast<unsigned long long *>(s) + ((x >> 8) & 4095), (unsigned long long)x); // scattered u64
        if (MODE == 8) { unsigned v = atomicAdd(&s[(x >> 8) & 8191], 1u); x += v; }   // scattered int, returning (dependent)
        if (MODE == 9) atomicAdd(reinterpret_cast<unsigned long long *>(s) + (bin & 15) * 256 + tid, (unsigned long long)x); // private u64
        if (MODE == 10) { float2 *p2 = reinterpret_cast<float2 *>(s) + ((x >> 8) & 4095); float2 v = *p2; v.x += 1.f; v.y += 2.f; *p2 = v; } // non-atomic b64 RMW scattered
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = s[5] + x;
}

template <int MODE>
float run(unsigned *d, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    unsigned *d; hipMalloc(&d, 1 << 20);
    const int blocks = 256 * 4, iters = 4096;   // 4 blocks (16 waves) per CU
    const char *names[] = {"ds_add_u32 private", "ds_add_f32 private", "ds_add_u32 scattered", "ds_add_f32 scattered",
                           "private RMW (read+write)", "scattered dependent read", "ds_add_f32 row-scattered",
                           "ds_add_u64 scattered", "ds_add_rtn_u32 scattered dep", "ds_add_u64 private", "b64 RMW scattered"};
    float ms[11] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters),
                   run<4>(d, blocks, iters), run<5>(d, blocks, iters), run<6>(d, blocks, iters), run<7>(d, blocks, iters),
                   run<8>(d, blocks, iters), run<9>(d, blocks, iters), run<10>(d, blocks, iters)};
    for (int m = 0; m < 11; ++m) {
        // wave-instructions per CU = 16 waves * iters ; cycles at 2.4 GHz
        const double cyc_per_winstr = ms[m] * 1e-3 * 2.4e9 / (16.0 * iters);
        printf("%-28s %8.3f ms  -> %6.1f cycles per wave-instruction per CU\n", names[m], ms[m], cyc_per_winstr);
    }
    return 0;
}
