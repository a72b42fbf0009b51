// Microbenchmark: does a ds_add_u64 with part of its lanes switched off cost less?  (decides how the KNN scatter
// backward walks a query's neighbour mask: one predicated atomic per slot, or one full atomic per set bit)
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_masked.bin lds_atomic_masked.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    __shared__ unsigned long long s[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) s[i] = 0;
    __syncthreads();
    unsigned x = tid * 2654435761u + blockIdx.x;
    unsigned y = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const int a = (x >> 8) & 4095;
        bool on = true;
        if (MODE == 1) on = (x >> 24) & 1;                 // 50 % of the lanes, at random
        if (MODE == 2) on = ((x >> 24) & 3) == 0;          // 25 %
        if (MODE == 3) on = (tid & 32) != 0;               // one half-wave
        if (MODE == 4) on = (tid & 1) != 0;                // every second lane
        if (MODE == 5) { y += s[a]; on = false; }          // (reference: a dependent b64 read instead)
        if (MODE == 6) { y += reinterpret_cast<unsigned short *>(s)[a]; on = false; }   // u16 read
        if (on) atomicAdd(&s[a], (unsigned long long)x);
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (unsigned)s[5] + x + y;
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
    const char *names[] = {"ds_add_u64 all lanes", "ds_add_u64 50% random lanes", "ds_add_u64 25% random lanes",
                           "ds_add_u64 one half-wave", "ds_add_u64 every 2nd lane", "ds_read_b64 dependent", "ds_read_u16 dependent"};
    float ms[7] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters),
                   run<4>(d, blocks, iters), run<5>(d, blocks, iters), run<6>(d, blocks, iters)};
    for (int m = 0; m < 7; ++m)
        printf("%-30s %8.3f ms  -> %6.1f cycles per wave-instruction per CU\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (16.0 * iters));
    return 0;
}
