// Microbenchmark: workgroup dispatch rate on gfx950 -- how fast can the device start (and retire) workgroups
// that do almost nothing?  Short tile kernels (12 us per workgroup) run into this ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NT>
__global__ __launch_bounds__(NT) void k_empty(float *out, int spin) {
    extern __shared__ float s_x[];
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0001f, 0.5f);
    if (a == 12345.678f) out[blockIdx.x] = a + s_x[threadIdx.x];
}

template <int NT>
void run(float *d, int blocks, size_t lds, int spin) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)k_empty<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty<NT>, dim3(blocks), dim3(NT), lds, 0, d, spin);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_empty<NT>, dim3(blocks), dim3(NT), lds, 0, d, spin);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double us = 1e3 * ms / reps;
    printf("threads %4d  blocks %6d  lds %6zu B  spin %5d : %8.1f us  = %7.1f workgroups/us (%6.1f waves/us)\n", NT, blocks, lds, spin,
           us, blocks / us, blocks * (NT / 64) / us);
}

int main() {
    float *d;
    hipMalloc(&d, 1 << 20);
    for (int spin : {0, 2000}) {
        run<64>(d, 67200, 0, spin);
        run<256>(d, 16800, 0, spin);
        run<256>(d, 16800, 18 * 1024, spin);
        run<256>(d, 16800, 39 * 1024, spin);
        run<1024>(d, 4200, 0, spin);
        run<1024>(d, 4200, 64 * 1024, spin);
    }
    return 0;
}
