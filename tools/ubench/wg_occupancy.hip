// How many 1024-thread (and 512-thread) workgroups does a CU of gfx950 hold at once as a function of their dynamic LDS?
// A workgroup spins for ~10 us of wall clock; 1470 workgroups; the kernel's duration / 10 us = rounds = 1470 / (256 x per CU).
// (k_lut_accum<ordered> asks for 69 KB with 1024 threads and was found to run ONE workgroup per CU -- stamps of
// tools/lut_accum_stamp_probe.py -- although two fit the 160 KB.)   hipcc --offload-arch=gfx950 -O2 wg_occupancy.hip -o wg_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NT>
__global__ __launch_bounds__(NT) void k_spin(unsigned long long *out, int ticks) {
    extern __shared__ unsigned long long s_buf[];
    if (threadIdx.x == 0) s_buf[0] = wall_clock64();
    __syncthreads();
    const unsigned long long t0 = s_buf[0];
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}

template <int NT>
static void run(size_t lds, int nwg) {
    (void)hipFuncSetAttribute((const void *)k_spin<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    unsigned long long *d;
    (void)hipMalloc(&d, nwg * sizeof(unsigned long long));
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spin<NT>, NT, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_spin<NT>, dim3(nwg), dim3(NT), lds, 0, d, 1000);      // 1000 ticks of 10 ns
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nwg);
    (void)hipMemcpy(h.data(), d, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull;
    for (auto v : h) if (v < mn) mn = v;
    int first = 0;
    for (auto v : h) if (v - mn < 100) ++first;                 // started within the first microsecond
    printf("threads %4d  LDS %6zu B: occupancy API %d per CU; %d of %d workgroups start in the first us (= %.2f per CU); kernel %.1f us = %.2f rounds of 10 us\n",
           NT, lds, occ, first, nwg, first / 256.0, best * 1e3f, best * 1e3f / 10.f);
    (void)hipFree(d);
}

int main() {
    const int nwg = 1470;
    for (size_t lds : {(size_t)8192, (size_t)32768, (size_t)49152, (size_t)65536, (size_t)69120, (size_t)75 * 1024, (size_t)80 * 1024 - 512, (size_t)81920}) run<1024>(lds, nwg);
    for (size_t lds : {(size_t)32768, (size_t)49152, (size_t)53248, (size_t)65536, (size_t)69120, (size_t)80 * 1024 - 512}) run<512>(lds, nwg);
    for (size_t lds : {(size_t)16384, (size_t)26624, (size_t)32768, (size_t)40960}) run<256>(lds, nwg * 4);
    return 0;
}
