// Calibration of the FETCH_SIZE / WRITE_SIZE counters on gfx950: kernels that read (and write) a KNOWN number of bytes with
// 4-, 8- and 16-byte accesses per lane.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes);
// tools/summarize_pmc.py divides the counter by the bytes printed here to get the correction factor per access width.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T>
__global__ void k_calib_copy(const T *__restrict__ src, T *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// gather of 8-byte words at a random permutation (one access per lane, a cache line each): the narrow-gather case
__global__ void k_calib_gather8(const float2 *__restrict__ src, const int *__restrict__ idx, float2 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

int main() {
    const size_t bytes = 1ull << 30;          // 1 GiB: beyond the 256 MiB Infinity Cache
    void *a, *b; int *idx;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    const size_t n8 = bytes / 8 / 4;          // gather: 256 MiB of 8-byte words, random order
    (void)hipMalloc(&idx, n8 * 4);
    {   // a multiplicative permutation of [0, n8): n8 is a power of two, odd multiplier
        int *h = (int *)malloc(n8 * 4);
        for (size_t i = 0; i < n8; ++i) h[i] = (int)((i * 2654435761ull) & (n8 - 1));
        (void)hipMemcpy(idx, h, n8 * 4, hipMemcpyHostToDevice); free(h);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_copy<float>, dim3(8192), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
        hipLaunchKernelGGL(k_calib_copy<float2>, dim3(8192), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, bytes / 8);
        hipLaunchKernelGGL(k_calib_copy<float4>, dim3(8192), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16);
        hipLaunchKernelGGL(k_calib_gather8, dim3(8192), dim3(256), 0, 0, (const float2 *)a, idx, (float2 *)b, n8);
    }
    (void)hipDeviceSynchronize();
    printf("{\"copy_bytes_read\": %zu, \"copy_bytes_written\": %zu, \"gather_bytes_read_payload\": %zu, \"gather_index_bytes\": %zu, \"gather_bytes_written\": %zu}\n",
           bytes, bytes, n8 * 8, n8 * 4, n8 * 8);
    return 0;
}
