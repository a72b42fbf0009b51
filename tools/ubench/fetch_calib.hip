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

// a 34 MB buffer (the size of the path's images at C3) written by one kernel and read by the next: producer -> consumer through
// the Infinity Cache, the pattern of k_iwe_accum -> k_contrast_march -> k_lut_accum
__global__ void k_calib_small_write(float *__restrict__ dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void k_calib_small_read(const float *__restrict__ src, float *__restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) out[0] = acc;
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
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_small_write, dim3(2048), dim3(256), 0, 0, (float *)b, (size_t)8500000, 1.0f + rep);
        hipLaunchKernelGGL(k_calib_small_read, dim3(2048), dim3(256), 0, 0, (const float *)b, (float *)a, (size_t)8500000);
    }
    (void)hipDeviceSynchronize();
    printf("{\"copy_bytes_read\": %zu, \"copy_bytes_written\": %zu, \"gather_bytes_read_payload\": %zu, \"gather_index_bytes\": %zu, \"gather_bytes_written\": %zu}\n",
           bytes, bytes, n8 * 8, n8 * 4, n8 * 8);
    return 0;
}
