// -DMPC_BOUNDS: the bounds-checked debug build of libmpcmax ("race detection / sanitizers": the reference has
// none; GPU AddressSanitizer is not available on this pool, so this build and the differential fuzzers are the memory-safety net
// of ~10k lines of hand index arithmetic into dynamic LDS and workspace sub-buffers).
//
// Every index into a carve-up of dynamic LDS or into a sub-buffer of the caller's workspace is written MPC_IDX(i, extent).  In the
// product build that is `(i)`: same code, same registers.  With -DMPC_BOUNDS it is a checked accessor: an index outside [0, extent)
// is RECORDED (first violation of the translation unit: source line, workgroup, index, extent; and a count) and replaced by 0, so the
// access itself stays inside the buffer -- no trap: a trapped wavefront takes the HIP context (and on this pool the box) with it.
// The host reads the records through mpc_bounds_check() (api.hip), which tests/conftest.py calls after every GPU test of a bounds
// build and tools/bounds_run.sh after every fuzzer: a violation fails the test with file:line through mpc_last_error_string().
//
//   MPC_EXTRA_HIPCC_FLAGS=-DMPC_BOUNDS python -m motionpriorcmax_amd.build     (tools/bounds_run.sh does this and restores the product build)
#pragma once
#include <hip/hip_runtime.h>

#ifdef MPC_BOUNDS
// per translation unit (no relocatable device code in this build): [0] violations, first one: [1] line, [2] workgroup,
// [3..4] index, [5..6] extent
static __device__ int mpc_bounds_dev[8];
__device__ __forceinline__ long long mpc_bounds_idx(long long i, long long n, int line) {
    if (i >= 0 && i < n) return i;
    if (atomicAdd(&mpc_bounds_dev[0], 1) == 0) {
        mpc_bounds_dev[1] = line; mpc_bounds_dev[2] = (int)blockIdx.x;
        mpc_bounds_dev[3] = (int)(i & 0xffffffffll); mpc_bounds_dev[4] = (int)(i >> 32);
        mpc_bounds_dev[5] = (int)(n & 0xffffffffll); mpc_bounds_dev[6] = (int)(n >> 32);
    }
    return 0;
}
#define MPC_IDX(i, n) mpc_bounds_idx((long long)(i), (long long)(n), __LINE__)
// a condition that must hold (not an index): recorded the same way, index = 0, extent = 0
#define MPC_EXPECT(cond) do { if (!(cond)) (void)mpc_bounds_idx(-1, 0, __LINE__); } while (0)

struct mpc_bounds_unit {
    const char *file;
    int (*read)(int *out8, int reset);
    mpc_bounds_unit *next;
};
void mpc_bounds_register(mpc_bounds_unit *u);      // api.hip
// one per translation unit, at file scope, after the kernels
#define MPC_BOUNDS_UNIT(file_)                                                                             \
    static int mpc_bounds_read_(int *out8, int reset) {                                                    \
        if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mpc_bounds_dev), 32) != hipSuccess) return 1;             \
        if (reset) { int z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mpc_bounds_dev), z, 32); } \
        return 0;                                                                                          \
    }                                                                                                      \
    static mpc_bounds_unit mpc_bounds_unit_ = {file_, mpc_bounds_read_, nullptr};                          \
    static const int mpc_bounds_reg_ = (mpc_bounds_register(&mpc_bounds_unit_), 0);
#else
#define MPC_IDX(i, n) (i)
#define MPC_EXPECT(cond) do { } while (0)
#define MPC_BOUNDS_UNIT(file_)
#endif
