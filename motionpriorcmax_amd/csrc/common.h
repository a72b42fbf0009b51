// Shared host/device helpers for libmpcmax (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/mpcmax.h"
#include "tuning.h"

#define MPC_WAVE 64
#define MPC_KNN_LDS_SORT_CELLS (150 * 1024 / 4)   // largest LUT grid the single-workgroup LDS counting sort holds

void mpc_set_error(const char *fmt, ...);
// Zero `bytes` bytes (a multiple of 4, 4-byte aligned) on `stream` with a kernel.  Used instead of hipMemsetAsync:
// memset NODES made repeated replays of a captured HIP graph fault on ROCm 7.2 (tools/graph_probe2.py).
int mpc_zero_async(void *ptr, size_t bytes, hipStream_t stream);

#define MPC_CHECK_ARG(cond, code, msg)                                   \
    do {                                                                 \
        if (!(cond)) {                                                   \
            mpc_set_error("%s: %s", __func__, msg);                      \
            return (code);                                               \
        }                                                                \
    } while (0)

#define MPC_CHECK_LAUNCH()                                               \
    do {                                                                 \
        hipError_t e__ = hipGetLastError();                              \
        if (e__ != hipSuccess) {                                         \
            mpc_set_error("%s: %s", __func__, hipGetErrorString(e__));   \
            return (int)e__;                                             \
        }                                                                \
    } while (0)

// Every kernel launch of the library goes through MPC_LAUNCH: the launch itself, and -- only while the diagnostics timer of
// mpc_profile_start() / mpc_profile_stop() is on (bench.py's instrumented pass; never inside a graph capture) -- a HIP event
// before and after it ON THE LAUNCH STREAM, so that the duration of every kernel is measured live, per launch.
bool mpc_prof_on();
void mpc_prof_pre(hipStream_t st);
void mpc_prof_post(const char *name, hipStream_t st);
#define MPC_LAUNCH(kern, grid, block, lds, st, ...)                                  \
    do {                                                                             \
        const bool pr__ = mpc_prof_on();                                             \
        if (pr__) mpc_prof_pre(st);                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                 \
        if (pr__) mpc_prof_post(#kern, st);                                          \
    } while (0)

// One-time, idempotent set-up that is PER DEVICE (raising the dynamic-LDS cap of a kernel): one bit per HIP device.
// need() is true until mark() ran for the current device; two threads (forward and autograd thread) may both run the
// set-up, which is harmless because it is idempotent -- the flag itself is atomic.
struct mpc_device_once {
    std::atomic<uint64_t> done{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((done.load(std::memory_order_acquire) >> dev()) & 1ull); }
    void mark() { done.fetch_or(1ull << dev(), std::memory_order_release); }
};

static inline int64_t mpc_align(int64_t x, int64_t a = 256) { return (x + a - 1) / a * a; }
static inline int mpc_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- workspace layout -------------------------------------------------------------------
// One layout function shared by every entry point, so that forward and backward calls agree.
struct mpc_ws_layout {
    // contrast / smoothness partial sums (fp64 pairs)
    int64_t off_cpart;   int32_t n_cblocks;     // [n_cblocks][2] double
    int64_t off_spart;   int32_t n_sblocks_max; // [n_sblocks_max][2] double
    int64_t off_counts;                         // int32[8] : n_sblocks used, ...
    // KNN
    int64_t off_cell_start;  // uint16 [B*nb][Gb+1]   first bucketed point of every cell of the bucket grid (query grid + margin)
    int64_t off_knn_sat;     // uint16 [B*nb][hb+1][wb+1]  summed-area table of the cell counts (strip forward only)
    int64_t off_spos;        // float2 [B*nb][n]
    int64_t off_sidx;        // uint16 [B*nb][n]
    int64_t off_knn_tmp_g;   // float2 [B*nb][n][T]  backward partials
    int64_t off_knn_tmp_a;   // float2 [B*nb][n]
    int64_t off_knn_cursor;  // int32  [B*nb][Gb]  fill cursors of the global-memory bucket sort (only when Gb*4 B exceeds the LDS sort)
    int64_t off_knn_reach;   // float  [B*nb][tiles of the bucket grid]  backward search reach per 16x16 tile
    int64_t off_knn_fail;    // int32  [1 + B*nb*G]  queries handed from the strip kernel to the fallback kernel
    int64_t off_knn_retry;   // int32  [1 + B*nb*ceil(wq/2)*ceil(hq/128)]  strips the strip kernel searches again in quarters (staging overflow)
    int64_t off_knn_farstrip; // int32 [1 + strips]  strips that hold far queries (k_knn_tail)
    int64_t off_knn_ftlist;  // int32 [1 + B*nb*tiles]  (sample, bin, tile) work items of k_knn_bwd_far;  off_knn_ftbits: uint32 [B*nb][ceil(tiles/32)] the same as bits
    int64_t off_knn_ftbits;
    int64_t off_knn_chord;   // uint8 [21][21]  chord table of the strip kernels (knn_device.h: knn_chord_cells)
    int64_t off_knn_again;   // uint32 [2][B*nb][hq][ceil(wq/32)]  queries the strip kernel's main launch hands to its second launch; those that need more rings
    int64_t off_knn_far;     // int32  [B*nb][1 + G]  per (sample, bin): the queries the fallback kernel served, for k_knn_bwd_far
    // event partition (LDS-tiled path)
    int64_t off_fcount;      // int32 [nfb + nbb + 8] bucket fill counters, marker; then [nbb] capacities and [nbb] first records of the backward buckets
    int64_t off_frec;        // float4 [nfb][fcap]
    int64_t off_brec;        // float4 [B][bcap = M]: the backward buckets of a sample back to back, each as large as its count of rows
    int32_t P, nimg, G;
    int32_t strip_rows, n_strips;   // destination strips of the IWE (forward buckets)
    int32_t cstrip_rows, n_cstrips; // source strips of LUT cell rows (backward buckets)
    int32_t nfb, nbb, fcap, bcap;
    int32_t b_exact;          // backward buckets sized by the counting pass (bcap = records of a SAMPLE) instead of M per bucket
    int64_t total;
};

// contrast tiles
#define MPC_CT_W 64
#define MPC_CF_TW 56   // fused kernel: tile + 2*4 halo columns = 64 = one wavefront row
// smoothness tiles (LUT cells)
#define MPC_SM_H 16
#define MPC_SM_W 60   // + 2*2 halo cells = 64 = one wavefront row

mpc_ws_layout mpc_layout(const mpc_shape *s);
// internal variants of two entry points used by mpc_focus_fwd (api.hip): the KNN forward's first kernel zeroes the event
// bucket counters, so that the event forward can skip its own zeroing launch
int mpc_knn_lut_fwd_ex(const mpc_shape *s, const float *traj, float *flow_lut, float *flow_next, float *knn_state,
                       int32_t *idx_out, void *ws, void *stream, int zero_event_counters, const float *events, int *done);
int mpc_event_splat_bwd_job(const mpc_shape *s, const float *events, const int32_t *offsets, const float *flow_lut,
                            const float *t_ref, const float *grad_iwe, const float *scal,
                            const float *grad_out, float *grad_flow_lut, const float *add_term,
                            void *ws, void *stream, const float *knn_state, int *reach_done);
int mpc_knn_lut_bwd_ex(const mpc_shape *s, const float *traj, const float *grad_flow_lut, const float *grad_flow_next,
                       const float *knn_state, float *grad_traj, void *ws, void *stream, int reach_ready,
                       const float *gnext_scale, float *gnext_scratch);
int mpc_event_splat_fwd_ex(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                           float *iwe_raw, void *ws, void *stream, int counters_zeroed, const int32_t *offsets);
int mpc_validate_shape(const mpc_shape *s);
int mpc_finalize_ex(const mpc_shape *s, int32_t smooth_nimg, int32_t smooth_C, float smooth_weight, float *scal, float *scal_out,
                    void *ws, void *stream);
// KNN (knn.hip): margin of the bucket grid; is the points' counting sort the global-memory one; does the forward keep a far list
int mpc_knn_margin(const mpc_shape *s);
int mpc_knn_tiles(const mpc_shape *s);          // 16 x 16 cell tiles of the bucket grid per (sample, bin)
bool mpc_knn_big_sort(const mpc_shape *s);
bool mpc_knn_uses_far_list(const mpc_shape *s);

// ---- device helpers ---------------------------------------------------------------------
#ifdef __HIPCC__
// v / sp in fp32 exactly as the division gives it, without the ~10-instruction IEEE division sequence when sp is a power of
// two (the shipped configurations: 4): the reciprocal is then exact and the product is the same scaling of the exponent.
// The branch is uniform (sp is a launch constant).
__device__ __forceinline__ float mpc_div_sp(float v, int sp) {
    return ((sp & (sp - 1)) == 0) ? v * (1.f / (float)sp) : v / (float)sp;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Sum of `v` over a workgroup of NT threads (NT multiple of 64); result valid in thread 0.
template <int NT>
__device__ __forceinline__ double block_sum_d(double v, double *lds /* NT/64 doubles */) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) lds[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += lds[i];
    }
    __syncthreads();
    return r;
}
#endif
