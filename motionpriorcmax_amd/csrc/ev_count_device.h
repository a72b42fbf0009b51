// Capacity of the BACKWARD event buckets, exactly: how many rows of a sample lie in each (time bin, LUT strip).  The key
// does not depend on the flow (it is the event's own LUT cell, focus.py:184-191), so it can be counted before anything is
// warped, and the rows of one sample fit M records in all -- instead of reserving room for all M rows in EVERY bucket
// (round 2: B * nb * strips * M * 16 bytes = 4.7 GB at C3).  A row that ends up writing no record (zero weight, no tap in
// the image) leaves its slot unused; the fill counters still say how many records a bucket holds.
// The counting pass reads the events once (67 MB at C3).  In the fused forward it rides in spare workgroups of the KNN
// strip kernel (VALU bound, HBM idle); stand-alone it is k_ev_count (events.hip).
#pragma once
#include "common.h"

struct EvCountArgs {
    const float *events;   // [B][M][6]; null: nothing to count
    int *cap;              // [B][nb * NCS] (zeroed by the caller); behind it [B][nb * NCS] first records (ev_prefix_block)
    int B, M, nb, sp, hq, CSR, NCS;
};

#define EV_COUNT_ROWS 4096      // rows per counting workgroup (256 threads x 16)

__host__ __device__ static inline int ev_count_blocks(const EvCountArgs &a) {
    return a.events ? (a.M + EV_COUNT_ROWS - 1) / EV_COUNT_ROWS * a.B : 0;
}

EvCountArgs mpc_event_count_args(const mpc_shape *s, const float *events, void *ws);      // events.hip

#ifdef __HIPCC__
// workgroup `blk` of ev_count_blocks(a), 256 threads; s_cnt: nb * NCS ints of LDS
__device__ __forceinline__ void ev_count_block(const EvCountArgs &a, int blk, int *s_cnt) {
    const int tid = threadIdx.x;
    const int chunks = (a.M + EV_COUNT_ROWS - 1) / EV_COUNT_ROWS;
    const int b = blk / chunks, chunk = blk - b * chunks;
    const int nk = a.nb * a.NCS;
    for (int i = tid; i < nk; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const float inv_CSR = 1.f / (float)a.CSR;
    const int r0 = chunk * EV_COUNT_ROWS, r1 = min(r0 + EV_COUNT_ROWS, a.M);
    // columns 0 (y) and 4 (bin) of the row: two 8-byte loads of the three that make up a row; eight rows of a thread in
    // flight together (the workgroup holds a slot of the kernel it rides in: it has to be short)
    for (int i0 = r0 + tid; i0 < r1; i0 += 8 * 256) {
        float2 ya[8], bn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float2 *row = reinterpret_cast<const float2 *>(a.events + ((size_t)b * a.M + min(i0 + u * 256, r1 - 1)) * 6);
            ya[u] = row[0]; bn[u] = row[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u * 256 >= r1) continue;
            // the cell exactly as warp_cell (events.hip) computes it
            int it = (int)bn[u].x;
            int iy = (int)floorf(mpc_div_sp(ya[u].x, a.sp));
            it = min(max(it, 0), a.nb - 1);
            iy = min(max(iy, 0), a.hq - 1);
            const int cst = (int)(((float)iy + 0.5f) * inv_CSR);
            atomicAdd(&s_cnt[it * a.NCS + cst], 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < nk; i += 256) {
        const int c = s_cnt[i];
        if (c) atomicAdd(&a.cap[(size_t)b * nk + i], c);
    }
}

// first record of every backward bucket of sample b = exclusive prefix of its capacities (once the counting is complete:
// a later kernel).  256 threads; s_tmp: 4 ints of LDS
__device__ __forceinline__ void ev_prefix_block(const EvCountArgs &a, int b, int *s_tmp) {
    const int tid = threadIdx.x, nk = a.nb * a.NCS;
    const int *cap = a.cap + (size_t)b * nk;
    int *start = a.cap + (size_t)a.B * nk + (size_t)b * nk;
    const int per = (nk + 255) >> 8;
    const int i0 = min(tid * per, nk), i1 = min(i0 + per, nk);
    int local = 0;
    for (int i = i0; i < i1; ++i) local += cap[i];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) s_tmp[tid >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (tid >> 6); ++w) run += s_tmp[w];
    for (int i = i0; i < i1; ++i) { start[i] = run; run += cap[i]; }
    __syncthreads();
}
#endif
