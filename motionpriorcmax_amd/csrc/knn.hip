// Exact K-nearest-neighbour flow look-up table and its backward.
//   forward : reference src/losses/focus.py:115-180 (KeOps argKmin/Kmin + gather + mean/iwd)
//   backward: scatter of dLUT/K (or the iwd weights) back to the K neighbours (autograd of the
//             gather at focus.py:145-163,170-176), restated as a GATHER per trajectory point.
//
// The reference evaluates all n x Q pair distances.  Here the n trajectory points of one
// (sample, bin) are bucketed into the LUT cells (cell centres == query points), each query
// searches a growing square of cells until K points lie provably closer than anything outside,
// and selects the K smallest (distance, index) keys with a per-thread radix histogram.  The
// result is the exact K-nearest set; ties resolve to the lowest index.
// No index tensor is materialised: the backward re-derives membership from the saved K-th key.
#include "common.h"
#include "knn_device.h"
#include "ev_count_device.h"
#include "bounds.h"
#include "diag/stamps.h"
#include <stdlib.h>


// ------------------------------------------------------------------------------------------
// bucket the points of one (sample, bin) by cell: counting sort in LDS.  With CACHED, every thread
// keeps the cells of its (up to KNN_BUCKET_NPT) points in registers: one round of global-load latency for the
// whole kernel instead of one per point and pass.  The kernel is latency- not bandwidth-limited, so the cell
// rows of a (sample, bin) are split over S workgroups: each reads all n points, counts those below its row
// range (its base offset in the bucketed arrays) and sorts the ones inside it -- no exchange between them.
// The cells are those of the BUCKET grid (knn_device.h: the query grid + a margin of p.m cells).
// grid B*nb*S, 1024 threads, dynamic LDS = ceil(hb/S)*wb * 4 + n * 2 bytes (+ n / 8 for the bitmap of a crowded cell)
// ------------------------------------------------------------------------------------------
#define KNN_BUCKET_NPT 24
#define KNN_BK_SMALL 6        // cells with up to this many points are ordered by their own thread (insertion sort)
#define KNN_BK_WAVE 128       // ... up to this many by one wavefront (bitonic network, two keys per lane); more: by the workgroup (bitmap)

// ascending bitonic sort of 128 keys held two per lane (k0 = element `lane`, k1 = element `lane + 64`)
__device__ __forceinline__ void knn_bitonic128(unsigned &k0, unsigned &k1, int lane) {
#pragma unroll
    for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride == 64) {
                // partners are the two keys of a lane; at size 128 every pair sorts ascending
                const unsigned lo = min(k0, k1), hi = max(k0, k1);
                k0 = lo; k1 = hi;
            } else {
                // element e (lane for k0, lane + 64 for k1): partner e ^ stride in the same half; ascending block <=> (e & size) == 0
                const unsigned p0 = (unsigned)__shfl_xor((int)k0, stride, 64), p1 = (unsigned)__shfl_xor((int)k1, stride, 64);
                const bool lower = (lane & stride) == 0;
                const bool asc0 = (lane & size) == 0, asc1 = ((lane + 64) & size) == 0;
                k0 = (lower == asc0) ? min(k0, p0) : max(k0, p0);
                k1 = (lower == asc1) ? min(k1, p1) : max(k1, p1);
            }
        }
    }
}

// NPT: points per thread held in registers (CACHED), a multiple of 4 >= ceil(n / 1024): 20 for the 19 200 points of a DSEC grid
template <bool CACHED, int NPT>
__global__ __launch_bounds__(1024) void k_knn_bucket(const KnnParams p, const float *__restrict__ traj,
                                                     knn_cs_t *__restrict__ cell_start, knn_cs_t *__restrict__ sat,
                                                     float2 *__restrict__ spos, knn_idx_t *__restrict__ sidx, int S,
                                                     float *__restrict__ tile_dkmax, int ntiles, const KnnLists ls,
                                                     int *__restrict__ zero_ptr, int zero_words) {
    extern __shared__ int s_cnt[];
    __shared__ int s_wave[16];
    __shared__ int s_low[16];
    __shared__ int s_nbig, s_nhuge;
    __shared__ int s_huge[32];
    __shared__ int s_big[1024];          // crowded cells of this workgroup (more than KNN_BK_SMALL points); the rest of them re-scan
    const int tid = threadIdx.x;
    BK_STAMP_DECL
    BK_STAMP(0);
    const int bt = blockIdx.x / S, part = blockIdx.x - bt * S, b = bt / p.nb, t = bt - b * p.nb;
    // set-up for the strip query kernel, which follows on the stream: its per-tile maxima are accumulated with
    // atomicMax and its fallback lists are appended to (knn_strip.hip)
    if (part == 0) for (int i = tid; i < ntiles * KNN_NCLS; i += 1024) tile_dkmax[(size_t)bt * ntiles * KNN_NCLS + i] = 0.f;
    // ... and the work lists of the forward start empty (knn_device.h: KnnLists)
    if (part == 0 && ls.far != nullptr) {
        if (tid == 0) ls.far[(size_t)bt * (p.G + 1)] = 0;
        for (int i = tid; i < ls.ftwords; i += 1024) ls.ftbits[(size_t)bt * ls.ftwords + i] = 0u;
    }
    if (part == 0 && ls.again != nullptr) for (int i = tid; i < ls.again_words; i += 1024) { ls.again[(size_t)bt * ls.again_words + i] = 0u; ls.grow[(size_t)bt * ls.again_words + i] = 0u; }
    if (blockIdx.x == 0 && tid == 0) { ls.fail[0] = 0; ls.retry[0] = 0; ls.farstrip[0] = 0; if (ls.ftlist) ls.ftlist[0] = 0; *knn_marked_count(ls) = 0; *knn_late_count(ls) = 0; *knn_tail_done(ls) = 0; }
    if (blockIdx.x == 0 && tid < (KNN_RFAR + 1) * (KNN_RFAR + 1))      // (the chord table of the strip kernels' row tables)
        ls.chord[tid] = (unsigned char)max(knn_chord_cells(tid / (KNN_RFAR + 1), tid % (KNN_RFAR + 1), p.sp, p.l1 != 0), 0);
    if (blockIdx.x == 0) for (int i = tid; i < zero_words; i += 1024) zero_ptr[i] = 0;      // (mpc_focus_fwd: the event bucket counters)
    const int rows_per = (p.hb + S - 1) / S;
    const int g_lo = min(part * rows_per, p.hb) * p.wb, g_hi = min((part + 1) * rows_per, p.hb) * p.wb, Gp = g_hi - g_lo;
    const float2 *pts = reinterpret_cast<const float2 *>(traj) + ((size_t)b * (p.T + p.nb) + p.T + t) * p.n;
    int qc[CACHED ? NPT : 1];
    int below = 0;                                  // points of this thread in cells before the range
    if (CACHED) {
        float2 q[NPT];
#pragma unroll
        // (slots beyond n read point 0, which always exists: n >= K >= 1; reading pts[tid] there ran past the end
        // of the trajectory tensor for the last (sample, bin) whenever n < 1024)
        for (int u = 0; u < NPT; ++u) { const int i = tid + u * 1024; q[u] = pts[i < p.n ? i : 0]; }
        // (skipping the rounds beyond the last point by a wave-uniform test instead of a compile-time count was measured:
        // 41.7 -> 45.5 us, the branches break up the batch of loads)
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            qc[u] = knn_cell_index(p, q[u].x, q[u].y);
            if (tid + u * 1024 >= p.n) qc[u] = 0x7fffffff;          // not a point
            below += qc[u] < g_lo;
        }
    }
    BK_STAMP(1);
    for (int g = tid; g < Gp; g += 1024) s_cnt[g] = 0;
    if (tid == 0) { s_nbig = 0; s_nhuge = 0; }
    __syncthreads();
    if (CACHED) {
#pragma unroll
        for (int u = 0; u < NPT; ++u)
            if (qc[u] >= g_lo && qc[u] < g_hi) atomicAdd(&s_cnt[MPC_IDX(qc[u] - g_lo, Gp)], 1);
    } else {
        for (int i = tid; i < p.n; i += 1024) {
            const float2 v = pts[i];
            const int c = knn_cell_index(p, v.x, v.y);
            below += c < g_lo;
            if (c >= g_lo && c < g_hi) atomicAdd(&s_cnt[MPC_IDX(c - g_lo, Gp)], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_down(below, o, 64);
    if ((tid & 63) == 0) s_low[tid >> 6] = below;
    __syncthreads();
    BK_STAMP(2);
    int base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) base += s_low[w];
    // exclusive scan over the Gp counters: each thread owns a contiguous chunk
    const int chunk = (Gp + 1023) / 1024;
    const int g0 = min(tid * chunk, Gp), g1 = min(g0 + chunk, Gp);
    int local = 0;
    for (int g = g0; g < g1; ++g) local += s_cnt[g];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (tid >> 6); ++w) wave_off += s_wave[w];
    int run = wave_off + incl - local;
    for (int g = g0; g < g1; ++g) {
        const int c = s_cnt[g];
        s_cnt[g] = run;        // becomes the fill cursor of the cell (local to the range)
        run += c;
    }
    __syncthreads();
    BK_STAMP(3);
    knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    for (int g = tid; g < Gp; g += 1024) cs[MPC_IDX(g_lo + g, p.Gb + 1)] = base + s_cnt[g];      // coalesced
    if (tid == 0 && part == S - 1) cs[p.Gb] = p.n;
    if (sat != nullptr && S == 1) {
        // the summed-area table of the cell counts while the first points of all cells are in LDS (one workgroup holds the
        // whole (sample, bin); otherwise k_knn_sat follows); the scratch is the index array's space, not yet in use
        knn_sat_build(p, [&](int y, int x) { const int g = y * p.wb + x; return g < Gp ? s_cnt[g] : p.n; },
                      sat + (size_t)bt * (p.hb + 1) * (p.wb + 1), s_cnt + Gp);
    }
    __syncthreads();
    BK_STAMP(4);
    float2 *sp_ = spos + (size_t)bt * p.n + base;
    knn_idx_t *si_ = sidx + (size_t)bt * p.n + base;
    // scatter the INDICES into LDS first: the slots inside a cell are handed out in the order the LDS atomics
    // happen to execute, so the (few) points of every cell are then ordered by trajectory index -- the bucket
    // order, and with it the fp32 summation order of the LUT, is the same in every run -- and only then are
    // the bucketed arrays written, with coalesced stores and the positions gathered by index.
    unsigned short *l_idx = reinterpret_cast<unsigned short *>(s_cnt + Gp);
    if (CACHED) {
#pragma unroll
        for (int u = 0; u < NPT; ++u)
            if (qc[u] >= g_lo && qc[u] < g_hi) l_idx[MPC_IDX(atomicAdd(&s_cnt[MPC_IDX(qc[u] - g_lo, Gp)], 1), p.n)] = (unsigned short)(tid + u * 1024);
    } else {
        for (int i = tid; i < p.n; i += 1024) {
            const float2 v = pts[i];
            const int c = knn_cell_index(p, v.x, v.y);
            if (c >= g_lo && c < g_hi) l_idx[MPC_IDX(atomicAdd(&s_cnt[MPC_IDX(c - g_lo, Gp)], 1), p.n)] = (unsigned short)i;
        }
    }
    __syncthreads();
    BK_STAMP(5);
    // Order the points of every cell by trajectory index.  A cell of the lattice holds about one point and its own thread
    // sorts it by insertion; a smooth flow field packs points (a contracting field: several per cell; the outermost ring of
    // the margin: everything that left the image by more than the margin, a hundred per cell) -- cells with more than
    // KNN_BK_SMALL points go on a list and are sorted by a whole wavefront (bitonic network, up to KNN_BK_WAVE keys), the few
    // with more than that by the whole workgroup (a bitmap over the trajectory indices: any number of keys).
    // (Round 5 measured the ordering PER POINT instead -- a point's place = the indices of its cell below its own, up to
    // KNN_BK_SMALL independent LDS reads -- : slowest workgroup 10.6 us in this phase instead of 13.2, but the mean 7.3 instead
    // of 6.2 and 20 spilled registers in the phase before: launch 53.8 us against 50.4.  Not kept; profiles/HISTORY_r05.md.)
    for (int g = tid; g < Gp; g += 1024) {          // s_cnt[g] is now the END of cell g
        const int e = s_cnt[g], a = g ? s_cnt[g - 1] : 0;
        if (e - a > KNN_BK_SMALL) {
            const int k = atomicAdd(&s_nbig, 1);
            if (k < 1024) s_big[k] = g;
            continue;
        }
        for (int i = a + 1; i < e; ++i) {
            const unsigned short key = l_idx[i];
            int j = i - 1;
            while (j >= a && l_idx[j] > key) { l_idx[MPC_IDX(j + 1, p.n)] = l_idx[j]; --j; }
            l_idx[MPC_IDX(j + 1, p.n)] = key;
        }
    }
    __syncthreads();
    {
        const int nbig = s_nbig;
        const int lane = tid & 63, wv = tid >> 6;
        // (more crowded cells than the list holds: every wavefront scans its share of the cells again instead)
        const int nitem = nbig <= 1024 ? nbig : Gp;
        // cells with up to 16 points -- nearly all crowded cells of a smooth field (an expansion by 45 %: up to 300 of them per
        // (sample, bin), a handful above 16) --: four at a time per wavefront, a quarter of the lanes each; a key's place is the
        // number of smaller keys of its cell (indices are distinct), 16 lane reads.  (The 128-key network below for every one of
        // them: 19 cells per wavefront one after the other, k_knn_bucket 98 us on that field against 49 on white noise.)
        for (int base = 4 * wv; base < nitem; base += 64) {
            const int it = base + (lane >> 4), l16 = lane & 15;
            int a = 0, c = 0;
            if (it < nitem) {
                const int g = nbig <= 1024 ? s_big[it] : it;
                a = g ? s_cnt[g - 1] : 0; c = s_cnt[g] - a;
            }
            const bool mine = c > KNN_BK_SMALL && c <= 16 && l16 < c;
            const unsigned key = mine ? (unsigned)l_idx[a + l16] : 0xffffffffu;
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) rank += (unsigned)__shfl((int)key, (lane & 48) | j, 64) < key ? 1 : 0;
            if (mine) l_idx[MPC_IDX(a + rank, p.n)] = (unsigned short)key;
        }
        for (int it = wv; it < nitem; it += 16) {
            const int g = nbig <= 1024 ? s_big[it] : it;
            const int e = s_cnt[g], a = g ? s_cnt[g - 1] : 0, c = e - a;
            if (c <= 16) continue;
            if (c > KNN_BK_WAVE) { if (lane == 0) { const int k = atomicAdd(&s_nhuge, 1); if (k < 32) s_huge[k] = g; } continue; }
            unsigned k0 = lane < c ? (unsigned)l_idx[a + lane] : 0xffffffffu;
            unsigned k1 = lane + 64 < c ? (unsigned)l_idx[a + 64 + lane] : 0xffffffffu;
            knn_bitonic128(k0, k1, lane);
            if (lane < c) l_idx[MPC_IDX(a + lane, p.n)] = (unsigned short)k0;
            if (lane + 64 < c) l_idx[MPC_IDX(a + 64 + lane, p.n)] = (unsigned short)k1;
        }
    }
    __syncthreads();
    if (s_nhuge > 0) {
        // cells beyond the wavefront sort, one at a time by the whole workgroup: mark the indices of the cell in a bitmap over
        // [0, n), then every word's bits go back in ascending order behind the bits of the words before it
        unsigned *bm = reinterpret_cast<unsigned *>(l_idx + ((p.n + 1) & ~1));
        const int nw = (p.n + 31) >> 5;
        const int nhuge = s_nhuge;
        const int nh_item = nhuge <= 32 ? nhuge : Gp;          // (more than the list holds: look at every cell)
        for (int hi_ = 0; hi_ < nh_item; ++hi_) {
            const int g = nhuge <= 32 ? s_huge[hi_] : hi_;
            const int e = s_cnt[g], a = g ? s_cnt[g - 1] : 0, c = e - a;
            if (c <= KNN_BK_WAVE) continue;                   // (workgroup-uniform)
            for (int w = tid; w < nw; w += 1024) bm[w] = 0u;
            __syncthreads();
            for (int i = a + tid; i < e; i += 1024) atomicOr(&bm[MPC_IDX(l_idx[MPC_IDX(i, p.n)] >> 5, nw)], 1u << (l_idx[i] & 31));
            __syncthreads();
            // prefix of the word popcounts: thread `tid` owns words [w0, w1)
            const int per = (nw + 1023) / 1024, w0 = min(tid * per, nw), w1 = min(w0 + per, nw);
            int mine = 0;
            for (int w = w0; w < w1; ++w) mine += __popc(bm[w]);
            int inc2 = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc2, o, 64); if ((tid & 63) >= o) inc2 += v; }
            if ((tid & 63) == 63) s_wave[tid >> 6] = inc2;
            __syncthreads();
            int pos = a + inc2 - mine;
            for (int w = 0; w < (tid >> 6); ++w) pos += s_wave[w];
            for (int w = w0; w < w1; ++w) {
                unsigned bits = bm[w];
                while (bits) { const int bit = __ffs(bits) - 1; bits &= bits - 1u; l_idx[MPC_IDX(pos, p.n)] = (unsigned short)(32 * w + bit); ++pos; }
            }
            __syncthreads();
        }
    }
    BK_STAMP(6);
    const int own = Gp > 0 ? s_cnt[Gp - 1] : 0;
    // (all gathers of a thread in flight together -- one batch of NPT with the points in registers, eight otherwise: one after
    // the other they were ~19 dependent L2 round trips, in batches of eight three)
    constexpr int GB = CACHED ? NPT : 8;
    for (int sl0 = tid; sl0 < own; sl0 += GB * 1024) {
        int i[GB];
        float2 v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) i[u] = (sl0 + u * 1024 < own) ? (int)l_idx[sl0 + u * 1024] : 0;
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = pts[MPC_IDX(i[u], p.n)];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int sl = sl0 + u * 1024;
            if (sl < own) { si_[MPC_IDX(base + sl, p.n) - base] = i[u]; sp_[sl] = v[u]; }
        }
    }
    BK_STAMP_WRITE(ls, tid);
}

// ------------------------------------------------------------------------------------------
// Summed-area table of the cell counts of one (sample, bin): sat[y][x] = points in cells (y', x') of the bucket grid with
// y' < y and x' < x, (hb + 1) x (wb + 1) entries.  The strip kernel and its fallback read the number of points in a query's
// search square from it (four loads) to choose the radius of the square.
// Built from the first bucketed point of every cell (rowstart(y, x), x in [0, wb]: x = wb is one past the row), whose
// differences along a row are the row prefixes; the sum over the rows in two levels: groups of 8 rows first (s_part), then
// every (column, group) item adds up its own rows behind the groups above it.  All threads of the workgroup call.
// ------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void knn_sat_build(const KnnParams &p, F rowstart, knn_cs_t *__restrict__ S, int *s_part, int x0 = 0, int ncol = -1) {
    // (columns [x0, x0 + ncol) of the table: the columns are independent of each other -- k_knn_sat splits them over workgroups)
    const int W1 = p.wb + 1, ngrp = (p.hb + 7) >> 3, nthr = blockDim.x;
    if (ncol < 0) ncol = W1;
    ncol = min(ncol, W1 - x0);
    const int nitem = ngrp * max(ncol, 0);
    // every (column, group of 8 rows) item keeps its 8 row prefixes in registers between the two phases; all 16 loads of an
    // item are in flight together (from global memory a dependent chain of them was the whole 18 us of the kernel at B = 1)
    constexpr int ITEMS = 4;                           // items per thread held in registers (more: a second round re-reads)
    int pre[ITEMS][8];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int it = threadIdx.x + u * nthr;
        const int g = it / ncol, x = x0 + (it - g * ncol);
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int y = g * 8 + k;
            const bool on = it < nitem && y < p.hb;
            const int a = on ? rowstart(y, x) : 0, z = on ? rowstart(y, 0) : 0;
            pre[u][k] = a - z;
            sum += pre[u][k];
        }
        if (it < nitem) s_part[MPC_IDX(it, nitem)] = sum;
    }
    for (int it = threadIdx.x + ITEMS * nthr; it < nitem; it += nthr) {        // (grids beyond ITEMS x threads items)
        const int g = it / ncol, x = x0 + (it - g * ncol);
        int sum = 0;
        for (int y = g * 8; y < min(g * 8 + 8, p.hb); ++y) sum += rowstart(y, x) - rowstart(y, 0);
        s_part[it] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int it = threadIdx.x + u * nthr;
        if (it >= nitem) continue;
        const int g = it / ncol, xl = it - g * ncol, x = x0 + xl;
        int acc = 0;
        for (int gg = 0; gg < g; ++gg) acc += s_part[gg * ncol + xl];
        if (g == 0) S[x] = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int y = g * 8 + k;
            if (y < p.hb) { acc += pre[u][k]; S[MPC_IDX((size_t)(y + 1) * W1 + x, (p.hb + 1) * W1)] = acc; }
        }
    }
    for (int it = threadIdx.x + ITEMS * nthr; it < nitem; it += nthr) {
        const int g = it / ncol, xl = it - g * ncol, x = x0 + xl;
        int acc = 0;
        for (int gg = 0; gg < g; ++gg) acc += s_part[gg * ncol + xl];
        for (int y = g * 8; y < min(g * 8 + 8, p.hb); ++y) {
            acc += rowstart(y, x) - rowstart(y, 0);
            S[(size_t)(y + 1) * W1 + x] = acc;
        }
    }
}
// from cell_start in global memory (the bucket sorts that do not hold a whole (sample, bin) in one workgroup's LDS)
// grid (B * nb, column chunks), 256 threads, one item per thread: KNN_SAT_COLS(hb) columns per workgroup (the launch only runs
// at small batches, where B * nb workgroups of 1 024 threads with three items each left the chip idle behind 15 chains of loads:
// 14.5 us at B = 1)
#define KNN_SAT_NT 256
__host__ __device__ static inline int knn_sat_cols(int hb) { const int c = KNN_SAT_NT / ((hb + 7) >> 3); return c < 1 ? 1 : c; }
__global__ __launch_bounds__(KNN_SAT_NT) void k_knn_sat(const KnnParams p, const knn_cs_t *__restrict__ cell_start, knn_cs_t *__restrict__ sat) {
    extern __shared__ int s_part[];
    const int bt = blockIdx.x, ncol = knn_sat_cols(p.hb);
    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    knn_sat_build(p, [&](int y, int x) { return cs[y * p.wb + x]; }, sat + (size_t)bt * (p.hb + 1) * (p.wb + 1), s_part, (int)blockIdx.y * ncol, ncol);
}


// ------------------------------------------------------------------------------------------
// LUT grids beyond the LDS sort (Gb > 38 400 cells, e.g. 1280x720 at superpixel 4): the same counting sort with
// the counters in global memory, as three launches (count / scan / scatter) and a fourth that orders the
// points of every cell by trajectory index, so that the bucket order -- and with it the fp32 summation order
// of the LUT -- does not depend on the order the atomics happened to execute in.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_bucket_count(const KnnParams p, const float *__restrict__ traj,
                                                          int *__restrict__ cursor) {
    const int bt = blockIdx.y, b = bt / p.nb, t = bt - b * p.nb;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float2 v = reinterpret_cast<const float2 *>(traj)[((size_t)b * (p.T + p.nb) + p.T + t) * p.n + i];
    atomicAdd(&cursor[(size_t)bt * p.Gb + knn_cell_index(p, v.x, v.y)], 1);
}

// grid B*nb, 1024 threads
__global__ __launch_bounds__(1024) void k_knn_bucket_scan(const KnnParams p, int *__restrict__ cursor,
                                                          knn_cs_t *__restrict__ cell_start,
                                                          float *__restrict__ tile_dkmax, int ntiles, const KnnLists ls,
                                                          int *__restrict__ zero_ptr, int zero_words) {
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, bt = blockIdx.x;
    for (int i = tid; i < ntiles * KNN_NCLS; i += 1024) tile_dkmax[(size_t)bt * ntiles * KNN_NCLS + i] = 0.f;       // (see k_knn_bucket)
    if (ls.far != nullptr) {
        if (tid == 0) ls.far[(size_t)bt * (p.G + 1)] = 0;
        for (int i = tid; i < ls.ftwords; i += 1024) ls.ftbits[(size_t)bt * ls.ftwords + i] = 0u;
    }
    if (ls.again != nullptr) for (int i = tid; i < ls.again_words; i += 1024) { ls.again[(size_t)bt * ls.again_words + i] = 0u; ls.grow[(size_t)bt * ls.again_words + i] = 0u; }
    if (bt == 0 && tid == 0) { ls.fail[0] = 0; ls.retry[0] = 0; ls.farstrip[0] = 0; if (ls.ftlist) ls.ftlist[0] = 0; *knn_marked_count(ls) = 0; *knn_late_count(ls) = 0; *knn_tail_done(ls) = 0; }
    if (bt == 0 && tid < (KNN_RFAR + 1) * (KNN_RFAR + 1))
        ls.chord[tid] = (unsigned char)max(knn_chord_cells(tid / (KNN_RFAR + 1), tid % (KNN_RFAR + 1), p.sp, p.l1 != 0), 0);
    if (bt == 0) for (int i = tid; i < zero_words; i += 1024) zero_ptr[i] = 0;
    int *cur = cursor + (size_t)bt * p.Gb;
    knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    const int chunk = (p.Gb + 1023) / 1024;
    const int g0 = min(tid * chunk, p.Gb), g1 = min(g0 + chunk, p.Gb);
    int local = 0;
    for (int g = g0; g < g1; ++g) local += cur[g];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (tid >> 6); ++w) run += s_wave[w];
    for (int g = g0; g < g1; ++g) {
        const int c = cur[g];
        cur[g] = run;           // becomes the fill cursor of the cell
        cs[g] = run;
        run += c;
    }
    if (tid == 0) cs[p.Gb] = p.n;
}

__global__ __launch_bounds__(256) void k_knn_bucket_scatter(const KnnParams p, const float *__restrict__ traj,
                                                            int *__restrict__ cursor,
                                                            float2 *__restrict__ spos, knn_idx_t *__restrict__ sidx) {
    const int bt = blockIdx.y, b = bt / p.nb, t = bt - b * p.nb;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float2 v = reinterpret_cast<const float2 *>(traj)[((size_t)b * (p.T + p.nb) + p.T + t) * p.n + i];
    const int pos = atomicAdd(&cursor[(size_t)bt * p.Gb + knn_cell_index(p, v.x, v.y)], 1);
    spos[(size_t)bt * p.n + pos] = v;
    sidx[(size_t)bt * p.n + pos] = i;
}

// one WAVEFRONT per cell: cells of up to 128 points by the bitonic network, larger ones by rank counting (every lane
// counts the keys below its own: quadratic, but spread over 64 lanes; the rare pile of the outermost margin ring)
__global__ __launch_bounds__(256) void k_knn_bucket_order(const KnnParams p, const knn_cs_t *__restrict__ cell_start,
                                                          float2 *__restrict__ spos, knn_idx_t *__restrict__ sidx,
                                                          const float *__restrict__ traj) {
    const int bt = blockIdx.y, b = bt / p.nb, t = bt - b * p.nb;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= p.Gb) return;
    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    float2 *sp_ = spos + (size_t)bt * p.n;
    knn_idx_t *si_ = sidx + (size_t)bt * p.n;
    const float2 *pts = reinterpret_cast<const float2 *>(traj) + ((size_t)b * (p.T + p.nb) + p.T + t) * p.n;
    const int a = cs[g], e = cs[g + 1], c = e - a;
    if (c <= 1) return;
    if (c <= KNN_BK_WAVE) {
        unsigned k0 = lane < c ? (unsigned)si_[a + lane] : 0xffffffffu;
        unsigned k1 = lane + 64 < c ? (unsigned)si_[a + 64 + lane] : 0xffffffffu;
        knn_bitonic128(k0, k1, lane);
        if (lane < c) { si_[a + lane] = (int)k0; sp_[a + lane] = pts[k0]; }
        if (lane + 64 < c) { si_[a + 64 + lane] = (int)k1; sp_[a + 64 + lane] = pts[k1]; }
        return;
    }
    // rank counting in rounds of 64 keys: all ranks first (reads), then the writes (the positions are re-read by index)
    for (int i0 = 0; i0 < c; i0 += 64) {
        const int mine = i0 + lane < c ? si_[a + i0 + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < c; ++j) rank += si_[a + j] < mine ? 1 : 0;
        // the ranks of the keys of this round are final, but writing them now would disturb the later rounds' counts: the
        // sorted indices go to the POSITION array's slots as bit patterns first, and are unpacked below
        if (i0 + lane < c) reinterpret_cast<int *>(sp_ + a + rank)[0] = mine;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < c; i0 += 64) {
        if (i0 + lane < c) {
            const int id = __hip_atomic_load(reinterpret_cast<int *>(sp_ + a + i0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            si_[a + i0 + lane] = id;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < c; i0 += 64) if (i0 + lane < c) sp_[a + i0 + lane] = pts[__hip_atomic_load(si_ + a + i0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
}


template <int NT>
__global__ __launch_bounds__(NT) void k_knn_query(const KnnParams p, const float *__restrict__ traj,
                                                   const knn_cs_t *__restrict__ cell_start,
                                                   const float2 *__restrict__ spos,
                                                   const knn_idx_t *__restrict__ sidx,
                                                   float *__restrict__ flow_lut,
                                                   float *__restrict__ flow_next,
                                                   float *__restrict__ knn_state,
                                                   int *__restrict__ idx_out,
                                                   float *__restrict__ tile_dkmax, int r_init, int RH,
                                                   int cap, int stage_flow, int gx, int gy) {      // (gx x gy tiles of the QUERY grid)
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_rowbase[64 + 1];   // RW <= 48
    __shared__ int s_rowg[64];
    __shared__ int s_use_lds;
    const int tid = threadIdx.x;
    // 1-D grid, XCD-contiguous: the tiles of one (sample, bin) run on one XCD, whose L2 then serves the
    // halo points that neighbouring tiles stage again
    const int nblk = gx * gy * p.B * p.nb;
    const int lblk = (int)(blockIdx.x & 7) * ((nblk + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (lblk >= nblk) return;
    const int bt = lblk / (gx * gy), bxy = lblk - bt * gx * gy;
    const int by_ = bxy / gx, bx_ = bxy - by_ * gx;
    const int b = bt / p.nb, t = bt - b * p.nb;
    constexpr int TY = NT / 16;                 // query rows per workgroup (16 columns)
    const int RW = 16 + 2 * RH, RWY = TY + 2 * RH;
    // dynamic LDS carve-up (all sizes multiples of 16 bytes)
    unsigned (*s_hist)[NT] = reinterpret_cast<unsigned (*)[NT]>(s_dyn);
    size_t o = (size_t)KNN_HW * NT * 4;
    float2 *lpos = reinterpret_cast<float2 *>(s_dyn + o); o += (size_t)cap * 8;
    const bool st_f0 = (p.T == 1) && stage_flow;
    float2 *lf0 = st_f0 ? reinterpret_cast<float2 *>(s_dyn + o) : nullptr; o += st_f0 ? (size_t)cap * 8 : 0;
    float2 *lf1 = reinterpret_cast<float2 *>(s_dyn + o); o += p.want_next ? (size_t)cap * 8 : 0;
    unsigned short *lidx = reinterpret_cast<unsigned short *>(s_dyn + o); o += (size_t)cap * 2;
    unsigned short *lcs = reinterpret_cast<unsigned short *>(s_dyn + o);

    QueryCtx c;
    c.cs = cell_start + (size_t)bt * (p.Gb + 1);
    c.spos = spos + (size_t)bt * p.n;
    c.sidx = sidx + (size_t)bt * p.n;
    c.traj_b = reinterpret_cast<const float2 *>(traj) + (size_t)b * (p.T + p.nb) * p.n;
    c.lcs = lcs; c.lpos = lpos; c.lidx = lidx; c.lf0 = lf0; c.lf1 = lf1;
    c.RW = RW; c.RWY = RWY; c.RH = RH;
    c.ry0 = by_ * TY - RH;
    c.rx0 = bx_ * 16 - RH;

    // ---- stage the region (tile + RH rings) ---------------------------------------------------
    const int xlo = max(c.rx0, -p.m), xhi = min(c.rx0 + RW - 1, p.wq + p.m - 1);
    if (tid < RWY) {
        const int yy = c.ry0 + tid;
        int gs = 0, ge = 0;
        if (yy >= -p.m && yy < p.hq + p.m) { gs = c.cs[knn_ci(p, yy, xlo)]; ge = c.cs[knn_ci(p, yy, xhi + 1)]; }
        s_rowg[tid] = gs;
        s_rowbase[tid + 1] = ge - gs;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        s_rowbase[0] = 0;
        for (int rr = 0; rr < RWY; ++rr) { const int cnt = s_rowbase[rr + 1]; s_rowbase[rr + 1] = run + cnt; run += cnt; }
        s_use_lds = (run <= cap) ? 1 : 0;
    }
    __syncthreads();
    const bool use_lds = s_use_lds != 0;
    if (use_lds) {
        const int total = s_rowbase[RWY];
        for (int i = tid; i < RWY * (RW + 1); i += NT) {
            const int rr = i / (RW + 1), cc = i - rr * (RW + 1);
            const int yy = c.ry0 + rr;
            int v = s_rowbase[rr];
            if (yy >= -p.m && yy < p.hq + p.m) {
                const int xx = min(max(c.rx0 + cc, xlo), xhi + 1);
                v += c.cs[knn_ci(p, yy, xx)] - s_rowg[rr];
            }
            lcs[i] = (unsigned short)v;
        }
        const float2 *tref0 = c.traj_b;                                   // T == 1: the reference time
        const float2 *tnext = c.traj_b + (size_t)(p.T + t + 1) * p.n;     // next bin (if any)
        const bool has_next = p.want_next && (t < p.nb - 1);
        for (int i = tid; i < total; i += NT) {
            int lo = 0, hi = RWY;             // row rr with s_rowbase[rr] <= i < s_rowbase[rr+1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rowbase[mid] <= i) lo = mid; else hi = mid; }
            const int g = s_rowg[lo] + (i - s_rowbase[lo]);
            const float2 pj = c.spos[g];
            const int id = c.sidx[g];
            lpos[i] = pj;
            lidx[i] = (unsigned short)id;
            if (st_f0) { const float2 a = tref0[id]; lf0[i] = make_float2(a.x - pj.x, a.y - pj.y); }
            if (has_next) { const float2 a = tnext[id]; lf1[i] = make_float2(a.x - pj.x, a.y - pj.y); }
        }
    }
    __syncthreads();

    const int cy = by_ * TY + (tid >> 4), cx = bx_ * 16 + (tid & 15);
    float dK = 0.f;
    if (cy < p.hq && cx < p.wq) {
        bool done = false;
        if (p.l1) {
            if (use_lds) done = knn_one_query<true, true, NT>(p, c, b, t, cy, cx, r_init, s_hist, flow_lut, flow_next, knn_state, idx_out, dK);
            if (!done) knn_one_query<false, true, NT>(p, c, b, t, cy, cx, r_init, s_hist, flow_lut, flow_next, knn_state, idx_out, dK);
        } else {
            if (use_lds) done = knn_one_query<true, false, NT>(p, c, b, t, cy, cx, r_init, s_hist, flow_lut, flow_next, knn_state, idx_out, dK);
            if (!done) knn_one_query<false, false, NT>(p, c, b, t, cy, cx, r_init, s_hist, flow_lut, flow_next, knn_state, idx_out, dK);
        }
    }
    // K-th distance into the tile maxima (they bound the backward's search windows)
    if (cy < p.hq && cx < p.wq) knn_tile_max_add(tile_dkmax, p, bt, cy, cx, knn_band_depth(r_init), dK);
}

__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// Gather over the query window of one trajectory point (num_tref == 1, 'mean'): one 16-byte LDS read
// {K-th distance, K-th index, dL/dLUT.y, dL/dLUT.x} per query cell; membership is a bitwise predicate
// (no short-circuit branches), so the unrolled body is straight-line code.
template <bool L1, bool NEXT>
__device__ __forceinline__ void bwd_window(const float4 *__restrict__ lq4, const float2 *__restrict__ lgn,
                                           int RW, int ry0, int rx0, int y0, int y1, int x0, int x1,
                                           int sp, float off, float2 pt, int i, float &ay, float &ax,
                                           float2 &an) {
    ay = 0.f; ax = 0.f;
    const float fsp = (float)sp;
    for (int cy = y0; cy <= y1; ++cy) {
        const float dy = ((float)(cy * sp) + off) - pt.x;
        const float dy2 = L1 ? fabsf(dy) : dy * dy;
        const float4 *row = lq4 + (cy - ry0) * RW - rx0;
        const float2 *rown = lgn + (cy - ry0) * RW - rx0;
        float qx = (float)(x0 * sp) + off;              // exact: small integers, same value as (float)(cx*sp)+off
#pragma unroll 4
        for (int cx = x0; cx <= x1; ++cx) {
            const float4 e = row[cx];
            const float dx = qx - pt.y;
            const float d = dy2 + (L1 ? fabsf(dx) : dx * dx);
            const int lt = d < e.x, eq = d == e.x, le = i <= __float_as_int(e.y);
            const bool in = (lt | (eq & le)) != 0;
            ay += in ? e.z : 0.f;
            ax += in ? e.w : 0.f;
            if (NEXT) { const float2 gq = rown[cx]; an.x += in ? gq.x : 0.f; an.y += in ? gq.y : 0.f; }
            qx += fsp;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, step 0: search reach of the points of every 16x16 cell tile = the largest K-th distance
// among the tiles whose queries can reach into the tile at all (Chebyshev gap between the tile's cell
// area and their query centres).  One workgroup per (sample, bin), the tile maxima staged in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_reach(const KnnParams p, const float *__restrict__ tile_dkmax,
                                                   float *__restrict__ reach) {
    extern __shared__ float s_lin[];          // linear K-th distance bound of every tile of this (sample, bin)
    const int ntx = knn_tiles_x(p.wq, p.m), nty = knn_tiles_y(p.hq, p.m), nt = ntx * nty;
    const int bt = blockIdx.x;
    for (int tb = threadIdx.x; tb < nt; tb += 256) {
        float dk = 0.f;
#pragma unroll
        for (int c = 0; c < KNN_NCLS; ++c) dk = fmaxf(dk, tile_dkmax[((size_t)bt * nt + tb) * KNN_NCLS + c]);
        s_lin[tb] = dk > 0.f ? (p.l1 ? dk : sqrtf(dk)) * 1.0001f + 0.01f : -1.f;
    }
    __syncthreads();
    for (int tile = threadIdx.x; tile < nt; tile += 256) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
        float ay0, ay1, ax0, ax1;
        knn_tile_area(p, ty, tx, ay0, ay1, ax0, ax1);
        float r = 0.f;
        for (int by = 0; by < nty; ++by) {
            for (int bx = 0; bx < ntx; ++bx) {
                int cy0, cy1, cx0, cx1;
                if (!knn_tile_class_cells(p, by, bx, 0, 0, cy0, cy1, cx0, cx1)) continue;
                const float lin = s_lin[by * ntx + bx];
                const float qy0 = (float)(cy0 * p.sp) + p.off, qy1 = (float)(cy1 * p.sp) + p.off;
                const float gy = fmaxf(0.f, fmaxf(qy0 - ay1, ay0 - qy1));
                const float qx0 = (float)(cx0 * p.sp) + p.off, qx1 = (float)(cx1 * p.sp) + p.off;
                const float gx = fmaxf(0.f, fmaxf(qx0 - ax1, ax0 - qx1));
                if (lin > 0.f && lin >= fmaxf(gy, gx)) r = fmaxf(r, lin);
            }
        }
        reach[(size_t)bt * nt + tile] = r;
    }
}

// ------------------------------------------------------------------------------------------
// backward, step 1: one thread per bucketed trajectory point of a TS x TS cell tile.  The K-th
// keys and LUT gradients of the tile and a halo of RQ cells are staged in LDS; each point scans
// the query cells within the reach of its 16x16 tile and gathers the gradient of every cell that
// has the point among its K nearest.  Writes per-(bin, point) partials.
// The kernel is a chain of short dependent phases (reach -> region -> points -> window), so large
// workgroups (TS = 32: 1024 threads, 4 points' worth of window work per phase) amortise the chain:
// measured at C3, 16x16 tiles 210 us of which only ~70 us is the window loop.
// grid (ceil(wq/TS), ceil(hq/TS), B*nb), TS*TS threads, dynamic LDS
// ------------------------------------------------------------------------------------------
#define KNN_RQ_MAX 9   // largest staged halo (cells); tiles whose reach needs more read the global arrays (7 -> 9: the 1 % of such tiles at C3 took 47 us each against 12, and were the whole duration of a B = 1 launch)
template <int TS>
__global__ __launch_bounds__(TS * TS) void k_knn_bwd_points(const KnnParams p, const knn_cs_t *__restrict__ cell_start,
                                                            const float2 *__restrict__ spos,
                                                            const knn_idx_t *__restrict__ sidx,
                                                            const float *__restrict__ glut,
                                                            const float *__restrict__ gnext,
                                                            const float *__restrict__ knn_state,
                                                            const float *__restrict__ reach,
                                                            float2 *__restrict__ tmp_g,   // [B*nb][n][T]
                                                            float2 *__restrict__ tmp_a,   // [B*nb][n]
                                                            int gx, int gy) {
    constexpr int NT = TS * TS, SUB = TS / 16;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_rowbase[KNN_TROWS + 1];
    __shared__ int s_rowg[KNN_TROWS];
    __shared__ float s_Rsub[SUB * SUB];
    const int tid = threadIdx.x;
    // 1-D grid, XCD-contiguous: the gx*gy tiles of one (sample, bin) run on one XCD, so the halo cells that
    // neighbouring tiles stage again are found in that XCD's L2
    const int nblk = gx * gy * p.B * p.nb;
    const int lblk = (int)(blockIdx.x & 7) * ((nblk + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (lblk >= nblk) return;
    const int bt = lblk / (gx * gy), bxy = lblk - bt * gx * gy;
    const int by_ = bxy / gx, bx_ = bxy - by_ * gx;
    const int b = bt / p.nb, t = bt - b * p.nb;
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    int tcy0, tcy1, tcx0, tcx1;           // cells of the tile's points: a border tile owns the margin beside it (knn_tile_cells)
    knn_tile_cells(by_, p.hq, p.m, tcy0, tcy1); knn_tile_cells(bx_, p.wq, p.m, tcx0, tcx1);
    const int nrow = tcy1 - tcy0;
    // phase 1 (independent loads): reach of the 16x16 sub-tiles; per tile row the contiguous range of
    // the bucketed arrays, turned into a running offset by a wavefront scan
    if (tid < SUB * SUB) {
        const int gx16 = knn_tiles_x(p.wq, p.m), gy16 = knn_tiles_y(p.hq, p.m);
        const int ty = by_ * SUB + tid / SUB, tx = bx_ * SUB + tid % SUB;
        s_Rsub[tid] = (ty < gy16 && tx < gx16) ? reach[((size_t)bt * gy16 + ty) * gx16 + tx] : 0.f;
    }
    if (tid < 64) {
        int gs = 0, ge = 0;
        if (tid < nrow) { gs = cs[knn_ci(p, tcy0 + tid, tcx0)]; ge = cs[knn_ci(p, tcy0 + tid, tcx1)]; }      // cell rows of the tile, margin included
        int run = ge - gs;                               // inclusive scan over the lanes
#pragma unroll
        for (int o2 = 1; o2 < KNN_TROWS; o2 <<= 1) { const int v = __shfl_up(run, o2, 64); if (tid >= o2) run += v; }
        if (tid < nrow) { s_rowg[tid] = gs; s_rowbase[tid + 1] = run; }
        if (tid == 0) s_rowbase[0] = 0;
    }
    __syncthreads();
    float R = s_Rsub[0];
#pragma unroll
    for (int k = 1; k < SUB * SUB; ++k) R = fmaxf(R, s_Rsub[k]);
    const int RQ_need = (int)ceilf(R / (float)p.sp) + 1;      // halo of the staged region, in cells
    const bool use_lds = RQ_need <= KNN_RQ_MAX;
    const int RQ = use_lds ? RQ_need : 0;
    const int RW = TS + 2 * RQ;
    const int ry0 = by_ * TS - RQ, rx0 = bx_ * TS - RQ;
    // LDS: only the fast path (num_tref == 1, 'mean') stages anything: one float4 per query cell
    // {K-th distance, K-th index, dL/dLUT.y, dL/dLUT.x} (+ float2 of the flow_to_next gradient)
    float4 *lq4 = reinterpret_cast<float4 *>(s_dyn);
    float2 *lgn = reinterpret_cast<float2 *>(s_dyn + (size_t)RW * RW * 16);
    const bool fast = use_lds && p.T == 1 && !p.iwd;
    const bool has_next = (gnext != nullptr) && (t < p.nb - 1);
    const float2 *gn2 = has_next ? reinterpret_cast<const float2 *>(gnext) + (size_t)(b * (p.nb - 1) + t) * p.G : nullptr;
    const float2 *gl2 = reinterpret_cast<const float2 *>(glut) + (size_t)bt * p.G * p.T;
    if (fast) {
        for (int i = tid; i < RW * RW; i += NT) {
            const int rr = i / RW, cc = i - rr * RW;
            const int yy = ry0 + rr, xx = rx0 + cc;
            float dk = -1.f; int ik = -1;
            float2 g = make_float2(0.f, 0.f), gn = make_float2(0.f, 0.f);
            if (yy >= 0 && yy < p.hq && xx >= 0 && xx < p.wq) {
                const size_t q = (size_t)bt * p.G + (size_t)yy * p.wq + xx;
                dk = knn_state[q];
                ik = reinterpret_cast<const int *>(knn_state)[BQ + q] & KNN_IDX_MASK;
                g = gl2[(size_t)yy * p.wq + xx];
                if (has_next) gn = gn2[(size_t)yy * p.wq + xx];
            }
            lq4[i] = make_float4(dk, __int_as_float(ik), g.x, g.y);
            if (has_next) lgn[i] = gn;
        }
    }
    __syncthreads();
    const int total = s_rowbase[nrow];
    const float invK = 1.f / (float)p.K, inv_sp = 1.f / (float)p.sp;
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const knn_idx_t *si_ = sidx + (size_t)bt * p.n;
    for (int pi = tid; pi < total; pi += NT) {
        int lo = 0, hi = nrow;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rowbase[mid] <= pi) lo = mid; else hi = mid; }
        const int g = s_rowg[lo] + (pi - s_rowbase[lo]);
        const float2 pt = sp_[g];
        const int i = si_[g];
        // query cells within reach: |q - p| <= R per axis.  R already carries a 0.01 px + 1e-4 relative
        // margin, which dominates the rounding of these four expressions, so no extra cell is added.
        // (a reciprocal multiply instead of four divisions: its 1e-7 relative error is far inside that margin too)
        int y0 = (int)ceilf((pt.x - R - p.off) * inv_sp);
        int y1 = (int)floorf((pt.x + R - p.off) * inv_sp);
        int x0 = (int)ceilf((pt.y - R - p.off) * inv_sp);
        int x1 = (int)floorf((pt.y + R - p.off) * inv_sp);
        y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, p.hq - 1); x1 = min(x1, p.wq - 1);
        if (fast) {      // the staged region always covers the window (see the note on clamped cells)
            y0 = max(y0, ry0); x0 = max(x0, rx0); y1 = min(y1, ry0 + RW - 1); x1 = min(x1, rx0 + RW - 1);
        }
        float2 an = make_float2(0.f, 0.f);
        if (fast) {
            float ay, ax;
            if (p.l1) {
                if (has_next) bwd_window<true, true>(lq4, lgn, RW, ry0, rx0, y0, y1, x0, x1, p.sp, p.off, pt, i, ay, ax, an);
                else bwd_window<true, false>(lq4, lgn, RW, ry0, rx0, y0, y1, x0, x1, p.sp, p.off, pt, i, ay, ax, an);
            } else {
                if (has_next) bwd_window<false, true>(lq4, lgn, RW, ry0, rx0, y0, y1, x0, x1, p.sp, p.off, pt, i, ay, ax, an);
                else bwd_window<false, false>(lq4, lgn, RW, ry0, rx0, y0, y1, x0, x1, p.sp, p.off, pt, i, ay, ax, an);
            }
            tmp_g[(size_t)bt * p.n + i] = make_float2(invK * ay, invK * ax);
            if (gnext != nullptr) tmp_a[(size_t)bt * p.n + i] = make_float2(invK * an.x, invK * an.y);
            continue;
        }
        for (int tr = 0; tr < p.T; ++tr) {
            float ay = 0.f, ax = 0.f;
            for (int cy = y0; cy <= y1; ++cy) {
                const float qy = (float)(cy * p.sp) + p.off;
                for (int cx = x0; cx <= x1; ++cx) {
                    const float qx = (float)(cx * p.sp) + p.off;
                    const float d = pair_dist(qy, qx, pt.x, pt.y, p.l1);
                    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
                    const float dk = knn_state[q];
                    if (d > dk) continue;
                    const int ik = reinterpret_cast<const int *>(knn_state)[BQ + q] & KNN_IDX_MASK;
                    if (d == dk && i > ik) continue;
                    const float nm = p.iwd ? knn_state[2 * BQ + q] : 1.f;
                    const float2 gq = gl2[((size_t)cy * p.wq + cx) * p.T + tr];
                    const float2 gnq = has_next ? gn2[(size_t)cy * p.wq + cx] : make_float2(0.f, 0.f);
                    const float w = p.iwd ? (1.f / (d + 1e-9f)) / nm : invK;
                    ay += w * gq.x; ax += w * gq.y;
                    if (tr == 0 && has_next) { an.x += invK * gnq.x; an.y += invK * gnq.y; }
                }
            }
            tmp_g[((size_t)bt * p.n + i) * p.T + tr] = make_float2(ay, ax);
        }
        tmp_a[(size_t)bt * p.n + i] = an;
    }
}

// window loop of k_knn_bwd_tile, CW columns wide (fully unrolled: the squared column offsets live in registers).
// Membership is `d <= K-th distance` (callers: no excluded tie among the staged queries).
// IWD ('iwd' interpolation, focus.py:155-163; round 6): a member's weight is (1 / (d + 1e-9)) / normaliser of the query instead of
// 1 / K -- the staged gradient of a query cell is dL/dLUT / normaliser (divided once per staged cell, not per visited cell: no fourth array
// in LDS, 12 bytes per cell as for 'mean'), and the weight is the membership flag times ONE hardware reciprocal of (d + 1e-9)
// (v_rcp_f32: 1 ulp; the two IEEE divisions of the formula as written are ~20 instructions per visited cell in a loop of 6, and the
// weights are constants of the backward: their rounding is a relative 2e-7 of the gradient).  The flow_to_next term stays a mean.
template <int CW, bool L1, bool NEXT, bool IWD>
__device__ __forceinline__ void bwd_window_fast(const KnnParams &p, const float2 pt, int x0, int y0, int nxw, int nyw, int nymax,
                                                int ry0, int RW, int RP, int xb, const float *ldk, const float2 *lg,
                                                const float2 *lgn, float &ay, float &ax, float2 &an) {
    float dx2[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float dx = ((float)__mul24(x0 + c, p.sp) + p.off) - pt.y;      // (24-bit multiplies: full rate; v_mul_lo_u32 is quarter rate)
        dx2[c] = (c < nxw) ? (L1 ? fabsf(dx) : dx * dx) : INFINITY;
    }
    for (int r = 0; r < nymax; ++r) {
        const int cyr = min(max(y0 + r, ry0), ry0 + RW - 1);
        const float dy = ((float)__mul24(cyr, p.sp) + p.off) - pt.x;
        const float dy2 = (r < nyw) ? (L1 ? fabsf(dy) : dy * dy) : INFINITY;
        const int ro = __mul24(cyr - ry0, RP) + xb;
        MPC_EXPECT(ro >= 0 && ro + CW <= RW * RP + 16);          // (every lane reads CW cells of the row: the slack behind the last row is there for that)
        const float *rdk = ldk + ro;
        const float2 *rg = lg + ro, *rown = lgn + ro;
        // The gradients as ONE ds_read_b64 each (knn_lds_f2: a volatile load the compiler may not pair).  Left alone it pairs
        // the reads of neighbouring cells into ds_read2_b64, which the LDS serves at half the rate of two single reads (8
        // cycles per pair: MI355X_MICROARCH.md, LDS table) -- and this loop lives on the LDS (SQ_LDS_IDX_ACTIVE: 76 % of the
        // kernel's cycles).  (Inline asm with its own s_waitcnt: 147 us; one 16-byte cell {K-th distance, gradient, index}
        // read as ds_read_b128: 164 us.)
        // (the loads of four cells first, then their arithmetic: a volatile load is a scheduling barrier for every other
        // memory operation, so in load-use-load-use order each cell would wait for its own LDS round trip)
#pragma unroll
        for (int c0 = 0; c0 < CW; c0 += KNN_BW_CH) {
            float2 e[KNN_BW_CH], gq[KNN_BW_CH];
            float dkc[KNN_BW_CH];
#pragma unroll
            for (int c = 0; c < KNN_BW_CH; ++c) {
                if (c0 + c < CW) {
                    e[c] = knn_lds_f2(rg + c0 + c);
                    if (NEXT) gq[c] = knn_lds_f2(rown + c0 + c);
                    dkc[c] = rdk[c0 + c];
                }
            }
#pragma unroll
            for (int c = 0; c < KNN_BW_CH; ++c) {
                if (c0 + c < CW) {
                    // member <=> d <= K-th distance, as ARITHMETIC: t = dk - d has the exact sign of the comparison (a
                    // difference of two floats is 0 only if they are equal, and it cannot round across 0), so
                    // clamp(t * 2^100 + 1, 0, 1) is 1 for t >= 0 and 0 for any t < 0 (|t| >= one ulp of a squared distance;
                    // d = inf and the dk = -1 of a cell that is no query give -inf / negative).  Two fp32 instructions, which
                    // this chip issues at twice the rate of the compare + select they replace (profiles/r02_ubench_valu_rate.txt).
                    const float d = dy2 + dx2[c0 + c];
                    const float w = fminf(fmaxf(fmaf(dkc[c] - d, 0x1p100f, 1.f), 0.f), 1.f);
                    if (IWD) {
                        // (d = inf beyond the window: the reciprocal is 0; the staged gradient of a query is dL/dLUT / normaliser)
                        const float wi = w * __builtin_amdgcn_rcpf(d + 1e-9f);
                        ay = fmaf(wi, e[c].x, ay); ax = fmaf(wi, e[c].y, ax);
                    } else { ay = fmaf(w, e[c].x, ay); ax = fmaf(w, e[c].y, ax); }
                    if (NEXT) { an.x = fmaf(w, gq[c].x, an.x); an.y = fmaf(w, gq[c].y, an.y); }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward for num_tref == 1, 'mean' (the shipped configurations): k_knn_reach and the gather in ONE launch, with a
// cheaper window loop.
//   * the workgroup derives the reach of its 16x16 tile from the tile maxima itself (one coalesced read);
//   * the squared column offsets of a point's window are computed once (registers, statically indexed: the window
//     is at most KNN_BW_WMAX cells wide in this variant), so a visited cell costs: one 16-byte LDS read, one add,
//     one compare, one select and two FMAs (0/1 weight: fma(1, a, b) == a + b);
//   * the membership test is `d <= K-th distance` unless a staged query has a point EXCLUDED at exactly its K-th
//     distance (tie resolved by index; the forward flags such queries in bit 30 of the K-th index): those
//     workgroups, and windows wider than KNN_BW_WMAX, take the exact (distance, index) loop of bwd_window.
// grid: 1-D, XCD-contiguous, 256 threads, dynamic LDS (RWmax^2 float4 + KNN_BW_WMAX of slack [+ float2 per cell])
// ------------------------------------------------------------------------------------------
#define KNN_BW_WMAX 16
// Workgroups per CU the register budget is set for (7 -> 72 VGPRs, 8 -> 64).  Round 2 needed all eight (the kernel waited for its
// LDS conflicts); without them, and with the reach phase gone, seven workgroups that spill 12 instead of 48 bytes per lane are
// faster: 118 -> 114 us at C3.  With the flow_to_next gradient (C4: a larger register set of its own) eight remain better: 57.3
// against 58.4 us.
// Reach of one 16x16 tile, computed by ONE wavefront (lane = threadIdx.x & 63, all 64 lanes call): step A, the largest K-th
// distance of any class of any tile of the slice; step B, the (tile, class) pairs within its D rings -- see k_knn_bwd_tile.
template <bool L1>
__device__ __forceinline__ float knn_tile_reach(const KnnParams &p, const float *__restrict__ tile_dkmax, int bt, int by_, int bx_,
                                                int gx, int gy, int bd) {
    const int tid = threadIdx.x & 63;
    const int ntx = gx, nty = gy, nt = ntx * nty;
    float m = 0.f;
    for (int i = tid; i < nt * KNN_NCLS; i += 64) m = fmaxf(m, tile_dkmax[(size_t)bt * nt * KNN_NCLS + i]);
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
    const float linmax = (L1 ? m : sqrtf(m)) * 1.0001f + 0.01f;
    const int D = (int)fminf(linmax / (float)(16 * p.sp), 1.0e6f) + 1;
    float ay0, ay1, ax0, ax1;
    knn_tile_area(p, by_, bx_, ay0, ay1, ax0, ax1);
    float r = 0.f;
    // one (source tile, class) pair: its K-th distance counts if its queries can reach this tile's area
    auto pair_reach = [&](int sy, int sx, int c) {
        const int tb = sy * ntx + sx;
        int cy0, cy1, cx0, cx1;
        if (!knn_tile_class_cells(p, sy, sx, c, bd, cy0, cy1, cx0, cx1)) return;
        const float dk = tile_dkmax[((size_t)bt * nt + tb) * KNN_NCLS + c];
        const float lin = (L1 ? dk : sqrtf(dk)) * 1.0001f + 0.01f;
        const float qy0 = (float)(cy0 * p.sp) + p.off, qy1 = (float)(cy1 * p.sp) + p.off;
        const float qx0 = (float)(cx0 * p.sp) + p.off, qx1 = (float)(cx1 * p.sp) + p.off;
        const float gyv = fmaxf(0.f, fmaxf(qy0 - ay1, ay0 - qy1)), gxv = fmaxf(0.f, fmaxf(qx0 - ax1, ax0 - qx1));
        if (dk > 0.f && lin >= fmaxf(gyv, gxv)) r = fmaxf(r, lin);
    };
    if (D == 1) {
        if (tid < 9 * KNN_NCLS) {
            const int nbr = tid / KNN_NCLS, c = tid - nbr * KNN_NCLS;
            const int sy = by_ + nbr / 3 - 1, sx = bx_ + (nbr % 3) - 1;
            if (sy >= 0 && sy < nty && sx >= 0 && sx < ntx) pair_reach(sy, sx, c);
        }
    } else {
        for (int tb = tid; tb < nt; tb += 64) {
            const int sy = tb / ntx, sx = tb - sy * ntx;
            if (abs(sy - by_) > D || abs(sx - bx_) > D) continue;
#pragma unroll
            for (int c = 0; c < KNN_NCLS; ++c) pair_reach(sy, sx, c);
        }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) r = fmaxf(r, __shfl_xor(r, o2, 64));
    return r;
}

// The reaches of all tiles from a launch of its own (the stage entry point mpc_knn_lut_bwd on large problems; mpc_focus_bwd lets
// the event backward's kernel do it on the side): knn_reach_slice, knn_device.h.  grid B * nb, 256 threads, dynamic LDS
// (nt * (KNN_NCLS + 1) + 16) floats
template <bool L1>
__global__ __launch_bounds__(256) void k_knn_reach_tiles(const KnnParams p, const float *__restrict__ tile_dkmax, float *__restrict__ reach,
                                                         int gx, int gy, int bd) {
    extern __shared__ float s_reach_mem[];
    knn_reach_slice<L1>(p, tile_dkmax, reach, blockIdx.x, gx, gy, bd, s_reach_mem);
}

template <bool L1, bool NEXT, bool IWD>
__global__ __launch_bounds__(256, NEXT ? KNN_BW_OCC_NEXT : (IWD ? KNN_BW_OCC_IWD : KNN_BW_OCC)) void k_knn_bwd_tile(const KnnParams p, const knn_cs_t *__restrict__ cell_start,
                                                      const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                                      const float *__restrict__ glut, const float *__restrict__ gnext,
                                                      const float *__restrict__ knn_state,
                                                      const float *__restrict__ tile_dkmax,
                                                      float2 *__restrict__ tmp_g, float2 *__restrict__ tmp_a,
                                                      float2 *__restrict__ gtraj_direct, const float *__restrict__ reach_in,
                                                      int gx, int gy, int bd, int *far_next, const float *__restrict__ gnext_scale KB_STAMP_PARAM) {
    constexpr int TS = 16;
    KB_STAMP_BEGIN
    extern __shared__ __align__(16) unsigned char s_dyn[];
    if (blockIdx.x == 0 && threadIdx.x == 0 && far_next != nullptr) *far_next = 0;       // (k_knn_bwd_far, the next launch: its dynamic items)
    __shared__ int s_rowbase[KNN_TROWS + 1];
    __shared__ int s_rowg[KNN_TROWS];
    __shared__ float s_wr[1];
    __shared__ int s_tiew[4];
    const int tid = threadIdx.x;
    const int nblk = gx * gy * p.B * p.nb;
    const int lblk = (int)(blockIdx.x & 7) * ((nblk + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (lblk >= nblk) return;
    const int bt = lblk / (gx * gy), bxy = lblk - bt * gx * gy;
    const int by_ = bxy / gx, bx_ = bxy - by_ * gx;
    const int b = bt / p.nb, t = bt - b * p.nb;
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    int tcy0, tcy1, tcx0, tcx1;           // cells of the tile's points: a border tile owns the margin beside it (knn_tile_cells)
    knn_tile_cells(by_, p.hq, p.m, tcy0, tcy1); knn_tile_cells(bx_, p.wq, p.m, tcx0, tcx1);
    const int nrow = tcy1 - tcy0;
    // ---- phase 1: reach of this tile = the largest linear K-th distance among the tiles whose queries can touch it
    //      (Chebyshev gap between this tile's cell area and their query centres); bucketed ranges of the tile rows --
    // Wavefront 0 alone.  Step A: the largest K-th distance of ANY class of ANY tile of the slice (coalesced reads).  A
    // source tile k tile rings away has its query centres at least (k - 1) * 16 cells from this tile's area, so only the tiles
    // within D = floor(linmax / (16 sp)) + 1 rings can touch it -- D = 1 in practice: the 3 x 3 tiles around this one.
    // Step B tests those 45 (tile, class) pairs on 45 lanes.  (Before: every tile of the slice on a lane of its own in a loop
    // over the classes -- ~300 vector instructions for one or two wavefronts of the workgroup, the maximum of each class a
    // dependent round trip behind the loop's `continue`.  With step A on all four wavefronts and a barrier before step B the
    // kernel was slower than that: 142 vs 139 us.)  Wavefront 1 meanwhile looks up the bucketed ranges of the tile rows.
    // (round 6) With the reaches precomputed (mpc_focus_bwd: by the event backward's kernel; large stage calls: by a launch of their
    // own) the tile's reach is ONE word at a workgroup-uniform address: every thread loads it itself -- a scalar load -- instead of
    // wavefront 0 loading it and handing it over through LDS behind a barrier.  The staging loads then go out at once, beside the row
    // ranges' round trip of wavefront 1 instead of behind it: the workgroup is a chain of dependent round trips at seven per CU, and
    // that chain -- not the window loop -- is the kernel's time (hack builds with two thirds of the loop's LDS reads or half of its
    // arithmetic removed: 119.5 -> 114 / 111 us; the pre-loop phase was 6.1 of a workgroup's 11.6 us, tools/bwd_stamp_probe.py).
    const bool pre = reach_in != nullptr;
    float Rpre = 0.f;
    if (pre) Rpre = reach_in[(size_t)bt * gx * gy + bxy];
    if (tid < 64) {
        if (!pre) {
            const float r = knn_tile_reach<L1>(p, tile_dkmax, bt, by_, bx_, gx, gy, bd);
            if (tid == 0) s_wr[0] = r;
        }
    } else if (tid < 128) {
        const int ln = tid - 64;
        int gs = 0, ge = 0;
        if (ln < nrow) { gs = cs[MPC_IDX(knn_ci(p, tcy0 + ln, tcx0), p.Gb + 1)]; ge = cs[MPC_IDX(knn_ci(p, tcy0 + ln, tcx1), p.Gb + 1)]; }      // cell rows of the tile, margin included
        int run = ge - gs;
#pragma unroll
        for (int o2 = 1; o2 < KNN_TROWS; o2 <<= 1) { const int v = __shfl_up(run, o2, 64); if (ln >= o2) run += v; }
        if (ln < nrow) { s_rowg[MPC_IDX(ln, KNN_TROWS)] = gs; s_rowbase[MPC_IDX(ln + 1, KNN_TROWS + 1)] = run; }
        if (ln == 0) s_rowbase[0] = 0;
    }
    // ---- phase 2 (after the reach is known; staging a guessed halo before it, so that the loads of the two phases
    //      travel together, measured slower: too many tiles stage twice): the K-th distance, K-th index | tie flag and
    //      dL/dLUT of the tile's query cells and the halo the reach asks for ------------------------------------------
    float *ldk; float2 *lg, *lgn;
    int RQ = 0, RW = TS, RP = 32, ry0 = by_ * TS, rx0 = bx_ * TS;
    const bool has_next = NEXT && (gnext != nullptr) && (t < p.nb - 1);
    const float2 *gn2 = has_next ? reinterpret_cast<const float2 *>(gnext) + (size_t)(b * (p.nb - 1) + t) * p.G : nullptr;
    const float2 *gl2 = reinterpret_cast<const float2 *>(glut) + (size_t)bt * p.G;
    // gnext_scale (mpc_focus_bwd): dL/dflow_next = grad_out * the smoothness gradient, multiplied as the cells are read -- the values
    // a pass of mpc_scale over the whole field would have written (round 5: 16 us of a C4 batch-6 step), bit for bit
    const float gns = (NEXT && gnext_scale != nullptr) ? gnext_scale[0] : 1.f;
    auto stage = [&]() {
        RW = TS + 2 * RQ;
        // Row pitch of the staged arrays.  The window loop reads one K-th distance (4 bytes) and one gradient (8 bytes) per
        // lane and cell; the LDS serves 32 lanes per cycle -- two cell rows of ~16 points each, whose windows start at about
        // the same column one row apart.  At a pitch of 32 entries (round 2) both rows sit on the same banks: a 2-way
        // conflict on every read (SQ_LDS_BANK_CONFLICT: 18.6 M cycles per C3 launch, a quarter of the kernel).  48 entries
        // put consecutive rows 16 banks (4-byte array: 32 banks) resp. 32 banks (8-byte array: 64 banks) apart: no conflict.
        // The K-th INDEX (only read on the exact path, for a tie) stays in global memory then, which keeps the workgroup
        // at 19.5 KB -- eight per CU as before.  With the flow_to_next gradient (8 more bytes per cell) the pitch stays 32.
        // With the flow_to_next gradient (20 bytes per cell; the allocation holds (16 + 2 KNN_RQ_MAX)^2 cells: six workgroups per
        // CU) the pitch is the widest conflict-free one the tile's own region leaves room for: 48 up to a halo of 4 cells -- nearly
        // every tile --, 40 up to 6, else the region's width (round 6; a fixed 32 before: the 2-way conflict of round 2).
        if (NEXT) {
            constexpr int CELLS = (TS + 2 * KNN_RQ_MAX) * (TS + 2 * KNN_RQ_MAX);
            RP = (RW * 48 <= CELLS) ? 48 : ((RW * 40 <= CELLS) ? 40 : (RW <= KNN_BW_PITCH_NEXT ? KNN_BW_PITCH_NEXT : RW));
        } else RP = KNN_BW_PITCH;
        ry0 = by_ * TS - RQ; rx0 = bx_ * TS - RQ;
        // separate arrays (the hot loop reads the K-th distance and the gradient only; neighbouring lanes then read
        // neighbouring 4- and 8-byte words): K-th distance, [K-th index,] dL/dLUT, [dL/dflow_next]
        const size_t ncell = (size_t)RW * RP + KNN_BW_WMAX;
        ldk = reinterpret_cast<float *>(s_dyn);
        if (NEXT) {
            // (the K-th INDEX -- read on the exact path only, for a tie -- stays in global memory here too, round 6: 20 bytes per cell
            // instead of 24, one workgroup more per CU)
            lg = reinterpret_cast<float2 *>(s_dyn + ncell * 4);
            lgn = reinterpret_cast<float2 *>(s_dyn + ncell * 12);
        } else {
            lgn = nullptr;
            lg = reinterpret_cast<float2 *>(s_dyn + ncell * 4);          // (ncell is even: 8-byte aligned)
        }
        int tie = 0;
        for (int rr = tid >> 5; rr < RW; rr += 8) {
            const int yy = ry0 + rr;
            for (int cc = tid & 31; cc < RP; cc += 32) {          // (columns RW..RP-1: padding, never a member)
                const int xx = rx0 + cc;
                float dk = -1.f, wn = 1.f; int ik = -1;
                float2 g = make_float2(0.f, 0.f), gn = make_float2(0.f, 0.f);
                if (cc < RW && yy >= 0 && yy < p.hq && xx >= 0 && xx < p.wq) {
                    const size_t q = (size_t)bt * p.G + (size_t)yy * p.wq + xx;
                    dk = knn_state[q];
                    ik = reinterpret_cast<const int *>(knn_state)[BQ + q];
                    if (IWD) wn = knn_state[2 * BQ + q];
                    g = gl2[(size_t)yy * p.wq + xx];
                    if (has_next) { gn = gn2[(size_t)yy * p.wq + xx]; if (gnext_scale != nullptr) gn = make_float2(gn.x * gns, gn.y * gns); }
                    if (ik & KNN_FAR_FLAG) dk = -1.f;        // served by the fallback kernel: k_knn_bwd_far adds its gradient
                    else tie |= ik & KNN_TIE_FLAG;
                    // 'iwd': the query's normaliser goes into its staged gradient (a cell that is no member of anything holds 0: its
                    // weight is 0, and 0 x a non-finite quotient would not be)
                    if (IWD) g = (dk >= 0.f && wn > 0.f) ? make_float2(g.x / wn, g.y / wn) : make_float2(0.f, 0.f);
                    ik &= KNN_IDX_MASK;
                }
                ldk[MPC_IDX(rr * RP + cc, ncell)] = dk; lg[MPC_IDX(rr * RP + cc, ncell)] = g;
                if (NEXT) lgn[MPC_IDX(rr * RP + cc, ncell)] = gn;
            }
        }
        if (tid < KNN_BW_WMAX) {                    // slack behind the last row: never a member
            ldk[MPC_IDX(RW * RP + tid, ncell)] = -1.f; lg[MPC_IDX(RW * RP + tid, ncell)] = make_float2(0.f, 0.f);
            if (NEXT) lgn[MPC_IDX(RW * RP + tid, ncell)] = make_float2(0.f, 0.f);
        }
        const bool wt = __ballot(tie != 0) != 0ull;
        if ((tid & 63) == 0) s_tiew[tid >> 6] = wt ? 1 : 0;
    };
    if ((tid & 63) == 0) s_tiew[tid >> 6] = 0;
    if (!pre) __syncthreads();                 // (workgroup-uniform; with `pre` the barrier behind the staging covers the row tables too)
    const float R = pre ? Rpre : s_wr[0];
    const int RQ_need = (int)ceilf(R / (float)p.sp) + 1;      // halo the reach asks for, in cells
    const bool use_lds = RQ_need <= KNN_RQ_MAX;
    if (use_lds) {
        RQ = RQ_need;
        stage();
    }
    __syncthreads();
    const bool anytie = (s_tiew[0] | s_tiew[1] | s_tiew[2] | s_tiew[3]) != 0;
    KB_STAMP_MID
    const int total = s_rowbase[nrow];
    const float invK = 1.f / (float)p.K, inv_sp = 1.f / (float)p.sp;
    const float wK = IWD ? 1.f : invK;                 // ('iwd': the weights are in the sums already)
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const knn_idx_t *si_ = sidx + (size_t)bt * p.n;
    for (int base = 0; base < total; base += 256) {
        const int pi = base + tid;
        const bool act = pi < total;
        if (__ballot(act) == 0ull) continue;          // (a tile holds ~256 points: a second round is mostly one wavefront)
        int lo = 0, hi = nrow;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rowbase[mid] <= pi) lo = mid; else hi = mid; }
        const int g = act ? s_rowg[lo] + (pi - s_rowbase[lo]) : 0;
        const float2 pt = sp_[MPC_IDX(g, p.n)];
        const int i = si_[MPC_IDX(g, p.n)];
        // query cells within reach: |q - p| <= R per axis (R carries a 0.01 px + 1e-4 relative margin, which dominates
        // the rounding of these expressions)
        int y0 = (int)ceilf((pt.x - R - p.off) * inv_sp), y1 = (int)floorf((pt.x + R - p.off) * inv_sp);
        int x0 = (int)ceilf((pt.y - R - p.off) * inv_sp), x1 = (int)floorf((pt.y + R - p.off) * inv_sp);
        y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, p.hq - 1); x1 = min(x1, p.wq - 1);
        float ay = 0.f, ax = 0.f;
        float2 an = make_float2(0.f, 0.f);
        if (use_lds) {
            y0 = max(y0, ry0); x0 = max(x0, rx0); y1 = min(y1, ry0 + RW - 1); x1 = min(x1, rx0 + RW - 1);
            if (!act) { y1 = y0 - 1; x1 = x0 - 1; }
            const int nxw = x1 - x0 + 1, nyw = y1 - y0 + 1;
            const bool narrow = __ballot(nxw > KNN_BW_WMAX) == 0ull;
            if (narrow && !anytie) {
                // (the unrolled column loop comes in three widths, picked by the widest window of the wavefront: an inner
                // tile's reach of ~3.3 cells gives windows of 7-8 columns, and a column beyond the window costs as much
                // as one inside it)
                const int nxmax = __builtin_amdgcn_readfirstlane(wave_max_int(nxw));
                const int nymax = __builtin_amdgcn_readfirstlane(wave_max_int(nyw));
                const int xb = min(x0, rx0 + RW - 1) - rx0;
                // (a point far outside the image sits in a border cell with an EMPTY window whose x0 lies beyond the
                // staged region: the unconditional reads must stay inside it -- xb, cyr)
                // windows wider than 8 columns (border tiles: clipped queries reach twice as far) go in two column chunks, 8 +
                // 2 / 5 / 8: the squared column offsets of 16 columns, live across the row loop, do not fit the 64-register
                // budget beside the single-read form of the gradient loads (bwd_window_fast)
                if (nxmax <= 7) bwd_window_fast<7, L1, NEXT, IWD>(p, pt, x0, y0, nxw, nyw, nymax, ry0, RW, RP, xb, ldk, lg, lgn, ay, ax, an);
                else {
                    bwd_window_fast<8, L1, NEXT, IWD>(p, pt, x0, y0, nxw, nyw, nymax, ry0, RW, RP, xb, ldk, lg, lgn, ay, ax, an);
                    if (nxmax > 8) {
                        if (nxmax <= 10) bwd_window_fast<2, L1, NEXT, IWD>(p, pt, x0 + 8, y0, nxw - 8, nyw, nymax, ry0, RW, RP, xb + 8, ldk, lg, lgn, ay, ax, an);
                        else if (nxmax <= 13) bwd_window_fast<5, L1, NEXT, IWD>(p, pt, x0 + 8, y0, nxw - 8, nyw, nymax, ry0, RW, RP, xb + 8, ldk, lg, lgn, ay, ax, an);
                        else bwd_window_fast<8, L1, NEXT, IWD>(p, pt, x0 + 8, y0, nxw - 8, nyw, nymax, ry0, RW, RP, xb + 8, ldk, lg, lgn, ay, ax, an);
                    }
                }
            } else if (act) {
                KB_STAMP_SLOW
                // exact (distance, index) membership: queries with an excluded tie, or a window wider than KNN_BW_WMAX
                for (int cy = y0; cy <= y1; ++cy) {
                    const float dy = ((float)(cy * p.sp) + p.off) - pt.x;
                    const float dy2 = L1 ? fabsf(dy) : dy * dy;
                    const int ro = (cy - ry0) * RP - rx0;
                    for (int cx = x0; cx <= x1; ++cx) {
                        const float dx = ((float)(cx * p.sp) + p.off) - pt.y;
                        const float d = dy2 + (L1 ? fabsf(dx) : dx * dx);
                        const float dk = ldk[MPC_IDX(ro + cx, (long long)RW * RP + KNN_BW_WMAX)];
                        // (the K-th index: staged only with the flow_to_next gradient, else read where it lives -- a tie is rare)
                        const bool in = (d < dk) || (d == dk && i <= (reinterpret_cast<const int *>(knn_state)[BQ + (size_t)bt * p.G + (size_t)cy * p.wq + cx] & KNN_IDX_MASK));
                        if (in) {
                            const float2 e = lg[ro + cx];
                            if (IWD) { const float wi = __builtin_amdgcn_rcpf(d + 1e-9f); ay = fmaf(wi, e.x, ay); ax = fmaf(wi, e.y, ax); }
                            else { ay += e.x; ax += e.y; }
                            if (NEXT) { const float2 gq = lgn[ro + cx]; an.x += gq.x; an.y += gq.y; }
                        }
                    }
                }
            }
        } else if (act) {
            for (int cy = y0; cy <= y1; ++cy) {
                const float qy = (float)(cy * p.sp) + p.off;
                for (int cx = x0; cx <= x1; ++cx) {
                    const float qx = (float)(cx * p.sp) + p.off;
                    const float d = pair_dist(qy, qx, pt.x, pt.y, L1);
                    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
                    const float dk = knn_state[q];
                    if (d > dk) continue;
                    const int ikf = reinterpret_cast<const int *>(knn_state)[BQ + q];
                    if ((ikf & KNN_FAR_FLAG) || (d == dk && i > (ikf & KNN_IDX_MASK))) continue;
                    const float2 gq = gl2[(size_t)cy * p.wq + cx];
                    if (IWD) { const float wi = __builtin_amdgcn_rcpf((d + 1e-9f) * knn_state[2 * BQ + q]); ay = fmaf(wi, gq.x, ay); ax = fmaf(wi, gq.y, ax); }
                    else { ay += gq.x; ax += gq.y; }
                    if (has_next) { float2 gnq = gn2[(size_t)cy * p.wq + cx]; if (gnext_scale != nullptr) gnq = make_float2(gnq.x * gns, gnq.y * gns); an.x += gnq.x; an.y += gnq.y; }
                }
            }
        }
        if (act) {
            // no flow_to_next term: d traj(t_mid)[t] = -g goes straight to its place in the trajectory gradient (the value
            // k_knn_bwd_combine would write: -g + 0 - 0); k_knn_bwd_combine_direct then only adds the bins up
            if (gtraj_direct != nullptr)
                gtraj_direct[MPC_IDX(((size_t)b * (p.T + p.nb) + 1 + t) * p.n + i, (long long)p.B * (p.T + p.nb) * p.n)] = make_float2(-(wK * ay) + 0.f, -(wK * ax) + 0.f);
            else {
                tmp_g[MPC_IDX((size_t)bt * p.n + i, (long long)p.B * p.nb * p.n)] = make_float2(wK * ay, wK * ax);
                if (gnext != nullptr) tmp_a[MPC_IDX((size_t)bt * p.n + i, (long long)p.B * p.nb * p.n)] = make_float2(invK * an.x, invK * an.y);
            }
        }
    }
    KB_STAMP_END(tid, lblk, RQ, use_lds, anytie, total);
}

// ------------------------------------------------------------------------------------------
// backward of the FAR queries (no square of up to KNN_RCAP cells around them holds K points: queries inside a band the flow
// field emptied, whose K neighbours lie in a thin segment of a disc tens of pixels away; served by k_knn_tail or
// the fallback kernel).  Counting them in the tile maxima would make every point of the tiles around such a band search a
// window of hundreds of query cells for the few dozen of them that hold it; k_knn_bwd_tile therefore leaves them out, and
// here the search runs the other way round, as in the forward: the tile's bucketed points and cell offsets are staged in LDS,
// one thread per far query of the (sample, bin) walks the cell rows of the tile that its disc (centre, K-th distance) touches,
// tests the points of the chord and adds dL/dLUT to the accumulators of the members -- 64-bit fixed point in LDS (scaled by a
// power of two from the largest gradient of the list: integer sums, any order, bitwise reproducible), then added to what the
// gather wrote for the point.
// Work items: the (sample, bin, tile) entries the forward put on the list (knn_far_mark_tiles: the tiles a far query's disc can
// touch).  grid: a fixed number of workgroups (the list length is only known on the device; the lattice-like point sets of
// the benchmark have a few hundred items per step)
// ------------------------------------------------------------------------------------------
#define KNN_FAR_CAP 640       // points of a tile per round of accumulators (a border tile with its margin cells: 24 x 24)
template <bool L1, bool NEXT, bool IWD>
__global__ __launch_bounds__(256) void k_knn_bwd_far(const KnnParams p, const knn_cs_t *__restrict__ cell_start,
                                                     const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                                     const float *__restrict__ glut, const float *__restrict__ gnext,
                                                     const float *__restrict__ knn_state, const KnnLists ls,
                                                     float2 *__restrict__ tmp_g, float2 *__restrict__ tmp_a,
                                                     float2 *__restrict__ gtraj_direct, const float *__restrict__ gnext_scale) {
    constexpr int NA = NEXT ? 4 : 2;
    const float gns = (NEXT && gnext_scale != nullptr) ? gnext_scale[0] : 1.f;          // (see k_knn_bwd_tile)
    __shared__ unsigned long long s_acc[KNN_FAR_CAP * NA];
    __shared__ float2 s_pos[KNN_FAR_CAP];
    __shared__ unsigned short s_idx[KNN_FAR_CAP];
    __shared__ int s_cs[KNN_TROWS][KNN_TROWS + 1];          // first point (numbered within the tile) of every cell of the tile
    __shared__ int s_rowbase[KNN_TROWS + 1];
    __shared__ int s_rowg[KNN_TROWS];
    __shared__ float s_wm[4];
    __shared__ int s_nq;                                   // far queries of the current batch that touch the tile: {cell, K-th index},
    __shared__ int2 s_qc[KNN_FAR_QB];                      // K-th distance, gradient(s) in fixed point
    __shared__ float s_qd[KNN_FAR_QB];
    __shared__ longlong2 s_qg[KNN_FAR_QB * (NA / 2)];
    __shared__ float4 s_qf[IWD ? KNN_FAR_QB : 1];          // 'iwd': the gradient as floats and the query's normaliser (the weight differs per member)
    const int tid = threadIdx.x;
    const int ntx = knn_tiles_x(p.wq, p.m), nty = knn_tiles_y(p.hq, p.m), nt = ntx * nty;
    const int nwork = min(ls.ftlist[0], p.B * p.nb * nt);
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const float invK = 1.f / (float)p.K;
    // A workgroup's first item is item blockIdx.x; further ones come from a counter (the launch has up to twice as many items as
    // workgroups on band-heavy fields, and an item takes 30-100 us depending on how many discs touch the tile: with the static deal
    // w, w + gridDim.x the slowest pair set the launch's time).  At most one read-modify-write per workgroup and item.
    __shared__ int s_next_item;
    for (int w = (int)blockIdx.x; w < nwork; ) {
        const int item = ls.ftlist[1 + w];
        const int bt = item / nt, tile = item - bt * nt;
        const int by_ = tile / ntx, bx_ = tile - by_ * ntx;
        const int b = bt / p.nb, t = bt - b * p.nb;
        const int *fl = ls.far + (size_t)bt * (p.G + 1);
        const int nfar = min(fl[0], p.G);
        const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
        const float2 *sp_ = spos + (size_t)bt * p.n;
        const knn_idx_t *si_ = sidx + (size_t)bt * p.n;
        const bool has_next = NEXT && (gnext != nullptr) && (t < p.nb - 1);
        const float2 *gn2 = has_next ? reinterpret_cast<const float2 *>(gnext) + (size_t)(b * (p.nb - 1) + t) * p.G : nullptr;
        const float2 *gl2 = reinterpret_cast<const float2 *>(glut) + (size_t)bt * p.G;
        const float *dks = knn_state + (size_t)bt * p.G;
        const int *iks = reinterpret_cast<const int *>(knn_state) + BQ + (size_t)bt * p.G;
        // cells of the tile's points (a border tile owns the margin beside it) and their pixel extent; the outermost cells also
        // hold what lies beyond the margin: open on that side
        int ty0, ty1, tx0, tx1;
        knn_tile_cells(by_, p.hq, p.m, ty0, ty1); knn_tile_cells(bx_, p.wq, p.m, tx0, tx1);
        const int nrow = ty1 - ty0, ncol = tx1 - tx0;
        float ay0 = (float)(ty0 * p.sp) - 0.5f, ay1 = (float)(ty1 * p.sp) - 0.5f, ax0 = (float)(tx0 * p.sp) - 0.5f, ax1 = (float)(tx1 * p.sp) - 0.5f;
        if (ty0 == -p.m) ay0 = -INFINITY;
        if (ty1 == p.hq + p.m) ay1 = INFINITY;
        if (tx0 == -p.m) ax0 = -INFINITY;
        if (tx1 == p.wq + p.m) ax1 = INFINITY;
        // bucketed ranges of the tile rows; cell offsets of the tile
        if (tid < 64) {
            int gs = 0, ge = 0;
            if (tid < nrow) { gs = cs[knn_ci(p, ty0 + tid, tx0)]; ge = cs[knn_ci(p, ty0 + tid, tx1)]; }
            int run = ge - gs;
#pragma unroll
            for (int o2 = 1; o2 < KNN_TROWS; o2 <<= 1) { const int v = __shfl_up(run, o2, 64); if (tid >= o2) run += v; }
            if (tid < nrow) { s_rowg[tid] = gs; s_rowbase[tid + 1] = run; }
            if (tid == 0) s_rowbase[0] = 0;
        }
        // largest gradient of the list -> the power of two that scales the fixed-point sums (values below 2^40, up to 2^20 of them)
        // (list entries four at a time: their dependent loads -- entry, then what it points at -- travel together)
        float gm = 0.f;
        for (int e0 = 0; e0 < nfar; e0 += 4 * 256) {
            int cellv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + u * 256 + tid; cellv[u] = e < nfar ? fl[1 + e] : -1; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cellv[u] < 0) continue;
                const float2 g = gl2[cellv[u]];
                gm = fmaxf(gm, fmaxf(fabsf(g.x), fabsf(g.y)));
                if (has_next) { float2 gn = gn2[cellv[u]]; if (gnext_scale != nullptr) gn = make_float2(gn.x * gns, gn.y * gns); gm = fmaxf(gm, fmaxf(fabsf(gn.x), fabsf(gn.y))); }
            }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o2, 64));
        if ((tid & 63) == 0) s_wm[tid >> 6] = gm;
        __syncthreads();
        gm = fmaxf(fmaxf(s_wm[0], s_wm[1]), fmaxf(s_wm[2], s_wm[3]));
        const int total = s_rowbase[nrow];
        const bool nothing = total == 0 || !(gm > 0.f) || !(gm < INFINITY);      // (no point to receive anything / nothing to add: uniform)
        int ex = 0;
        (void)frexpf(nothing ? 1.f : gm, &ex);                              // gm = f * 2^ex, f in [0.5, 1)
        const double scale = ldexp(1.0, 40 - ex), inv_scale = ldexp(1.0, ex - 40);
        for (int i = tid; i < nrow * (ncol + 1); i += 256) {
            const int rr = i / (ncol + 1), cc = i - rr * (ncol + 1);
            s_cs[rr][cc] = s_rowbase[rr] + cs[knn_ci(p, ty0 + rr, tx0 + cc)] - s_rowg[rr];
        }
        for (int c0 = 0; c0 < total && !nothing; c0 += KNN_FAR_CAP) {
            for (int i = tid; i < KNN_FAR_CAP * NA; i += 256) s_acc[i] = 0ull;
            for (int li = tid; li < min(KNN_FAR_CAP, total - c0); li += 256) {          // this round's points: position and index
                const int pi = c0 + li;
                int lo = 0, hi = nrow;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rowbase[mid] <= pi) lo = mid; else hi = mid; }
                const int g = s_rowg[lo] + (pi - s_rowbase[lo]);
                s_pos[li] = sp_[g]; s_idx[li] = (unsigned short)si_[g];
            }
            __syncthreads();
            // The far queries of the (sample, bin) in batches of KNN_FAR_QB: first every thread tests two of them against the tile's
            // area and puts the ones that touch it on a list in LDS (centre, K-th key, gradient in fixed point); then the list is
            // worked off as (query, cell row of the tile) units spread evenly over the threads -- a thread per QUERY walked all
            // rows of the tile for the few queries that touch it while most lanes had none: the walk was the kernel's time.
            for (int e0 = 0; e0 < nfar; e0 += KNN_FAR_QB) {
              int cellv[KNN_FAR_QB / 256]; float dkv[KNN_FAR_QB / 256];
#pragma unroll
              for (int u = 0; u < KNN_FAR_QB / 256; ++u) { const int e = e0 + u * 256 + tid; cellv[u] = e < nfar ? fl[1 + e] : -1; }
#pragma unroll
              for (int u = 0; u < KNN_FAR_QB / 256; ++u) dkv[u] = cellv[u] >= 0 ? dks[cellv[u]] : -1.f;
              if (tid == 0) s_nq = 0;
              __syncthreads();
#pragma unroll
              for (int u = 0; u < KNN_FAR_QB / 256; ++u) {
                const int cell = cellv[u];
                if (cell < 0) continue;
                const int cy = cell / p.wq, cx = cell - cy * p.wq;
                const float dk = dkv[u];
                const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
                // does the disc of the query touch the tile's area at all?
                const float gy_ = fmaxf(0.f, fmaxf(qy - ay1, ay0 - qy)), gx_ = fmaxf(0.f, fmaxf(qx - ax1, ax0 - qx));
                if ((L1 ? gy_ + gx_ : gy_ * gy_ + gx_ * gx_) > dk) continue;
                const int ik = iks[cell] & KNN_IDX_MASK;
                const float2 g = gl2[cell];
                float2 gn = has_next ? gn2[cell] : make_float2(0.f, 0.f);
                if (has_next && gnext_scale != nullptr) gn = make_float2(gn.x * gns, gn.y * gns);
                const int k = atomicAdd(&s_nq, 1);
                s_qc[k] = make_int2(cell, ik);
                s_qd[k] = dk;
                s_qg[k * (NA / 2) + 0] = make_longlong2(__double2ll_rn((double)g.x * scale), __double2ll_rn((double)g.y * scale));
                if (IWD) s_qf[k] = make_float4(g.x, g.y, knn_state[2 * BQ + (size_t)bt * p.G + cell], 0.f);
                if (NEXT) s_qg[k * (NA / 2) + 1] = make_longlong2(__double2ll_rn((double)gn.x * scale), __double2ll_rn((double)gn.y * scale));
              }
              __syncthreads();
              const int units = s_nq * nrow;
              const float inv_nrow = 1.f / (float)nrow;
              for (int un = tid; un < units; un += 256) {
                const int qi = min((int)(((float)un + 0.5f) * inv_nrow), s_nq - 1), rr = un - qi * nrow;      // (un < 2^16: the margin 0.5 / nrow dwarfs the rounding)
                const int2 qc = s_qc[qi];
                const int cell = qc.x, ik = qc.y;
                const int cy = cell / p.wq, cx = cell - cy * p.wq;
                const float dk = s_qd[qi];
                const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
                // a point of this cell row is at least dyc away along y; along x it then lies within wx of the query
                const float dyc = fmaxf((float)abs(ty0 + rr - cy) - 0.5f, 0.f) * (float)p.sp;
                const float w2 = L1 ? dk - dyc : dk - dyc * dyc;
                if (w2 < 0.f) continue;
                // cells cx - xr .. cx + xr: (|x - cx| - 0.5) sp < wx, i.e. |x - cx| <= floor(wx / sp + 0.5) (one ulp of slack in wx)
                const int xr = (int)((L1 ? w2 : sqrtf(w2) * 1.000001f) / (float)p.sp + 0.5f);
                int xa = max(cx - xr, tx0), xb = min(cx + xr, tx1 - 1);
                if (cx - xr <= -p.m) xa = tx0;                   // (the outermost column holds everything beyond it)
                if (cx + xr >= p.wq + p.m - 1) xb = tx1 - 1;
                if (xa > xb) continue;
                const int ja = max(s_cs[rr][xa - tx0] - c0, 0), jb = min(s_cs[rr][xb + 1 - tx0] - c0, KNN_FAR_CAP);
                if (ja >= jb) continue;
                const longlong2 fg = s_qg[qi * (NA / 2) + 0];
                for (int li = ja; li < jb; ++li) {
                    const float2 pj = s_pos[li];
                    const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                    if (d > dk || (d == dk && (int)s_idx[li] > ik)) continue;
                    if (IWD) {
                        // the member's weight as the formula has it (focus.py:159-161): (1 / (d + 1e-9)) / sum -- at most 1, so the
                        // fixed-point scale of the largest gradient still holds
                        const float4 qf = s_qf[qi];
                        const float wgt = (1.f / (d + 1e-9f)) / qf.z;
                        atomicAdd(&s_acc[li * NA + 0], (unsigned long long)__double2ll_rn((double)(wgt * qf.x) * scale));
                        atomicAdd(&s_acc[li * NA + 1], (unsigned long long)__double2ll_rn((double)(wgt * qf.y) * scale));
                    } else {
                        atomicAdd(&s_acc[li * NA + 0], (unsigned long long)fg.x);
                        atomicAdd(&s_acc[li * NA + 1], (unsigned long long)fg.y);
                    }
                    if (NEXT && has_next) {
                        const longlong2 fn = s_qg[qi * (NA / 2) + 1];
                        atomicAdd(&s_acc[li * NA + 2], (unsigned long long)fn.x); atomicAdd(&s_acc[li * NA + 3], (unsigned long long)fn.y);
                    }
                }
              }
              __syncthreads();          // (the list is rewritten by the next batch)
            }
            __syncthreads();
            for (int li = tid; li < min(KNN_FAR_CAP, total - c0); li += 256) {
                const long long ay = (long long)s_acc[li * NA + 0], ax = (long long)s_acc[li * NA + 1];
                const long long any_ = NEXT ? (long long)s_acc[li * NA + 2] : 0ll, anx = NEXT ? (long long)s_acc[li * NA + 3] : 0ll;
                if ((ay | ax | any_ | anx) == 0ll) continue;
                const int i = (int)s_idx[li];
                const float wKf = IWD ? 1.f : invK;              // ('iwd': the weights are in the sums)
                const float vy = wKf * (float)((double)ay * inv_scale), vx = wKf * (float)((double)ax * inv_scale);
                if (gtraj_direct != nullptr) {
                    float2 *dst = gtraj_direct + ((size_t)b * (p.T + p.nb) + 1 + t) * p.n + i;
                    const float2 o = *dst;
                    *dst = make_float2(o.x - vy, o.y - vx);
                } else {
                    float2 *dst = tmp_g + (size_t)bt * p.n + i;
                    const float2 o = *dst;
                    *dst = make_float2(o.x + vy, o.y + vx);
                    if (NEXT && gnext != nullptr) {
                        float2 *da = tmp_a + (size_t)bt * p.n + i;
                        const float2 oa = *da;
                        *da = make_float2(oa.x + invK * (float)((double)any_ * inv_scale), oa.y + invK * (float)((double)anx * inv_scale));
                    }
                }
            }
            __syncthreads();
        }
        if (tid == 0) s_next_item = (int)gridDim.x + atomicAdd(knn_bwd_far_next(ls), 1);
        __syncthreads();          // (the next item reuses the tables)
        w = s_next_item;
    }
}

// backward, step 2: one thread per (sample, trajectory point): combine the per-bin partials.
//   d traj(t_ref)[tr] = sum_t g[t][tr];   d traj(t_mid)[t] = -sum_tr g[t][tr] - a[t] + a[t-1]
__global__ __launch_bounds__(256) void k_knn_bwd_combine(const KnnParams p, const float2 *__restrict__ tmp_g,
                                                         const float2 *__restrict__ tmp_a,   // nullptr: no flow_to_next term
                                                         float *__restrict__ gtraj) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (size_t)p.B * p.n) return;
    const int b = (int)(gi / p.n), i = (int)(gi - (size_t)b * p.n);
    float2 *g2 = reinterpret_cast<float2 *>(gtraj) + (size_t)b * (p.T + p.nb) * p.n;
    if (p.T == 1) {                               // one pass over the partials
        float sy = 0.f, sx = 0.f;
        float2 carry = make_float2(0.f, 0.f);
#pragma unroll 5
        for (int t = 0; t < p.nb; ++t) {
            const float2 g = tmp_g[(size_t)(b * p.nb + t) * p.n + i];
            const float2 a = tmp_a ? tmp_a[(size_t)(b * p.nb + t) * p.n + i] : make_float2(0.f, 0.f);
            sy += g.x; sx += g.y;
            g2[(size_t)(1 + t) * p.n + i] = make_float2(-g.x + carry.x - a.x, -g.y + carry.y - a.y);
            carry = a;
        }
        g2[i] = make_float2(sy, sx);
        return;
    }
    for (int tr = 0; tr < p.T; ++tr) {
        float sy = 0.f, sx = 0.f;
        for (int t = 0; t < p.nb; ++t) {
            const float2 g = tmp_g[((size_t)(b * p.nb + t) * p.n + i) * p.T + tr];
            sy += g.x; sx += g.y;
        }
        g2[(size_t)tr * p.n + i] = make_float2(sy, sx);
    }
    float2 carry = make_float2(0.f, 0.f);
    for (int t = 0; t < p.nb; ++t) {
        float my = 0.f, mx = 0.f;
        for (int tr = 0; tr < p.T; ++tr) {
            const float2 g = tmp_g[((size_t)(b * p.nb + t) * p.n + i) * p.T + tr];
            my -= g.x; mx -= g.y;
        }
        const float2 a = tmp_a[(size_t)(b * p.nb + t) * p.n + i];
        my += carry.x - a.x; mx += carry.y - a.y;
        carry = a;
        g2[(size_t)(p.T + t) * p.n + i] = make_float2(my, mx);
    }
}

// The same for num_tref == 1 WITH the flow_to_next term (both EVIMO2 configs), spread over the bins (round 6): a workgroup = 64 points
// x four wavefronts, wavefront w takes the bins t = w, w + 4, ... (every load coalesced over the points, a[t - 1] read again instead of
// carried), the four partial sums of d traj(t_ref) are added in wavefront order.  One thread per point walked all 41 bins of C4 one
// after the other on 75 workgroups: 17 us of a 0.26 ms step (B = 1), a chain of dependent rounds on an idle chip.
__global__ __launch_bounds__(256) void k_knn_bwd_combine_bins(const KnnParams p, const float2 *__restrict__ tmp_g,
                                                              const float2 *__restrict__ tmp_a, float *__restrict__ gtraj) {
    __shared__ float2 s_part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t gi = (size_t)blockIdx.x * 64 + lane;
    const bool on = gi < (size_t)p.B * p.n;
    const int b = on ? (int)(gi / p.n) : 0, i = on ? (int)(gi - (size_t)b * p.n) : 0;
    float2 *g2 = reinterpret_cast<float2 *>(gtraj) + (size_t)b * (1 + p.nb) * p.n;
    float sy = 0.f, sx = 0.f;
    if (on) {
#pragma unroll 4
        for (int t = w; t < p.nb; t += 4) {
            const float2 g = tmp_g[(size_t)(b * p.nb + t) * p.n + i];
            const float2 a = tmp_a[(size_t)(b * p.nb + t) * p.n + i];
            const float2 c = t > 0 ? tmp_a[(size_t)(b * p.nb + t - 1) * p.n + i] : make_float2(0.f, 0.f);
            sy += g.x; sx += g.y;
            g2[(size_t)(1 + t) * p.n + i] = make_float2(-g.x + c.x - a.x, -g.y + c.y - a.y);
        }
    }
    s_part[w][lane] = make_float2(sy, sx);
    __syncthreads();
    if (w == 0 && on) {
        const float2 p0 = s_part[0][lane], p1 = s_part[1][lane], p2 = s_part[2][lane], p3 = s_part[3][lane];
        g2[i] = make_float2(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y);
    }
}

// num_tref == 1 without the flow_to_next term, after a k_knn_bwd_tile that wrote d traj(t_mid) in place: only
//   d traj(t_ref)[i] = sum_t g[t][i] = sum_t -(d traj(t_mid)[t][i])    (same order, same bits: a - (-g) == a + g)
// is left -- 15 coalesced reads and one write per point instead of 15 + 16.
__global__ __launch_bounds__(256) void k_knn_bwd_combine_direct(const KnnParams p, float *__restrict__ gtraj) {
    // (one thread per point: with 15 bins the launch is bound by its 32 MB at C3; the bin-parallel form of k_knn_bwd_combine_bins was
    // measured here too, round 6: 10.4 us against 9.8 at C3, 6.4 against 6.4 at B = 1)
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (size_t)p.B * p.n) return;
    const int b = (int)(gi / p.n), i = (int)(gi - (size_t)b * p.n);
    float2 *g2 = reinterpret_cast<float2 *>(gtraj) + (size_t)b * (1 + p.nb) * p.n;
    float sy = 0.f, sx = 0.f;
#pragma unroll 5
    for (int t = 0; t < p.nb; ++t) {
        const float2 m = g2[(size_t)(1 + t) * p.n + i];
        sy -= m.x; sx -= m.y;
    }
    g2[i] = make_float2(sy, sx);
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
// search radius of a query (cells) at the MEAN point density: the smallest square that can hold K points and pass the ring
// bound.  The strip kernel starts from it (and chooses every query's own radius from the summed-area table of the cell
// counts: a flow field thins the points out where it diverges and packs them where it converges); it is also the depth of the
// border classes of the tile maxima.
int mpc_knn_r_init(const mpc_shape *s) {
    const double dens = (double)s->n / ((double)s->hq * s->wq);
    // (the ball of K points at the mean density: a disc, pi r^2; with the L1 norm a diamond, 2 r^2)
    const double ball = (s->flags & MPC_F_DIST_L1) ? 2.0 : 3.14159265;
    int r_init = (int)ceil(sqrt((double)s->K / ball / (dens > 0 ? dens : 1.0)) - 0.5);
    return r_init < 1 ? 1 : r_init;
}

// margin of the bucket grid (knn_device.h): the strip kernel never reaches the outermost ring
int mpc_knn_margin(const mpc_shape *) { return KNN_MARGIN; }
int mpc_knn_tiles(const mpc_shape *s) { return knn_tiles_x(s->wq, KNN_MARGIN) * knn_tiles_y(s->hq, KNN_MARGIN); }

// true where the forward is the strip kernel + fallback (num_tref == 1: the shipped configurations)
static bool knn_fwd_is_strip(const mpc_shape *s, const int32_t *idx_out) {
    return idx_out == nullptr && mpc_knn_strip_usable(s, mpc_knn_r_init(s));
}
// true where the backward of this shape is k_knn_bwd_tile + k_knn_bwd_far (the gather with per-tile reaches)
static bool knn_bwd_is_tile(const mpc_shape *s) {
    return s->B > 0 && s->T == 1;          // (round 6: 'iwd' too -- the gather weights a member by its distance)
}
// the fallback kernel lists the queries it served for k_knn_bwd_far (and keeps them out of the tile maxima)
bool mpc_knn_uses_far_list(const mpc_shape *s) { return knn_bwd_is_tile(s) && knn_fwd_is_strip(s, nullptr); }

// The counting sort of the points of a (sample, bin): S workgroups per (sample, bin) with the counters in LDS -- as many as
// keep the launch within one workgroup per CU (measured at C3: 210 workgroups 39 us, split three ways 68 us; at B = 1: 15
// workgroups 33 us, split eight ways 19 us) -- or, returned as true, the global-memory sort (bucket grids beyond the LDS).
static bool knn_sort_plan(const mpc_shape *s, int *S_out, size_t *lds_out) {
    const int m = mpc_knn_margin(s), hb = s->hq + 2 * m, wb = s->wq + 2 * m;
    int S = 256 / ((s->B > 0 ? s->B : 1) * s->nb);
    if (S > 8) S = 8;
    if (S > hb) S = hb;
    if (S < 1) S = 1;
    size_t tail = (size_t)((s->n + 1) / 2 * 2) * 2 + (size_t)((s->n + 31) / 32) * 4;              // index array + bitmap of a crowded cell
    const size_t sat_tmp = (size_t)((hb + 7) / 8) * (wb + 1) * 4;                                   // scratch of the summed-area table (S == 1)
    if (S == 1 && sat_tmp > tail) tail = sat_tmp;
    const size_t lds = (size_t)((hb + S - 1) / S) * wb * 4 + tail;
    *S_out = S; *lds_out = lds;
    return (int64_t)hb * wb > MPC_KNN_LDS_SORT_CELLS || lds > 145 * 1024;
}
bool mpc_knn_big_sort(const mpc_shape *s) { int S; size_t l; return knn_sort_plan(s, &S, &l); }

// the work lists of the forward in the workspace (knn_device.h: KnnLists)
static KnnLists knn_lists(const mpc_shape *s, const mpc_ws_layout &L, void *ws) {
    KnnLists ls;
    ls.fail = (int *)((char *)ws + L.off_knn_fail);
    ls.retry = (int *)((char *)ws + L.off_knn_retry);
    ls.farstrip = (int *)((char *)ws + L.off_knn_farstrip);
    const bool far = mpc_knn_uses_far_list(s);
    ls.far = far ? (int *)((char *)ws + L.off_knn_far) : nullptr;
    ls.ftlist = far ? (int *)((char *)ws + L.off_knn_ftlist) : nullptr;
    ls.ftbits = far ? (unsigned *)((char *)ws + L.off_knn_ftbits) : nullptr;
    ls.ftwords = (mpc_knn_tiles(s) + 31) / 32;
    ls.again = (unsigned *)((char *)ws + L.off_knn_again);
    ls.again_words = s->hq * ((s->wq + 31) / 32);
    ls.chord = (unsigned char *)ws + L.off_knn_chord;
    ls.grow = ls.again + (size_t)(s->B > 0 ? s->B : 1) * s->nb * ls.again_words;
    return ls;
}

static int set_max_lds(const void *fn, const char *who) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { mpc_set_error("%s: %s", who, hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int mpc_knn_lut_fwd(const mpc_shape *s, const float *traj, float *flow_lut, float *flow_next,
                               float *knn_state, int32_t *idx_out, void *ws, void *stream) {
    return mpc_knn_lut_fwd_ex(s, traj, flow_lut, flow_next, knn_state, idx_out, ws, stream, 0, nullptr, nullptr);
}

// zero_event_counters: the first kernel also zeroes the bucket counters of mpc_event_splat_fwd (mpc_focus_fwd: one launch less);
// events (with it): the strip kernel also counts these event rows per backward bucket (ev_count_device.h).  *done receives
// what of the two was done: bit 0 zeroed, bit 1 counted.
int mpc_knn_lut_fwd_ex(const mpc_shape *s, const float *traj, float *flow_lut, float *flow_next, float *knn_state,
                       int32_t *idx_out, void *ws, void *stream, int zero_event_counters, const float *events, int *done) {
    if (done) *done = 0;
    MPC_CHECK_ARG(s && traj && flow_lut && knn_state && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(!(s->flags & MPC_F_WANT_NEXT) || flow_next, MPC_E_NULL, "flow_next is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    MPC_CHECK_ARG(s->n >= s->K && s->K >= 1, MPC_E_SHAPE, "need 1 <= K <= n");
    MPC_CHECK_ARG(s->n < 65536, MPC_E_UNSUPPORTED, "more than 65535 trajectories per sample");
    const KnnParams p = knn_params(s);
    if (s->B == 0) return 0;
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    knn_cs_t *cell_start = (knn_cs_t *)((char *)ws + L.off_cell_start);
    knn_cs_t *sat = (knn_cs_t *)((char *)ws + L.off_knn_sat);
    float2 *spos = (float2 *)((char *)ws + L.off_spos);
    knn_idx_t *sidx = (knn_idx_t *)((char *)ws + L.off_sidx);
    float *tile_dkmax = knn_state + 3 * (size_t)s->B * s->nb * p.G;
    const KnnLists ls = knn_lists(s, L, ws);
    const int ntiles = knn_tiles_x(s->wq, p.m) * knn_tiles_y(s->hq, p.m);
    int *zero_ptr = (int *)((char *)ws + L.off_fcount);
    const int zero_words = (zero_event_counters && L.strip_rows > 0) ? L.nfb + 2 * L.nbb + 8 : 0;
    EvCountArgs evc = (zero_words > 0) ? mpc_event_count_args(s, events, ws) : EvCountArgs{};
    const bool strip = knn_fwd_is_strip(s, idx_out);
    // (the fallback kernel's first B workgroups finish the count; the count rides in workgroups of the strip kernel)
    if (s->B > 256 || !strip || !mpc_knn_strip_counts_events(s, &evc)) evc = EvCountArgs{};
    static mpc_device_once attr_once;   // raising the dynamic-LDS cap: idempotent, once per device
    if (attr_once.need()) {
        if ((rc = set_max_lds((const void *)k_knn_bucket<true, 8>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bucket<true, 16>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bucket<true, 20>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bucket<true, 24>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bucket<false, 1>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_query<256>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_query<512>, __func__))) return rc;
        attr_once.mark();
    }
    int S; size_t sort_lds;
    bool sat_launch = true;          // (the LDS sort of a whole (sample, bin) builds the summed-area table itself)
    if (knn_sort_plan(s, &S, &sort_lds)) {
        int *cursor = (int *)((char *)ws + L.off_knn_cursor);
        const int e = mpc_zero_async(cursor, (size_t)s->B * s->nb * p.Gb * sizeof(int), st);
        if (e) return e;
        const dim3 gp(mpc_cdiv(s->n, 256), s->B * s->nb), gc(mpc_cdiv(p.Gb, 4), s->B * s->nb);
        MPC_LAUNCH(k_knn_bucket_count, gp, dim3(256), 0, st, p, traj, cursor);
        MPC_LAUNCH(k_knn_bucket_scan, dim3(s->B * s->nb), dim3(1024), 0, st, p, cursor, cell_start, tile_dkmax, ntiles, ls, zero_ptr, zero_words);
        MPC_LAUNCH(k_knn_bucket_scatter, gp, dim3(256), 0, st, p, traj, cursor, spos, sidx);
        MPC_LAUNCH(k_knn_bucket_order, gc, dim3(256), 0, st, p, cell_start, spos, sidx, traj);
    } else {
        sat_launch = S != 1;
#define KB_LAUNCH(C_, N_) MPC_LAUNCH((k_knn_bucket<C_, N_>), dim3(s->B * s->nb * S), dim3(1024), sort_lds, st, p, traj, cell_start, strip ? sat : nullptr, spos, sidx, S, tile_dkmax, ntiles, ls, zero_ptr, zero_words)
        if (s->n <= 8 * 1024) KB_LAUNCH(true, 8);
        else if (s->n <= 16 * 1024) KB_LAUNCH(true, 16);
        else if (s->n <= 20 * 1024) KB_LAUNCH(true, 20);
        else if (s->n <= KNN_BUCKET_NPT * 1024) KB_LAUNCH(true, 24);
        else KB_LAUNCH(false, 1);
#undef KB_LAUNCH
    }
    MPC_CHECK_LAUNCH();
    const int r_init = mpc_knn_r_init(s);
    // fast path (num_tref == 1, the shipped configurations): strip kernel + fallback (knn_strip.hip)
    if (strip) {
        if (sat_launch) {
            const int scol = knn_sat_cols(p.hb);
            MPC_LAUNCH(k_knn_sat, dim3(s->B * s->nb, (p.wb + 1 + scol - 1) / scol), dim3(KNN_SAT_NT), (size_t)((p.hb + 7) / 8) * scol * sizeof(int), st, p, cell_start, sat);
            MPC_CHECK_LAUNCH();
        }
        rc = mpc_knn_strip_launch(s, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, &ls, r_init,
                                  evc.events ? &evc : nullptr, st);
        if (!rc && done) *done = (zero_words > 0 ? 1 : 0) | (evc.events ? 2 : 0);
        return rc;
    }
    // everything else (num_tref > 1, idx_out wanted, K or densities the strip kernel does not hold): the tile kernel, one
    // thread per query with a per-thread radix histogram (knn_one_query)
    const double dens = (double)s->n / (double)p.G;
    int RH = r_init + 1;                                   // halo of the staged region: one ring of slack
    if (RH > 16) RH = 16;
    const int RW = 16 + 2 * RH;
    // LDS per workgroup decides how many wavefronts a CU holds, and the kernel is latency bound: measured at
    // C3 with 256-thread workgroups, 2 / 3 / 4 / 6 per CU = 1180 / 850 / 685 / 550 us.  The staging capacity is
    // what is left of the per-workgroup share after the histogram columns, provided it still holds 1.25x the
    // mean region; 16x32-query workgroups (512 threads) carry less halo per query and are used when they
    // reach more wavefronts per CU.
    const size_t per_pt = 8 + 2 + (s->T == 1 ? 8 : 0) + (p.want_next ? 8 : 0);
    int best_nt = 256, best_cap = 64, best_waves = 0;
    for (int nt = 256; nt <= 512; nt *= 2) {
        if (nt == 512 && s->hq <= 16) continue;
        const int RWY = nt / 16 + 2 * RH;
        const size_t fixed = (size_t)KNN_HW * nt * 4 + (size_t)RWY * (RW + 1) * 2 + 64;
        const double mean_pts = dens * RW * RWY;
        for (int blocks = 2048 / nt; blocks >= 1; --blocks) {
            const size_t share = (size_t)160 * 1024 / blocks;
            const size_t budget = (share > 64 * 1024 ? 64 * 1024 : share) - 1024;  // static LDS + allocation granule
            if (budget <= fixed) continue;
            int c = (int)((budget - fixed) / per_pt / 64 * 64);
            if (c > 65472) c = 65472;                                              // 16-bit staged offsets
            if (c >= (int)(1.25 * mean_pts) + 64 || blocks == 1) {
                const int full = ((int)(1.5 * mean_pts) + 128 + 63) / 64 * 64;
                if (blocks == 1 && c > full) c = full;
                if (c < 64) c = 64;
                if (blocks * nt / 64 > best_waves) { best_waves = blocks * nt / 64; best_nt = nt; best_cap = c; }
                break;
            }
        }
    }
    const int cap = best_cap;
    const int RWYb = best_nt / 16 + 2 * RH;
    const size_t lds = (size_t)KNN_HW * best_nt * 4 + (size_t)RWYb * (RW + 1) * 2 + 64 + per_pt * cap;
    const int gx = mpc_cdiv(s->wq, 16), gy = mpc_cdiv(s->hq, best_nt / 16);
    const dim3 grid(((int64_t)gx * gy * s->B * s->nb + 7) / 8 * 8);
    if (best_nt == 512)
        MPC_LAUNCH(k_knn_query<512>, grid, dim3(512), lds, st, p, traj, cell_start, spos, sidx, flow_lut,
                           flow_next, knn_state, idx_out, tile_dkmax, r_init, RH, cap, 1, gx, gy);
    else
        MPC_LAUNCH(k_knn_query<256>, grid, dim3(256), lds, st, p, traj, cell_start, spos, sidx, flow_lut,
                           flow_next, knn_state, idx_out, tile_dkmax, r_init, RH, cap, 1, gx, gy);
    MPC_CHECK_LAUNCH();
    if (done) *done = zero_words > 0 ? 1 : 0;
    return 0;
}

bool mpc_knn_reach_job(const mpc_shape *s, const float *knn_state, void *ws, KnnReachJob *job) {
    job->on = 0;
    if (!s || !knn_state || !ws || mpc_validate_shape(s)) return false;
    const mpc_ws_layout L = mpc_layout(s);
    if (!knn_bwd_is_tile(s)) return false;
    job->p = knn_params(s);
    job->tile_dkmax = knn_state + 3 * (size_t)s->B * s->nb * job->p.G;
    job->reach = (float *)((char *)ws + L.off_knn_reach);
    job->gx = knn_tiles_x(s->wq, job->p.m); job->gy = knn_tiles_y(s->hq, job->p.m);
    job->bd = knn_band_depth(mpc_knn_r_init(s));
    // (below one round of the gather's workgroups the side job costs the host kernel more than the gather gains: C2, B = 1 x 15
    // bins = 1 200 workgroups: +3.5 us on k_lut_accum's slowest workgroup for -1.5 on the gather; C4, 3 280: +3.7 / -6)
    job->on = ((int64_t)job->gx * job->gy * s->B * s->nb >= 2048) ? 1 : 0;
    return job->on != 0;
}

extern "C" int mpc_knn_lut_bwd(const mpc_shape *s, const float *traj, const float *grad_flow_lut,
                               const float *grad_flow_next, const float *knn_state, float *grad_traj,
                               void *ws, void *stream) {
    return mpc_knn_lut_bwd_ex(s, traj, grad_flow_lut, grad_flow_next, knn_state, grad_traj, ws, stream, 0, nullptr, nullptr);
}

// reach_ready: the reaches of the tiles are in the workspace already (the event backward's kernel computed them: mpc_focus_bwd)
// gnext_scale (device scalar or null): dL/dflow_next = gnext_scale[0] * grad_flow_next.  The tile gather and the far backward
// multiply as they read; the general gather gets the product from a pass of mpc_scale into gnext_scratch [like grad_flow_next].
int mpc_knn_lut_bwd_ex(const mpc_shape *s, const float *traj, const float *grad_flow_lut,
                       const float *grad_flow_next, const float *knn_state, float *grad_traj,
                       void *ws, void *stream, int reach_ready, const float *gnext_scale, float *gnext_scratch) {
    MPC_CHECK_ARG(s && traj && grad_flow_lut && knn_state && grad_traj && ws, MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    if (s->B == 0) return 0;
    if (grad_flow_next && gnext_scale && !knn_bwd_is_tile(s)) {
        MPC_CHECK_ARG(gnext_scratch, MPC_E_NULL, "grad_next_scratch is null");
        if ((rc = mpc_scale(grad_flow_next, gnext_scale, gnext_scratch, (int64_t)s->B * (s->nb - 1) * s->hq * s->wq * 2, stream))) return rc;
        grad_flow_next = gnext_scratch; gnext_scale = nullptr;
    }
    if (!grad_flow_next) gnext_scale = nullptr;
    const KnnParams p = knn_params(s);
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    const knn_cs_t *cell_start = (const knn_cs_t *)((char *)ws + L.off_cell_start);
    const float2 *spos = (const float2 *)((char *)ws + L.off_spos);
    const knn_idx_t *sidx = (const knn_idx_t *)((char *)ws + L.off_sidx);
    float2 *tmp_g = (float2 *)((char *)ws + L.off_knn_tmp_g);
    float2 *tmp_a = (float2 *)((char *)ws + L.off_knn_tmp_a);
    const float *tile_dkmax = knn_state + 3 * (size_t)s->B * s->nb * p.G;
    float *reach = (float *)((char *)ws + L.off_knn_reach);
    static mpc_device_once attr_once;   // raising the dynamic-LDS cap: idempotent, once per device
    if (attr_once.need()) {
        if ((rc = set_max_lds((const void *)k_knn_bwd_points<16>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<false, false, false>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<false, true, false>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<true, false, false>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<true, true, false>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<false, false, true>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<false, true, true>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<true, false, true>, __func__))) return rc;
        if ((rc = set_max_lds((const void *)k_knn_bwd_tile<true, true, true>, __func__))) return rc;
        attr_once.mark();
    }
    const int gxb = knn_tiles_x(s->wq, p.m), gyb = knn_tiles_y(s->hq, p.m), ntiles = gxb * gyb;
    if (knn_bwd_is_tile(s)) {
        const int RWm = 16 + 2 * KNN_RQ_MAX;
        const size_t ldsb = grad_flow_next ? ((size_t)RWm * (RWm > KNN_BW_PITCH_NEXT ? RWm : KNN_BW_PITCH_NEXT) + KNN_BW_WMAX) * 20
                                           : ((size_t)RWm * KNN_BW_PITCH + KNN_BW_WMAX) * 12;          // ('iwd': the same arrays -- the normaliser is in the staged gradient)
        const dim3 gridb(((int64_t)gxb * gyb * s->B * s->nb + 7) / 8 * 8);
        float2 *direct = (grad_flow_next == nullptr) ? reinterpret_cast<float2 *>(grad_traj) : nullptr;
        // The reach of every tile NOT as the first phase of every workgroup: the gather is a chain of phases per workgroup at
        // eight workgroups per CU -- 8.1 of a workgroup's 14.6 us pass before its window loop (tools/bwd_stamp_probe.py) -- and
        // without the two dependent round trips of the reach phase it runs 132 -> 117 us at C3.  mpc_focus_bwd has the event
        // backward's kernel compute the reaches on the side (reach_ready); called alone, a launch of its own does it (~6-8 us)
        // where that pays: from about four rounds of workgroups (C3: 8.2; B = 1: 0.6 -- there it would cost 3 us).
        const bool reach_launch = !reach_ready && (int64_t)gxb * gyb * s->B * s->nb >= 4 * 2048;
        const float *reach_pre = reach_ready ? reach : nullptr;
        int *far_next = nullptr;              // (the counter of the far backward's dynamic items: zeroed by the gather's launch)
        if (mpc_knn_uses_far_list(s)) { const KnnLists ls0 = knn_lists(s, L, ws); far_next = reinterpret_cast<int *>(ls0.chord + 896); }
        if (reach_launch) {
            const size_t rl = ((size_t)gxb * gyb * (KNN_NCLS + 1) + 16) * sizeof(float);
            if (p.l1) MPC_LAUNCH(k_knn_reach_tiles<true>, dim3(s->B * s->nb), dim3(256), rl, st, p, tile_dkmax, reach, gxb, gyb, knn_band_depth(mpc_knn_r_init(s)));
            else MPC_LAUNCH(k_knn_reach_tiles<false>, dim3(s->B * s->nb), dim3(256), rl, st, p, tile_dkmax, reach, gxb, gyb, knn_band_depth(mpc_knn_r_init(s)));
            reach_pre = reach;
        }
#define KB_LAUNCH(L1_, NEXT_)                                                                                            \
        do { if (p.iwd) KB_LAUNCH3(L1_, NEXT_, true); else KB_LAUNCH3(L1_, NEXT_, false); } while (0)
#define KB_LAUNCH3(L1_, NEXT_, IWD_)                                                                                     \
        MPC_LAUNCH((k_knn_bwd_tile<L1_, NEXT_, IWD_>), gridb, dim3(256), ldsb, st, p, cell_start, spos, sidx, grad_flow_lut, \
                           grad_flow_next, knn_state, tile_dkmax, tmp_g, tmp_a, direct, reach_pre, gxb, gyb, knn_band_depth(mpc_knn_r_init(s)), far_next, gnext_scale KB_STAMP_ARG(ws, L))
        if (p.l1) { if (grad_flow_next) KB_LAUNCH(true, true); else KB_LAUNCH(true, false); }
        else { if (grad_flow_next) KB_LAUNCH(false, true); else KB_LAUNCH(false, false); }
#undef KB_LAUNCH
#undef KB_LAUNCH3
        MPC_CHECK_LAUNCH();
        if (mpc_knn_uses_far_list(s)) {
            // the queries the forward's fallback kernel served (left out of the gather above)
            const KnnLists ls = knn_lists(s, L, ws);
            // (no more workgroups than the list can hold items: at B = 1 a quarter of KNN_FAR_BLOCKS, and the launch usually finds
            // a few items or none)
            const long long far_items_max = (long long)s->B * s->nb * knn_tiles_x(p.wq, p.m) * knn_tiles_y(p.hq, p.m);
            const unsigned far_blocks = (unsigned)(far_items_max < KNN_FAR_BLOCKS ? (far_items_max < 1 ? 1 : far_items_max) : KNN_FAR_BLOCKS);
#define KF_LAUNCH(L1_, NEXT_)                                                                                            \
            do { if (p.iwd) KF_LAUNCH3(L1_, NEXT_, true); else KF_LAUNCH3(L1_, NEXT_, false); } while (0)
#define KF_LAUNCH3(L1_, NEXT_, IWD_)                                                                                     \
            MPC_LAUNCH((k_knn_bwd_far<L1_, NEXT_, IWD_>), dim3(far_blocks), dim3(256), 0, st, p, cell_start, spos, sidx, grad_flow_lut, grad_flow_next, \
                               knn_state, ls, tmp_g, tmp_a, direct, gnext_scale)
            if (p.l1) { if (grad_flow_next) KF_LAUNCH(true, true); else KF_LAUNCH(true, false); }
            else { if (grad_flow_next) KF_LAUNCH(false, true); else KF_LAUNCH(false, false); }
#undef KF_LAUNCH
#undef KF_LAUNCH3
            MPC_CHECK_LAUNCH();
        }
        const int64_t totalb = (int64_t)s->B * s->n;
        if (direct) MPC_LAUNCH(k_knn_bwd_combine_direct, dim3(mpc_cdiv(totalb, 256)), dim3(256), 0, st, p, grad_traj);
        else MPC_LAUNCH(k_knn_bwd_combine_bins, dim3(mpc_cdiv(totalb, 64)), dim3(256), 0, st, p, tmp_g, tmp_a, grad_traj);          // (not direct: grad_flow_next given)
        MPC_CHECK_LAUNCH();
        return 0;
    }
    // num_tref > 1 or 'iwd': the general gather, one thread per bucketed point and tile (k_knn_bwd_points)
    MPC_LAUNCH(k_knn_reach, dim3(s->B * s->nb), dim3(256), (size_t)ntiles * sizeof(float), st, p, tile_dkmax, reach);
    MPC_CHECK_LAUNCH();
    const int RWmax = 16 + 2 * KNN_RQ_MAX;
    const size_t lds = (s->T == 1 && !p.iwd) ? (size_t)RWmax * RWmax * (16 + (grad_flow_next ? 8 : 0)) : 0;
    const dim3 grid(((int64_t)gxb * gyb * s->B * s->nb + 7) / 8 * 8);
    MPC_LAUNCH(k_knn_bwd_points<16>, grid, dim3(256), lds, st, p, cell_start, spos, sidx, grad_flow_lut,
                       grad_flow_next, knn_state, reach, tmp_g, tmp_a, gxb, gyb);
    MPC_CHECK_LAUNCH();
    const int64_t total = (int64_t)s->B * s->n;
    MPC_LAUNCH(k_knn_bwd_combine, dim3(mpc_cdiv(total, 256)), dim3(256), 0, st, p, tmp_g,
                       (grad_flow_next || s->T != 1 || p.iwd) ? tmp_a : nullptr, grad_traj);
    MPC_CHECK_LAUNCH();
    return 0;
}

MPC_BOUNDS_UNIT("knn.hip")
