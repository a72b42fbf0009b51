// Device-side building blocks of the exact K-nearest-neighbour search shared by the tile kernel (knn.hip), the strip
// kernel and its per-query fallback (knn_strip.hip): parameters, cell assignment, the pair distance exactly as the
// reference computes it (focus.py:132-135), and the generic two-scan search of one query (knn_one_query).
#pragma once
#include "common.h"
#include <math.h>
#define KNN_HW (KNN_BINS / 4)   // histogram words per query: four 8-bit bins per word
#define KNN_LIST KNN_HW          // {candidate slot, trajectory index} pairs kept in a (dead) histogram column
// bit 30 of the saved K-th index: some point lies at exactly the K-th distance but was excluded (tie resolved by
// index); without it the backward's membership test is simply `d <= K-th distance`
#define KNN_TIE_FLAG 0x40000000
// bit 29: the query was served by the fallback kernel and is on the far list of its (sample, bin): the backward's gather
// skips it (its K-th distance is not in the tile maxima either) and k_knn_bwd_far adds its gradient
#define KNN_FAR_FLAG 0x20000000
#define KNN_IDX_MASK 0x1fffffff
// Largest K-th distance per 16x16 cell tile, kept per CLASS of query so that the backward's search reach stays tight:
// queries next to the image border have clipped neighbourhoods and therefore K-th distances up to 2-4x larger than
// inner ones, but they sit in a thin band; a per-tile maximum over all of them would inflate the reach of every tile
// adjacent to a border tile.  Class 0: every query of the tile that lies at least `bd` cells from all borders;
// 1 / 2 / 3 / 4: queries within `bd` cells of the top / bottom / left / right border (a corner query counts for both).
// Layout: tile_dkmax[((b*nb + bin) * ntiles + tile) * KNN_NCLS + class].
#define KNN_NCLS 5
#define KNN_SLACK 0.01f   // px, absorbs fp32 rounding of the cell assignment in the ring bound

// The points are bucketed into a grid of cells that extends `m` cells beyond the query grid on every side (cell (y, x),
// y in [-m, hq + m), x in [-m, wq + m): the query cells are y in [0, hq), x in [0, wq)).  Trajectory points that a flow has
// carried out of the image keep their own cells there; only what lies beyond the margin is clamped into the outermost ring,
// which no search of the fast path reaches (m = KNN_RCAP + 1).  Without the margin every point outside the image piled up in
// the border cells of the query grid (an expanding flow field: a hundred points per corner cell), and every border query
// had to look at all of them.
// The bucket tables are 16-bit: every value is a count of points of one (sample, bin) or a trajectory index, at most n <= 65535
// (mpc_knn_lut_fwd refuses more).  Round 5: int32 tables were 56 of the 89 MB the bucket kernel wrote per C3 step, and that kernel's
// table phases run at the HBM write rate.
typedef unsigned short knn_cs_t;      // cell_start [B*nb][Gb + 1], summed-area table [B*nb][hb + 1][wb + 1]
typedef unsigned short knn_idx_t;     // trajectory index of a bucketed point, sidx [B*nb][n]
struct KnnParams {
    int B, nb, T, n, hq, wq, sp, K, G;      // G = hq * wq query cells
    int m, hb, wb, Gb;                       // margin; bucket grid hb x wb = (hq + 2m) x (wq + 2m), Gb cells
    int l1, iwd, want_next;
    float off;   // sp/2 - 0.5 : centre of cell 0 (focus.py:117)
};
#define KNN_RFAR 20                // ... of its launch for the FAR queries (no square up to KNN_RCAP cells holds enough points: the
                                   // inside of a band the flow field emptied); beyond it a query goes to the fallback kernel
#define KNN_MARGIN (KNN_RCAP + 1)
// smallest count of points in the (2r + 1)^2 cell square of a query for which radius r is tried (K / (pi / 4) at K = 32: the
// disc of the ring bound holds K points if they are spread evenly over the square; calibrated on smooth flow fields)
// (L1: the ball of the ring bound is a diamond, half of the square)
__host__ __device__ static inline int knn_square_need(int K, int l1 = 0) { return (int)((float)K * (l1 ? 2.05f : 1.28f) + 0.5f); }

// count of points in the square for which a FAR query tries radius r: the K neighbours of a query inside an emptied band lie
// in a segment of the disc, the square also holds what the disc cuts off (calibrated as above: 2.6 K misses 4 % of them)
__host__ __device__ static inline int knn_square_need_far(int K, int l1 = 0) { return (int)((float)K * (l1 ? 4.2f : 2.65f) + 0.5f); }

int mpc_knn_margin(const mpc_shape *s);

// Work lists of the KNN forward, all in the workspace, counters zeroed by the bucket kernels:
//   fail     int [1 + B*nb*G]      queries for the fallback workgroups of k_knn_tail (bits 0..29 the query, bits 30..31 why)
//   retry    int [1 + strips]      strips whose points overflowed the staging area: k_knn_strip_retry searches them in quarters
//   farstrip int [1 + strips]      strips that hold far queries: k_knn_strip_far
//   far      int [B*nb][1 + G]     per (sample, bin): the cells (cy * wq + cx) of the far queries that were served (K-th key
//                                  saved, flagged KNN_FAR_FLAG), for k_knn_bwd_far; null where the backward is not the tile gather
//   ftbits   u32 [B*nb][ftwords]   tiles whose points a far query's disc can touch; ftlist int [1 + B*nb*tiles]: the same as a
//                                  list of (sample, bin) * tiles + tile: the work items of k_knn_bwd_far
//   again    u32 [B*nb][hq][ceil(wq/32)]  queries the main launch of the strip kernel could not finish (fewer than K candidates
//                                  below the ring bound after all, more slots than its registers hold): k_knn_tail
//                                  searches them with one more ring, 128 slots and chord-shaped rows
//   grow     u32, same shape: of those, the ones that need more RINGS (too few candidates below the bound; far queries)
//   chord    u8 [KNN_RFAR + 1][KNN_RFAR + 1]  chord[r][j] = knn_chord_cells(r, j): written by the bucket kernels, read by the strip kernels
struct KnnLists {
    int *fail, *retry, *farstrip, *far, *ftlist;
    unsigned *ftbits, *again, *grow;
    unsigned char *chord;
    int ftwords, again_words;       // words per (sample, bin)
};
// number of queries the main launch marked for the second one (behind the chord table, in the same 1 KB of the workspace); next to it
// the counters of the tail kernel (k_knn_tail): queries its strip workgroups handed to its fallback workgroups -- the LATE list,
// which grows downwards from the end of the `fail` array (main list and late list together never exceed the number of queries) --
// and the strip workgroups that have finished.  All zeroed by the bucket kernels.
#ifdef __HIPCC__
// (one 128-byte line each: knn_late_count and knn_tail_done take device-scope atomics from every XCD all through the tail launch, and a
// plain load of knn_marked_count in the same line waited behind them -- the far pass of a UNet-like field 187 us instead of 146)
__device__ __forceinline__ int *knn_marked_count(const KnnLists &ls) { return reinterpret_cast<int *>(ls.chord + 512); }
__device__ __forceinline__ int *knn_late_count(const KnnLists &ls) { return reinterpret_cast<int *>(ls.chord + 640); }
__device__ __forceinline__ int *knn_tail_done(const KnnLists &ls) { return reinterpret_cast<int *>(ls.chord + 768); }
__device__ __forceinline__ int *knn_bwd_far_next(const KnnLists &ls) { return reinterpret_cast<int *>(ls.chord + 896); }      // k_knn_bwd_far: items beyond a workgroup's first (zeroed by k_knn_bwd_tile)
// the MARKED list: every query the main launch marks for the tail's strip workgroups is also listed (same place as the late list, the
// end of the `fail` array downwards; its length is knn_marked_count), with the radius its search would start from.  With few marked
// queries in the whole launch (KS_FORWARD_MAX) the tail's fallback workgroups take the marked list straight away, the strip
// workgroups skip their far pass and the late list (what the retry quarters cannot finish) starts behind the marked entries;
// otherwise the far pass runs and the late list overwrites the (then dead) marked entries.
#endif
// Blocks of queries of the second launch of the strip kernel: the main launch's strips, 2 columns x 128 rows.  The far queries of
// a band along the top or bottom border are a few rows of EVERY strip, so wider, shorter blocks were tried: 4 x 64 halves the
// workgroups of such a band and 8 x 32 quarters them, but every region row is 2 / 6 cells wider and two / three times as many
// queries need more than the 128 slots and end up in the fallback kernel (C3, 40 px translation: 1.08 ms with 2 x 128, 1.15 with
// 4 x 64, 1.21 with 8 x 32).
#define KNN_FAR_WS 2
#define KNN_FAR_TH 128
__host__ __device__ static inline int knn_far_items_x(int wq) { return (wq + KNN_FAR_WS - 1) / KNN_FAR_WS; }
__host__ __device__ static inline int knn_far_items_y(int hq) { return (hq + KNN_FAR_TH - 1) / KNN_FAR_TH; }

static KnnParams knn_params(const mpc_shape *s) {
    KnnParams p;
    p.B = s->B; p.nb = s->nb; p.T = s->T; p.n = s->n; p.hq = s->hq; p.wq = s->wq; p.sp = s->sp;
    p.K = s->K; p.G = s->hq * s->wq;
    p.m = mpc_knn_margin(s); p.hb = s->hq + 2 * p.m; p.wb = s->wq + 2 * p.m; p.Gb = p.hb * p.wb;
    p.l1 = (s->flags & MPC_F_DIST_L1) ? 1 : 0;
    p.iwd = ((s->flags & MPC_F_SCHEME_IWD) && s->K > 1) ? 1 : 0;   // focus.py:145-147: K == 1 is a plain gather
    p.want_next = (s->flags & MPC_F_WANT_NEXT) ? 1 : 0;
    p.off = (float)s->sp / 2.f - 0.5f;
    return p;
}

__device__ __forceinline__ int cell_of(float v, int sp, int ncell, int m) {
    // cells are centred on the query points: cell c covers [c*sp - 0.5, (c+1)*sp - 0.5); c in [-m, ncell + m), clamped
    const float c = floorf(mpc_div_sp(v + 0.5f, sp));
    return (int)fminf(fmaxf(c, (float)-m), (float)(ncell + m - 1));
}
// index of cell (y, x) -- query-grid coordinates, -m <= y < hq + m, -m <= x <= wq + m -- in cell_start
__device__ __forceinline__ int knn_ci(const KnnParams &p, int y, int x) { return (y + p.m) * p.wb + (x + p.m); }
// bucket cell of a point
__device__ __forceinline__ int knn_cell_index(const KnnParams &p, float py, float px) {
    return knn_ci(p, cell_of(py, p.sp, p.hq, p.m), cell_of(px, p.sp, p.wq, p.m));
}
// 16 x 16 cell tiles (the backward's workgroups; the tile maxima of the K-th distance): the tiles of the QUERY grid; the
// tiles along the image border also own the margin cells beside them (up to KNN_TROWS cell rows / columns of points, still
// 16 x 16 queries), so that the tile count -- and the work of the lattice-like point sets -- is that of the query grid.
#define KNN_TROWS (16 + 2 * KNN_MARGIN)
__host__ __device__ static inline int knn_tiles_x(int wq, int) { return (wq + 15) >> 4; }
__host__ __device__ static inline int knn_tiles_y(int hq, int) { return (hq + 15) >> 4; }
// tile of cell c (query numbering, may lie in the margin) along an axis of n query cells
__host__ __device__ static inline int knn_tile_of(int c, int n) { return (c < 0 ? 0 : (c > n - 1 ? n - 1 : c)) >> 4; }
// cells [c0, c1) of tile t along an axis of n query cells and margin m
__host__ __device__ static inline void knn_tile_cells(int t, int n, int m, int &c0, int &c1) {
    c0 = t == 0 ? -m : 16 * t;
    c1 = 16 * t + 16 >= n ? n + m : 16 * t + 16;
}
// summed-area table of the cell counts: sat[(y + m) * (wb + 1) + (x + m)] = points in cells (y', x') with y' < y and x' < x
__device__ __forceinline__ int knn_square_count(const KnnParams &p, const knn_cs_t *__restrict__ sat, int cy, int cx, int r) {
    const int y0 = max(cy - r, -p.m) + p.m, y1 = min(cy + r, p.hq + p.m - 1) + p.m + 1;
    const int x0 = max(cx - r, -p.m) + p.m, x1 = min(cx + r, p.wq + p.m - 1) + p.m + 1;
    const int W1 = p.wb + 1;
    return sat[y1 * W1 + x1] - sat[y0 * W1 + x1] - sat[y1 * W1 + x0] + sat[y0 * W1 + x0];
}

__device__ __forceinline__ float pair_dist(float qy, float qx, float py, float px, int l1) {
    // focus.py:132-135: (grid - traj) ** 2 summed over (y, x), or abs
    const float dy = qy - py, dx = qx - px;
    return l1 ? (fabsf(dy) + fabsf(dx)) : (dy * dy + dx * dx);
}

// ------------------------------------------------------------------------------------------
// query: one thread per LUT cell; 16 x (NT/16) cells per workgroup (NT = 256 or 512).  The candidate
// points of the tile and a halo of RH cell rings are staged in LDS (positions, 16-bit offsets and indices,
// flows), so the selection passes never leave the CU; a thread that needs a square outside the staged
// region (or a workgroup whose region overflows the staging capacity) finishes on the global arrays
// with the same code.  Per thread: a private column of 8 LDS words = 32 histogram bins of 8 bits, later
// reused for the short list of keys inside the K-th bin.
// 1-D grid of gx*gy*B*nb workgroups in XCD-contiguous order, NT threads, dynamic LDS sized by the launcher
// so that as many workgroups as possible fit a CU (the kernel is latency bound below ~24 wavefronts per CU)
// ------------------------------------------------------------------------------------------
struct QueryCtx {
    // global
    const knn_cs_t *cs;          // cell_start of this (sample, bin)
    const float2 *spos;
    const knn_idx_t *sidx;
    const float2 *traj_b;   // trajectories of this sample: [T+nb][n]
    // LDS
    const unsigned short *lcs;   // [RW][RW+1]  (staged offsets < cap <= 65535)
    const float2 *lpos;
    const unsigned short *lidx;  // trajectory index (n < 65536)
    const float2 *lf0;      // flow to t_ref (T == 1 and staged), else null
    const float2 *lf1;      // flow to the next bin (want_next only)
    int ry0, rx0, RW, RWY, RH;   // staged region: cell rows [ry0, ry0+RWY), columns [rx0, rx0+RW); RH = halo
};

template <bool LDS>
struct Acc {
    const KnnParams &p;
    const QueryCtx &c;
    int t;
    __device__ __forceinline__ void range(int yy, int x0, int x1, int &js, int &je) const {
        if (LDS) {
            const unsigned short *row = c.lcs + (yy - c.ry0) * (c.RW + 1);
            js = row[x0 - c.rx0];
            je = row[x1 + 1 - c.rx0];
        } else {
            js = c.cs[knn_ci(p, yy, x0)];
            je = c.cs[knn_ci(p, yy, x1 + 1)];
        }
    }
    __device__ __forceinline__ float2 pos(int j) const { return LDS ? c.lpos[j] : c.spos[j]; }
    __device__ __forceinline__ int idx(int j) const { return LDS ? c.lidx[j] : c.sidx[j]; }
    __device__ __forceinline__ float2 flow_ref(int j, int tr, float2 pj) const {
        if (LDS && c.lf0 != nullptr) return c.lf0[j];
        const float2 a = c.traj_b[(size_t)tr * p.n + idx(j)];
        return make_float2(a.x - pj.x, a.y - pj.y);          // traj(t_ref) - traj(t_mid)
    }
    __device__ __forceinline__ float2 flow_next(int j, float2 pj) const {
        if (LDS) return c.lf1[j];
        const float2 a = c.traj_b[(size_t)(p.T + t + 1) * p.n + idx(j)];
        return make_float2(a.x - pj.x, a.y - pj.y);          // traj(t_mid[i+1]) - traj(t_mid[i])
    }
};

// Returns false if the search needs more rings than the LDS halo holds (LDS variant only).
// r_init: first radius of the search square (cells).  DISC (global-array variant, the bulk fallback of the strip kernel): the
// cells of a row that lie wholly beyond the ring bound are skipped -- a query deep inside an empty band (a flow field that
// carried the points away from an image border) has its K neighbours in a thin segment of a large disc, and the corners of
// the square hold as many points again; `flag_or` is OR-ed into the saved K-th index (KNN_FAR_FLAG).
template <bool LDS, bool L1, int NT, bool DISC = false>
__device__ bool knn_one_query(const KnnParams &p, const QueryCtx &c, int b, int t, int cy, int cx,
                              int r_init, unsigned (*s_hist)[NT], float *__restrict__ flow_lut,
                              float *__restrict__ flow_next, float *__restrict__ knn_state,
                              int *__restrict__ idx_out, float &dK_out, int flag_or = 0) {
    const Acc<LDS> A{p, c, t};
    const int tid = threadIdx.x;
    const int bt = b * p.nb + t;
    const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
    const int ylo = -p.m, yhi = p.hq + p.m - 1, xlo = -p.m, xhi = p.wq + p.m - 1;      // the bucket grid
    // ---- 1. grow the search square until K candidates are provably the nearest ---------------
    int r = r_init, y0, y1, x0, x1, cnt;
    if (!DISC) {   // next to the border of the bucket grid the square is clipped: start with one of the same cell count
        const int want = (2 * r_init + 1) * (2 * r_init + 1);
        for (;;) {
            const int hh = min(cy + r, yhi) - max(cy - r, ylo) + 1;
            const int ww = min(cx + r, xhi) - max(cx - r, xlo) + 1;
            if (hh * ww >= want || (hh == p.hb && ww == p.wb) || (LDS && r >= c.RH)) break;
            ++r;
        }
    }
    float upper = INFINITY, scale;
    bool whole;
    // bucketed range of row yy of the search square; DISC: only the cells that can hold a point below the ring bound
    auto row_range = [&](int yy, int &js, int &je) {
        int xa = x0, xb = x1;
        if (DISC && upper < INFINITY) {
            // a point of cell row yy is at least dyc away along y; along x it must then lie within wx of the query
            const float dyc = fmaxf((float)abs(yy - cy) - 0.5f, 0.f) * (float)p.sp;
            const float w2 = L1 ? upper - dyc : upper - dyc * dyc;
            if (!(w2 > 0.f)) { js = je = 0; return; }
            const int xr = (int)((L1 ? w2 : sqrtf(w2)) / (float)p.sp + 0.5f) + 1;      // (+1: rounding of the square root)
            xa = max(xa, cx - xr); xb = min(xb, cx + xr);
        }
        A.range(yy, xa, xb, js, je);
    };
    for (;;) {
        y0 = max(cy - r, ylo); y1 = min(cy + r, yhi);
        x0 = max(cx - r, xlo); x1 = min(cx + r, xhi);
        // the LDS variant serves any square that lies inside the staged region: next to the image border a
        // grown square is wider than the halo but, clipped, still inside the tile (otherwise such queries finish
        // on the global arrays, one dependent L2 round trip per step: the tail of small launches)
        if (LDS && (y0 < c.ry0 || y1 >= c.ry0 + c.RWY || x0 < c.rx0 || x1 >= c.rx0 + c.RW)) return false;
        whole = (y0 == ylo && x0 == xlo && y1 == yhi && x1 == xhi);
        if (whole) {
            if (LDS) return false;
            // every point is a candidate: range of the histogram = largest distance
            float dmax = 0.f;
            for (int j = 0; j < p.n; ++j) {
                const float2 q = c.spos[j];
                dmax = fmaxf(dmax, pair_dist(qy, qx, q.x, q.y, L1));
            }
            upper = INFINITY;
            scale = dmax > 0.f ? (float)KNN_BINS / dmax : 0.f;
        } else {
            // anything outside the square is at least lb away along one axis
            const float lb = ((float)r + 0.5f) * (float)p.sp - KNN_SLACK;
            upper = L1 ? lb : lb * lb;
            scale = (float)KNN_BINS / upper;
        }
#pragma unroll
        for (int h = 0; h < KNN_HW; ++h) s_hist[h][tid] = 0u;
        cnt = 0;
        // software pipeline: the candidate positions of the next step and the cell range of the next row are
        // requested before the current ones are consumed (the wavefront otherwise parks on every LDS round trip)
        int njs, nje;
        row_range(y0, njs, nje);
        for (int yy = y0; yy <= y1; ++yy) {
            const int js = njs, je = nje;
            row_range(min(yy + 1, y1), njs, nje);
            if (js >= je) continue;
            float2 q[KNN_BATCH];
#pragma unroll
            for (int u = 0; u < KNN_BATCH; ++u) q[u] = A.pos(js + u);      // reads past the row are masked below
            for (int j = js; j < je; j += KNN_BATCH) {
                float2 nq[KNN_BATCH];
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) nq[u] = A.pos(j + KNN_BATCH + u);
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) {
                    // predicated, not branched: an out-of-range candidate adds 0 (the kernel is bound by
                    // instruction issue, and exec-mask branches cost more than the spare LDS atomic)
                    const float d = pair_dist(qy, qx, q[u].x, q[u].y, L1);
                    const int in = (j + u < je) & (d < upper);
                    const int bin = min((int)(d * scale), KNN_BINS - 1);     // d < upper <= FLT_MAX wherever `in` holds
                    // word (bin >> 2) of the private column: byte offset (bin & 0x1c) * NT, one AND + one shift-add
                    unsigned *hw = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(&s_hist[0][tid]) + (bin & 0x1c) * NT);
                    atomicAdd(hw, in ? (1u << ((bin << 3) & 31)) : 0u);
                    cnt += in;
                }
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) q[u] = nq[u];
            }
        }
        if (cnt >= p.K || whole) break;
        r += 1 + (r >> 2);
    }
    // ---- 2. bin holding the K-th smallest ----------------------------------------------------
    int bstar = KNN_BINS - 1, before = 0;
    if (cnt > 255) {
        // an 8-bit bin may have wrapped (dense clusters, K > 255): treat every candidate as one bin,
        // which sends the selection to the repeated-minimum path below
        scale = 0.f; bstar = 0;
    } else {
        // word holding the K-th smallest (v_sad_u8 sums the four 8-bit bins of a word), then the bin inside it
        int cum = 0;
        unsigned wstar = 0u;
        bool found = false;
#pragma unroll
        for (int h = 0; h < KNN_HW; ++h) {
            const unsigned wv = s_hist[h][tid];
            const int nc = (int)__builtin_amdgcn_sad_u8(wv, 0u, (unsigned)cum);
            if (!found && nc >= p.K) { bstar = 4 * h; before = cum; wstar = wv; found = true; }
            cum = nc;
        }
        if (found) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ck = (int)((wstar >> (8 * k)) & 0xffu);
                if (before + ck < p.K) { before += ck; ++bstar; } else break;
            }
        }
    }
    // ---- 3. second scan: sum the flows of the bins below bstar (num_tref == 1), and list the
    //         keys inside bstar in the thread's (now dead) histogram column -----------------------
    const int need = p.K - before;
    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const bool fuse = (p.T == 1);
    const bool do_next0 = p.want_next && (t < p.nb - 1);
    float sy = 0.f, sx = 0.f, sw = 0.f, ny = 0.f, nx = 0.f;
    const float fb = (float)bstar, fb1 = (bstar == KNN_BINS - 1) ? INFINITY : (float)(bstar + 1);
    int m = 0;
    int njs, nje;
    row_range(y0, njs, nje);
    for (int yy = y0; yy <= y1; ++yy) {
        const int js = njs, je = nje;
        row_range(min(yy + 1, y1), njs, nje);
        if (js >= je) continue;
        float2 qq[KNN_BATCH];
#pragma unroll
        for (int u = 0; u < KNN_BATCH; ++u) qq[u] = A.pos(js + u);
        for (int j0 = js; j0 < je; j0 += KNN_BATCH) {
            float2 cur[KNN_BATCH];
#pragma unroll
            for (int u = 0; u < KNN_BATCH; ++u) { cur[u] = qq[u]; qq[u] = A.pos(j0 + KNN_BATCH + u); }
#pragma unroll
            for (int u = 0; u < KNN_BATCH; ++u) {
                const int j = j0 + u;
                const float2 pj = cur[u];
                const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                if (j >= je || !(d < upper)) continue;      // (a predicated form of this scan measured slower: 751 vs 673 us)
                const float ds = d * scale;                 // bin = min(floor(ds), KNN_BINS - 1)
                if (ds < fb) {
                    if (fuse) {
                        const float2 f = A.flow_ref(j, 0, pj);
                        if (p.iwd) { const float w = 1.f / (d + 1e-9f); sy += w * f.x; sx += w * f.y; sw += w; }
                        else { sy += f.x; sx += f.y; }
                        if (do_next0) { const float2 g = A.flow_next(j, pj); ny += g.x; nx += g.y; }
                    }
                } else if (ds < fb1) {
                    if (m < KNN_LIST) s_hist[m][tid] = ((unsigned)j << 16) | (unsigned)A.idx(j);
                    ++m;
                }
            }
        }
    }
    float dK = 0.f; int iK = -1;
    bool listed = (m <= KNN_LIST);
    if (m <= 4) {
        // the usual case: rank up to four keys by (distance, index) in registers, straight-line
        float dd[4]; int ii[4], jj[4]; float2 pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned w = s_hist[u][tid];
            const bool valid = u < m;
            jj[u] = valid ? (int)(w >> 16) : 0;
            ii[u] = valid ? (int)(w & 0xffffu) : 0x7fffffff;
            pp[u] = A.pos(jj[u]);
            dd[u] = valid ? pair_dist(qy, qx, pp[u].x, pp[u].y, L1) : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int rank = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e != u) rank += ((dd[e] < dd[u]) | ((dd[e] == dd[u]) & (ii[e] < ii[u]))) ? 1 : 0;
            if (u < m && rank < need) {
                if (fuse) {
                    const float2 f = A.flow_ref(jj[u], 0, pp[u]);
                    if (p.iwd) { const float w = 1.f / (dd[u] + 1e-9f); sy += w * f.x; sx += w * f.y; sw += w; }
                    else { sy += f.x; sx += f.y; }
                    if (do_next0) { const float2 g = A.flow_next(jj[u], pp[u]); ny += g.x; nx += g.y; }
                }
                if (rank == need - 1) { dK = dd[u]; iK = ii[u]; }
            }
        }
    } else if (listed) {
        // rank the listed keys by (distance, index); the first `need` of them are neighbours
        for (int a = 0; a < m; ++a) {
            const unsigned wa = s_hist[a][tid];
            const int ja = (int)(wa >> 16), ia = (int)(wa & 0xffffu);
            const float2 pj = A.pos(ja);
            const float da = pair_dist(qy, qx, pj.x, pj.y, L1);
            int rank = 0;
            for (int e = 0; e < m; ++e) {
                const unsigned we = s_hist[e][tid];
                const float2 pe = A.pos((int)(we >> 16));
                const float de = pair_dist(qy, qx, pe.x, pe.y, L1);
                rank += (de < da || (de == da && (int)(we & 0xffffu) < ia)) ? 1 : 0;
            }
            if (rank < need) {
                if (fuse) {
                    const float2 f = A.flow_ref(ja, 0, pj);
                    if (p.iwd) { const float w = 1.f / (da + 1e-9f); sy += w * f.x; sx += w * f.y; sw += w; }
                    else { sy += f.x; sx += f.y; }
                    if (do_next0) { const float2 g = A.flow_next(ja, pj); ny += g.x; nx += g.y; }
                }
                if (rank == need - 1) { dK = da; iK = ia; }
            }
        }
    } else if (m <= 2 * KNN_LIST) {
        // The bin holds more keys than the packed list (a tight cluster; one workgroup in a thousand at C3, but
        // it used to cost `need` + 1 further scans and was the tail of small launches): collect the slots once
        // more as 16-bit entries in the same column, then rank them by (distance, index) as above.
        int mm = 0;
        for (int yy = y0; yy <= y1; ++yy) {
            int js, je;
            row_range(yy, js, je);
            for (int j = js; j < je; ++j) {
                const float2 pj = A.pos(j);
                const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                if (!(d < upper)) continue;
                const float ds = d * scale;
                if (ds < fb || !(ds < fb1)) continue;
                reinterpret_cast<unsigned short *>(&s_hist[mm >> 1][tid])[mm & 1] = (unsigned short)j;
                ++mm;
            }
        }
        for (int a = 0; a < mm; ++a) {
            const int ja = (int)reinterpret_cast<const unsigned short *>(&s_hist[a >> 1][tid])[a & 1];
            const int ia = A.idx(ja);
            const float2 pj = A.pos(ja);
            const float da = pair_dist(qy, qx, pj.x, pj.y, L1);
            int rank = 0;
            for (int e = 0; e < mm; ++e) {
                const int je = (int)reinterpret_cast<const unsigned short *>(&s_hist[e >> 1][tid])[e & 1];
                const float2 pe = A.pos(je);
                const float de = pair_dist(qy, qx, pe.x, pe.y, L1);
                rank += (de < da || (de == da && A.idx(je) < ia)) ? 1 : 0;
            }
            if (rank < need) {
                if (fuse) {
                    const float2 f = A.flow_ref(ja, 0, pj);
                    if (p.iwd) { const float w = 1.f / (da + 1e-9f); sy += w * f.x; sx += w * f.y; sw += w; }
                    else { sy += f.x; sx += f.y; }
                    if (do_next0) { const float2 g = A.flow_next(ja, pj); ny += g.x; nx += g.y; }
                }
                if (rank == need - 1) { dK = da; iK = ia; }
            }
        }
        listed = true;
    } else {
        // far more keys in the bin than any list holds (heavy ties): select by repeated minimum
        float ld = -1.f; int li = -1;
        for (int it = 0; it < need; ++it) {
            float bd = INFINITY; int bi = 0x7fffffff;
            for (int yy = y0; yy <= y1; ++yy) {
                int js, je;
                row_range(yy, js, je);
                for (int j = js; j < je; ++j) {
                    const float2 pj = A.pos(j);
                    const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                    if (!(d < upper)) continue;
                    if (min((int)(d * scale), KNN_BINS - 1) != bstar) continue;
                    if (d < ld || d > bd) continue;
                    const int id = A.idx(j);
                    const bool gt_last = (d > ld) || (id > li);
                    const bool lt_best = (d < bd) || (id < bi);
                    if (gt_last && lt_best) { bd = d; bi = id; }
                }
            }
            ld = bd; li = bi;
        }
        dK = ld; iK = li;
    }
    // ---- 4. outputs; a full membership scan per reference time where the sums were not fused ----
    float norm = 0.f;
    if (fuse && listed) {
        float2 ov;
        if (p.iwd) { ov.x = sy / sw; ov.y = sx / sw; norm = sw; }
        else { ov.x = sy / (float)p.K; ov.y = sx / (float)p.K; }
        reinterpret_cast<float2 *>(flow_lut)[q] = ov;
        if (do_next0) {
            float2 on; on.x = ny / (float)p.K; on.y = nx / (float)p.K;
            reinterpret_cast<float2 *>(flow_next)[((size_t)(b * (p.nb - 1) + t)) * p.G + (size_t)cy * p.wq + cx] = on;
        }
    } else {
        for (int tr = 0; tr < p.T; ++tr) {
            sy = sx = sw = ny = nx = 0.f;
            const bool do_next = (tr == 0) && do_next0;
            for (int yy = y0; yy <= y1; ++yy) {
                int js, je;
                row_range(yy, js, je);
                for (int j = js; j < je; ++j) {
                    const float2 pj = A.pos(j);
                    const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                    if (d > dK) continue;
                    if (d == dK && A.idx(j) > iK) continue;
                    const float2 f = A.flow_ref(j, tr, pj);
                    if (p.iwd) {
                        const float w = 1.f / (d + 1e-9f);
                        sy += w * f.x; sx += w * f.y; sw += w;
                    } else {
                        sy += f.x; sx += f.y;
                    }
                    if (do_next) {
                        const float2 g = A.flow_next(j, pj);
                        ny += g.x; nx += g.y;
                    }
                }
            }
            float2 ov;
            if (p.iwd) { ov.x = sy / sw; ov.y = sx / sw; norm = sw; }
            else { ov.x = sy / (float)p.K; ov.y = sx / (float)p.K; }
            reinterpret_cast<float2 *>(flow_lut)[q * p.T + tr] = ov;
            if (do_next) {
                float2 on; on.x = ny / (float)p.K; on.y = nx / (float)p.K;
                reinterpret_cast<float2 *>(flow_next)[((size_t)(b * (p.nb - 1) + t)) * p.G + (size_t)cy * p.wq + cx] = on;
            }
        }
    }
    knn_state[q] = dK;
    reinterpret_cast<int *>(knn_state)[BQ + q] = iK | KNN_TIE_FLAG | flag_or;      // (this routine does not look for excluded ties: flagged conservatively)
    knn_state[2 * BQ + q] = norm;
    // ---- 5. optional: the K indices in ascending (distance, index) order ---------------------
    if (idx_out != nullptr) {
        float pd = -1.f; int pi = -1;
        for (int k = 0; k < p.K; ++k) {
            float bd = INFINITY; int bi = 0x7fffffff;
            for (int yy = y0; yy <= y1; ++yy) {
                int js, je;
                row_range(yy, js, je);
                for (int j = js; j < je; ++j) {
                    const float2 pj = A.pos(j);
                    const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                    const int id = A.idx(j);
                    const bool gt_last = (d > pd) || (d == pd && id > pi);
                    const bool lt_best = (d < bd) || (d == bd && id < bi);
                    if (gt_last && lt_best) { bd = d; bi = id; }
                }
            }
            pd = bd; pi = bi;
            idx_out[q * p.K + k] = bi;
        }
    }
    dK_out = dK;
    return true;
}

// band depth (cells) of the border classes of the tile maxima
__host__ __device__ static inline int knn_band_depth(int r_init) { return r_init + 1; }
// search radius of a query at the MEAN point density (cells): the first radius tried, and the depth of the border classes
int mpc_knn_r_init(const mpc_shape *s);

#ifdef __HIPCC__
// classes of the query (cy, cx): bit c set <=> it counts for class c
__device__ __forceinline__ unsigned knn_query_classes(const KnnParams &p, int cy, int cx, int bd) {
    const unsigned m = (cy < bd ? 2u : 0u) | (cy >= p.hq - bd ? 4u : 0u) | (cx < bd ? 8u : 0u) | (cx >= p.wq - bd ? 16u : 0u);
    return m ? m : 1u;
}
// one query's K-th distance into the tile maxima (atomicMax on the bits of a non-negative float); tiles of the bucket grid
__device__ __forceinline__ void knn_tile_max_add(float *__restrict__ tile_dkmax, const KnnParams &p, int bt, int cy, int cx,
                                                 int bd, float dK) {
    if (tile_dkmax == nullptr) return;
    const int gx16 = knn_tiles_x(p.wq, p.m), gy16 = knn_tiles_y(p.hq, p.m);
    int *dst = reinterpret_cast<int *>(tile_dkmax) + (((size_t)bt * gy16 + (cy >> 4)) * gx16 + (cx >> 4)) * KNN_NCLS;
    const unsigned m = knn_query_classes(p, cy, cx, bd);
#pragma unroll
    for (int c = 0; c < KNN_NCLS; ++c) if ((m >> c) & 1u) atomicMax(dst + c, __float_as_int(dK));
}
// A far query (cy, cx) with K-th distance dK has been served: its cell onto the far list of (sample, bin) bt is the caller's
// business (one atomic per wavefront); here the tiles whose points its disc can touch go onto the work list of k_knn_bwd_far
// (each tile once: the bit map).  Called by the lane that owns the query.
// the tiles (ta..tb) x (tc..td) whose points the disc of a far query can touch
__device__ __forceinline__ void knn_far_tile_range(const KnnParams &p, int cy, int cx, float dK, int &ta, int &tb, int &tc, int &td) {
    const float R = (p.l1 ? dK : sqrtf(dK)) * 1.0001f + 0.01f;
    const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
    ta = knn_tile_of(cell_of(qy - R, p.sp, p.hq, p.m), p.hq); tb = knn_tile_of(cell_of(qy + R, p.sp, p.hq, p.m), p.hq);
    tc = knn_tile_of(cell_of(qx - R, p.sp, p.wq, p.m), p.wq); td = knn_tile_of(cell_of(qx + R, p.sp, p.wq, p.m), p.wq);
}
// The same for the far queries of a whole WORKGROUP (all of one (sample, bin)): every lane that owns one ORs its tiles into a
// bit map in LDS (s_ft: KNN_FT_LDS_WORDS words, zeroed by the caller before a barrier); knn_far_flush_tiles, after a barrier,
// lets one thread per word add what is new to the global map and the work list -- one load and at most one atomic per word of
// the workgroup instead of a chain of a load and an atomic per tile of every query.
#define KNN_FT_LDS_WORDS 64
__device__ __forceinline__ void knn_far_mark_tiles_lds(const KnnParams &p, unsigned *s_ft, int cy, int cx, float dK) {
    int ta, tb, tc, td;
    knn_far_tile_range(p, cy, cx, dK, ta, tb, tc, td);
    const int ntx = knn_tiles_x(p.wq, p.m);
    for (int ty = ta; ty <= tb; ++ty)
        for (int tx = tc; tx <= td; ++tx) { const int tile = ty * ntx + tx; atomicOr(&s_ft[tile >> 5], 1u << (tile & 31)); }
}
__device__ __forceinline__ void knn_far_flush_tiles(const KnnParams &p, const KnnLists &ls, int bt, const unsigned *s_ft, int word) {
    const unsigned need = s_ft[word];
    if (need == 0u) return;
    unsigned *w = ls.ftbits + (size_t)bt * ls.ftwords + word;
    unsigned fresh = need & ~*w;
    if (fresh == 0u) return;                                  // (set already: the usual case inside a band)
    fresh &= ~atomicOr(w, fresh);
    if (fresh == 0u) return;
    const int ntx = knn_tiles_x(p.wq, p.m), nty = knn_tiles_y(p.hq, p.m);
    int k = atomicAdd(&ls.ftlist[0], __popc(fresh));
    while (fresh) { const int bit = __ffs(fresh) - 1; fresh &= fresh - 1u; ls.ftlist[1 + k++] = bt * ntx * nty + 32 * word + bit; }
}
__device__ __forceinline__ void knn_far_mark_tiles(const KnnParams &p, const KnnLists &ls, int bt, int cy, int cx, float dK) {
    int ta, tb, tc, td;
    knn_far_tile_range(p, cy, cx, dK, ta, tb, tc, td);
    const int ntx = knn_tiles_x(p.wq, p.m), nty = knn_tiles_y(p.hq, p.m);
    for (int ty = ta; ty <= tb; ++ty)
        for (int tx = tc; tx <= td; ++tx) {
            const int tile = ty * ntx + tx;
            const unsigned bit = 1u << (tile & 31);
            unsigned *w = ls.ftbits + (size_t)bt * ls.ftwords + (tile >> 5);
            if ((*w & bit) != 0u) continue;                       // (set already: the usual case inside a band)
            if ((atomicOr(w, bit) & bit) == 0u) ls.ftlist[1 + atomicAdd(&ls.ftlist[0], 1)] = bt * ntx * nty + tile;
        }
}
// Is a query's K-th distance beyond what the backward's gather should carry in its tile maxima?  One ring more than the radius
// of the mean density: the odd query the fallback kernel finishes with a slightly larger square stays in the gather (its
// tile searches a window two cells wider), a query of an emptied band goes on the far list (k_knn_bwd_far).
__device__ __forceinline__ bool knn_is_far_dk(const KnnParams &p, float dK, int r_init) {
    const float lim = ((float)(r_init + KNN_FAR_RINGS) + 0.5f) * (float)p.sp;
    return dK > (p.l1 ? lim : lim * lim);
}
// Search radius of a query from the summed-area table of the cell counts: the smallest r in [rmin, KNN_RCAP] whose square
// of (2r + 1)^2 cells (clipped to the bucket grid) holds at least `need` points; KNN_RCAP + 1 if none does.
__device__ __forceinline__ int knn_sat_radius(const KnnParams &p, const knn_cs_t *__restrict__ sat, int cy, int cx, int rmin, int need, int rcap = KNN_RCAP) {
    int r = rmin;
    while (r <= rcap && knn_square_count(p, sat, cy, cx, r) < need) ++r;
    return r;
}
// half width, in cells, of the part of cell row cy + j (0 <= j <= r) that can hold a point below the ring bound of radius r
// around a query of cell (cy, cx): the cells cx - w .. cx + w.  A point of cell column cx + w (w >= 1) is at least (w - 0.5) sp
// away along x and one of row cy + j at least (j - 0.5) sp along y: the column counts only while that corner distance is below
// the bound -- the comparison the search itself makes (`d < upper`), on a distance no point of the cell can undercut.
__device__ __forceinline__ int knn_chord_cells(int r, int j, int sp, bool l1) {
    const float lb = ((float)r + 0.5f) * (float)sp - KNN_SLACK;
    const float upper = l1 ? lb : lb * lb;
    const float dyc = fmaxf((float)j - 0.5f, 0.f) * (float)sp;
    int w = r;
    while (w >= 1) {
        const float dxc = ((float)w - 0.5f) * (float)sp;
        // (the corner itself is not in the cell -- cells are half open -- and rounding of the points' own distances is far
        // below the 0.01 px of KNN_SLACK; one ulp of margin here keeps the cell on a tie)
        const float dc = l1 ? dxc + dyc : dxc * dxc + dyc * dyc;
        if (dc * 0.999999f < upper) break;
        --w;
    }
    return w;
}
#endif

// ---- strip kernel (knn_strip.hip): the fast path of the forward for num_tref == 1 ---------------------------
bool mpc_knn_strip_usable(const mpc_shape *s, int r_init);
bool mpc_knn_uses_far_list(const mpc_shape *s);
struct EvCountArgs;      // ev_count_device.h: event rows to count per backward bucket in spare workgroups of the strip kernel, or null
int mpc_knn_strip_launch(const mpc_shape *s, const float *traj, const knn_cs_t *cell_start, const knn_cs_t *sat, const float2 *spos, const knn_idx_t *sidx,
                         float *flow_lut, float *flow_next, float *knn_state, float *tile_dkmax, const KnnLists *lists, int r_init,
                         const EvCountArgs *evc, hipStream_t st);
bool mpc_knn_strip_counts_events(const mpc_shape *s, const EvCountArgs *evc);

// ------------------------------------------------------------------------------------------
// Reach of every 16x16 tile of ONE (sample, bin) for the backward gather (k_knn_bwd_tile): the largest linear K-th distance
// among the (tile, class) pairs whose queries can touch the tile's cell area (Chebyshev gap between that area and their
// query centres; classes: knn_query_classes).  Tiles are those of the BUCKET grid (tile (ty, tx) = cells 16 ty - m ...).
// Step A: the linear bounds of all pairs into LDS and the largest of them -- a source tile k tile rings away has its query
// centres at least (k - 1) * 16 cells from a tile's area, so only D = floor(linmax / (16 sp)) + 1 rings matter (one, in
// practice).  Step B: one (tile, neighbour tile) item per thread and round.
// Called by all threads of a workgroup of any size (<= 1024); s_mem: nt * (KNN_NCLS + 1) + 16 floats of LDS; the caller
// synchronises before it reuses s_mem.  Hosts: k_knn_reach_tiles (a launch of its own), k_lut_accum (the event backward's
// kernel, which runs just before the gather in mpc_focus_bwd: one workgroup per slice does it on the side).
// ------------------------------------------------------------------------------------------
struct KnnReachJob {
    KnnParams p;
    const float *tile_dkmax;
    float *reach;
    int gx, gy, bd, on;
};

#ifdef __HIPCC__
// query cells of class c inside tile (sy, sx): false if there are none
__device__ __forceinline__ bool knn_tile_class_cells(const KnnParams &p, int sy, int sx, int c, int bd, int &cy0, int &cy1, int &cx0, int &cx1) {
    cy0 = sy * 16; cy1 = min(sy * 16 + 16, p.hq) - 1;
    cx0 = sx * 16; cx1 = min(sx * 16 + 16, p.wq) - 1;
    if (c == 1) cy1 = min(cy1, bd - 1);
    if (c == 2) cy0 = max(cy0, p.hq - bd);
    if (c == 3) cx1 = min(cx1, bd - 1);
    if (c == 4) cx0 = max(cx0, p.wq - bd);
    return cy0 <= cy1 && cx0 <= cx1;
}
// pixel extent of the cells of tile (ty, tx), margin included (its bucketed points lie inside, or -- clamped into the
// outermost ring -- farther out, which only makes them farther from every query)
__device__ __forceinline__ void knn_tile_area(const KnnParams &p, int ty, int tx, float &ay0, float &ay1, float &ax0, float &ax1) {
    int y0, y1, x0, x1;
    knn_tile_cells(ty, p.hq, p.m, y0, y1); knn_tile_cells(tx, p.wq, p.m, x0, x1);
    ay0 = (float)(y0 * p.sp) - 0.5f; ay1 = (float)(y1 * p.sp) - 0.5f;
    ax0 = (float)(x0 * p.sp) - 0.5f; ax1 = (float)(x1 * p.sp) - 0.5f;
}

template <bool L1>
__device__ __forceinline__ void knn_reach_slice(const KnnParams &p, const float *__restrict__ tile_dkmax, float *__restrict__ reach,
                                                int bt, int gx, int gy, int bd, float *s_mem) {
    const int tid = threadIdx.x, nthr = blockDim.x, ntx = gx, nty = gy, nt = ntx * nty;
    float *s_lin = s_mem;                                            // [nt * NCLS] linear bound of a pair (-1: no query of that class)
    int *s_r = reinterpret_cast<int *>(s_mem + nt * KNN_NCLS);      // [nt] reaches (float bits: non-negative floats order like their bits)
    float *s_m = s_mem + nt * (KNN_NCLS + 1);                        // [16] per-wavefront maxima
    float m = 0.f;
    for (int i = tid; i < nt * KNN_NCLS; i += nthr) {
        const float dk = tile_dkmax[(size_t)bt * nt * KNN_NCLS + i];
        s_lin[i] = dk > 0.f ? (L1 ? dk : sqrtf(dk)) * 1.0001f + 0.01f : -1.f;
        m = fmaxf(m, dk);
    }
    for (int i = tid; i < nt; i += nthr) s_r[i] = 0;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
    if ((tid & 63) == 0) s_m[tid >> 6] = m;
    __syncthreads();
    float dkmax = 0.f;
    for (int w = 0; w < (nthr + 63) / 64; ++w) dkmax = fmaxf(dkmax, s_m[w]);
    const float linmax = (L1 ? dkmax : sqrtf(dkmax)) * 1.0001f + 0.01f;
    const int D = min((int)fminf(linmax / (float)(16 * p.sp), 1.0e6f) + 1, max(ntx, nty));      // (more rings than tiles: all of them)
    const int W = 2 * D + 1;
    for (int it = tid; it < nt * W * W; it += nthr) {
        const int tile = it / (W * W), nbr = it - tile * W * W;
        const int by_ = tile / ntx, bx_ = tile - by_ * ntx;
        const int sy = by_ + nbr / W - D, sx = bx_ + nbr % W - D;
        if (sy < 0 || sy >= nty || sx < 0 || sx >= ntx) continue;
        float ay0, ay1, ax0, ax1;
        knn_tile_area(p, by_, bx_, ay0, ay1, ax0, ax1);
        const int tb = sy * ntx + sx;
        float r = 0.f;
#pragma unroll
        for (int c = 0; c < KNN_NCLS; ++c) {
            int cy0, cy1, cx0, cx1;
            if (!knn_tile_class_cells(p, sy, sx, c, bd, cy0, cy1, cx0, cx1)) continue;
            const float lin = s_lin[tb * KNN_NCLS + c];
            const float qy0 = (float)(cy0 * p.sp) + p.off, qy1 = (float)(cy1 * p.sp) + p.off;
            const float qx0 = (float)(cx0 * p.sp) + p.off, qx1 = (float)(cx1 * p.sp) + p.off;
            const float gyv = fmaxf(0.f, fmaxf(qy0 - ay1, ay0 - qy1)), gxv = fmaxf(0.f, fmaxf(qx0 - ax1, ax0 - qx1));
            if (lin > 0.f && lin >= fmaxf(gyv, gxv)) r = fmaxf(r, lin);
        }
        if (r > 0.f) atomicMax(&s_r[tile], __float_as_int(r));
    }
    __syncthreads();
    for (int i = tid; i < nt; i += nthr) reach[(size_t)bt * nt + i] = __int_as_float(s_r[i]);
}
#endif
// what the event backward needs to host the reach computation (on = 0: not this shape / not this backward)
bool mpc_knn_reach_job(const mpc_shape *s, const float *knn_state, void *ws, KnnReachJob *job);
int mpc_knn_lut_bwd_ex(const mpc_shape *s, const float *traj, const float *grad_flow_lut, const float *grad_flow_next,
                       const float *knn_state, float *grad_traj, void *ws, void *stream, int reach_ready);

#ifdef __HIPCC__
// One 8-byte LDS read as ONE ds_read_b64 (volatile: the compiler may not pair it with its neighbour into ds_read2_b64, which
// the LDS serves at half the rate of two single reads -- MI355X_MICROARCH.md, LDS table).  p must point into LDS.
__device__ __forceinline__ float2 knn_lds_f2(const float2 *p) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned long long b = *(lds_u64 *)(uintptr_t)(unsigned)(size_t)p;        // (low half of a flat LDS address = the LDS offset)
    return make_float2(__uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
#endif
