// Exact K-nearest-neighbour flow look-up table, strip kernel (num_tref == 1): the fast path of mpc_knn_lut_fwd.
//   reference: src/losses/focus.py:115-180 (KeOps argKmin/Kmin over all n x Q pairs, gather, mean / iwd, flow_to_next)
//
// Same search as the tile kernel of knn.hip (square of cells around the query, ring bound, exact (distance, index)
// order), restated so that the per-candidate work is a handful of vector instructions with no divergence:
//   * a workgroup owns a vertical STRIP of WS query columns x TH query rows; the bucketed points of the strip's cell
//     columns +- r are staged row by row, so the candidates of a query -- the rows [cy - r, cy + r] of the strip --
//     are ONE contiguous slot range (points outside the query's square are farther than the ring bound and drop out
//     by the same `d < upper` test that makes the search exact);
//   * EVERY QUERY HAS ITS OWN RADIUS r, read off the summed-area table of the cell counts (k_knn_sat): the smallest square
//     that holds about K / (pi / 4) points.  A trained network's flow field thins the trajectory points out where it
//     diverges, packs them where it converges and empties a band along the borders it moves away from; one radius from the
//     mean density (rounds 1-3) served white-noise coefficients only.  A region row is staged as wide as the widest square
//     that uses it;
//   * pass 1 walks the range in groups of eight slots with a wave-uniform trip count and keeps a NEARNESS byte of
//     every slot (a monotone, DEcreasing map of the exact fp32 distance: 63 levels over [0.4 upper, upper), one more
//     for everything nearer, 0 for anything at or beyond the ring bound -- the float -> byte conversion saturates
//     negative values to 0, so the map needs no clamp instruction) in a register array (statically indexed: the loop
//     is fully unrolled);
//   * the level holding the K-th nearest is found by bisection with SWAR byte compares on those registers
//     (count(byte >= beta) = popcount((w + (128 - beta) * 0x01010101) & 0x80808080) per word) -- no LDS histogram,
//     no atomics, no second distance evaluation;
//   * pass 2 re-reads the bytes: slots nearer than the K-th level add their flow (the SWAR flag byte 0x80 / 0 converted
//     to 128.0 / 0.0 is the weight of an fma; the factor 128 is a power of two, every rounding is that of the plain sum),
//     slots at the K-th level are noted in a bit mask and ranked afterwards by exact (distance, index).
// A query the fast path cannot serve (no square up to KNN_RCAP cells holds enough points: the inside of an emptied band;
// fewer than K candidates below the ring bound after all; more slots than the registers hold; a strip whose points overflow
// the staging area even in quarters) is marked for the strip workgroups of the tail launch (k_knn_tail: radii up to KNN_RFAR, chord-shaped
// regions, 192 slots) or appended to a list and searched by its fallback workgroups, one wavefront per query: the result is the exact
// K-nearest set for any input, ties to the lowest index.
// (Names in the comments below, from the rounds in which these were launches of their own: "second launch" / "far pass" = the strip
// workgroups of k_knn_tail, strip_more_body; "fallback kernel" = its one-wavefront-per-query workgroups, fallback_one_query.)
#include "knn_device.h"
#include "ev_count_device.h"
#include "bounds.h"
#include "diag/stamps.h"
#include <stdlib.h>

#define KS_NT 256
#define KS_MAXCH 24                 // words of 4 slots per query on the fast path (96 slots)
#ifndef KS_OCC_WIDE
#define KS_OCC_WIDE 5               // workgroups per CU of the variants with the wider register set (flow_to_next, L1): 96 VGPRs
#endif
#ifndef KS_MAXCH_L1
#define KS_MAXCH_L1 32              // ... with the L1 norm (128 slots): the ball of the ring bound is a diamond, HALF of its square -- a square that
                                    // holds K points in its diamond holds ~2 K (99 slots at the DSEC density); eight more nearness words per lane,
                                    // the 96-register budget of the flow_to_next variant (five workgroups per CU)
#endif
#define KS_BASECH 16                // words every bisection step counts; the rest only in wavefronts that use them
#define KS_NLEV 64                  // nearness levels: 1 .. 63 over [0.4 upper, upper), 64 = nearer than that; byte 0 = not a candidate
#define KS_LMAX 4                   // keys of the K-th level a lane ranks in registers; more: served by the whole wavefront
#define KS_TAIL(MAXCH_) (4 * (MAXCH_) + 8)  // slots of far-away dummy points behind the staged ones (reads beyond a range)
#define KS_FAR 1.0e18f              // coordinate of a dummy point: its distance is finite and beyond any bound
#define KS_FB_SLOTS 4               // fallback, wavefront per query: candidates per lane (64 * 4 per query)
#define KS_FB_BLOCKS 1024           // one-wavefront-per-query workgroups of the tail launch
#ifndef KS_RETRY_BLOCKS
#define KS_RETRY_BLOCKS (256 * KS_MORE_OCC)   // strip workgroups of the tail launch
#endif
static_assert(KNN_FAR_WS * KNN_FAR_TH == KS_NT, "a block of queries of the second launch = one workgroup");
static_assert(KNN_RCAP <= 6, "the packed chord widths of the main launch");
static_assert(KNN_MARGIN > KNN_RCAP, "the strip kernel must not reach the outermost ring of the bucket grid");

// 'iwd' weight of a neighbour at distance d (focus.py:159-161: 1 / (d + 1e-9)) on the strip path: ONE hardware reciprocal (v_rcp_f32,
// 1 ulp) instead of the ~10 instructions of the IEEE division, per visited slot of pass 2 (round 6: `k_knn_strip<*, *, true>` 299 -> 270 us at
// C3, 262 with the branch around a slot's arithmetic gone too; the weights' rounding is a relative 6e-8 of a term of a 32-term weighted
// mean, the LUT is held to 1e-5 against the reference)
#ifndef KS_IWD_PJ
#define KS_IWD_PJ 4          // positions of pass 2 in flight together ('iwd')
#endif
#ifndef KS_IWD_RCP
#define KS_IWD_RCP 1
#endif
__device__ __forceinline__ float ks_iwd_weight(float d) { return KS_IWD_RCP ? __builtin_amdgcn_rcpf(d + 1e-9f) : 1.f / (d + 1e-9f); }
// value of lane `l` (wave-uniform l): v_readlane, no LDS round trip
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned lane_u(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// (8-byte LDS reads go through knn_lds_f2 -- knn_device.h: single ds_read_b64; the compiler's ds_read2_b64 pairs run at half
// the LDS rate, and the LDS is busy for 62 % of this kernel's cycles: SQ_LDS_IDX_ACTIVE)
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// One strip (workgroup-wide): the query rows [pr0, pr1) of strip `lblk`.  SPLIT = false: the whole strip (pr0 = 0, pr1 = TH); a
// strip whose points do not fit the staging area (a place where the flow field packs the points) goes on the retry list and
// is searched again by k_knn_strip_retry in quarters (SPLIT = true); what does not fit a quarter goes to the fallback list.
// MODE 0: the main launch -- the queries whose square of up to KNN_RCAP cells holds enough points; a strip that holds others (FAR
// queries) is noted on the farstrip list.  MODE 1: k_knn_strip_retry (SPLIT).  MODE 2: k_knn_strip_far -- the far queries of the
// strips on that list alone, with radii up to KNN_RFAR, the region rows staged as wide as the widest CHORD that uses them (the
// disc of the ring bound, not the square: above and below a horizontal band front the squares are a hundred cells wide);
// served far queries go on the far list of their (sample, bin) for k_knn_bwd_far and stay out of the tile maxima.
#define KS_NR_MAX (KS_NT / 2 + 2 * KNN_RFAR)
template <int WS, bool L1, bool NEXT, bool IWD, int MODE>
__device__ __forceinline__ void strip_body(const KnnParams &p, const float *__restrict__ traj,
                                           const knn_cs_t *__restrict__ cell_start, const knn_cs_t *__restrict__ sat,
                                           const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                           float *__restrict__ flow_lut, float *__restrict__ flow_next,
                                           float *__restrict__ knn_state, float *__restrict__ tile_dkmax,
                                           const KnnLists &ls, int r_init, int cap, int gx, int gy,
                                           int lblk, int pr0, int pr1, unsigned char *s_dyn, int *s_wsum, int *s_wmax, unsigned char *s_rq,
                                           int unused_ = 0) {
    // (Round 5 measured the far pass at WAVEFRONT level -- a wavefront per block of 2 x 24 queries with its own LDS slice, no
    // workgroup barrier, four independent searches per workgroup instead of one wavefront searching while three wait: slower,
    // UNet-like mixture 185 us against 160, 30 % contraction 725 against 492 -- six blocks per strip each stage their own +-20-row
    // halo with 64 lanes instead of 256, and most blocks of a listed strip hold no marked query.  profiles/HISTORY_r05.md)
    constexpr bool SPLIT = MODE == 1, FARK = MODE == 2;
    constexpr int NT = KS_NT;
#define KS_SYNC() __syncthreads()
    KS_STP_DECL
    KS_STP();
    constexpr int MAXCH = FARK ? KS_MAXCH_FAR : (L1 ? KS_MAXCH_L1 : KS_MAXCH);      // words of four slots per query
    (void)SPLIT;
    constexpr int RC = FARK ? KNN_RFAR : KNN_RCAP;
    constexpr int TH = KS_NT / WS;
    constexpr int NR = TH + 2 * RC;                     // region rows: the strip's query rows and the largest radius above and below
    static_assert(NR <= NT && NR <= KS_NR_MAX, "one thread per region row");
    // queries for the one-wavefront-per-query search: the main launch appends to the `fail` list (counter fail[0], entries upwards from
    // fail[1]); the strip workgroups of the tail kernel append to the LATE list (counter knn_late_count, entries downwards from the end
    // of the same array, behind the marked list where that one is alive), which the tail kernel's fallback workgroups take once all
    // strip workgroups are done.  (Everything about the lists is worked out where a push happens -- a rare path --, not kept live
    // across the search: the kernel has no scalar register to spare, and a spilled one costs a vector register.)
    int *const fail = ls.fail;
    (void)unused_;
    // main launch: the queries a workgroup marks for the tail's strip workgroups also go on the MARKED list (knn_device.h) with the
    // radius that search would start from.  Workgroup-level: the ONE atomic with which the workgroup adds its marked queries to the
    // launch's count (it did that before) also reserves their places -- a list append per wavefront (a second global counter) cost
    // the main launch 4 us on white noise and 50-220 us on band-heavy inputs.  All threads call (n = the workgroup's marked queries).
    auto mark_list = [&](bool on, size_t qid, int rad, int n) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const unsigned long long pm = __ballot(on);
        __syncthreads();                                   // (s_wsum / s_wmax are free again)
        if (lane == 0) s_wsum[wv] = __popcll(pm);
        if (threadIdx.x == 0) s_wmax[0] = atomicAdd(knn_marked_count(ls), n);
        __syncthreads();
        int rank = s_wmax[0] + __popcll(pm & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wv; ++w2) rank += s_wsum[w2];
        const long long fcap = (long long)p.B * p.nb * p.G;
        const unsigned hint = (fcap < (1ll << 24)) ? (unsigned)min(max(rad, 1), 63) << 24 : 0u;
        if (on && rank < fcap) fail[MPC_IDX(fcap - (long long)rank, 1 + fcap)] = (int)((unsigned)qid | hint | (1u << 30));
    };
    const int tid = threadIdx.x;
    if (MODE != 1) {
        // (main and second launch) the width of every region row, collected below with LDS atomics: starts at 0.  The table lives where
        // the first bucketed slots of the rows go later (s_row); this early barrier costs nothing: every wavefront is here at once
        if (tid < NR) reinterpret_cast<int2 *>(s_dyn)[MPC_IDX(tid, NR)].x = 0;
        if (tid == 0) s_rq[0] = 0;                    // (main launch: "this strip holds far queries", set below; it has no other use for s_rq)
        KS_SYNC();
    }
    const int bt = lblk / (gx * gy), bxy = lblk - bt * gx * gy;
    const int sy = bxy / gx, sx = bxy - sy * gx;
    const int b = bt / p.nb, t = bt - b * p.nb;
    const int qx0 = sx * WS, qy0 = sy * TH;
    const int qx1 = min(qx0 + WS, p.wq) - 1, qy1 = min(qy0 + TH, p.hq) - 1;
    const int ry_base = qy0 - RC;                // grid row of region row 0 (may lie outside the bucket grid: an empty row)
    // ---- LDS carve-up ----------------------------------------------------------------------------
    int2 *s_row = reinterpret_cast<int2 *>(s_dyn);             // [NR] {first bucketed slot, points (-1: no such row)}
    int *s_rowstart = reinterpret_cast<int *>(s_row + NR);     // [NR + 1] first slot of every region row
    size_t o = ((size_t)NR * 8 + (size_t)(NR + 1) * 4 + 15) & ~(size_t)15;
    float2 *lpos = reinterpret_cast<float2 *>(s_dyn + o); o += (size_t)(cap + KS_TAIL(MAXCH)) * 8;
    float2 *lflow = reinterpret_cast<float2 *>(s_dyn + o); o += (size_t)(cap + KS_TAIL(MAXCH)) * 8;
    float2 *lnext = reinterpret_cast<float2 *>(s_dyn + o); o += NEXT ? (size_t)(cap + KS_TAIL(MAXCH)) * 8 : 0;
    unsigned short *lidx = reinterpret_cast<unsigned short *>(s_dyn + o);

    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    const knn_cs_t *sat_bt = sat + (size_t)bt * (p.hb + 1) * (p.wb + 1);
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const knn_idx_t *si_ = sidx + (size_t)bt * p.n;
    const float2 *traj_b = reinterpret_cast<const float2 *>(traj) + (size_t)b * (p.T + p.nb) * p.n;

    // ---- the query of this thread and its search radius -------------------------------------------
    // Thread -> query: 16-row groups of the strip (a wavefront = two groups).  The queries next to the top and bottom
    // border of the image are the expensive ones (half of their neighbourhood lies outside the image: larger squares, up to
    // 96 slots against 63) and a wavefront pays for its most expensive lane, so the group holding the bottom border trades
    // places with group 1: ONE wavefront of a full-height strip carries both borders, not two.
    int grp = tid / (16 * WS);
    if (qy0 == 0 && p.hq <= TH) {
        const int gb = (p.hq - 1) >> 4;
        if (gb >= 2) grp = (grp == 1) ? gb : ((grp == gb) ? 1 : grp);
    }
    int cy = qy0 + grp * 16 + (tid % (16 * WS)) / WS, cx = qx0 + tid % WS;
    const bool valid = cy <= qy1 && cx <= qx1;
    // Radius from the summed-area table (four loads per radius tried): the smallest r whose square holds `need` points.
    // Starts at the radius of the mean density; only a clearly denser place tries smaller ones.
    const int aw = (p.wq + 31) >> 5;
    const int aoff = bt * ls.again_words + min(cy, p.hq - 1) * aw + (min(cx, p.wq - 1) >> 5);      // this query's word of the `again` / `grow` maps
    // (the second launch: only the queries the main launch marked look at the table at all -- `grow` alone: a far query, no
    // square up to KNN_RCAP holds enough points; `again`: one it could not finish, with `grow` if for too few candidates)
    // (the radius hint of a marked query -- below -- is requested with the two words, for every query: one round trip, not two)
    int r_hint = 0;
    if (FARK && valid) r_hint = reinterpret_cast<const int *>(knn_state)[(size_t)bt * p.G + (size_t)cy * p.wq + cx];
    const bool bit_again = FARK && valid && ((ls.again[aoff] >> (cx & 31)) & 1u) != 0u;
    const bool bit_grow = FARK && valid && ((ls.grow[aoff] >> (cx & 31)) & 1u) != 0u;
    const bool marked = bit_again || bit_grow, farq = bit_grow && !bit_again;
    int r = 0, nr = 0;                                  // radius and the points in its square
    bool served = false;
    const int need_q = knn_square_need(p.K, p.l1);
    // (second launch: the radius the main launch worked out for a far query / tried for one it could not finish, left in the
    // query's K-th distance slot of knn_state -- whichever kernel serves the query overwrites it.  Reading the table again here
    // was a chain of four to five dependent round trips in front of everything else a work item does: 12 of its 29 us.)
    if (FARK && marked) r = r_hint;
    if (!FARK && valid) {
        const int need = need_q;
        r = min(r_init, KNN_RCAP);
        nr = knn_square_count(p, sat_bt, cy, cx, r);
        if (nr >= need) {
            if (2 * nr >= 3 * need) {
                for (;;) {
                    const int nm = r > 1 ? knn_square_count(p, sat_bt, cy, cx, r - 1) : 0;
                    if (nm < need) break;
                    --r; nr = nm;
                }
            }
        } else {
            for (++r; r <= KNN_RCAP; ++r) { nr = knn_square_count(p, sat_bt, cy, cx, r); if (nr >= need) break; }
        }
        served = r <= KNN_RCAP;
    }
    bool isfar = FARK ? farq : (valid && !served);      // no square up to KNN_RCAP holds enough points: the second launch's query
    if (!FARK) {
        // The count says how many points the SQUARE holds; the disc of the ring bound holds pi / 4 of them, give or take the
        // scatter of the positions.  With fewer than need + need / 7 points in the square (47 for K = 32: the lattice has 49
        // in its 7 x 7 square) one query in ten .. a hundred comes up short.  Where that is the rare lane (white-noise
        // coefficients: 9 % of the lanes) the radius stands -- a wavefront pays for its widest lane, one more ring is 90 slots
        // instead of 63 -- and the query that does come up short is searched again; where it is the rule (an expanding flow
        // field: every square holds ~44) those lanes take one more ring right away.
        const int need = need_q;
        // (a square with barely `need` points comes up short every other time: one more ring whatever the neighbours do)
        const bool marginal = served && r < KNN_RCAP && nr < need + need / 7;
        const int nm = __popcll(__ballot(marginal)), nv = __popcll(__ballot(served));
        if (marginal && (4 * nm >= nv || nr < need + 2)) ++r;
        // (for the launch that follows: `grow` without `again` = far; the strip goes onto its work list when this workgroup ends)
        // (MODE 0 only: the quarters of an overflowed strip run beside the second launch's far pass -- which may have served the
        // query and written its K-th distance by now -- and the main launch marked the strip's far queries before it gave up)
        if (MODE == 0 && isfar) {
            // smallest radius in (KNN_RCAP, KNN_RFAR] whose square holds the far queries' count, for the second launch: bisection
            // (the count grows with the radius), four probes instead of up to fourteen -- here, where five other workgroups of
            // the CU cover the round trips (KNN_RFAR + 1: nothing within KNN_RFAR, the fallback kernel)
            // (a LOWER BOUND of the candidates below the ring bound -- the points of the cells inside a cross of rectangles
            // inscribed in the disc, five rectangle counts -- instead of this estimate was measured: no fewer queries on the
            // fallback list, larger radii, queries in dense places beyond the 256 slots: not kept)
            const int need_far = knn_square_need_far(p.K, p.l1);
            int lo = KNN_RCAP, hi = KNN_RFAR + 1;       // (lo: too few; hi: enough)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (knn_square_count(p, sat_bt, cy, cx, mid) >= need_far) hi = mid; else lo = mid;
            }
            reinterpret_cast<int *>(knn_state)[MPC_IDX((size_t)bt * p.G + (size_t)cy * p.wq + cx, (long long)p.B * p.nb * p.G)] = hi;
            atomicOr(ls.grow + MPC_IDX(aoff, (long long)p.B * p.nb * ls.again_words), 1u << (cx & 31)); s_rq[0] = 1;
            r = hi;                                      // (the far pass starts there; so does the marked-list entry below)
        }

    }
    if (FARK) {
        served = false;
        if (isfar) served = r <= KNN_RFAR;              // (else: the fallback kernel)
        else if (bit_again) {
            // the main launch could not finish it: too few candidates below the ring bound -- two more rings -- or more slots
            // than its registers hold -- the same radius with this launch's 192 slots
            if (bit_grow) r = min(r + 2, KNN_RFAR);
            served = true;
        }
    }
    bool mine = FARK ? marked : (valid && !isfar);
    {   // radius of the widest square of every query row (the WS lanes of a row are neighbours)
        int rr = (mine && served) ? r : 0;
#pragma unroll
        for (int o2 = 1; o2 < WS; o2 <<= 1) rr = max(rr, __shfl_xor(rr, o2, 64));
        if (MODE == 1) { if ((tid % WS) == 0 && cy - qy0 < TH) s_rq[MPC_IDX(cy - qy0, KS_NT / WS)] = (unsigned char)rr; }
        else if (MODE == 2) { }                          // (pushed per query after the compaction below)
        else if ((tid % WS) == 0 && rr > 0) {
            // main launch: every query row pushes the width of its disc's chord onto the region rows it uses -- 2 r + 1 LDS
            // atomics per row of queries instead of every region row looking at 13 query rows (the loop of the other launches)
            int2 *s_w = reinterpret_cast<int2 *>(s_dyn);
            const int rc = cy - qy0 + RC;                   // region row of this query row
            for (int j = -rr; j <= rr; ++j) {
                const int aj = abs(j);
                int wc;
                // the chord of the query's disc at row offset j (the corner cells of the square hold nothing below the ring
                // bound): only the outermost row of a square (and the one before it from r = 5) is narrower --
                // (2w - 1)^2 + (2j - 1)^2 < (2r + 1)^2 in half cells, whatever the cell size: the widths packed four bits per
                // radius; the L1 ball is the diamond w <= r - j + 1
                if (!KS_MAIN_CHORD) wc = rr;
                else if (L1) wc = min(rr, rr - aj + 1);
                else wc = aj == rr ? (int)((0x3332210u >> (4 * rr)) & 15u) : (aj == rr - 1 ? (int)((0x5443210u >> (4 * rr)) & 15u) : rr);
                atomicMax(&s_w[MPC_IDX(rc + j, NR)].x, wc);
            }
        }
    }
    // chord of the disc of radius r at row offset j, for the row tables below (knn_device.h)
    __shared__ unsigned char s_chord[(RC + 1) * (RC + 1)];
    __shared__ unsigned s_ft[FARK ? KNN_FT_LDS_WORDS : 1];          // far pass: tiles the served far queries' discs touch (knn_far_mark_tiles_lds)
    if (FARK && tid < KNN_FT_LDS_WORDS) s_ft[tid] = 0u;
    if (FARK) for (int i = tid; i < (RC + 1) * (RC + 1); i += NT) s_chord[i] = ls.chord[(i / (RC + 1)) * (KNN_RFAR + 1) + i % (RC + 1)];
    // (the barrier the row tables need anyway) MODE 0: any far query in this strip?  Then it goes on the list of
    // k_knn_tail when this workgroup ends
    KS_SYNC();
    KS_STP();        // 1: marks read, radius from the table
    const bool anyfar = MODE == 0 && s_rq[0] != 0;
    if (FARK) {
        // The marked queries of the strip -- a few dozen of its 256 -- move into the first lanes: one wavefront searches them
        // instead of four that each carry a handful (the row tables above are per query ROW: they do not care).
        int *s_list = reinterpret_cast<int *>(lpos);            // (the staging area is not in use yet)
        const int lane = tid & 63, wv = tid >> 6;
        const unsigned long long mm = __ballot(mine);
        if (lane == 0) s_wsum[wv] = __popcll(mm);
        __syncthreads();
        int base = 0, nmine = 0;
#pragma unroll
        for (int w2 = 0; w2 < KS_NT / 64; ++w2) { const int c = s_wsum[w2]; if (w2 < wv) base += c; nmine += c; }
        if (mine) s_list[MPC_IDX(base + __popcll(mm & ((1ull << lane) - 1ull)), KS_NT)] = (cy - qy0) | ((cx - qx0) << 8) | (r << 12) | (served ? 1 << 20 : 0) | (isfar ? 1 << 21 : 0);
        __syncthreads();
        // (round 5: WHICH wavefront takes the first 64 entries -- wavefront 0 of every workgroup, or a different one per workgroup so that
        // the few workgroups of a CU search on different SIMDs -- makes no difference: 107.0 against 106.1 us on the UNet-like mixture)
        const int vt = tid;
        mine = vt < nmine;
        served = false; isfar = false; r = 0;
        if (mine) {
            const int e = s_list[vt];
            cy = qy0 + (e & 0xff); cx = qx0 + ((e >> 8) & 0xf); r = (e >> 12) & 0xff; served = ((e >> 20) & 1) != 0; isfar = ((e >> 21) & 1) != 0;
        }
    }
    if (FARK) {
        // every query pushes the chords of its disc onto the region rows it uses (as the main launch does per query row)
        if (mine && served) {
            int2 *s_w = reinterpret_cast<int2 *>(s_dyn);
            const int rc = cy - qy0 + RC;
            for (int j = -r; j <= r; ++j) atomicMax(&s_w[MPC_IDX(rc + j, NR)].x, (int)s_chord[MPC_IDX(r * (RC + 1) + abs(j), (RC + 1) * (RC + 1))]);
        }
        KS_SYNC();                                               // (before the row tables are read and the staging area is written)
        KS_STP();    // 2: compaction, chords pushed
    }
    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;       // global query id
    // ---- column extent of every region row = the widest square (of the query rows [pr0, pr1) of the strip) that uses the
    //      row; slots of the region rows: an exclusive scan of the row lengths; a row of even length gets one dummy slot so
    //      that the row pitch is odd (consecutive rows then start in different LDS banks: with the 8 points per row of a
    //      regular lattice an unpadded pitch puts every fourth row on the same banks).  Returns the slots of the region.
    int pitch = 1;
    auto region_rows = [&](int pr0, int pr1) -> int {
        int len = -1, padded = 0, gs = 0;
        if (tid < NR) {
            const int y = ry_base + tid;
            int R = 0;
            if (MODE != 1) R = s_row[tid].x;                // (collected with atomics above)
            else {
                const int c0 = max(tid - 2 * RC, pr0), c1 = min(tid, pr1 - 1);       // query rows within RC of this row
                for (int c = c0; c <= c1; ++c) {
                    const int rq = (int)s_rq[c], j = abs(c - (tid - RC));
                    if (j > rq) continue;
                    // the chord of the query's disc at this row: far queries from the table, the quarters of an overflowed strip
                    // the square
                    R = max(R, FARK ? (int)s_chord[rq * (RC + 1) + j] : rq);
                }
            }
            if (R > 0 && y >= -p.m && y < p.hq + p.m) {
                const int xl = max(qx0 - R, -p.m), xh = min(qx1 + R, p.wq + p.m - 1);
                gs = cs[MPC_IDX(knn_ci(p, y, xl), p.Gb + 1)];
                len = cs[MPC_IDX(knn_ci(p, y, xh + 1), p.Gb + 1)] - gs;
                // (an EMPTY row takes no slot at all: a far query's square is mostly empty rows -- 28 of the 33 at radius 16 --
                // and a dummy slot for each used to eat a quarter of its 128 slots)
                padded = len == 0 ? 0 : len + ((len & 1) ? 0 : 1);
            }
        }
        int incl = padded;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(incl, o2, 64); if ((tid & 63) >= o2) incl += v; }
        const int wmx = wave_max_i(padded);
        if ((tid & 63) == 63) { s_wsum[tid >> 6] = incl; s_wmax[tid >> 6] = wmx; }
        KS_SYNC();
        int run = incl - padded;
        for (int w = 0; w < (tid >> 6); ++w) run += s_wsum[w];
        if (tid < NR) { s_rowstart[MPC_IDX(tid, NR + 1)] = run; s_row[MPC_IDX(tid, NR)] = make_int2(gs, len); }
        if (tid == NR - 1) s_rowstart[NR] = run + padded;
        // row pitch of the staging loop = the longest row of the region
        pitch = max(max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])), 1);
        KS_SYNC();
        return s_rowstart[NR];
    };
    const bool inpass = mine && (cy - qy0) >= pr0 && (cy - qy0) < pr1;
    const int total = region_rows(pr0, pr1);
    KS_STP();        // 3: row table
    bool overflow = false;
    if (total > cap) {
        if (MODE == 0) {              // (workgroup-uniform) again in quarters: k_knn_tail
            const int nfq = anyfar ? __syncthreads_count(isfar ? 1 : 0) : 0;
            if (tid == 0) {
                ls.retry[MPC_IDX(1 + atomicAdd(&ls.retry[0], 1), 1 + (long long)gx * gy * p.B * p.nb)] = lblk;
                if (anyfar) ls.farstrip[MPC_IDX(1 + atomicAdd(&ls.farstrip[0], 1), 1 + (long long)gx * gy * p.B * p.nb)] = lblk;
            }
            if (anyfar) mark_list(isfar, (size_t)bt * p.G + (size_t)cy * p.wq + cx, r, nfq);      // (workgroup-uniform)
            return;
        }
        overflow = true;              // even a quarter of the strip (or the far queries' region) does not fit: to the fallback list
    }
    {
        const bool act = inpass && served && !overflow;
        // ---- stage positions, flows and indices: item = (region row, k-th slot of the row), KS_SB items per thread
        //      with their (dependent) global loads in flight together ----------------------------------------------
        if (!overflow) {
            const float2 *tref0 = traj_b;                                     // T == 1: the reference time
            const float2 *tnext = traj_b + (size_t)(p.T + t + 1) * p.n;       // next bin (if any)
            const bool has_next = NEXT && (t < p.nb - 1);
            // item -> (row, slot of the row) with a reciprocal multiply: (it + 0.5) / pitch is at least 0.5 / pitch away from
            // an integer, far more than the rounding of the product (it < 2^14)
            // (second launch: item = staged slot, its row by a search over the first slots of the rows -- the rows of a far
            // query's region are up to 42 cells wide and nearly all of them empty: on the (row, slot of the row) grid of the
            // main launch a workgroup walked 7 000 items for 300 points, nine rounds of dependent loads)
            const int items = FARK ? total : NR * pitch;
            const float inv_pitch = __builtin_amdgcn_rcpf((float)pitch);        // (1 ulp: the margin below is 0.5 / pitch)
            for (int base = 0; base < items; base += NT * KS_SB) {
                int slot[KS_SB], id[KS_SB];
                bool in[KS_SB], real[KS_SB];
                float2 pj[KS_SB], f0[KS_SB], f1[KS_SB];
#pragma unroll
                for (int u = 0; u < KS_SB; ++u) {
                    const int it = base + u * NT + tid;
                    int rr, k;
                    if (FARK) {
                        // the last row whose first slot is <= it (an empty row shares its first slot with the row behind it)
                        int lo = 0, hi = NR;
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rowstart[mid] <= it) lo = mid; else hi = mid; }
                        rr = lo; k = it - s_rowstart[lo];
                    } else {
                        rr = min((int)(((float)it + 0.5f) * inv_pitch), NR - 1); k = it - rr * pitch;
                    }
                    const int2 row = s_row[MPC_IDX(rr, NR)];                                     // {first bucketed slot, points}
                    in[u] = it < items && row.y > 0 && k < (row.y | 1);             // (an even row has one dummy slot: odd pitch; an empty row none)
                    real[u] = in[u] && k < row.y;
                    slot[u] = FARK ? it : s_rowstart[MPC_IDX(rr, NR + 1)] + k;
                    pj[u] = make_float2(KS_FAR, KS_FAR); id[u] = 0;
                    if (real[u]) { pj[u] = sp_[MPC_IDX(row.x + k, p.n)]; id[u] = si_[MPC_IDX(row.x + k, p.n)]; }
                }
#pragma unroll
                for (int u = 0; u < KS_SB; ++u) {
                    f0[u] = f1[u] = make_float2(0.f, 0.f);
                    if (real[u]) {
                        const float2 a = tref0[MPC_IDX(id[u], p.n)];
                        f0[u] = make_float2(a.x - pj[u].x, a.y - pj[u].y);      // traj(t_ref) - traj(t_mid)  focus.py:140-141
                        if (has_next) { const float2 c = tnext[MPC_IDX(id[u], p.n)]; f1[u] = make_float2(c.x - pj[u].x, c.y - pj[u].y); }
                    }
                }
#pragma unroll
                for (int u = 0; u < KS_SB; ++u) {
                    if (in[u]) {
                        lpos[MPC_IDX(slot[u], cap + KS_TAIL(MAXCH))] = pj[u];
                        lflow[MPC_IDX(slot[u], cap + KS_TAIL(MAXCH))] = f0[u];
                        if (NEXT) lnext[MPC_IDX(slot[u], cap + KS_TAIL(MAXCH))] = f1[u];
                        lidx[MPC_IDX(slot[u], cap)] = (unsigned short)id[u];
                    }
                }
            }
            // the tail behind the staged slots: far-away positions, zero flows (lanes whose range is shorter than the
            // wavefront's trip count read them, flagged off)
            for (int i = total + tid; i < total + KS_TAIL(MAXCH); i += NT) {
                lpos[MPC_IDX(i, cap + KS_TAIL(MAXCH))] = make_float2(KS_FAR, KS_FAR);
                lflow[MPC_IDX(i, cap + KS_TAIL(MAXCH))] = make_float2(0.f, 0.f);
                if (NEXT) lnext[MPC_IDX(i, cap + KS_TAIL(MAXCH))] = make_float2(0.f, 0.f);
            }
        }
        KS_SYNC();
        KS_STP();    // 4: staged

        // ---- search --------------------------------------------------------------------------------------
        const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
        bool failed = inpass && (!served || overflow);       // far query beyond KNN_RFAR (why 0) / staging overflow (why 2)
        unsigned why = (inpass && served && overflow) ? 2u : 0u;      // diagnostics: top two bits of a list entry (0 few candidates, 1 too many slots, 2 staging overflow)
        int s = 0, nsl = 0;
        if (act) {
            s = s_rowstart[MPC_IDX(cy - r - ry_base, NR + 1)];
            nsl = s_rowstart[MPC_IDX(cy + r - ry_base + 1, NR + 1)] - s;
            if (nsl > 4 * MAXCH) { failed = true; why = 1u; nsl = 0; s = 0; }
        }
        // anything outside the square is at least lb away along one axis
        const float lb = ((float)r + 0.5f) * (float)p.sp - KNN_SLACK;
        const float upper = L1 ? lb : lb * lb;
        // NEARNESS of a slot, one byte: (upper - d) * (NLEV - 1) / (upper - lo) converted with saturation -- a monotone map of the
        // exact fp32 distance (that is all exactness needs: a slot with a smaller byte is at least as far as every slot with a
        // larger one).  0 <=> not below the ring bound (or a dummy slot: the conversion saturates negative values to 0, which
        // is why the map runs downwards -- no clamp instruction per slot); 1 .. NLEV - 1 resolve [lo, upper); everything nearer
        // than lo lands in NLEV .. 127 and counts as ONE level (NLEV).  lo = upper / 2: the K-th distance of an unclipped
        // square sits near 0.83 upper; a clipped square was enlarged by whole rings, its K-th distance can be as low as
        // upper / 2: lo = upper / 4.  Largest byte: upper * (NLEV - 1) / (upper - lo) = 126 or 84 -- bytes stay below 128, which
        // the SWAR compares rely on.  (A candidate within half a level of the ring bound converts to 0: it is treated as
        // outside, which only makes the fast path give up earlier -- the byte of anything at or beyond the bound is 0 for sure:
        // the rounding of the fma is ~1e-5 of a level.)
        const float lo_d = 0.4f * upper;
        // (hardware reciprocal, 1 ulp: any constant near this one gives a monotone map; the largest byte stays below 127.5)
        const float nscale = -(float)(KS_NLEV - 1) * __builtin_amdgcn_rcpf(upper - lo_d), loff = -upper * nscale;
        const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(nsl));      // wave-uniform trip count (slots)
        const float2 *pp = lpos + s;
        // (every lane reads its range rounded up to the wavefront's trip count: the dummy slots behind the staged ones are there for that)
        MPC_EXPECT(s + ((nmax + 7) & ~7) <= cap + KS_TAIL(MAXCH));
        // pass 1: nearness byte of every slot; groups of 8 slots whose loads are issued together
        unsigned w[MAXCH];
#pragma unroll
        for (int g = 0; g < MAXCH / 2; ++g) {
            w[2 * g] = w[2 * g + 1] = 0u;
            if (8 * g < nmax) {
                float2 pj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) pj[u] = knn_lds_f2(pp + 8 * g + u);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    unsigned acc = 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d = pair_dist(qy, qx, pj[4 * h + u].x, pj[4 * h + u].y, L1);
                        acc = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, nscale, loff), u, acc);       // saturating: negative -> 0
                    }
                    w[2 * g + h] = acc;
                }
            }
        }
        // number of slots with byte >= beta (1 <= beta <= 128): bytes are <= 127, so (byte + 128 - beta) has bit 7 exactly when
        // byte >= beta, and no carry crosses a byte (words of groups not visited hold 0)
        auto count_ge = [&](unsigned beta) {
            const unsigned C = (128u - beta) * 0x01010101u;
            // (two v_bcnt_u32_b32 accumulate chains: the compiler's own form is bcnt + a tree of adds, half an instruction
            // more per word)
            int acc = 0, acc1 = 0;
#pragma unroll
            for (int c = 0; c < KS_BASECH; c += 2) {
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"((w[c] + C) & 0x80808080u));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc1) : "v"((w[c + 1] + C) & 0x80808080u));
            }
            acc += acc1;
            // (an inner query has 7 rows of 9 slots; only wavefronts next to the image border, whose rows are wider, get
            // here: real branches on the wave-uniform trip count -- the empty asm keeps them from being if-converted)
            if (nmax > 4 * KS_BASECH) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = KS_BASECH; c < KS_BASECH + 4; ++c) acc += __popc((w[c] + C) & 0x80808080u);
                if (nmax > 4 * KS_BASECH + 16) {
                    asm volatile("" ::: "memory");
                    constexpr int T2 = MAXCH < 32 ? MAXCH : 32;
#pragma unroll
                    for (int c = KS_BASECH + 4; c < T2; ++c) acc += __popc((w[c] + C) & 0x80808080u);
                    // (the second launch: up to 192 or 256 slots, in two more steps)
                    if (MAXCH > 32 && nmax > 128) {
                        asm volatile("" ::: "memory");
                        constexpr int T3 = MAXCH < 48 ? MAXCH : 48;
#pragma unroll
                        for (int c = 32; c < T3; ++c) acc += __popc((w[c] + C) & 0x80808080u);
                        if (MAXCH > 48 && nmax > 192) {
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int c = 48; c < MAXCH; ++c) acc += __popc((w[c] + C) & 0x80808080u);
                        }
                    }
                }
            }
            return acc;
        };
        int bstar = 0, before = 0, inbin = 0;
        if (act && !failed) {
            // the level of the K-th nearest = the largest beta in [1, NLEV] with count_ge(beta) >= K  (NLEV and everything above
            // it is one level: `hi` starts behind it with "nothing is nearer")
            int lo = 1, clo = -1, hi = KS_NLEV + 1, chi = 0;      // count_ge(lo) >= K > count_ge(hi) -- assumed for lo = 1, checked below
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const int cm = count_ge((unsigned)mid);
                if (cm >= p.K) { lo = mid; clo = cm; } else { hi = mid; chi = cm; }
            }
            // the number of candidates below the ring bound is only needed when the search ends at the bottom level (rare)
            if (clo < 0) {
                clo = count_ge(1u);
                if (clo < p.K) failed = true;          // fewer than K candidates below the ring bound: the square must grow
            }
            bstar = lo; before = chi; inbin = clo - chi;     // level of the K-th nearest, slots nearer than it, slots in it
        }
        const bool live = act && !failed;
        // pass 2: flows of the slots NEARER than level bstar, slots AT level bstar into a bit mask.  Per word of four bytes:
        // ge1 = bit 7 of (byte + 128 - bstar) <=> byte >= bstar; ge2 likewise for bstar + 1 (for bstar = NLEV -- the one level of
        // everything nearer than lo -- nothing is nearer: beta 128).  ge2's bytes are 0x80 / 0: converted to 128.0 / 0.0 they are
        // the weight of the slot's flow, so the sums below carry a factor of 128 (a power of two: every rounding is that of the
        // plain sum) which the final division removes.  Dead lanes: beta 128 twice -- no slot anywhere.
        const bool do_next = NEXT && (t < p.nb - 1);
        float sy_ = 0.f, sx_ = 0.f, sw_ = 0.f, ny_ = 0.f, nx_ = 0.f;
        constexpr int NE = (MAXCH + 7) / 8;
        unsigned E[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) E[e] = 0u;
        const unsigned beta2 = (live && bstar < KS_NLEV) ? (unsigned)bstar + 1u : 128u;
        const unsigned c1 = (128u - (live ? (unsigned)bstar : 128u)) * 0x01010101u, c2 = (128u - beta2) * 0x01010101u;
        // The four flags of word j (of the eight words of a mask) go to bits j, 8 + j, 16 + j, 24 + j: slot 32 m + 4 j + u <->
        // bit j + 8 u of E[m].
        const float2 *pf = lflow + s, *pn = lnext + s;
#pragma unroll
        for (int g = 0; g < MAXCH / 2; ++g) {
            if (8 * g < nmax) {
                float2 fj[8], gj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { fj[u] = knn_lds_f2(pf + 8 * g + u); if (NEXT) gj[u] = knn_lds_f2(pn + 8 * g + u); }
                unsigned near4[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = (2 * g + h) & 7;
                    const unsigned ge1 = w[2 * g + h] + c1, ge2 = w[2 * g + h] + c2;
                    near4[h] = ge2 & 0x80808080u;
                    E[g / 4] |= ((ge1 ^ ge2) >> (7 - j)) & (0x01010101u << j);      // (ge2 implies ge1: the xor is "at level bstar")
                }
                float2 pj[IWD ? KS_IWD_PJ : 1];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (IWD && (u % KS_IWD_PJ) == 0) {
#pragma unroll
                        for (int v = 0; v < KS_IWD_PJ; ++v) pj[v] = knn_lds_f2(pp + 8 * g + u + v);      // (positions requested together, no branch around them)
                    }
                    // byte u & 3 of the word as a float, 128.0 or 0.0 (spelled out: with the bytes known to be 0x80 / 0 the
                    // compiler rewrites the shift-and-mask form into a shift, an SDWA and, and a conversion of byte 0)
                    float flag;
                    if ((u & 3) == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(flag) : "v"(near4[u >> 2]));
                    else if ((u & 3) == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(flag) : "v"(near4[u >> 2]));
                    else if ((u & 3) == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(flag) : "v"(near4[u >> 2]));
                    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(flag) : "v"(near4[u >> 2]));
                    if (IWD) {
                        // focus.py:159-161, branch-free (round 6): weight = flag x 1 / (d + 1e-9) -- 128 / (d + 1e-9) for a neighbour, 0 for
                        // any other slot (a dummy slot's distance is finite); the factor 128, a power of two, is in numerator and
                        // normaliser alike and leaves their quotient's bits alone
                        const float wgt = flag * ks_iwd_weight(pair_dist(qy, qx, pj[IWD ? u % KS_IWD_PJ : 0].x, pj[IWD ? u % KS_IWD_PJ : 0].y, L1));
                        sy_ = fmaf(wgt, fj[u].x, sy_); sx_ = fmaf(wgt, fj[u].y, sx_); sw_ += wgt;
                    } else { sy_ = fmaf(flag, fj[u].x, sy_); sx_ = fmaf(flag, fj[u].y, sx_); }
                    if (NEXT) { ny_ = fmaf(flag, gj[u].x, ny_); nx_ = fmaf(flag, gj[u].y, nx_); }
                }
            }
        }
        // ---- the K-th level: rank its keys by exact (distance, index); the first `need` are neighbours -----------------
        // Lanes with at most KS_LMAX keys there (nearly all) rank them in registers.  The others -- a lattice with little
        // flow has shells of up to eight points at practically one distance -- are HEAVY: rare per lane but present in
        // most wavefronts, so the whole wavefront serves them one at a time, a lane per slot (any number of keys).
        const int need = p.K - before;
        float dK = 0.f; int iK = -1;
        bool tie = false;                 // a point at exactly the K-th distance that is NOT a neighbour (higher index): KNN_TIE_FLAG
        const bool heavy = live && inbin > KS_LMAX;
        {
            float dd[KS_LMAX]; int ii[KS_LMAX], jj[KS_LMAX], kraw[KS_LMAX];
            const bool light = live && !heavy;
            const int mmax = __builtin_amdgcn_readfirstlane(wave_max_i(light ? inbin : 0));
            // the masks of the K-th level, 64 slots each (NM of them: two on the main launch)
            unsigned long long em = ((unsigned long long)E[1] << 32) | E[0];
            unsigned long long e2 = ((unsigned long long)(NE > 3 ? E[3] : 0u) << 32) | E[2];      // slots 64 ..
            unsigned long long e3 = 0ull, e4 = 0ull;                                               // slots 128 .., 192 .. (second launch)
            if (NE > 4) e3 = ((unsigned long long)(NE > 5 ? E[NE > 5 ? 5 : 0] : 0u) << 32) | E[NE > 4 ? 4 : 0];
            if (NE > 6) e4 = ((unsigned long long)(NE > 7 ? E[NE > 7 ? 7 : 0] : 0u) << 32) | E[NE > 6 ? 6 : 0];
#pragma unroll
            for (int a = 0; a < KS_LMAX; ++a) {
                dd[a] = INFINITY; ii[a] = 0x7fffffff; jj[a] = 0; kraw[a] = 0;
                if (a < mmax) {
                    int k = em ? __ffsll((long long)em) - 1 : (e2 ? 64 + __ffsll((long long)e2) - 1 : -1);      // a bit still in the mask ...
                    if (NE > 4 && !em && !e2) k = e3 ? 128 + __ffsll((long long)e3) - 1 : (e4 ? 192 + __ffsll((long long)e4) - 1 : -1);
                    if (em) em &= em - 1ull; else if (e2 || NE <= 4) e2 &= e2 - 1ull; else if (e3) e3 &= e3 - 1ull; else e4 &= e4 - 1ull;
                    kraw[a] = k;
                    k = (k & ~31) + 4 * (k & 7) + ((k & 31) >> 3);                                  // ... and its slot
                    if (light && k >= 0) {
                        const float2 pj = pp[MPC_IDX(k, cap + KS_TAIL(MAXCH) - s)];
                        jj[a] = k;
                        dd[a] = pair_dist(qy, qx, pj.x, pj.y, L1);
                        ii[a] = (int)lidx[MPC_IDX(s + k, cap)];
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < KS_LMAX; ++a) {
                if (a < mmax) {
                    int rank = 0;
#pragma unroll
                    for (int e = 0; e < KS_LMAX; ++e)
                        if (e != a) rank += ((dd[e] < dd[a]) | ((dd[e] == dd[a]) & (ii[e] < ii[a]))) ? 1 : 0;
                    if (light && a < inbin && rank < need) {
                        const float2 f = pf[jj[a]];
                        if (IWD) { const float wgt = 128.f * ks_iwd_weight(dd[a]); sy_ = fmaf(wgt, f.x, sy_); sx_ = fmaf(wgt, f.y, sx_); sw_ += wgt; }
                        else { sy_ = fmaf(128.f, f.x, sy_); sx_ = fmaf(128.f, f.y, sx_); }       // (the sums carry the factor 128 of pass 2)
                        if (do_next) { const float2 g2 = pn[jj[a]]; ny_ = fmaf(128.f, g2.x, ny_); nx_ = fmaf(128.f, g2.y, nx_); }
                        if (rank == need - 1) { dK = dd[a]; iK = ii[a]; }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < KS_LMAX; ++a) tie = tie || (light && a < inbin && dd[a] == dK && ii[a] > iK);
        }
        {
            unsigned long long hm = __ballot(heavy);
            const int lane = tid & 63;
            while (hm != 0ull) {                                       // wave-uniform loop over the heavy lanes
                const int h = __ffsll((long long)hm) - 1;
                hm &= hm - 1ull;
                const int hs = lane_i(s, h), hneed = lane_i(need, h);
                constexpr int NM = (NE + 1) / 2;                  // chunks of 64 slots (two on the main launch, four on the second)
                unsigned hw[2 * NM];
#pragma unroll
                for (int e = 0; e < 2 * NM; ++e) hw[e] = e < NE ? lane_u(E[e < NE ? e : 0], h) : 0u;
                const float hqy = lane_f(qy, h), hqx = lane_f(qx, h);
                // this lane's slots of the heavy query: lane, lane + 64, ...
                const int mybit = ((lane >> 2) & 7) + 8 * (lane & 3);            // bit of slot `lane` (and of slot 64 c + lane) in its mask word
                bool bb[NM]; float dc[NM]; int ic[NM], rc[NM];
#pragma unroll
                for (int c = 0; c < NM; ++c) {
                    bb[c] = (((lane < 32 ? hw[2 * c] : hw[2 * c + 1]) >> mybit) & 1u) != 0u;
                    dc[c] = INFINITY; ic[c] = 0x7fffffff; rc[c] = 0;
                    if (bb[c]) { const float2 pj = lpos[hs + 64 * c + lane]; dc[c] = pair_dist(hqy, hqx, pj.x, pj.y, L1); ic[c] = (int)lidx[hs + 64 * c + lane]; }
                }
#pragma unroll
                for (int c2 = 0; c2 < NM; ++c2) {
                    unsigned long long km = ((unsigned long long)hw[2 * c2 + 1] << 32) | hw[2 * c2];
                    while (km != 0ull) {
                        const int kb = __ffsll((long long)km) - 1;
                        km &= km - 1ull;
                        const int k = (kb & ~31) + 4 * (kb & 7) + ((kb & 31) >> 3);      // slot - 64 c2 of the bit
                        const float kd = lane_f(dc[c2], k); const int ki = lane_i(ic[c2], k);
#pragma unroll
                        for (int c = 0; c < NM; ++c) rc[c] += ((kd < dc[c]) | ((kd == dc[c]) & (ki < ic[c]))) ? 1 : 0;
                    }
                }
                float cy_ = 0.f, cx_ = 0.f, cw_ = 0.f, cny = 0.f, cnx = 0.f, cdk = 0.f;
                int cik = -1;
                const bool hnext = NEXT && (t < p.nb - 1);
#pragma unroll
                for (int c = 0; c < NM; ++c) {
                    if (bb[c] && rc[c] < hneed) {
                        const float2 f = lflow[hs + 64 * c + lane];
                        if (IWD) { const float wgt = ks_iwd_weight(dc[c]); cy_ += wgt * f.x; cx_ += wgt * f.y; cw_ += wgt; } else { cy_ += f.x; cx_ += f.y; }
                        if (hnext) { const float2 g2 = lnext[hs + 64 * c + lane]; cny += g2.x; cnx += g2.y; }
                        if (rc[c] == hneed - 1) { cdk = dc[c]; cik = ic[c]; }
                    }
                }
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) {
                    cy_ += __shfl_xor(cy_, o2, 64); cx_ += __shfl_xor(cx_, o2, 64);
                    if (IWD) cw_ += __shfl_xor(cw_, o2, 64);
                    if (NEXT) { cny += __shfl_xor(cny, o2, 64); cnx += __shfl_xor(cnx, o2, 64); }
                    cdk = fmaxf(cdk, __shfl_xor(cdk, o2, 64)); cik = max(cik, __shfl_xor(cik, o2, 64));
                }
                bool mytie = false;
#pragma unroll
                for (int c = 0; c < NM; ++c) mytie = mytie || (bb[c] && dc[c] == cdk && ic[c] > cik);
                const bool htie = __ballot(mytie) != 0ull;
                if (lane == h) {
                    if (IWD) { sy_ = fmaf(128.f, cy_, sy_); sx_ = fmaf(128.f, cx_, sx_); sw_ = fmaf(128.f, cw_, sw_); }
                    else { sy_ = fmaf(128.f, cy_, sy_); sx_ = fmaf(128.f, cx_, sx_); }
                    ny_ = fmaf(128.f, cny, ny_); nx_ = fmaf(128.f, cnx, nx_);
                    dK = cdk; iK = cik; tie = htie;
                }
            }
        }
        // ---- outputs ---------------------------------------------------------------------------------------
        if (live) {
            const size_t BQ = (size_t)p.B * p.nb * p.G;
            float2 ov; float norm = 0.f;
            if (IWD) { ov.x = sy_ / sw_; ov.y = sx_ / sw_; norm = sw_ * 0.0078125f; }      // (the sums carry the factor 128 of pass 2; the normaliser is saved without it)
            // mean = sum / K (focus.py:166) with the factor 128 of pass 2 taken out first (exact).  K a power of two (the shipped
            // 32): the division is a multiplication by an exact reciprocal -- same bits, a tenth of the instructions
            const bool kpow2 = (p.K & (p.K - 1)) == 0;
            const float rK = __int_as_float((120 - (__ffs(p.K) - 1)) << 23);        // 2^-(7 + log2 K), built from its exponent (used if kpow2)
            if (IWD) { }
            else if (kpow2) { ov.x = sy_ * rK; ov.y = sx_ * rK; }
            else { ov.x = (sy_ * 0.0078125f) / (float)p.K; ov.y = (sx_ * 0.0078125f) / (float)p.K; }
            reinterpret_cast<float2 *>(flow_lut)[MPC_IDX(q, (long long)p.B * p.nb * p.G)] = ov;
            if (do_next) {
                float2 on;
                if (kpow2) { on.x = ny_ * rK; on.y = nx_ * rK; }
                else { on.x = (ny_ * 0.0078125f) / (float)p.K; on.y = (nx_ * 0.0078125f) / (float)p.K; }
                reinterpret_cast<float2 *>(flow_next)[((size_t)(b * (p.nb - 1) + t)) * p.G + (size_t)cy * p.wq + cx] = on;
            }
            knn_state[MPC_IDX(q, BQ)] = dK;
            reinterpret_cast<int *>(knn_state)[BQ + MPC_IDX(q, BQ)] = iK | (tie ? KNN_TIE_FLAG : 0) | ((FARK && ls.far != nullptr && knn_is_far_dk(p, dK, r_init)) ? KNN_FAR_FLAG : 0);
            KS_INBIN_STAT(knn_state, BQ, q, inbin, nsl, IWD, norm);         // (the 'mean' backward never reads the normaliser)
        }
        {   // queries for the fallback kernel: one atomic per wavefront reserves their places in the list
            // (main launch: a query with too few candidates below the ring bound or too many slots gets a second chance in
            // k_knn_tail -- more rings, 128 slots, only such queries staged: its bit in the `again` map -- if the
            // strip has more than KS_MORE_MIN of them or far queries; the odd one: the fallback list)
            const bool late = MODE == 0 && inpass && !live && why < 2u;
            int nlate = 0;
            if (MODE == 0) nlate = __syncthreads_count(late ? 1 : 0);
            // ... or the strip goes there anyway for its far queries
            const bool to_more = MODE == 0 && (nlate > KS_MORE_MIN || anyfar);
            if (late && to_more) {
                reinterpret_cast<int *>(knn_state)[MPC_IDX(q, (long long)p.B * p.nb * p.G)] = r;          // (the radius that was tried, for the second launch)
                atomicOr(ls.again + MPC_IDX(aoff, (long long)p.B * p.nb * ls.again_words), 1u << (cx & 31));
                if (why == 0u) atomicOr(ls.grow + MPC_IDX(aoff, (long long)p.B * p.nb * ls.again_words), 1u << (cx & 31));
            }

            // (and the number of queries marked for it: with only a handful in the whole launch it hands them on to the fallback kernel)
            const int nfq = (MODE == 0 && anyfar) ? __syncthreads_count(isfar ? 1 : 0) : 0;
            if (to_more && tid == 0) ls.farstrip[MPC_IDX(1 + atomicAdd(&ls.farstrip[0], 1), 1 + (long long)gx * gy * p.B * p.nb)] = lblk;
            // (the far pass gives a query that held too few candidates two more rings, one with too many slots the same radius, a far
            // query the radius the bisection above found)
            if (to_more) mark_list((late && to_more) || (MODE == 0 && isfar), q, isfar ? r : (why == 0u ? min(r + 2, KNN_RFAR) : r), nlate + nfq);      // (workgroup-uniform)
            const bool push = inpass && !live && !(late && to_more);
            const unsigned long long pm = __ballot(push);
            if (pm != 0ull) {
                const int lane = tid & 63, first = __ffsll((long long)pm) - 1;
                int base = 0;
                const long long fcap = (long long)p.B * p.nb * p.G;
                // (late list: behind the marked list where the fallback workgroups take that one -- few marked queries in the launch)
                const int late_base = (MODE != 0 && *knn_marked_count(ls) <= KS_FORWARD_MAX) ? (int)min((long long)*knn_marked_count(ls), fcap) : 0;
                if (lane == first) base = atomicAdd(MODE == 0 ? &fail[0] : knn_late_count(ls), __popcll(pm));
                base = __shfl(base, first, 64);
                // (entry: the query, why in bits 30..31 and -- where the query ids leave room: fewer than 2^24 queries -- the radius
                // that was tried in bits 24..29, so that the fallback kernel need not read it off the summed-area table again)
                const unsigned hint = ((size_t)p.B * p.nb * p.G < (1u << 24) && served) ? (unsigned)min(r, 63) << 24 : 0u;
                const int k = base + __popcll(pm & ((1ull << lane) - 1ull));
                if (push) fail[MPC_IDX(MODE == 0 ? 1 + (long long)k : fcap - (long long)(late_base + k), 1 + fcap)] = (int)((unsigned)q | hint | (why << 30));
                if (MODE != 0 && lane == first) s_wsum[KS_NT / 64] = 1;      // (late list: this workgroup pushed; one fence before it counts itself done, k_knn_tail)
            }
        }
        // largest K-th distance per 16x16 tile of the bucket grid and class of query (bounds the search windows of the gather
        // backward): a wavefront covers 64 / WS consecutive rows of one tile column, i.e. 64 / (16 WS) tiles of 16 WS lanes each
        // (the tiles of the query grid).  Only wavefronts next to the image border hold anything but class 0 (knn_device.h).
        if (!FARK) {
            static_assert(FARK || WS == 2, "32 lanes = one tile");
            const int bd = knn_band_depth(r_init);
            const unsigned cls = live ? knn_query_classes(p, cy, cx, bd) : 0u;
            const bool plain = __ballot(cls > 1u) == 0ull;
            const int gx16 = knn_tiles_x(p.wq, p.m), gy16 = knn_tiles_y(p.hq, p.m);
            int *dst = reinterpret_cast<int *>(tile_dkmax) + (((size_t)bt * gy16 + (min(cy, p.hq - 1) >> 4)) * gx16 + (qx0 >> 4)) * KNN_NCLS;
            const bool writer = (tid & (16 * WS - 1)) == 0 && cy <= qy1;
#pragma unroll
            for (int c = 0; c < KNN_NCLS; ++c) {
                if (c > 0 && plain) break;
                float m = ((cls >> c) & 1u) ? dK : 0.f;
#pragma unroll
                for (int o2 = 8 * WS; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
                if (writer && m > 0.f) atomicMax(dst + c, __float_as_int(m));
            }
        } else if (ls.far != nullptr) {
            // a served query of the second launch with a K-th distance the gather should not carry: onto the far list of its
            // (sample, bin) (one atomic per wavefront) and its tiles onto the work list of k_knn_bwd_far; the others into the
            // tile maxima like any query
            const bool isf = live && knn_is_far_dk(p, dK, r_init);
            if (live && !isf) knn_tile_max_add(tile_dkmax, p, bt, cy, cx, knn_band_depth(r_init), dK);
            const unsigned long long pm = __ballot(isf);
            if (pm != 0ull) {
                const int lane = tid & 63, first = __ffsll((long long)pm) - 1;
                int *fl = ls.far + (size_t)bt * (p.G + 1);
                int base = 0;
                if (lane == first) base = atomicAdd(&fl[0], __popcll(pm));
                base = __shfl(base, first, 64);
                if (isf) {
                    fl[MPC_IDX(1 + base + __popcll(pm & ((1ull << lane) - 1ull)), p.G + 1)] = cy * p.wq + cx;
                    // its tiles onto the work list of k_knn_bwd_far: through the workgroup's bit map in LDS where the (sample,
                    // bin)'s tiles fit it (a per-query chain of a load and an atomic per tile was a third of a work item's time)
                    if (ls.ftwords <= KNN_FT_LDS_WORDS) knn_far_mark_tiles_lds(p, s_ft, cy, cx, dK);
                    else knn_far_mark_tiles(p, ls, bt, cy, cx, dK);
                }
            }
            if (ls.ftwords <= KNN_FT_LDS_WORDS) {           // (workgroup-uniform)
                KS_SYNC();
                if (tid < ls.ftwords) { knn_far_flush_tiles(p, ls, bt, s_ft, tid); s_ft[tid] = 0u; }
            }
        } else if (live) knn_tile_max_add(tile_dkmax, p, bt, cy, cx, knn_band_depth(r_init), dK);      // (far query, general gather backward)
        KS_STP_WRITE(ls, tid, mine, total);      // 5: searched, lists written
    }
#undef KS_SYNC
}

// 1-D grid of gx * gy * B * nb workgroups (gx strips, gy row blocks) in XCD-contiguous order, 256 threads,
// dynamic LDS sized by the launcher
template <int WS, bool L1, bool NEXT, bool IWD>
__global__ __launch_bounds__(KS_NT, (NEXT || L1) ? KS_OCC_WIDE : 6) void k_knn_strip(const KnnParams p, const float *__restrict__ traj,
                                                     const knn_cs_t *__restrict__ cell_start, const knn_cs_t *__restrict__ sat,
                                                     const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                                     float *__restrict__ flow_lut, float *__restrict__ flow_next,
                                                     float *__restrict__ knn_state, float *__restrict__ tile_dkmax,
                                                     const KnnLists ls, int r_init, int cap, int gx, int gy,
                                                     const EvCountArgs evc, int n_evc, int evc_stride) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_wsum[KS_NT / 64 + 1], s_wmax[KS_NT / 64];      // (s_wsum[KS_NT / 64]: "pushed onto the late list", tail kernel only)
    __shared__ unsigned char s_rq[KS_NT / WS];          // radius of the widest square of every query row of the strip (0: none)
    // mpc_focus_fwd: some workgroups do not search -- they count the event rows per backward bucket for the event kernels
    // that follow (ev_count_device.h).  This kernel is bound by vector-instruction issue and leaves HBM idle, so the 67 MB of
    // C3's events are read beside it.  The counting workgroups come in groups of 8 (one per XCD: the search workgroups keep
    // their XCD mapping), a group every `evc_stride` workgroups -- thinly spread: all at the front they held 45 % of the
    // workgroup slots for the ~25 us a bandwidth-bound start takes (+12 us); spread they hold ~4 % at any time.
    int pblk = (int)blockIdx.x;
    if (n_evc > 0) {
        const int grp = pblk / evc_stride, in_grp = pblk - grp * evc_stride, ngrp = n_evc >> 3;
        if (grp < ngrp && in_grp < 8) {
            const int cb = grp * 8 + in_grp;
            if (cb < ev_count_blocks(evc)) ev_count_block(evc, cb, reinterpret_cast<int *>(s_dyn));
            return;
        }
        pblk -= 8 * min(grp + 1, ngrp);
    }
    const int nblk = gx * gy * p.B * p.nb;
    const int lblk = (pblk & 7) * ((nblk + 7) >> 3) + (pblk >> 3);
    if (lblk >= nblk) return;
    strip_body<WS, L1, NEXT, IWD, 0>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls,
                                     r_init, cap, gx, gy, lblk, 0, KS_NT / WS, s_dyn, s_wsum, s_wmax, s_rq);
}

// strip workgroups of the tail kernel that have anything to do (the others neither work nor count themselves done): workgroup wg
// takes the retry items wg, wg + nwg, ... and the far items likewise
// few marked queries in the whole launch: no far pass, the fallback workgroups take the marked list
__device__ __forceinline__ bool strip_forward(const KnnLists &ls) { return *knn_marked_count(ls) <= KS_FORWARD_MAX; }
__device__ __forceinline__ int strip_more_busy(const KnnParams &p, const KnnLists &ls, int gx, int gy, int nwg) {
    const int nstrips = gx * gy * p.B * p.nb;
    const int gxf = knn_far_items_x(p.wq), gyf = knn_far_items_y(p.hq);
    const int nretry = 4 * min(ls.retry[0], nstrips), nfar = strip_forward(ls) ? 0 : min(ls.farstrip[0], gxf * gyf * p.B * p.nb);
    return min(nwg, max(nretry, nfar));
}

// Strip workgroups of the tail kernel (k_knn_tail): the strips on the retry list (a quarter of the query rows per workgroup and
// round), then the strips on the farstrip list (their far queries and the queries the main launch could not finish).  `nwg` of them,
// this one is number `wg` (the list lengths are only known on the device; nothing to do for the lattice-like point sets of the
// benchmark: both lists empty or nearly)
template <int WS, bool L1, bool NEXT, bool IWD>
__device__ __forceinline__ void strip_more_body(const KnnParams &p, const float *__restrict__ traj,
                                                const knn_cs_t *__restrict__ cell_start, const knn_cs_t *__restrict__ sat,
                                                const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                                float *__restrict__ flow_lut, float *__restrict__ flow_next,
                                                float *__restrict__ knn_state, float *__restrict__ tile_dkmax,
                                                const KnnLists &ls, int r_init, int cap, int gx, int gy, int wg, int nwg,
                                                unsigned char *s_dyn, int *s_wsum, int *s_wmax, unsigned char *s_rq) {
    constexpr int TH = KS_NT / WS;
    const int nstrips = gx * gy * p.B * p.nb;
    const int gxf = knn_far_items_x(p.wq), gyf = knn_far_items_y(p.hq);
    // (only a few marked queries in the whole launch -- a B = 1 step with a dozen of them in three strips paid a full work item's
    // 25 us for them: the fallback workgroups take them from the MARKED list beside what they have anyway, no far pass.  As a zero
    // trip count read before both loops: an early return between the loops cost the far pass 25 % -- 14 spilled registers instead
    // of 8, reloaded inside its search)
    const int nretry = 4 * min(ls.retry[0], nstrips), nfar = strip_forward(ls) ? 0 : min(ls.farstrip[0], gxf * gyf * p.B * p.nb);
    for (int w = wg; w < nretry; w += nwg) {
        const int quarter = w & 3;
        strip_body<WS, L1, NEXT, IWD, 1>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls,
                                         r_init, cap, gx, gy, ls.retry[1 + (w >> 2)], quarter * (TH / 4), (quarter + 1) * (TH / 4), s_dyn, s_wsum, s_wmax, s_rq);
        __syncthreads();
    }
    for (int w = wg; w < nfar; w += nwg) {
        strip_body<KNN_FAR_WS, L1, NEXT, IWD, 2>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls,
                                                 r_init, cap, gxf, gyf, ls.farstrip[1 + w], 0, KNN_FAR_TH, s_dyn, s_wsum, s_wmax, s_rq);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The queries the strip kernel could not serve: ONE WAVEFRONT per query -- a thread per query spends ~100 us in dependent global
// loads (measured in round 4: 5-7 ns per query against ~3.5).  The lanes take the rows of the query's search region, then its
// candidates (up to 64 * KS_FB_SLOTS of them; more: rows and candidates in rounds of 64, those below the ring bound compacted in
// LDS), and the K-th key is found by a bitwise search with ballots.  The search starts from the radius the strip kernel tried (hint in
// the list entry) or from one read off the summed-area table, and looks only at the cells of each row that can hold a point below
// the ring bound (a band query's neighbours lie in a thin segment of a large disc).
// With `far` (the backward is the tile gather) every query served here goes on the far list of its (sample, bin), is flagged
// KNN_FAR_FLAG and stays out of the tile maxima: k_knn_bwd_far computes its gradient (knn.hip).
// grid: KS_FB_BLOCKS workgroups (the list length is only known on the device), 256 threads
// ------------------------------------------------------------------------------------------
// first radius of a fallback search from the summed-area table: up to KNN_RCAP the smallest square with 1.25 x the strip
// kernel's count (which it has tried); beyond it the far queries' rule, found by doubling and bisection (a band 20 rings deep
// costs 8 probes of four loads, not 20)
__device__ __forceinline__ int fallback_radius(const KnnParams &p, const knn_cs_t *__restrict__ sat, int cy, int cx, int r0) {
    const int need = knn_square_need(p.K, p.l1) + (knn_square_need(p.K, p.l1) >> 2), need_far = knn_square_need_far(p.K, p.l1);
    const int rmax = max(p.hb, p.wb);
    int r = max(r0, 2);
    while (r <= KNN_RCAP && knn_square_count(p, sat, cy, cx, r) < need) ++r;
    if (r <= KNN_RCAP) return r;
    int lo = KNN_RCAP, hi = KNN_RCAP + 2;
    while (hi < rmax && knn_square_count(p, sat, cy, cx, hi) < need_far) { lo = hi; hi += max(2, hi >> 1); }
    hi = min(hi, rmax);
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (knn_square_count(p, sat, cy, cx, mid) >= need_far) hi = mid; else lo = mid;
    }
    return hi;
}

// one served far query: onto the far list of its (sample, bin), its tiles onto the work list of k_knn_bwd_far.  Called by ALL
// lanes of the wavefront that served it: lane 0 appends the query, every lane takes one of the tiles its disc can touch (their
// loads and atomics travel together -- one lane walking them was a chain of two round trips per tile behind every query)
__device__ __forceinline__ void far_list_add(const KnnParams &p, const KnnLists &ls, int bt, int cy, int cx, float dK, int lane) {
    if (lane == 0) {
        int *fl = ls.far + (size_t)bt * (p.G + 1);
        const int k = atomicAdd(&fl[0], 1);
        fl[MPC_IDX(1 + k, p.G + 1)] = cy * p.wq + cx;
    }
    int ta, tb, tc, td;
    knn_far_tile_range(p, cy, cx, dK, ta, tb, tc, td);
    const int ntx = knn_tiles_x(p.wq, p.m), nty = knn_tiles_y(p.hq, p.m), nx = td - tc + 1, cnt = (tb - ta + 1) * nx;
    for (int i = lane; i < cnt; i += 64) {
        const int ty = ta + i / nx, tx = tc + i % nx, tile = ty * ntx + tx;
        const unsigned bit = 1u << (tile & 31);
        unsigned *w = ls.ftbits + (size_t)bt * ls.ftwords + (tile >> 5);
        if ((*w & bit) != 0u) continue;                       // (set already: the usual case inside a band)
        if ((atomicOr(w, bit) & bit) == 0u) ls.ftlist[1 + atomicAdd(&ls.ftlist[0], 1)] = bt * ntx * nty + tile;
    }
}

// the K-th smallest of the wavefront's keys (KS_FB_SLOTS per lane, the first `ns` in use; (distance bits << 32) | index, all
// distinct; an empty slot holds (inf, 0x7fffffff)) by a bitwise search from the top: distances are non-negative floats (their
// bit patterns order like unsigned integers), indices < 2^16.  Bit by bit: the smallest V with count(key <= V) >= K.  47
// wave-uniform steps of one 64-bit compare per slot.
__device__ __forceinline__ unsigned long long fb_kth_key(const unsigned long long (&key)[KS_FB_SLOTS], int ns, int K) {
    unsigned long long V = 0ull;
    for (int bit = 62; bit >= 0; --bit) {
        if (bit == 31) bit = 15;                                // index bits 16..31 are zero for every candidate
        const unsigned long long cand = V | ((1ull << bit) - 1ull);
        int c = 0;
#pragma unroll
        for (int m = 0; m < KS_FB_SLOTS; ++m)
            if (m < ns) c += __popcll(__ballot(key[m] <= cand));
        if (c < K) V |= 1ull << bit;
    }
    return V;
}

template <bool L1>
__device__ void fallback_one_query(const KnnParams &p, const float *__restrict__ traj, const knn_cs_t *__restrict__ cell_start,
                                   const knn_cs_t *__restrict__ sat, const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                   float *__restrict__ flow_lut, float *__restrict__ flow_next,
                                   float *__restrict__ knn_state, float *__restrict__ tile_dkmax, const KnnLists &ls, int q, int r_init, int r_start,
                                   float4 (*s_comp)[256]) {
    int *const far = ls.far;
    const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
    const int bt = q / p.G, cell = q - bt * p.G;
    const int cy = cell / p.wq, cx = cell - cy * p.wq;
    const int b = bt / p.nb, t = bt - b * p.nb;
    const knn_cs_t *cs = cell_start + (size_t)bt * (p.Gb + 1);
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const knn_idx_t *si_ = sidx + (size_t)bt * p.n;
    const float2 *traj_b = reinterpret_cast<const float2 *>(traj) + (size_t)b * (p.T + p.nb) * p.n;
    const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
    const int ylo = -p.m, yhi = p.hq + p.m - 1, xlo = -p.m, xhi = p.wq + p.m - 1;
    int r = r_start > 0 ? r_start : fallback_radius(p, sat + (size_t)bt * (p.hb + 1) * (p.wb + 1), cy, cx, r_init);
    float dd[KS_FB_SLOTS]; int ii[KS_FB_SLOTS];
    float2 pq[KS_FB_SLOTS];
    int ns = KS_FB_SLOTS;                  // slots per lane actually in use (wave-uniform): ceil(candidates / 64)
    for (;;) {
        const int y0 = max(cy - r, ylo), y1 = min(cy + r, yhi), x0 = max(cx - r, xlo), x1 = min(cx + r, xhi);
        const bool whole = (y0 == ylo && x0 == xlo && y1 == yhi && x1 == xhi);
        const int nrows = y1 - y0 + 1;
        const float lb = ((float)r + 0.5f) * (float)p.sp - KNN_SLACK;
        const float upper = whole ? INFINITY : (L1 ? lb : lb * lb);
        // bucketed range of cell row y -- only the cells that can hold a point below the ring bound
        auto row_range = [&](int y, int &js, int &ln) {
            int xa = x0, xb = x1;
            if (!whole) {
                const float dyc = fmaxf((float)abs(y - cy) - 0.5f, 0.f) * (float)p.sp;
                const float w2 = L1 ? upper - dyc : upper - dyc * dyc;
                const int xr = w2 > 0.f ? (int)((L1 ? w2 : sqrtf(w2)) / (float)p.sp + 0.5f) + 1 : -1;
                xa = max(xa, cx - xr); xb = min(xb, cx + xr);
            }
            js = 0; ln = 0;
            if (xa <= xb) { js = cs[MPC_IDX(knn_ci(p, y, xa), p.Gb + 1)]; ln = cs[MPC_IDX(knn_ci(p, y, xb + 1), p.Gb + 1)] - js; }
        };
        // lane l < nrows: the range of row y0 + l; exclusive scan over the lanes -> flat candidate numbering
        int js = 0, ln = 0;
        if (lane < min(nrows, 64)) row_range(y0 + lane, js, ln);
        int incl = ln;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(incl, o2, 64); if (lane >= o2) incl += v; }
        const int N = __shfl(incl, 63, 64);
        int cnt = 0;
        if (nrows <= 64 && N <= 64 * KS_FB_SLOTS) {
            ns = (N + 63) >> 6;
            const int excl = incl - ln;
#pragma unroll
            for (int m = 0; m < KS_FB_SLOTS; ++m) {
                const int k = lane + 64 * m;                    // flat candidate number of this lane's m-th slot
                dd[m] = INFINITY; ii[m] = 0x7fffffff; pq[m] = make_float2(0.f, 0.f);
                if (m >= ns) continue;
                // row of candidate k: the last lane whose exclusive offset is <= k (offsets are non-decreasing)
                int lo = 0, hi = 64;
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const int mid = (lo + hi) >> 1;
                    const int ev = __shfl(excl, mid, 64);
                    if (ev <= k) lo = mid; else hi = mid;
                }
                const int rjs = __shfl(js, lo, 64), rex = __shfl(excl, lo, 64);
                if (k < N) {
                    const int g = rjs + (k - rex);
                    const float2 pj = sp_[MPC_IDX(g, p.n)];
                    const int id = si_[MPC_IDX(g, p.n)];                          // (with the position: one round trip, not two)
                    const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                    if (d < upper) { dd[m] = d; ii[m] = id; pq[m] = pj; ++cnt; }
                }
            }
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) cnt += __shfl_xor(cnt, o2, 64);
        } else {
            // more rows than lanes (a band deeper than 30 rings) or more candidates than the lanes' slots (the outermost ring of
            // the margin in reach: everything that left the image lies there, nearly all of it beyond the ring bound): rows in
            // rounds of 64, candidates in rounds of 64, those below the bound compacted into LDS -- up to 256 of them.  MORE than
            // that below the bound (round 5; a query deep inside a band a contracting field emptied: one more ring of a large square
            // brings in a long stretch of the dense front): the K smallest of the 256 at hand are kept -- their K-th key is a bound
            // no neighbour exceeds -- and from there on only candidates below THAT key are admitted; exact for any number of
            // candidates.  (Rounds 1-4 handed such a query to one lane's thread-serial search: ~170 us per query, 7 ms of a C3 step on
            // a 45 % contraction.)
            float tauD = upper; int tauI = 0;                   // admitted: d < tauD, or d == tauD and index < tauI
            for (int rb = 0; rb < nrows; rb += 64) {
                if (rb > 0) { js = 0; ln = 0; if (rb + lane < nrows) row_range(y0 + rb + lane, js, ln); }
                int inc2 = ln;
#pragma unroll
                for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc2, o2, 64); if (lane >= o2) inc2 += v; }
                const int Nr = __shfl(inc2, 63, 64), exc2 = inc2 - ln;
                for (int k0 = 0; k0 < Nr; k0 += 64) {
                    const int k = k0 + lane;
                    int lo = 0, hi = 64;
#pragma unroll
                    for (int st = 0; st < 6; ++st) {
                        const int mid = (lo + hi) >> 1;
                        const int ev = __shfl(exc2, mid, 64);
                        if (ev <= k) lo = mid; else hi = mid;
                    }
                    const int rjs = __shfl(js, lo, 64), rex = __shfl(exc2, lo, 64);
                    bool have = false;
                    float d = 0.f; int id = 0; float2 pj = make_float2(0.f, 0.f);
                    if (k < Nr) {
                        const int g = rjs + (k - rex);
                        pj = sp_[MPC_IDX(g, p.n)]; id = si_[MPC_IDX(g, p.n)];
                        d = pair_dist(qy, qx, pj.x, pj.y, L1);
                        have = true;
                    }
                    bool keep = have && (d < tauD || (d == tauD && id < tauI));
                    unsigned long long km = __ballot(keep);
                    if (cnt + __popcll(km) > 64 * KS_FB_SLOTS) {            // (wave-uniform) no room: keep the K smallest of what is there
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        unsigned long long kk[KS_FB_SLOTS];
#pragma unroll
                        for (int m = 0; m < KS_FB_SLOTS; ++m) {
                            kk[m] = 0x7f8000007fffffffull;
                            if (lane + 64 * m < cnt) {
                                const float4 e = s_comp[wvi][lane + 64 * m];
                                kk[m] = ((unsigned long long)__float_as_uint(e.x) << 32) | (unsigned)__float_as_int(e.y);
                            }
                        }
                        const unsigned long long V = fb_kth_key(kk, KS_FB_SLOTS, p.K);
                        // the survivors move to the front, 64 entries at a time: a round's destinations lie at or before its own
                        // sources (read into registers, fenced) and behind nothing that is still to be read
                        int nk = 0;
#pragma unroll
                        for (int m = 0; m < KS_FB_SLOTS; ++m) {
                            const bool sv = kk[m] <= V;
                            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (sv) e = s_comp[wvi][lane + 64 * m];
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            const unsigned long long sm = __ballot(sv);
                            if (sv) s_comp[wvi][MPC_IDX(nk + __popcll(sm & ((1ull << lane) - 1ull)), 64 * KS_FB_SLOTS)] = e;
                            nk += __popcll(sm);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        }
                        cnt = nk;                                           // (= K: the keys are distinct)
                        tauD = __uint_as_float((unsigned)(V >> 32)); tauI = (int)(unsigned)(V & 0xffffffffull);
                        keep = have && (d < tauD || (d == tauD && id < tauI));
                        km = __ballot(keep);
                    }
                    const int slot = cnt + __popcll(km & ((1ull << lane) - 1ull));
                    if (keep) s_comp[wvi][MPC_IDX(slot, 64 * KS_FB_SLOTS)] = make_float4(d, __int_as_float(id), pj.x, pj.y);
                    cnt += __popcll(km);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (the other lanes' LDS writes, before the reads below)
            ns = (cnt + 63) >> 6;
#pragma unroll
            for (int m = 0; m < KS_FB_SLOTS; ++m) {
                dd[m] = INFINITY; ii[m] = 0x7fffffff; pq[m] = make_float2(0.f, 0.f);
                if (lane + 64 * m < cnt) {
                    const float4 e = s_comp[wvi][lane + 64 * m];
                    dd[m] = e.x; ii[m] = __float_as_int(e.y); pq[m] = make_float2(e.z, e.w);
                }
            }
        }
        if (cnt >= p.K || whole) break;
        r += 1 + (r >> 3);
    }
    // the K-th smallest key (distance bits, index): fb_kth_key
    unsigned long long key[KS_FB_SLOTS];
#pragma unroll
    for (int m = 0; m < KS_FB_SLOTS; ++m) key[m] = ((unsigned long long)__float_as_uint(dd[m]) << 32) | (unsigned)ii[m];
    const unsigned long long V = fb_kth_key(key, ns, p.K);
    // neighbours: rank < K (indices are distinct, so ranks are); sums in lane order
    const bool do_next = p.want_next && (t < p.nb - 1);
    float sy_ = 0.f, sx_ = 0.f, sw_ = 0.f, ny_ = 0.f, nx_ = 0.f;
    const float dK = __uint_as_float((unsigned)(V >> 32));
    const int iK = (int)(unsigned)(V & 0xffffffffull);
#pragma unroll
    for (int m = 0; m < KS_FB_SLOTS; ++m) {
        if (dd[m] < INFINITY && key[m] <= V) {
            const float2 pj = pq[m];
            const float2 a = traj_b[MPC_IDX(ii[m], p.n)];                                   // T == 1
            const float fy = a.x - pj.x, fx = a.y - pj.y;
            if (p.iwd) { const float wgt = 1.f / (dd[m] + 1e-9f); sy_ += wgt * fy; sx_ += wgt * fx; sw_ += wgt; }
            else { sy_ += fy; sx_ += fx; }
            if (do_next) { const float2 c = traj_b[(size_t)(p.T + t + 1) * p.n + ii[m]]; ny_ += c.x - pj.x; nx_ += c.y - pj.y; }
        }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        sy_ += __shfl_xor(sy_, o2, 64); sx_ += __shfl_xor(sx_, o2, 64); sw_ += __shfl_xor(sw_, o2, 64);
        ny_ += __shfl_xor(ny_, o2, 64); nx_ += __shfl_xor(nx_, o2, 64);
    }
    bool tie = false;
#pragma unroll
    for (int m = 0; m < KS_FB_SLOTS; ++m) tie = tie || (dd[m] == dK && dd[m] < INFINITY && ii[m] > iK);
    tie = __ballot(tie) != 0ull;
    if (lane == 0) {
        const size_t BQ = (size_t)p.B * p.nb * p.G;
        float2 ov; float norm = 0.f;
        if (p.iwd) { ov.x = sy_ / sw_; ov.y = sx_ / sw_; norm = sw_; }
        else { ov.x = sy_ / (float)p.K; ov.y = sx_ / (float)p.K; }
        reinterpret_cast<float2 *>(flow_lut)[q] = ov;
        if (do_next) {
            float2 on; on.x = ny_ / (float)p.K; on.y = nx_ / (float)p.K;
            reinterpret_cast<float2 *>(flow_next)[((size_t)(b * (p.nb - 1) + t)) * p.G + (size_t)cy * p.wq + cx] = on;
        }
        const bool isf = far != nullptr && knn_is_far_dk(p, dK, r_init);
        knn_state[q] = dK;
        reinterpret_cast<int *>(knn_state)[BQ + q] = iK | (tie ? KNN_TIE_FLAG : 0) | (isf ? KNN_FAR_FLAG : 0);
        knn_state[2 * BQ + q] = norm;
        if (!isf) knn_tile_max_add(tile_dkmax, p, bt, cy, cx, knn_band_depth(r_init), dK);
    }
    if (far != nullptr && knn_is_far_dk(p, dK, r_init)) far_list_add(p, ls, bt, cy, cx, dK, lane);      // (dK is wave-uniform)
}

// One entry of the fail / late list: the query and where its search starts
template <bool L1>
__device__ __forceinline__ void fallback_entry(const KnnParams &p, const float *__restrict__ traj, const knn_cs_t *__restrict__ cell_start,
                                               const knn_cs_t *__restrict__ sat, const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                               float *__restrict__ flow_lut, float *__restrict__ flow_next, float *__restrict__ knn_state,
                                               float *__restrict__ tile_dkmax, const KnnLists &ls, unsigned ent, int nq, int r_init, float4 (*s_comp)[256]) {
    const bool hinted = (size_t)nq < (1u << 24);
    const int q = (int)(ent & (hinted ? 0x00ffffffu : 0x3fffffffu));
    // radius to start from: the one the strip kernel tried (one more ring if it held too few candidates), else from the table
    const int why = (int)(ent >> 30), rh = hinted ? (int)((ent >> 24) & 63u) : 0;
    const int r_start = rh > 0 ? rh + (why == 0 ? 1 : 0) : 0;
    fallback_one_query<L1>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls, q, r_init, r_start, s_comp);
}

// ------------------------------------------------------------------------------------------
// The TAIL of the KNN forward, one launch (round 5; rounds 2-4: k_knn_strip_more, then k_knn_fallback): everything the main launch
// of the strip kernel left over.
//   workgroups [0, KS_RETRY_BLOCKS): the strip work (strip_more_body) -- overflowed strips in quarters, the far queries; what they
//     cannot finish goes on the LATE list; when a workgroup is through, it counts itself done;
//   workgroups [KS_RETRY_BLOCKS, + KS_FB_BLOCKS): one wavefront per query of the main launch's `fail` list -- at once, BESIDE the strip
//     workgroups: both halves are chains of dependent round trips at low occupancy (a work item of the far pass ~25 us on one
//     wavefront of four, a fallback query ~14 us), side by side they cost what the longer one costs -- then, once every strip
//     workgroup is done, the late list.
//     FORWARD PROGRESS of that wait (round 6): the strip workgroups wait for nobody, so they finish as soon as they are resident; a
//     waiting fallback workgroup holds a slot of the chip while it spins.  The launcher therefore starts FEWER fallback workgroups
//     than the chip can hold workgroups of this kernel at once (tail_fallback_blocks: occupancy query x CU count minus 32 slots -- 992
//     of 1 024 on an MI355X), so a strip workgroup that has not started always finds a free slot whatever order the hardware
//     dispatches in.  Round 5 relied on dispatch in blockIdx order, which HIP does not promise; the late list in a launch
//     of its own was measured too: +9 us on the white-noise step (profiles/HISTORY_r06.md section 2).
// One launch instead of two for the lattice-like point sets of the benchmark, whose lists are (nearly) empty.
// grid: KS_RETRY_BLOCKS + KS_FB_BLOCKS workgroups (the list lengths are only known on the device), 256 threads, dynamic LDS of the
// far pass (the fallback workgroups use its first 16 KB)
// ------------------------------------------------------------------------------------------
template <int WS, bool L1, bool NEXT, bool IWD>
__global__ __launch_bounds__(KS_NT, KS_MORE_OCC) void k_knn_tail(const KnnParams p, const float *__restrict__ traj,
                                                     const knn_cs_t *__restrict__ cell_start, const knn_cs_t *__restrict__ sat,
                                                     const float2 *__restrict__ spos, const knn_idx_t *__restrict__ sidx,
                                                     float *__restrict__ flow_lut, float *__restrict__ flow_next,
                                                     float *__restrict__ knn_state, float *__restrict__ tile_dkmax,
                                                     const KnnLists ls, int r_init, int cap, int gx, int gy, const EvCountArgs evc) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_wsum[KS_NT / 64 + 1], s_wmax[KS_NT / 64];      // (s_wsum[KS_NT / 64]: this strip workgroup put queries on the late list)
    __shared__ unsigned char s_rq[KS_NT / WS];
    KT_DECL
    KT_T(0);
    if ((int)blockIdx.x < KS_RETRY_BLOCKS) {
        if (threadIdx.x == 0) s_wsum[KS_NT / 64] = 0;
        __syncthreads();
        strip_more_body<WS, L1, NEXT, IWD>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls, r_init, cap, gx, gy,
                                           (int)blockIdx.x, KS_RETRY_BLOCKS, s_dyn, s_wsum, s_wmax, s_rq);
        // (its late-list entries go out with one device-wide fence -- an L2 write-back on this chip: only a workgroup that pushed any
        // pays for it; 1 024 unconditional fences were 40 us of a 60 us launch, one per push 67 us of a 30 % contraction's 490 --
        // then its count; a workgroup without work does not count: 1 024 atomics on one word were 17 us of a B = 1 launch)
        __syncthreads();
        if (threadIdx.x == 0 && (int)blockIdx.x < strip_more_busy(p, ls, gx, gy, KS_RETRY_BLOCKS)) {
            if (s_wsum[KS_NT / 64]) __threadfence();
            atomicAdd(knn_tail_done(ls), 1);
        }
        KT_T(1); KT_T(3); KT_WRITE(ls);
        return;
    }
    const int fb = (int)blockIdx.x - KS_RETRY_BLOCKS;
    float4 (*s_comp)[256] = reinterpret_cast<float4 (*)[256]>(s_dyn);      // per wavefront: the candidates below the ring bound, compacted (fallback_one_query)
    // mpc_focus_fwd: the strip kernel before this one counted the event rows per backward bucket; the first B fallback workgroups
    // turn the counts of their sample into first records (the event kernels follow on the stream)
    if (evc.events != nullptr && fb < evc.B) {
        ev_prefix_block(evc, fb, s_wsum);
        __syncthreads();
    }
    const int *fail = ls.fail;
    const int nq = p.B * p.nb * p.G;
    const int wv = fb * 4 + (threadIdx.x >> 6), nw = ((int)gridDim.x - KS_RETRY_BLOCKS) * 4;
    const int nfail = min(fail[0], nq);
    // ... and the marked list, where the strip workgroups leave it to us (written by the main launch: complete): one round-robin over
    // both, so that with two short lists no wavefront takes an entry of each, one behind the other
    const int nmark = strip_forward(ls) ? min(*knn_marked_count(ls), nq - nfail) : 0;
    for (int i = wv; i < nfail + nmark; i += nw)
        fallback_entry<L1>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls,
                           (unsigned)fail[MPC_IDX(i < nfail ? 1 + i : nq - (i - nfail), 1 + (long long)nq)], nq, r_init, s_comp);
    // the late list: complete once every strip workgroup has counted itself done
    // (counter, list length and entries are read with device-scope atomic loads, which do not hit a stale line of this XCD's L2:
    // no acquire fence -- an L2 invalidate per wavefront)
    KT_T(1); KT_COUNT(4, (nfail + nmark - wv + nw - 1) / nw);
    const int busy = strip_more_busy(p, ls, gx, gy, KS_RETRY_BLOCKS);
    if (busy == 0) { KT_T(3); KT_WRITE(ls); return; }            // (no strip work at all: no late list either)
    if (threadIdx.x == 0)
        while (__hip_atomic_load(knn_tail_done(ls), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < busy) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    KT_T(2);
    const int nlate = min(__hip_atomic_load(knn_late_count(ls), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nq - nfail - nmark);
    for (int i = wv; i < nlate; i += nw)
        fallback_entry<L1>(p, traj, cell_start, sat, spos, sidx, flow_lut, flow_next, knn_state, tile_dkmax, ls,
                           (unsigned)__hip_atomic_load(&fail[MPC_IDX(nq - nmark - i, 1 + (long long)nq)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nq, r_init, s_comp);
    KT_COUNT(5, (nlate - wv + nw - 1) / nw); KT_T(3); KT_WRITE(ls);
}

// ------------------------------------------------------------------------------------------
// launcher (called by mpc_knn_lut_fwd once the points are bucketed; `fail[0]`, the far-list counters and tile_dkmax zeroed by
// the bucket kernel)
// ------------------------------------------------------------------------------------------
// staging capacity (slots) of a strip and the dynamic LDS of the main launch / of the launch for the far queries (more region
// rows, more dummy slots behind the staged ones)
static bool strip_geometry(const mpc_shape *s, int r_init, int WS, int *cap_out, size_t *lds_out, size_t *lds_far_out = nullptr) {
    const int TH = KS_NT / WS;
    if (TH + 2 * KNN_RFAR > KS_NT || r_init > KNN_RCAP) return false;
    const double dens = (double)s->n / ((double)s->hq * s->wq);
    const double row_pts = dens * (WS + 2 * r_init);                       // points per region row of an inner strip
    // slots of an inner query: its rows, one dummy slot per even row; must leave room for denser places
    const int maxch = (s->flags & MPC_F_DIST_L1) ? KS_MAXCH_L1 : KS_MAXCH;
    if ((2 * r_init + 1) * (row_pts + 0.5) * 1.3 > 4 * maxch) return false;
    const int rows = (TH < s->hq ? TH : s->hq) + 2 * r_init;
    int cap = (int)(1.15 * rows * (row_pts + 0.5)) + 32;
    cap = (cap + 15) / 16 * 16;
    const bool next = (s->flags & MPC_F_WANT_NEXT) != 0;
    auto lds_of = [&](int NR, int tail) {
        return (((size_t)NR * 8 + (size_t)(NR + 1) * 4 + 15) & ~(size_t)15) + (size_t)(cap + tail) * 8 * (next ? 3 : 2) + (size_t)cap * 2 + 16;
    };
    const size_t lds = lds_of(TH + 2 * KNN_RCAP, KS_TAIL(maxch)), lds_far = lds_of(TH + 2 * KNN_RFAR, KS_TAIL(KS_MAXCH_FAR));
    if (lds_far > 64 * 1024) return false;
    *cap_out = cap; *lds_out = lds;
    if (lds_far_out) *lds_far_out = lds_far;
    return true;
}

bool mpc_knn_strip_usable(const mpc_shape *s, int r_init) {
    int cap; size_t lds;
    if (s->T != 1 || s->n >= 65536 || s->K < 1 || s->K > 4 * ((s->flags & MPC_F_DIST_L1) ? KS_MAXCH_L1 : KS_MAXCH)) return false;
    if (s->hq < 2 * r_init + 2 || s->wq < 2 * r_init + 2) return false;       // tiny grids: the square is the whole grid
    return strip_geometry(s, r_init, 2, &cap, &lds);
}

// can the strip kernel's spare workgroups count the event rows of `evc` (their LDS holds one counter per backward bucket of a sample)?
bool mpc_knn_strip_counts_events(const mpc_shape *s, const EvCountArgs *evc) {
    int cap; size_t lds;
    if (!evc || !evc->events || !strip_geometry(s, mpc_knn_r_init(s), 2, &cap, &lds)) return false;
    return (size_t)evc->nb * evc->NCS * sizeof(int) <= lds && evc->B <= 256;
}

// Fallback workgroups of a tail launch: as many as KS_FB_BLOCKS, but at least 32 workgroup slots fewer than the chip holds
// workgroups of this kernel at once (see k_knn_tail: a fallback workgroup may spin until the strip workgroups are done, and those
// must always find a free slot), and no fewer than `min_fb` (the event-count prefix of mpc_focus_fwd rides in the first B of them).
static int tail_fallback_blocks(const void *kernel, size_t lds_tail, int min_fb) {
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, KS_NT, lds_tail) != hipSuccess) {
        (void)hipGetLastError();
        mpc_set_error("mpc_knn_strip_launch: occupancy query failed");
        return MPC_E_UNSUPPORTED;
    }
    const long long slots = (long long)ncu * per_cu;
    // (32 slots -- one per group of 8 CUs -- stay free of spinners whatever happens: enough for forward progress in ANY dispatch order;
    // with the hardware's usual order the strip workgroups hold their slots before a fallback workgroup starts, and what counts is how
    // many one-wavefront searchers the late list of a band-heavy field gets: a whole slot per CU kept free cost a 45 % contraction 7 %)
    long long fb = slots - 32;
    if (fb > KS_FB_BLOCKS) fb = KS_FB_BLOCKS;
    if (fb < 64 || fb < min_fb) {
        mpc_set_error("mpc_knn_strip_launch: %d workgroup(s) of the tail kernel per CU on %d CUs leave no room for its fallback workgroups", per_cu, ncu);
        return MPC_E_UNSUPPORTED;
    }
    return (int)fb;
}

int mpc_knn_strip_launch(const mpc_shape *s, const float *traj, const knn_cs_t *cell_start, const knn_cs_t *sat, const float2 *spos, const knn_idx_t *sidx,
                         float *flow_lut, float *flow_next, float *knn_state, float *tile_dkmax, const KnnLists *lists, int r_init,
                         const EvCountArgs *evc, hipStream_t st) {
    const KnnParams p = knn_params(s);
    const KnnLists ls = *lists;
    int cap = 0; size_t lds = 0, lds_far = 0;
    if (!strip_geometry(s, r_init, 2, &cap, &lds, &lds_far)) { mpc_set_error("mpc_knn_strip_launch: shape not served by the strip kernel"); return MPC_E_UNSUPPORTED; }
    constexpr int WS = 2, TH = KS_NT / WS;
    const int gx = mpc_cdiv(s->wq, WS), gy = mpc_cdiv(s->hq, TH);
    EvCountArgs ec{};
    if (evc) ec = *evc;                 // (the caller checked mpc_knn_strip_counts_events)
    const int n_evc = (ev_count_blocks(ec) + 7) / 8 * 8;
    const int64_t total = ((int64_t)gx * gy * s->B * s->nb + 7) / 8 * 8 + n_evc;
    int evc_stride = n_evc > 0 ? (int)(total / (n_evc >> 3) / 8 * 8) : 8;       // a group of 8 counting workgroups every `stride` workgroups
    if (evc_stride < 8) evc_stride = 8;
    const dim3 grid((unsigned)total);
    // (the tail launch: nothing to do unless a strip overflowed its staging area / holds far queries / the main launch listed
    // queries for the one-wavefront-per-query search -- workgroups that read two words)
    size_t lds_tail = lds_far > lds ? lds_far : lds;       // (far pass; the quarters of an overflowed strip use the main launch's carve-up)
    if (lds_tail < 4 * 256 * sizeof(float4)) lds_tail = 4 * 256 * sizeof(float4);
#define KS_LAUNCH(L1_, NEXT_, IWD_)                                                                                       \
    do {                                                                                                                  \
        MPC_LAUNCH((k_knn_strip<WS, L1_, NEXT_, IWD_>), grid, dim3(KS_NT), lds, st, p, traj, cell_start, sat, spos, sidx, \
                           flow_lut, flow_next, knn_state, tile_dkmax, ls, r_init, cap, gx, gy, ec, n_evc, evc_stride);   \
        const int fb_ = tail_fallback_blocks((const void *)k_knn_tail<WS, L1_, NEXT_, IWD_>, lds_tail, ec.events ? ec.B : 0); \
        if (fb_ < 0) return fb_;                                                                                          \
        MPC_LAUNCH((k_knn_tail<WS, L1_, NEXT_, IWD_>), dim3(KS_RETRY_BLOCKS + fb_), dim3(KS_NT), lds_tail, st, p, traj, cell_start, sat, spos, sidx, \
                           flow_lut, flow_next, knn_state, tile_dkmax, ls, r_init, cap, gx, gy, ec);                      \
    } while (0)
    switch ((p.l1 ? 4 : 0) | (p.want_next ? 2 : 0) | (p.iwd ? 1 : 0)) {
    case 0: KS_LAUNCH(false, false, false); break;
    case 1: KS_LAUNCH(false, false, true); break;
    case 2: KS_LAUNCH(false, true, false); break;
    case 3: KS_LAUNCH(false, true, true); break;
    case 4: KS_LAUNCH(true, false, false); break;
    case 5: KS_LAUNCH(true, false, true); break;
    case 6: KS_LAUNCH(true, true, false); break;
    default: KS_LAUNCH(true, true, true); break;
    }
#undef KS_LAUNCH
    MPC_CHECK_LAUNCH();
    return 0;
}

MPC_BOUNDS_UNIT("knn_strip.hip")
