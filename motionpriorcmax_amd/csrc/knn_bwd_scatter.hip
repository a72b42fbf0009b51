// Backward of the K-nearest-neighbour flow look-up table as a query-centric SCATTER (num_tref == 1, 'mean'):
//   d flow[point] = 1/K * sum over the queries q that have the point among their K neighbours of dLUT[q]
//   reference: autograd of the gather + mean at src/losses/focus.py:140-168 (and :170-176 for flow_to_next)
//
// The forward strip kernel (knn_strip.hip, LEAN) leaves each query's neighbour set as a bit mask over the query's
// contiguous slot range in its strip, plus the strip's row table (first bucketed point and first slot of every region
// row).  Here a workgroup OWNS the trajectory points of G strips' worth of LUT columns (x one block of TH rows): their
// accumulators live in LDS, in bucket order.  It walks the strips whose queries can reach its columns (its own and
// r_init on either side); per strip every wavefront translates the slots of the rows its 64 queries use into
// accumulator addresses (a 16-bit table in LDS; 0 = not one of this workgroup's points), and every lane then walks the
// bits of its query's mask -- set bit => one integer LDS atomic that adds the query's gradient to the neighbour.
//
// Fixed point: both components of a gradient are packed into one 64-bit word (y in the high, x in the low 32 bits, the
// sum is a plain 64-bit add: ds_add_u64 costs 8-12 cycles per wavefront where ds_add_f32 costs ~190), scaled by a
// power of two taken from the largest |dLUT| the workgroup can see so that 2^F bounds every addend.  A point is a
// neighbour only of queries within 2 r_init cells (a served query's neighbours lie inside its search square), so at
// most (4 r_init + 1)^2 mask additions plus KB_FB fallback additions (below) reach one word per round: with F = 30 -
// ceil(log2 of that) no field can overflow, the integer sums are exact and independent of the order of the atomics --
// the gradient is bitwise reproducible -- and the rounding of an addend is 2^-(F+1) of the largest gradient in sight.
//
// Queries the strip kernel handed to its fallback (flagged in `fbits`; 0.01-0.2 % of them) have no mask: the
// workgroup lists, in query order, those whose K-th distance reaches its points, a wavefront per query re-derives the
// membership of the workgroup's points from the saved K-th key, KB_FB queries per round, each round closed by an output
// pass (store, then add).  A workgroup whose points do not fit its LDS share (heavily clustered trajectories) keeps its
// accumulators in global memory (`gacc`, 64-bit integer atomics): slow, exact, never taken by the shipped shapes.
#include "knn_device.h"
#include <stdlib.h>

#define KB_NT 256
#define KB_FB 64           // fallback queries per round
#define KB_FLIST 2048      // listed fallback queries per pass over the map

struct KbGeom {
    int G, ngx;                       // strips owned per workgroup, workgroups per row of strips
    int gx, gy, WS, TH, NR, capT;     // the forward's strips (KnnStripGeom)
    int r_init, cap_grp, F, fwpr;
};

__device__ __forceinline__ int kb_wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// power-of-two scale that brings |v| <= m to |v * sc| <= 2^F, and its inverse
__device__ __forceinline__ float2 kb_scale(float m, int F) {
    if (!(m > 0.f) || !(m < INFINITY)) return make_float2(1.f, 1.f);
    int e;
    (void)frexpf(m, &e);                       // m = f * 2^e, 0.5 <= f < 1
    int sh = F - e;
    sh = min(max(sh, -100), 100);
    return make_float2(ldexpf(1.f, sh), ldexpf(1.f, -sh));
}

__device__ __forceinline__ unsigned long long kb_pack(float2 g, float sc) {
    const long long iy = (long long)__float2int_rn(g.x * sc), ix = (long long)__float2int_rn(g.y * sc);
    return (unsigned long long)((iy << 32) + ix);
}

__device__ __forceinline__ float2 kb_unpack(unsigned long long v, float inv, float invK) {
    const int lo = (int)(unsigned)(v & 0xffffffffull);
    const int hi = (int)(((long long)v - (long long)lo) >> 32);
    return make_float2(invK * ((float)hi * inv), invK * ((float)lo * inv));
}

// One word of the masks of a PAIR of queries (the two columns of a strip at one row: same slot range, so bit b of both
// masks is the same slot): slot 32 wd + k <-> bit 8 (k & 3) + (k >> 2).  A slot that is a neighbour of both queries gets
// ONE atomic with the sum of their gradients: neighbouring queries share ~85 % of their neighbours, and the kernel is
// bound by the number of LDS instructions, not by their width.  `sq` = accumulators of the strip at the pair's first slot
// (strip layout: slot s + j of the strip kernel is word s + j, no translation).
template <bool NEXT>
__device__ __forceinline__ void kb_scatter_pair_word(unsigned ma, unsigned mb, unsigned long long *sq, unsigned long long *sqn, int wd,
                                                     unsigned long long ga, unsigned long long gb, unsigned long long gab,
                                                     unsigned long long gna, unsigned long long gnb, unsigned long long gnab,
                                                     bool has_next) {
    const unsigned mu = ma | mb;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const unsigned bit = 1u << (8 * (k & 3) + (k >> 2));
        if (mu & bit) {
            const bool A = (ma & bit) != 0u, Bq = (mb & bit) != 0u;
            atomicAdd(sq + 32 * wd + k, A ? (Bq ? gab : ga) : gb);
            if (NEXT) { if (has_next) atomicAdd(sqn + 32 * wd + k, A ? (Bq ? gnab : gna) : gnb); }
        }
    }
}

// grid: 1-D, ngx * gy * B * nb workgroups in XCD-contiguous order, 256 threads, dynamic LDS sized by the launcher
template <bool L1, bool NEXT>
__global__ __launch_bounds__(KB_NT) void k_knn_bwd_scatter(const KnnParams p, const KbGeom kg,
                                                           const int *__restrict__ cell_start,
                                                           const float2 *__restrict__ spos, const int *__restrict__ sidx,
                                                           const float *__restrict__ glut, const float *__restrict__ gnext,
                                                           const float *__restrict__ knn_state,
                                                           const unsigned *__restrict__ masks, const int2 *__restrict__ rowtab,
                                                           const unsigned *__restrict__ fbits,
                                                           unsigned long long *__restrict__ gacc,
                                                           float2 *__restrict__ tmp_g, float2 *__restrict__ tmp_a) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_scan[KB_NT / 64], s_mx[KB_NT / 64];
    __shared__ float s_fm[2][KB_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = kg.ngx * kg.gy * p.B * p.nb;
    const int lblk = (int)(blockIdx.x & 7) * ((nblk + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (lblk >= nblk) return;
    const int bt = lblk / (kg.ngx * kg.gy), rem = lblk - bt * kg.ngx * kg.gy;
    const int sy = rem / kg.ngx, jx = rem - sy * kg.ngx;
    const int b = bt / p.nb, t = bt - b * p.nb;
    const int WS = kg.WS, TH = kg.TH, R2 = 2 * kg.r_init, NR = kg.NR;
    // the points this workgroup owns: LUT rows [oy0, oy1] x columns [c0, c1]
    const int oy0 = sy * TH, oy1 = min(oy0 + TH, p.hq) - 1, nrows = oy1 - oy0 + 1;
    const int c0 = jx * kg.G * WS, c1 = min(c0 + kg.G * WS, p.wq) - 1;
    const int *cs = cell_start + (size_t)bt * (p.G + 1);
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const int *si_ = sidx + (size_t)bt * p.n;
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const float2 *gl2 = reinterpret_cast<const float2 *>(glut) + (size_t)bt * p.G;
    const bool has_next = NEXT && (gnext != nullptr) && (t < p.nb - 1);
    const float2 *gn2 = has_next ? reinterpret_cast<const float2 *>(gnext) + (size_t)(b * (p.nb - 1) + t) * p.G : nullptr;
    // ---- LDS carve-up ------------------------------------------------------------------------------
    int2 *grow = reinterpret_cast<int2 *>(s_dyn);                              // [TH + 1] {first owned point of the row (bucket order), first accumulator}
    size_t o = (size_t)(TH + 1) * 8;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(s_dyn + o); o += (size_t)kg.cap_grp * 8;
    unsigned long long *accn = reinterpret_cast<unsigned long long *>(s_dyn + o); o += NEXT ? (size_t)kg.cap_grp * 8 : 0;
    // two strips are walked at a time (wavefronts 0-1 / 2-3): their accumulators in the strip kernel's slot order (+ the
    // tail a mask word can reach beyond the last slot), and the first slot of every region row
    const size_t sacc_n = (size_t)kg.capT + 96;
    unsigned long long *sacc_all = reinterpret_cast<unsigned long long *>(s_dyn + o); o += 2 * sacc_n * 8 * (NEXT ? 2 : 1);
    unsigned short *rsl_all = reinterpret_cast<unsigned short *>(s_dyn + o);          // [2][NR]
    const int half = wv >> 1, hl = tid & 127;
    unsigned long long *sacc = sacc_all + (size_t)half * sacc_n * (NEXT ? 2 : 1), *saccn = sacc + sacc_n;
    unsigned short *rsl = rsl_all + (size_t)half * ((NR + 1) & ~1);
    unsigned *flist = reinterpret_cast<unsigned *>(sacc_all);                         // (after the strips) listed fallback queries

    // ---- 1. rows of the owned region: first point and number of points, exclusive scan -> accumulator slots ------
    int ra = 0, rc = 0;
    if (tid < nrows) { const int y = oy0 + tid; ra = cs[y * p.wq + c0]; rc = cs[y * p.wq + c1 + 1] - ra; }
    int incl = rc;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(incl, o2, 64); if (lane >= o2) incl += v; }
    const int wmx = kb_wave_max(rc);
    if (lane == 63) { s_scan[wv] = incl; s_mx[wv] = wmx; }
    // ---- 2. largest |dLUT| among the queries that can reach the region -> fixed-point scales ----------------------
    {
        const int ry0 = max(oy0 - R2, 0), ry1 = min(oy1 + R2, p.hq - 1), rx0 = max(c0 - R2, 0), rx1 = min(c1 + R2, p.wq - 1);
        const int rw = rx1 - rx0 + 1, items = (ry1 - ry0 + 1) * rw;
        float m = 0.f, mn = 0.f;
        int yy = ry0 + tid / rw, xx = rx0 + tid % rw;
        const int dyy = KB_NT / rw, dxx = KB_NT % rw;
        for (int it = tid; it < items; it += KB_NT) {
            const float2 g = gl2[yy * p.wq + xx];
            m = fmaxf(m, fmaxf(fabsf(g.x), fabsf(g.y)));
            if (has_next) { const float2 g2 = gn2[yy * p.wq + xx]; mn = fmaxf(mn, fmaxf(fabsf(g2.x), fabsf(g2.y))); }
            yy += dyy; xx += dxx;
            if (xx > rx1) { xx -= rw; ++yy; }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { m = fmaxf(m, __shfl_xor(m, o2, 64)); mn = fmaxf(mn, __shfl_xor(mn, o2, 64)); }
        if (lane == 0) { s_fm[0][wv] = m; s_fm[1][wv] = mn; }
    }
    __syncthreads();
    int run = incl - rc;
    for (int w = 0; w < wv; ++w) run += s_scan[w];
    if (tid < nrows) grow[tid] = make_int2(ra, run);
    if (tid == nrows - 1) grow[nrows] = make_int2(ra + rc, run + rc);
    const int maxcnt = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    const float2 scl = kb_scale(fmaxf(fmaxf(s_fm[0][0], s_fm[0][1]), fmaxf(s_fm[0][2], s_fm[0][3])), kg.F);
    const float2 scn = kb_scale(fmaxf(fmaxf(s_fm[1][0], s_fm[1][1]), fmaxf(s_fm[1][2], s_fm[1][3])), kg.F);
    __syncthreads();
    const int total = grow[nrows].y;
    const bool use_global = total > kg.cap_grp;
    unsigned long long *ga = gacc + (size_t)bt * p.n, *gan = gacc + ((size_t)p.B * p.nb + bt) * p.n;
    // lanes per row of the passes over the owned points: the power of two that holds the longest row (4 .. 64)
    int lsub = 6;
    while (lsub > 2 && (1 << (lsub - 1)) >= maxcnt) --lsub;
    const int sub = 1 << lsub, rpi = KB_NT >> lsub;
    // f(accumulator slot, bucketed point) for every owned point; the same thread gets the same point in every pass
    auto for_owned = [&](auto f) {
        for (int r0 = 0; r0 < nrows; r0 += rpi) {
            const int rr = r0 + (tid >> lsub);
            if (rr < nrows) {
                const int2 gr = grow[rr];
                const int cnt = grow[rr + 1].y - gr.y;
                for (int k = tid & (sub - 1); k < cnt; k += sub) f(gr.y + k, gr.x + k);
            }
        }
    };
    auto zero_acc = [&]() {
        if (!use_global) {
            for (int i = tid; i < total; i += KB_NT) { acc[i] = 0ull; if (NEXT) accn[i] = 0ull; }
        } else {
            for_owned([&](int, int g) { atomicExch(ga + g, 0ull); if (NEXT) atomicExch(gan + g, 0ull); });
            __threadfence();                                          // (performed before the barrier that follows)
        }
    };
    zero_acc();
    __syncthreads();

    // ---- 3. the strips whose queries can reach the region, two at a time ----------------------------------------------
    // Per pair of strips: [first slots of the rows -> LDS] barrier [scatter: lane = the pair of queries of one row, one
    // predicated atomic per slot into the strip's accumulators] barrier [fold: lane = region row, every slot of the row is
    // taken (exchange with 0) and, where it is one of this workgroup's points, added to the point's accumulator].
    // Everything the NEXT pair reads from global memory is requested before the current one is worked on.
    for (size_t i = tid; i < 2 * sacc_n * (NEXT ? 2 : 1); i += KB_NT) sacc_all[i] = 0ull;
    const int i_lo = max(jx * kg.G - kg.r_init, 0), i_hi = min(jx * kg.G + kg.G + kg.r_init, kg.gx);
    const int npair = (i_hi - i_lo + 1) >> 1;
    const int sy_lo = max(sy - 1, 0), nsy = min(sy + 1, kg.gy - 1) - sy_lo + 1;
    struct Pre {
        int2 e, e2; int tot, ry_base, cy, ra, rb; bool on, acta, actb;
        unsigned ma0, ma1, ma2, mb0, mb1, mb2; float2 ga, gb, gna, gnb;
    };
    auto prefetch = [&](int it) {
        Pre v;
        const int sy2 = sy_lo + it / npair, i = i_lo + 2 * (it - (it / npair) * npair) + half;
        v.on = i < i_hi;                                          // (an odd number of strips: the last pair is half empty)
        const int ic = min(i, i_hi - 1);
        const int qy0 = sy2 * TH, qy1 = min(qy0 + TH, p.hq) - 1, qx0 = ic * WS, qx1 = min(qx0 + WS, p.wq) - 1;
        const int sid = (bt * kg.gy + sy2) * kg.gx + ic;
        v.ry_base = qy0 - R2;
        v.cy = qy0 + hl;
        const bool va = v.on && v.cy <= qy1, vb = va && qx0 + 1 <= qx1;
        v.ra = v.rb = kg.r_init;
        if (va) v.ra = query_radius(p, v.cy, qx0, kg.r_init);
        if (vb) v.rb = query_radius(p, v.cy, qx0 + 1, kg.r_init);
        v.acta = va && qx0 + v.ra >= c0 && qx0 - v.ra <= c1 && v.cy + v.ra >= oy0 && v.cy - v.ra <= oy1;
        v.actb = vb && qx0 + 1 + v.rb >= c0 && qx0 + 1 - v.rb <= c1 && v.cy + v.rb >= oy0 && v.cy - v.rb <= oy1;
        const int2 *rt = rowtab + (size_t)sid * (NR + 1);
        v.tot = rt[NR].x;
        v.e = rt[min(hl, NR)];
        v.e2 = rt[min(128 + hl, NR)];
        const unsigned *mq = masks + (size_t)sid * (3 * KB_NT) + min(hl, TH - 1) * WS;
        v.ma0 = mq[0]; v.ma1 = mq[KB_NT]; v.ma2 = mq[2 * KB_NT];
        v.mb0 = mq[1]; v.mb1 = mq[KB_NT + 1]; v.mb2 = mq[2 * KB_NT + 1];
        const int gi = min(v.cy, p.hq - 1) * p.wq + qx0, gj = min(v.cy, p.hq - 1) * p.wq + min(qx0 + 1, p.wq - 1);
        v.ga = gl2[gi]; v.gb = gl2[gj];
        v.gna = v.gnb = make_float2(0.f, 0.f);
        if (has_next) { v.gna = gn2[gi]; v.gnb = gn2[gj]; }
        return v;
    };
    const int nit = nsy * npair;
    __syncthreads();
    Pre cur = prefetch(0);
    for (int it = 0; it < nit; ++it) {
        const Pre v = cur;
        // rows of this half's strip: lane = region row (two rows for the first NR - 128 lanes)
        const bool live = v.on && v.tot <= kg.capT;            // (a strip that overflowed its staging area served no query)
        const int rs1 = v.e.y & 0xffff, len1 = (live && hl < NR) ? (int)((unsigned)v.e.y >> 16) : 0;
        const int rs2 = v.e2.y & 0xffff, len2 = (live && 128 + hl < NR) ? (int)((unsigned)v.e2.y >> 16) : 0;
        if (hl < NR) rsl[hl] = (unsigned short)rs1;
        if (128 + hl < NR) rsl[128 + hl] = (unsigned short)rs2;
        __syncthreads();
        cur = prefetch(min(it + 1, nit - 1));
        // ---- scatter ----
        {
            const bool acta = live && v.acta, actb = live && v.actb;
            unsigned ma0 = acta ? v.ma0 : 0u, ma1 = acta ? v.ma1 : 0u, ma2 = acta ? v.ma2 : 0u;
            unsigned mb0 = actb ? v.mb0 : 0u, mb1 = actb ? v.mb1 : 0u, mb2 = actb ? v.mb2 : 0u;
            if (__ballot(acta || actb) != 0ull) {
                const unsigned long long pa = kb_pack(v.ga, scl.x), pb = kb_pack(v.gb, scl.x);
                const unsigned long long pna = has_next ? kb_pack(v.gna, scn.x) : 0ull, pnb = has_next ? kb_pack(v.gnb, scn.x) : 0ull;
                const int rowa = max(v.cy - v.ra, 0) - v.ry_base, rowb = max(v.cy - v.rb, 0) - v.ry_base;
                const int sa = (acta || actb) ? (int)rsl[min(max(rowa, 0), NR - 1)] : 0, sb = (acta || actb) ? (int)rsl[min(max(rowb, 0), NR - 1)] : 0;
                // (the two columns of a strip next to the left or right image border can have squares of different size:
                // their slot ranges then start at different rows, and they are walked one after the other)
                const bool split = __ballot(acta && actb && sa != sb) != 0ull;
                for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
                    unsigned a0 = ma0, a1 = ma1, a2 = ma2, b0 = mb0, b1 = mb1, b2 = mb2;
                    int s0 = acta ? sa : sb;
                    if (split) {
                        if (pass == 0) { b0 = b1 = b2 = 0u; s0 = sa; } else { a0 = a1 = a2 = 0u; s0 = sb; }
                    }
                    unsigned long long *sq = sacc + s0, *sqn = saccn + s0;
                    kb_scatter_pair_word<NEXT>(a0, b0, sq, sqn, 0, pa, pb, pa + pb, pna, pnb, pna + pnb, has_next);
                    kb_scatter_pair_word<NEXT>(a1, b1, sq, sqn, 1, pa, pb, pa + pb, pna, pnb, pna + pnb, has_next);
                    if (__ballot((a2 | b2) != 0u) != 0ull)
                        kb_scatter_pair_word<NEXT>(a2, b2, sq, sqn, 2, pa, pb, pa + pb, pna, pnb, pna + pnb, has_next);
                }
            }
        }
        __syncthreads();
        // ---- fold ----
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int rr = h2 * 128 + hl, gs = h2 ? v.e2.x : v.e.x, rs = h2 ? rs2 : rs1, len = h2 ? len2 : len1;
            const int lmax = __builtin_amdgcn_readfirstlane(kb_wave_max(len));
            if (lmax == 0) continue;
            int gfirst = 0, gslot = 0, cnt = 0;
            const int y = v.ry_base + rr;
            if (len > 0 && y >= oy0 && y <= oy1) { const int2 gr = grow[y - oy0]; gfirst = gr.x; gslot = gr.y; cnt = grow[y - oy0 + 1].y - gr.y; }
            for (int k = 0; k < lmax; ++k) {
                if (k < len) {
                    const unsigned long long val = atomicExch(sacc + rs + k, 0ull);
                    const int og = gs + k - gfirst;
                    const bool own = og >= 0 && og < cnt;
                    if (own && val != 0ull) { if (use_global) atomicAdd(ga + gs + k, val); else atomicAdd(acc + gslot + og, val); }
                    if (NEXT && has_next) {
                        const unsigned long long vn = atomicExch(saccn + rs + k, 0ull);
                        if (own && vn != 0ull) { if (use_global) atomicAdd(gan + gs + k, vn); else atomicAdd(accn + gslot + og, vn); }
                    }
                }
            }
        }
        // (the next pair's first-slot table is written after this fold by the same threads that read it above; the
        // accumulators it scatters into are separated from this fold by the barrier at the top of the loop)
    }
    if (use_global) __threadfence();
    __syncthreads();

    // ---- 4. fallback queries, in rounds of KB_FB; every round ends with an output pass ----------------------------
    const float invK = 1.f / (float)p.K;
    auto output = [&](bool first) {
        for_owned([&](int slot, int g) {
            const int idx = si_[g];
            const unsigned long long v = use_global ? atomicAdd(ga + g, 0ull) : acc[slot];
            float2 ov = kb_unpack(v, scl.y, invK);
            float2 *dst = tmp_g + (size_t)bt * p.n + idx;
            if (!first) { const float2 c = *dst; ov.x += c.x; ov.y += c.y; }
            *dst = ov;
            if (NEXT) {
                if (gnext != nullptr) {
                    const unsigned long long vn = use_global ? atomicAdd(gan + g, 0ull) : accn[slot];
                    float2 on = has_next ? kb_unpack(vn, scn.y, invK) : make_float2(0.f, 0.f);
                    float2 *dn = tmp_a + (size_t)bt * p.n + idx;
                    if (!first) { const float2 c = *dn; on.x += c.x; on.y += c.y; }
                    *dn = on;
                }
            }
        });
    };
    // the (sample, bin)'s map of fallback queries, a contiguous run of words per thread: the list comes out in query order
    const int fw = p.hq * kg.fwpr, fchunk = (fw + KB_NT - 1) / KB_NT;
    const unsigned *fb = fbits + (size_t)bt * fw;
    const float ylo = (oy0 == 0) ? -INFINITY : (float)(oy0 * p.sp) - 0.5f, yhi = (oy1 == p.hq - 1) ? INFINITY : (float)((oy1 + 1) * p.sp) - 0.5f;
    const float xlo = (c0 == 0) ? -INFINITY : (float)(c0 * p.sp) - 0.5f, xhi = (c1 == p.wq - 1) ? INFINITY : (float)((c1 + 1) * p.sp) - 0.5f;
    auto reach_of = [&](float dk) { return (L1 ? dk : sqrtf(dk)) * 1.0001f + 0.01f; };
    auto relevant = [&](int cy, int cx) {
        const float dk = knn_state[(size_t)bt * p.G + (size_t)cy * p.wq + cx];
        const float R = reach_of(dk);
        const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
        const float dy = fmaxf(fmaxf(ylo - qy, qy - yhi), 0.f), dx = fmaxf(fmaxf(xlo - qx, qx - xhi), 0.f);
        return fmaxf(dy, dx) <= R;
    };
    int mine = 0;
    for (int wdx = tid * fchunk; wdx < min((tid + 1) * fchunk, fw); ++wdx) {
        unsigned bits = fb[wdx];
        const int cy = wdx / kg.fwpr, cxb = (wdx - cy * kg.fwpr) * 32;
        while (bits) { const int bpos = __ffs(bits) - 1; bits &= bits - 1u; mine += relevant(cy, cxb + bpos) ? 1 : 0; }
    }
    int fincl = mine;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(fincl, o2, 64); if (lane >= o2) fincl += v; }
    if (lane == 63) s_scan[wv] = fincl;
    __syncthreads();
    int fbase = fincl - mine;
    for (int w = 0; w < wv; ++w) fbase += s_scan[w];
    const int nf = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    bool first = true;
    for (int pos0 = 0; pos0 < nf; pos0 += KB_FLIST) {
        // list the fallback queries number pos0 .. pos0 + KB_FLIST - 1
        __syncthreads();
        if (mine > 0 && fbase < pos0 + KB_FLIST && fbase + mine > pos0) {
            int ord = fbase;
            for (int wdx = tid * fchunk; wdx < min((tid + 1) * fchunk, fw); ++wdx) {
                unsigned bits = fb[wdx];
                const int cy = wdx / kg.fwpr, cxb = (wdx - cy * kg.fwpr) * 32;
                while (bits) {
                    const int bpos = __ffs(bits) - 1; bits &= bits - 1u;
                    if (relevant(cy, cxb + bpos)) {
                        if (ord >= pos0 && ord < pos0 + KB_FLIST) flist[ord - pos0] = ((unsigned)cy << 16) | (unsigned)(cxb + bpos);
                        ++ord;
                    }
                }
            }
        }
        __syncthreads();
        const int m = min(KB_FLIST, nf - pos0);
        for (int b0 = 0; b0 < m; b0 += KB_FB) {
            if (!first) { zero_acc(); __syncthreads(); }
            for (int j = b0 + wv; j < min(b0 + KB_FB, m); j += KB_NT / 64) {
                const unsigned ent = flist[j];
                const int cy = (int)(ent >> 16), cx = (int)(ent & 0xffffu);
                const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
                const float dK = knn_state[q];
                const int iK = reinterpret_cast<const int *>(knn_state)[BQ + q] & ~KNN_TIE_FLAG;
                const unsigned long long gp = kb_pack(gl2[cy * p.wq + cx], scl.x);
                const unsigned long long gpn = has_next ? kb_pack(gn2[cy * p.wq + cx], scn.x) : 0ull;
                const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
                const float R = reach_of(dK);
                const int ya = max(cell_of(qy - R, p.sp, p.hq), oy0), yb = min(cell_of(qy + R, p.sp, p.hq), oy1);
                const int xa = max(cell_of(qx - R, p.sp, p.wq), c0), xb = min(cell_of(qx + R, p.sp, p.wq), c1);
                if (xa > xb) continue;
                for (int rb = ya; rb <= yb; rb += 64) {
                    // lane = row: its run of owned points inside the reach; flat numbering of the candidates
                    const int y = rb + lane;
                    int js = 0, ln = 0;
                    if (y <= yb) { js = cs[y * p.wq + xa]; ln = cs[y * p.wq + xb + 1] - js; }
                    int ci = ln;
#pragma unroll
                    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(ci, o2, 64); if (lane >= o2) ci += v; }
                    const int N = __shfl(ci, 63, 64), excl = ci - ln;
                    for (int k0 = 0; k0 < N; k0 += 64) {
                        const int k = k0 + lane;
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
                            const float2 pj = sp_[g];
                            const int id = si_[g];
                            const float d = pair_dist(qy, qx, pj.x, pj.y, L1);
                            if (d < dK || (d == dK && id <= iK)) {
                                const int2 gr = grow[rb + lo - oy0];
                                if (use_global) {
                                    atomicAdd(ga + g, gp);
                                    if (NEXT) { if (has_next) atomicAdd(gan + g, gpn); }
                                } else {
                                    const int slot = gr.y + (g - gr.x);
                                    atomicAdd(acc + slot, gp);
                                    if (NEXT) { if (has_next) atomicAdd(accn + slot, gpn); }
                                }
                            }
                        }
                    }
                }
            }
            if (use_global) __threadfence();
            __syncthreads();
            output(first);
            first = false;
            __syncthreads();
        }
    }
    if (first) output(true);
}

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
static bool kb_geometry(const mpc_shape *s, bool next, KbGeom *kg, size_t *lds_out) {
    const int r_init = mpc_knn_r_init(s);
    KnnStripGeom g;
    if (!mpc_knn_strip_geom(s, r_init, &g)) return false;
    if (g.TH != 128 || g.WS != 2) return false;                 // (lane = the pair of queries of one strip row; two strips per workgroup pass)
    const int64_t bt = (int64_t)s->B * s->nb;
    static const int g_env = getenv("MPC_KNN_BWD_G") ? atoi(getenv("MPC_KNN_BWD_G")) : 0;
    const double dens = (double)s->n / ((double)s->hq * s->wq);
    // Strips per workgroup, G: a workgroup walks G + 2 r_init strips (r_init on either side belong to its neighbours), so a
    // large G repeats fewer queries; a small G gives more, shorter workgroups.  Model: rounds of 256 workgroups x pairs of
    // strips walked, among the G whose accumulators leave room for two workgroups per CU.
    int best = 0, best_cap = 0; size_t best_lds = 0; double best_cost = 0.0;
    for (int G = 1; G <= g.gx && G <= 64; ++G) {
        if (g_env > 0 && G != g_env) continue;
        const int cols = (G * g.WS < s->wq) ? G * g.WS : s->wq, rows = g.TH < s->hq ? g.TH : s->hq;
        int cap_grp = (int)(1.25 * dens * cols * rows) + 192;
        cap_grp = (cap_grp + 63) / 64 * 64;
        if (cap_grp > 65534) break;                               // (16-bit accumulator addresses)
        size_t sbytes = 2 * ((size_t)g.cap + 96) * 8 * (next ? 2 : 1);          // strip accumulators of the two halves (later: the fallback list)
        if (sbytes < (size_t)KB_FLIST * 4) sbytes = (size_t)KB_FLIST * 4;
        const size_t lds = (size_t)(g.TH + 1) * 8 + (size_t)cap_grp * 8 * (next ? 2 : 1) + sbytes + 2 * (size_t)((g.NR + 1) & ~1) * 2 + 64;
        if (lds > (G == 1 || g_env > 0 ? (size_t)150 * 1024 : (size_t)64 * 1024)) break;
        const int64_t nwg = bt * g.gy * mpc_cdiv(g.gx, G);
        const double cost = (double)((nwg + 255) / 256) * (double)(((G < g.gx ? G : g.gx) + 2 * r_init + 1) / 2 + 2);
        if (best == 0 || cost < best_cost) { best = G; best_cap = cap_grp; best_lds = lds; best_cost = cost; }
    }
    if (best == 0) return false;
    kg->G = best; kg->ngx = mpc_cdiv(g.gx, best);
    kg->gx = g.gx; kg->gy = g.gy; kg->WS = g.WS; kg->TH = g.TH; kg->NR = g.NR; kg->capT = g.cap;
    kg->r_init = r_init; kg->cap_grp = best_cap;
    // additions one accumulator word can receive per round: the queries within 2 r_init cells, plus KB_FB fallback queries
    const int adds = (4 * r_init + 1) * (4 * r_init + 1) + KB_FB;
    int hb = 0;
    while ((1 << hb) < adds) ++hb;
    kg->F = 30 - hb;
    kg->fwpr = (s->wq + 31) / 32;
    *lds_out = best_lds;
    return true;
}

bool mpc_knn_bwd_scatter_usable(const mpc_shape *s) {
    KbGeom kg; size_t lds;
    return kb_geometry(s, (s->flags & MPC_F_WANT_NEXT) != 0, &kg, &lds);
}

int mpc_knn_bwd_scatter_launch(const mpc_shape *s, const int *cell_start, const float2 *spos, const int *sidx,
                               const float *grad_flow_lut, const float *grad_flow_next, const float *knn_state,
                               const KnnLeanBufs *lean, unsigned long long *gacc, float2 *tmp_g, float2 *tmp_a, hipStream_t st) {
    const KnnParams p = knn_params(s);
    KbGeom kg; size_t lds = 0;
    const bool flag_next = (s->flags & MPC_F_WANT_NEXT) != 0;
    if (grad_flow_next && !flag_next) { mpc_set_error("mpc_knn_bwd_scatter_launch: grad_flow_next without MPC_F_WANT_NEXT"); return MPC_E_SHAPE; }
    if (!kb_geometry(s, flag_next, &kg, &lds)) { mpc_set_error("mpc_knn_bwd_scatter_launch: shape not served"); return MPC_E_UNSUPPORTED; }
    static mpc_device_once attr_once;
    if (attr_once.need()) {
        const void *fns[4] = {(const void *)k_knn_bwd_scatter<false, false>, (const void *)k_knn_bwd_scatter<false, true>,
                              (const void *)k_knn_bwd_scatter<true, false>, (const void *)k_knn_bwd_scatter<true, true>};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
        }
        attr_once.mark();
    }
    const int64_t nblk = (int64_t)kg.ngx * kg.gy * s->B * s->nb;
    const dim3 grid((nblk + 7) / 8 * 8);
#define KB_LAUNCH(L1_, NEXT_)                                                                                              \
    MPC_LAUNCH((k_knn_bwd_scatter<L1_, NEXT_>), grid, dim3(KB_NT), lds, st, p, kg, cell_start, spos, sidx,        \
                       grad_flow_lut, grad_flow_next, knn_state, lean->masks, lean->rowtab, lean->fbits, gacc, tmp_g, tmp_a)
    const bool next = grad_flow_next != nullptr;
    if (p.l1) { if (next) KB_LAUNCH(true, true); else KB_LAUNCH(true, false); }
    else { if (next) KB_LAUNCH(false, true); else KB_LAUNCH(false, false); }
#undef KB_LAUNCH
    MPC_CHECK_LAUNCH();
    return 0;
}
