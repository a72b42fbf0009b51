// Per-event part of the CMax loss: LUT warp, event weight, bilinear vote into the Image of
// Warped Events, and its hand-derived backward into the flow look-up table.
//   forward : reference src/losses/focus.py:182-230 + src/utils/event_image_converter.py:333-391
//   backward: SURVEY.md 8a row A11
// Arithmetic follows SURVEY.md Appendix A op for op (fp32, no FMA contraction: this file is
// compiled with -ffp-contract=off).
#include "common.h"
#include "ev_count_device.h"
#include "knn_device.h"
#include "bounds.h"
#include "diag/stamps.h"

struct EvParams {
    int B, M, Mp, nb, T, H, W, sp, hq, wq, P;
    unsigned flags;
};

struct Warped {
    float y, x, w;     // warped position and event weight
    int lut;           // index of the LUT cell (b,bin,iy,ix) -> element offset / (2T) ... see below
    float fy, fx;      // fractional parts
    int y0, x0;        // top-left tap
    int it, iy, ix;    // time bin and LUT cell of the (unwarped) event
};

__device__ __forceinline__ EvParams make_params(const mpc_shape s) {
    EvParams p;
    p.B = s.B; p.M = s.M; p.Mp = s.Mp; p.nb = s.nb; p.T = s.T; p.H = s.H; p.W = s.W;
    p.sp = s.sp; p.hq = s.hq; p.wq = s.wq;
    p.P = (s.flags & MPC_F_POLARITY_SPLIT) ? 2 : 1;
    p.flags = s.flags;
    return p;
}

// Warp one event for reference time `tr`.  `e` = the 6 columns of the event row.
// Returns false if the event contributes nothing (weight exactly 0).
// LUT cell of an event (focus.py:184-191): it = int(bin), iy = int(y // sp), ix = int(x // sp)
__device__ __forceinline__ void warp_cell(const EvParams &p, const float e[6], int b, int tr, Warped &o) {
    int it = (int)e[4];
    int iy = (int)floorf(mpc_div_sp(e[0], p.sp));
    int ix = (int)floorf(mpc_div_sp(e[1], p.sp));
    // torch indexing would raise on out-of-range indices; clamp instead of faulting
    it = min(max(it, 0), p.nb - 1);
    iy = min(max(iy, 0), p.hq - 1);
    ix = min(max(ix, 0), p.wq - 1);
    o.it = it; o.iy = iy; o.ix = ix;
    o.lut = (((b * p.nb + it) * p.hq + iy) * p.wq + ix) * p.T + tr;
}

// Warp with the flow `f` of the event's LUT cell already at hand (warp_cell + a read of the table).
__device__ __forceinline__ bool warp_event_with(const EvParams &p, const float e[6], const float2 f, float t_ref, Warped &o) {
    float w = (p.flags & MPC_F_UNIT_WEIGHT) ? 1.0f : e[5];
    float y = e[0], x = e[1];
    if (!(p.flags & MPC_F_NO_WARP)) {
        // focus.py:191: pos = lut + event
        y = f.x + e[0];
        x = f.y + e[1];
    }
    if (p.flags & MPC_F_SCALE_BY_DT) {
        // focus.py:204-206: (1 - clamp(|t - t_ref|, 0, 1)) * w
        const float dt = fminf(fmaxf(fabsf(e[2] - t_ref), 0.f), 1.f);
        w = (1.f - dt) * w;
    }
    if (p.flags & MPC_F_MASK_BORDER) {
        // focus.py:208-214: strict comparisons
        if (y > (float)p.H || x > (float)p.W || y < 0.f || x < 0.f) w = 0.f;
    }
    o.y = y; o.x = x; o.w = w;
    // event_image_converter.py:357-359
    const float y0f = floorf(y + 1e-6f), x0f = floorf(x + 1e-6f);
    o.fy = y - y0f;
    o.fx = x - x0f;
    // positions far outside the image cannot touch it; keep the int conversion defined
    o.y0 = (int)fminf(fmaxf(y0f, -4.f), (float)p.H + 4.f);
    o.x0 = (int)fminf(fmaxf(x0f, -4.f), (float)p.W + 4.f);
    return w != 0.f;
}

__device__ __forceinline__ bool warp_event(const EvParams &p, const float e[6], int b, int tr,
                                           const float *__restrict__ lut, float t_ref, Warped &o) {
    o.lut = -1;
    float2 f = make_float2(0.f, 0.f);
    if (!(p.flags & MPC_F_NO_WARP)) {
        warp_cell(p, e, b, tr, o);
        f = reinterpret_cast<const float2 *>(lut)[o.lut];
    }
    return warp_event_with(p, e, f, t_ref, o);
}

__device__ __forceinline__ void load_event(const float *__restrict__ events, size_t row, float e[6]) {
    const float2 *p = reinterpret_cast<const float2 *>(events + row * 6);
    const float2 a = p[0], b = p[1], c = p[2];
    e[0] = a.x; e[1] = a.y; e[2] = b.x; e[3] = b.y; e[4] = c.x; e[5] = c.y;
}

// ------------------------------------------------------------------------------------------
// v0: one thread per event, global float atomics (debug / cross-check path, MPC_F_ATOMIC_PATH)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_splat_fwd_atomic(const mpc_shape s,
                                                          const float *__restrict__ events,
                                                          const float *__restrict__ lut,
                                                          const float *__restrict__ t_ref,
                                                          float *__restrict__ iwe) {
    const EvParams p = make_params(s);
    const size_t total = (size_t)p.B * p.M;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / p.M), i = (int)(row - (size_t)b * p.M);
        float e[6];
        load_event(events, row, e);
        const int pol = (p.P == 2 && i >= p.Mp) ? 1 : 0;
        for (int tr = 0; tr < p.T; ++tr) {
            Warped o;
            const float tref = (p.flags & MPC_F_SCALE_BY_DT) ? t_ref[tr] : 0.f;
            if (!warp_event(p, e, b, tr, lut, tref, o)) continue;
            float *img = iwe + ((size_t)(b * p.T + tr) * p.P + pol) * p.H * p.W;
            const float w00 = (1.f - o.fy) * (1.f - o.fx) * o.w;
            const float w10 = o.fy * (1.f - o.fx) * o.w;
            const float w01 = (1.f - o.fy) * o.fx * o.w;
            const float w11 = o.fy * o.fx * o.w;
            const bool yin0 = o.y0 >= 0 && o.y0 < p.H, yin1 = o.y0 + 1 >= 0 && o.y0 + 1 < p.H;
            const bool xin0 = o.x0 >= 0 && o.x0 < p.W, xin1 = o.x0 + 1 >= 0 && o.x0 + 1 < p.W;
            if (yin0 && xin0) atomicAdd(img + (size_t)o.y0 * p.W + o.x0, w00);
            if (yin1 && xin0) atomicAdd(img + (size_t)(o.y0 + 1) * p.W + o.x0, w10);
            if (yin0 && xin1) atomicAdd(img + (size_t)o.y0 * p.W + o.x0 + 1, w01);
            if (yin1 && xin1) atomicAdd(img + (size_t)(o.y0 + 1) * p.W + o.x0 + 1, w11);
        }
    }
}

// per-event gradient w.r.t. the warped position, unscaled
__device__ __forceinline__ void event_pos_grad(const EvParams &p, const Warped &o,
                                               const float *__restrict__ g /* image */, float &gy,
                                               float &gx) {
    const bool yin0 = o.y0 >= 0 && o.y0 < p.H, yin1 = o.y0 + 1 >= 0 && o.y0 + 1 < p.H;
    const bool xin0 = o.x0 >= 0 && o.x0 < p.W, xin1 = o.x0 + 1 >= 0 && o.x0 + 1 < p.W;
    const float g00 = (yin0 && xin0) ? g[(size_t)o.y0 * p.W + o.x0] : 0.f;
    const float g10 = (yin1 && xin0) ? g[(size_t)(o.y0 + 1) * p.W + o.x0] : 0.f;
    const float g01 = (yin0 && xin1) ? g[(size_t)o.y0 * p.W + o.x0 + 1] : 0.f;
    const float g11 = (yin1 && xin1) ? g[(size_t)(o.y0 + 1) * p.W + o.x0 + 1] : 0.f;
    gy = o.w * ((1.f - o.fx) * (g10 - g00) + o.fx * (g11 - g01));
    gx = o.w * ((1.f - o.fy) * (g01 - g00) + o.fy * (g11 - g10));
}

__global__ __launch_bounds__(256) void k_splat_bwd_atomic(const mpc_shape s,
                                                          const float *__restrict__ events,
                                                          const float *__restrict__ lut,
                                                          const float *__restrict__ t_ref,
                                                          const float *__restrict__ gimg,
                                                          const float *__restrict__ scal,
                                                          const float *__restrict__ grad_out,
                                                          float *__restrict__ glut) {
    const EvParams p = make_params(s);
    const float coef = scal[MPC_SCAL_GCOEF] * (grad_out ? grad_out[0] : 1.f);
    const size_t total = (size_t)p.B * p.M;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / p.M), i = (int)(row - (size_t)b * p.M);
        float e[6];
        load_event(events, row, e);
        const int pol = (p.P == 2 && i >= p.Mp) ? 1 : 0;
        for (int tr = 0; tr < p.T; ++tr) {
            Warped o;
            const float tref = (p.flags & MPC_F_SCALE_BY_DT) ? t_ref[tr] : 0.f;
            if (!warp_event(p, e, b, tr, lut, tref, o)) continue;
            const float *img = gimg + ((size_t)(b * p.T + tr) * p.P + pol) * p.H * p.W;
            float gy, gx;
            event_pos_grad(p, o, img, gy, gx);
            atomicAdd(glut + 2 * (size_t)o.lut, coef * gy);
            atomicAdd(glut + 2 * (size_t)o.lut + 1, coef * gx);
        }
    }
}

// ------------------------------------------------------------------------------------------
// UNPINNED EXTENSION (default off, `FocusLoss.calc_per_event_basis`): gradient of the objective with respect to the WARPED
// POSITION of every event -- what autograd computes for `warped = differences + events[..., :2]` (focus.py:191) through
// create_iwe (event_image_converter.py:333-391).  The rows carry positions that are warped already (MPC_F_NO_WARP): a per-event
// continuous-time warp (the motion basis evaluated at the event's own timestamp, no flow LUT and no KNN -- BASELINE.json's
// north_star; the reference has none, focus.py:182-195 gathers a binned LUT) is formed by the caller in plain torch, whose
// autograd carries grad_pos on to the coefficient grid.  One thread per event row, no atomics.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_event_pos_grad(const mpc_shape s, const float *__restrict__ events,
                                                        const float *__restrict__ t_ref, const float *__restrict__ gimg,
                                                        const float *__restrict__ scal, const float *__restrict__ grad_out,
                                                        float2 *__restrict__ gpos) {
    const EvParams p = make_params(s);
    const float coef = scal[MPC_SCAL_GCOEF] * (grad_out ? grad_out[0] : 1.f);
    const float tref = (p.flags & MPC_F_SCALE_BY_DT) ? t_ref[0] : 0.f;
    const size_t total = (size_t)p.B * p.M;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / p.M), i = (int)(row - (size_t)b * p.M);
        float e[6];
        load_event(events, row, e);
        const int pol = (p.P == 2 && i >= p.Mp) ? 1 : 0;
        Warped o;
        float2 g = make_float2(0.f, 0.f);
        if (warp_event_with(p, e, make_float2(0.f, 0.f), tref, o)) {
            float gy, gx;
            event_pos_grad(p, o, gimg + ((size_t)b * p.P + pol) * p.H * p.W, gy, gx);
            g = make_float2(coef * gy, coef * gx);
        }
        gpos[row] = g;
    }
}

// ==========================================================================================
// v1: LDS-tiled path (num_tref == 1)
//
//   forward   k_ev_bin     one pass over the events: warp, weight, and append a 16-byte record to
//                          (a) the bucket of the destination strip(s) of the image it votes into
//                          and (b) the bucket of its (time bin, source LUT strip) for the backward
//             k_iwe_accum  one workgroup per (image, strip): the strip lives in LDS as 64-bit
//                          fixed point (Q33.30), records are voted with ds_add_u64 -- the global
//                          scatter becomes an on-chip accumulate -- and the strip is written once
//                          with plain coalesced stores (no zero-fill, no global atomics)
//   backward  k_lut_accum  one workgroup per (sample, bin, LUT strip): gathers the 4 taps of the
//                          adjoint image per record and accumulates d/dLUT in LDS (Q33.30)
//   capacity  a bucket holds every record that can reach it (all events of a polarity block can vote into one
//             strip; all events of a sample can sit in one (bin, LUT strip)): address space, not traffic -- only the
//             filled part is ever touched -- so no spill path exists and the sums are bitwise reproducible for any
//             event distribution; a concentrated distribution costs a long tail of the busiest workgroup instead
//
// Why fixed point: on gfx950 ds_add_f32 serialises the wave (~193 cycles per instruction) while
// ds_add_u64 takes ~8-12 (profiles/r01_ubench_lds_atomics.txt).  Q33.30 resolves 9.3e-10, finer
// than fp32 accumulation of the same taps, and integer sums are order independent, so the
// image is bitwise reproducible from run to run.
// ==========================================================================================
#define EV_FIX_SHIFT 30
#define EV_MARKER 0x6d706331   // 'mpc1': backward records of this workspace are valid

struct BinLayout {
    int SR, NS, CSR, NCS, NF, NBk, fcap, bcap, P, exact;
    int *gcount;            // [NF + NBk + 8]
    int *bcapcnt;           // [NBk] rows of the sample in each backward bucket (ev_count_device.h): its capacity; then [NBk] first records
    float4 *frec, *brec;    // bucket storage: frec [NF][fcap] of 12-byte records {y, x, w} (the image is implied by the bucket); brec: 16-byte records
};

struct __attribute__((packed, aligned(4))) rec3 { float y, x, w; };       // forward record (one 12-byte store / load)

__device__ __forceinline__ long long ev_to_fixed_small(float v) {   // |v| < 2
    return (long long)(int)(v * (float)(1 << EV_FIX_SHIFT));
}
__device__ __forceinline__ long long ev_to_fixed(float v) {         // |v| < 2^31
    const float hi = truncf(v);
    return ((long long)(int)hi << EV_FIX_SHIFT) + (long long)(int)((v - hi) * (float)(1 << EV_FIX_SHIFT));
}
__device__ __forceinline__ float ev_from_fixed(long long a) {
    return (float)((double)a * (1.0 / (double)(1 << EV_FIX_SHIFT)));
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Map physical block p to
// logical block l so that every XCD works on ONE contiguous range of logical blocks: neighbours in
// that range touch the same LUT slices / adjoint-image rows, which then stay in that XCD's L2
// instead of being fetched by all eight (speed only; correct for any placement).
__device__ __forceinline__ int xcd_swizzle(int p, int n) {
    const int per = (n + 7) >> 3;
    return (p & 7) * per + (p >> 3);
}


// one record into slot `slot` of local bucket `lb` (forward buckets first, then backward ones)
__device__ __forceinline__ void ev_emit(const EvParams &p, const BinLayout &L, int b, int nf_loc, int lb, int slot,
                                        float4 rec) {
    if (lb < nf_loc) {
        const int g = (b * p.P + lb / L.NS) * L.NS + (lb % L.NS);
        MPC_EXPECT(g >= 0 && g < L.NF);
        reinterpret_cast<rec3 *>(L.frec)[(size_t)g * L.fcap + MPC_IDX(slot, L.fcap)] = rec3{rec.x, rec.y, rec.z};
    } else {
        // (exact buckets: `slot` counts from the sample's first record -- the bucket starts behind the rows of the buckets
        // before it; else every bucket has room for all the rows of its sample)
        const size_t base = L.exact ? (size_t)b : (size_t)(b * p.nb * L.NCS + (lb - nf_loc));
        MPC_EXPECT((long long)base < (L.exact ? (long long)p.B : (long long)L.NBk));
        L.brec[base * L.bcap + MPC_IDX(slot, L.bcap)] = rec;
    }
}

// grid (ceil(ceil(M / (256*EV_PER_THREAD)) * B / 8) * 8), 256 threads, dynamic LDS = (P*NS + nb*NCS) * 2 ints
__global__ __launch_bounds__(256) void k_ev_bin(const mpc_shape s, const BinLayout L,
                                                const float *__restrict__ events,
                                                const float *__restrict__ lut,
                                                const float *__restrict__ t_ref, int want_bwd,
                                                const int *__restrict__ offsets) {
    extern __shared__ __align__(16) int s_cnt[];          // [nloc] local counts, [nloc] global bases, [nloc+1] local offsets, ids, records
    __shared__ int s_wsum[4];
    const EvParams p = make_params(s);
    const int chunks = (p.M + 256 * EV_PER_THREAD - 1) / (256 * EV_PER_THREAD);
    const int lblk = xcd_swizzle(blockIdx.x, chunks * p.B);
    if (lblk >= chunks * p.B) return;
    const int tid = threadIdx.x, b = lblk / chunks, chunk = lblk - b * chunks;
    // bucket-ordered events (offsets table of ingest / mpc_event_bucket_order): the last entry of a polarity block's table is its
    // first PADDING row -- rows from there to the end of the block are zero rows that vote for nothing: not even read
    int pad0 = p.M, pad1 = p.M, blk1 = p.M;                // [pad0, blk1) and [pad1, M) are padding
    if (offsets != nullptr) {
        const int nk1 = p.nb * L.NCS + 1;
        const int *ob = offsets + (size_t)b * 2 * nk1;
        blk1 = p.Mp;                                        // the two blocks are [0, Mp) and [Mp, M) whatever P is
        pad0 = min(max(ob[nk1 - 1], 0), blk1);
        pad1 = min(max(ob[2 * nk1 - 1], blk1), p.M);
        const int c0 = chunk * EV_PER_THREAD * 256, c1 = min(c0 + EV_PER_THREAD * 256, p.M);
        const bool all_pad = (c0 >= pad0 && c1 <= blk1) || c0 >= pad1;
        if (all_pad) return;                                  // (the marker of the backward records is not this workgroup's to write: ordered events have none)
    }
    const int nf_loc = p.P * L.NS, nb_loc = want_bwd ? p.nb * L.NCS : 0, nloc = nf_loc + nb_loc;
    int *s_base = s_cnt + nloc;
    const int stage_off = ((3 * nloc + 1 + EV_STAGE / 2) + 3) & ~3;     // in ints; 16-byte aligned record area
    for (int i = tid; i < nloc; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const float tref = (p.flags & MPC_F_SCALE_BY_DT) ? t_ref[0] : 0.f;
    const float inv_SR = 1.f / (float)L.SR, inv_CSR = 1.f / (float)L.CSR;

    float ry[EV_PER_THREAD], rx[EV_PER_THREAD], rw[EV_PER_THREAD];
    int f0[EV_PER_THREAD], f1[EV_PER_THREAD], bk[EV_PER_THREAD];     // local bucket ids (-1: none)
    int r0[EV_PER_THREAD], r1[EV_PER_THREAD], rb[EV_PER_THREAD];     // ranks inside the block
    unsigned aux[EV_PER_THREAD];
    // the rows of a thread are requested together, then their LUT gathers, and only then does the bucket logic
    // (with its branches and LDS atomics) run: two dependent memory round trips per workgroup instead of four
    float ev[EV_PER_THREAD][6];
    Warped wo[EV_PER_THREAD];
    bool live[EV_PER_THREAD];
#pragma unroll
    for (int k = 0; k < EV_PER_THREAD; ++k) {
        const int i = (chunk * EV_PER_THREAD + k) * 256 + tid;
        load_event(events, (size_t)b * p.M + min(i, p.M - 1), ev[k]);
    }
#pragma unroll
    for (int k = 0; k < EV_PER_THREAD; ++k) {
        const int i = (chunk * EV_PER_THREAD + k) * 256 + tid;
        live[k] = warp_event(p, ev[k], b, 0, lut, tref, wo[k]) && (i < p.M) && !((i >= pad0 && i < blk1) || i >= pad1);
    }
#pragma unroll
    for (int k = 0; k < EV_PER_THREAD; ++k) {
        const int i = (chunk * EV_PER_THREAD + k) * 256 + tid;
        f0[k] = f1[k] = bk[k] = -1;
        r0[k] = r1[k] = rb[k] = 0;
        ry[k] = rx[k] = rw[k] = 0.f; aux[k] = 0u;
        if (!live[k]) continue;
        const Warped &o = wo[k];
        const bool xin = (o.x0 + 1 >= 0) && (o.x0 < p.W);
        const bool yin0 = o.y0 >= 0 && o.y0 < p.H, yin1 = o.y0 + 1 >= 0 && o.y0 + 1 < p.H;
        if (!xin || !(yin0 || yin1)) continue;            // no tap inside the image
        const int pol = (p.P == 2 && i >= p.Mp) ? 1 : 0;
        ry[k] = o.y; rx[k] = o.x; rw[k] = o.w;
        // strip = row / SR without an integer division: (row + 0.5) / SR is at least 0.5 / SR away from an
        // integer, far more than the rounding of the float product, so the truncation is exact
        const int s0 = yin0 ? (int)(((float)o.y0 + 0.5f) * inv_SR) : -1, s1 = yin1 ? (int)(((float)o.y0 + 1.5f) * inv_SR) : -1;
        if (s0 >= 0) { f0[k] = pol * L.NS + s0; r0[k] = atomicAdd(&s_cnt[MPC_IDX(f0[k], nf_loc)], 1); }
        if (s1 >= 0 && s1 != s0) { f1[k] = pol * L.NS + s1; r1[k] = atomicAdd(&s_cnt[MPC_IDX(f1[k], nf_loc)], 1); }
        if (want_bwd && o.lut >= 0) {
            const int cst = (int)(((float)o.iy + 0.5f) * inv_CSR);
            bk[k] = nf_loc + o.it * L.NCS + cst;
            rb[k] = atomicAdd(&s_cnt[MPC_IDX(bk[k], nloc)], 1);
            aux[k] = ((unsigned)pol << 31) | (unsigned)((o.iy - cst * L.CSR) * p.wq + o.ix);
        }
    }
    __syncthreads();
    // reserve the block's slots in the global buckets
    for (int i = tid; i < nloc; i += 256) {
        const int c = s_cnt[i];
        int g;
        int first = 0;            // backward bucket: its first record among the sample's (ev_prefix_block, before this kernel)
        if (i < nf_loc) g = (b * p.P + i / L.NS) * L.NS + (i % L.NS);
        else { g = L.NF + b * p.nb * L.NCS + (i - nf_loc); if (L.exact) first = L.bcapcnt[L.NBk + b * p.nb * L.NCS + (i - nf_loc)]; }
        s_base[i] = (c > 0 ? atomicAdd(&L.gcount[MPC_IDX(g, L.NF + L.NBk)], c) : 0) + first;
    }
    // Local exclusive prefix of the counts: the records are first laid out in LDS bucket by bucket and then
    // written out in that order, so that neighbouring lanes store to neighbouring slots of the same bucket
    // (a 64-lane store touches a few runs of 16-byte records instead of 64 separate cache lines; the kernel
    // was bound by the issue of scattered stores: SQ_WAIT_INST_ANY 46 % of the wave cycles).
    int *s_loc = s_base + nloc;                              // [nloc + 1]
    unsigned short *s_bid = reinterpret_cast<unsigned short *>(s_loc + nloc + 1);     // [EV_STAGE]
    float4 *s_rec = reinterpret_cast<float4 *>(s_cnt + stage_off);                     // [EV_STAGE]
    {
        const int per = (nloc + 255) >> 8;
        const int i0 = min(tid * per, nloc), i1 = min(i0 + per, nloc);
        int local = 0;
        for (int i = i0; i < i1; ++i) local += s_cnt[i];
        int incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < (tid >> 6); ++w) run += s_wsum[w];
        for (int i = i0; i < i1; ++i) { s_loc[i] = run; run += s_cnt[i]; }
        if (tid == 255) s_loc[nloc] = run;
    }
    __syncthreads();
    const int total = s_loc[nloc];
#pragma unroll
    for (int k = 0; k < EV_PER_THREAD; ++k) {
        if (f0[k] >= 0 || f1[k] >= 0) {
            const int polf = (f0[k] >= 0 ? f0[k] : f1[k]) / L.NS;
            const float4 rec = make_float4(ry[k], rx[k], rw[k], __int_as_float(b * p.P + polf));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lb = h ? f1[k] : f0[k];
                if (lb < 0) continue;
                const int at = s_loc[lb] + (h ? r1[k] : r0[k]);
                if (at < EV_STAGE) { s_rec[MPC_IDX(at, EV_STAGE)] = rec; s_bid[MPC_IDX(at, EV_STAGE)] = (unsigned short)lb; }
                else ev_emit(p, L, b, nf_loc, lb, s_base[lb] + (h ? r1[k] : r0[k]), rec);
            }
        }
        if (bk[k] >= 0) {
            const float4 rec = make_float4(ry[k], rx[k], rw[k], __uint_as_float(aux[k]));
            const int at = s_loc[bk[k]] + rb[k];
            if (at < EV_STAGE) { s_rec[MPC_IDX(at, EV_STAGE)] = rec; s_bid[MPC_IDX(at, EV_STAGE)] = (unsigned short)bk[k]; }
            else ev_emit(p, L, b, nf_loc, bk[k], s_base[bk[k]] + rb[k], rec);
        }
    }
    __syncthreads();
    for (int r = tid; r < min(total, EV_STAGE); r += 256) {
        const int lb = s_bid[r];
        ev_emit(p, L, b, nf_loc, lb, s_base[MPC_IDX(lb, nloc)] + (r - s_loc[lb]), s_rec[r]);
    }
    if (want_bwd && lblk == 0 && tid == 0) L.gcount[L.NF + L.NBk + 2] = EV_MARKER;
}

// taps of one record restricted to rows [row0, row1): calls f(yy, xx, value)
template <typename F>
__device__ __forceinline__ void record_taps(float y, float x, float w, int H, int W, int row0, int row1, F f) {
    const float y0f = floorf(y + 1e-6f), x0f = floorf(x + 1e-6f);
    const float fy = y - y0f, fx = x - x0f;
    const int y0 = (int)fminf(fmaxf(y0f, -4.f), (float)H + 4.f), x0 = (int)fminf(fmaxf(x0f, -4.f), (float)W + 4.f);
    const bool r0 = y0 >= row0 && y0 < row1, r1 = y0 + 1 >= row0 && y0 + 1 < row1;
    const bool c0 = x0 >= 0 && x0 < W, c1 = x0 + 1 >= 0 && x0 + 1 < W;
    if (r0 && c0) f(y0, x0, (1.f - fy) * (1.f - fx) * w);
    if (r1 && c0) f(y0 + 1, x0, fy * (1.f - fx) * w);
    if (r0 && c1) f(y0, x0 + 1, (1.f - fy) * fx * w);
    if (r1 && c1) f(y0 + 1, x0 + 1, fy * fx * w);
}

// grid NF, 1024 threads, dynamic LDS = SR * W * 8 bytes.  FIXED: the accumulators themselves (Q33.30, int64) are the
// output -- partial images of event shards add up exactly, in any order (mpc_event_splat_fwd_fixed)
template <bool FIXED>
__global__ __launch_bounds__(1024) void k_iwe_accum(const BinLayout L, float *__restrict__ iwe, int H, int W) {
    extern __shared__ unsigned long long s_acc[];
    const int tid = threadIdx.x;
    const int g = blockIdx.x, img = g / L.NS, strip = g - img * L.NS;
    const int row0 = strip * L.SR, row1 = min(row0 + L.SR, H);
    const int npix = (row1 - row0) * W;
    for (int i = tid; i < npix; i += 1024) s_acc[i] = 0ull;
    __syncthreads();
    const int n = L.gcount[g];
    const rec3 *rec = reinterpret_cast<const rec3 *>(L.frec) + (size_t)g * L.fcap;
    for (int r0 = tid; r0 < n; r0 += 4 * 1024) {           // four record loads in flight per thread
        rec3 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = rec[min(r0 + u * 1024, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 + u * 1024 >= n) continue;
            // |tap| <= (1 + 1e-6)^2 |w|: the one-conversion form is exact below 2, any other weight (create_iwe(weight=
            // tensor), a weighted `valid` column; event_image_converter.py:45-74 accepts any) takes the hi/lo split
            if (fabsf(e[u].w) <= 1.5f)
                record_taps(e[u].y, e[u].x, e[u].w, H, W, row0, row1, [&](int yy, int xx, float v) {
                    atomicAdd(&s_acc[MPC_IDX((yy - row0) * W + xx, npix)], (unsigned long long)ev_to_fixed_small(v));
                });
            else
                record_taps(e[u].y, e[u].x, e[u].w, H, W, row0, row1, [&](int yy, int xx, float v) {
                    atomicAdd(&s_acc[MPC_IDX((yy - row0) * W + xx, npix)], (unsigned long long)ev_to_fixed(v));
                });
        }
    }
    __syncthreads();
    if (FIXED) {
        long long *dst = reinterpret_cast<long long *>(iwe) + ((size_t)img * H + row0) * W;
        for (int i = tid; i < npix; i += 1024) dst[i] = (long long)s_acc[i];
    } else {
        float *dst = iwe + ((size_t)img * H + row0) * W;
        for (int i = tid; i < npix; i += 1024) dst[i] = ev_from_fixed((long long)s_acc[i]);
    }
}

__global__ __launch_bounds__(256) void k_iwe_from_fixed(const long long *__restrict__ src, float *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = ev_from_fixed(src[i]);
}

struct __attribute__((packed, aligned(4))) pair4 { float x, y; };

// gradient of one record w.r.t. its warped position from the adjoint image (unscaled)
__device__ __forceinline__ void record_grad(float y, float x, float w, const float *__restrict__ g, int H, int W,
                                            float &gy, float &gx) {
    const float y0f = floorf(y + 1e-6f), x0f = floorf(x + 1e-6f);
    const float fy = y - y0f, fx = x - x0f;
    const int y0 = (int)fminf(fmaxf(y0f, -4.f), (float)H + 4.f), x0 = (int)fminf(fmaxf(x0f, -4.f), (float)W + 4.f);
    const bool r0 = y0 >= 0 && y0 < H, r1 = y0 + 1 >= 0 && y0 + 1 < H;
    const bool c0 = x0 >= 0 && x0 < W, c1 = x0 + 1 >= 0 && x0 + 1 < W;
    // Branch-free: two unconditional 8-byte (4-byte aligned) gathers at clamped coordinates, the taps picked
    // and masked afterwards.  With branches around the loads the compiler could not start the gathers of the
    // four records a thread has in flight together: four dependent L2 round trips instead of one (measured:
    // 13.7 us of the 16.5 us a workgroup of k_lut_accum lives).
    const int xa = min(max(x0, 0), W - 2);
    const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const pair4 a = *reinterpret_cast<const pair4 *>(g + MPC_IDX((size_t)ya * W + xa, (long long)H * W - 1));
    const pair4 b = *reinterpret_cast<const pair4 *>(g + MPC_IDX((size_t)yb * W + xa, (long long)H * W - 1));
    const bool lo = (xa == x0);                 // column x0 is the first element of the pair
    const float g00 = (r0 && c0) ? (lo ? a.x : a.y) : 0.f;
    const float g01 = (r0 && c1) ? (lo ? a.y : a.x) : 0.f;
    const float g10 = (r1 && c0) ? (lo ? b.x : b.y) : 0.f;
    const float g11 = (r1 && c1) ? (lo ? b.y : b.x) : 0.f;
    gy = w * ((1.f - fx) * (g10 - g00) + fx * (g11 - g01));
    gx = w * ((1.f - fy) * (g01 - g00) + fy * (g11 - g10));
}

// grid NBk, 512 threads, dynamic LDS = CSR * wq * 2 * 8 bytes.
// glut = grad_out * (GCOEF * sum + add_term): the smoothness gradient is folded in here.
// ORDERED: the events were ordered by mpc_event_bucket_order -- the rows of bucket (bin, LUT strip) of each polarity
// block are contiguous (offsets table) -- so the kernel reads the event rows themselves and redoes the warp instead of
// reading records that the forward would have had to write (section 7 of DESIGN.md).
template <bool ORDERED>
__global__ __launch_bounds__(ORDERED ? EV_LUT_THREADS_ORD : EV_LUT_THREADS, ORDERED ? 1 : EV_LUT_MINWAVES) void k_lut_accum(const mpc_shape s, const BinLayout L,
                                                    const float *__restrict__ gimg,
                                                    const float *__restrict__ scal,
                                                    const float *__restrict__ grad_out,
                                                    float *__restrict__ glut,
                                                    const float *__restrict__ add_term,
                                                    const float *__restrict__ events, const float *__restrict__ lut,
                                                    const float *__restrict__ t_ref, const int *__restrict__ offsets,
                                                    const KnnReachJob job) {
    extern __shared__ unsigned long long s_acc[];
    LA_STAMP_DECL
    constexpr int NT = ORDERED ? EV_LUT_THREADS_ORD : EV_LUT_THREADS, NIF = ORDERED ? EV_LUT_INFLIGHT_ORD : EV_LUT_INFLIGHT;
    const EvParams p = make_params(s);
    const int tid = threadIdx.x;
    // logical order (sample, LUT strip, bin): the bins of one strip read the same adjoint-image rows
    const int lg = xcd_swizzle(blockIdx.x, L.NBk);
    if (lg >= L.NBk) return;
    const int b = lg / (L.NCS * p.nb), rem = lg - b * (L.NCS * p.nb);
    const int cst = rem / p.nb, it = rem - cst * p.nb;
    const int bt = b * p.nb + it;
    const int g = bt * L.NCS + cst;                   // bucket id: (b*nb + it)*NCS + cstrip
    const int crow0 = cst * L.CSR, crow1 = min(crow0 + L.CSR, p.hq);
    const int ncell = (crow1 - crow0) * p.wq;
    // mpc_focus_bwd: the KNN backward follows on the stream and needs the reach of every 16x16 tile of every (sample, bin); the
    // first workgroup of each (sample, bin) here works it out on the side (knn_device.h) -- this kernel leaves the vector pipes
    // almost idle, and the gather that follows loses two dependent round trips per workgroup (132 -> 117 us at C3)
    if (job.on && cst == 0) {
        if (job.p.l1) knn_reach_slice<true>(job.p, job.tile_dkmax, job.reach, bt, job.gx, job.gy, job.bd, reinterpret_cast<float *>(s_acc));
        else knn_reach_slice<false>(job.p, job.tile_dkmax, job.reach, bt, job.gx, job.gy, job.bd, reinterpret_cast<float *>(s_acc));
        __syncthreads();
    }
    for (int i = tid; i < 2 * ncell; i += NT) s_acc[i] = 0ull;
    // ORDERED: the bucket's strip of the table goes to LDS (coalesced, while the event rows are on their way), so the
    // warp costs no dependent round trip of its own
    float2 *s_lut = reinterpret_cast<float2 *>(s_acc + 2 * ncell);
    if (ORDERED) {
        const float2 *src = reinterpret_cast<const float2 *>(lut) + ((size_t)(b * p.nb + it) * p.hq + crow0) * p.wq;
        for (int i = tid; i < ncell; i += NT) s_lut[i] = src[i];
    }
    __syncthreads();
    LA_STAMP(1);
    const bool valid = ORDERED || L.gcount[L.NF + L.NBk + 2] == EV_MARKER;
    const int NK = p.nb * L.NCS, key = it * L.NCS + cst;
    int n, n_pos = 0, o_pos = 0, o_neg = 0;
    if (ORDERED) {
        const int *ob = offsets + (size_t)b * 2 * (NK + 1);
        // (clamped: a table that does not belong to this tensor must not make the kernel read outside it)
        o_pos = min(max(ob[key], 0), p.M); n_pos = min(max(ob[key + 1] - o_pos, 0), p.M - o_pos);
        o_neg = min(max(ob[NK + 1 + key], 0), p.M);
        n = n_pos + min(max(ob[NK + 1 + key + 1] - o_neg, 0), p.M - o_neg);
    } else n = valid ? L.gcount[L.NF + g] : 0;
    LA_STAMP(2);
    // the bucket's records: behind those of the sample's buckets before it (first records: ev_prefix_block)
    const float4 *rec = L.brec;
    if (!ORDERED) rec = L.exact ? L.brec + (size_t)b * L.bcap + L.bcapcnt[L.NBk + (size_t)b * NK + key] : L.brec + (size_t)g * L.bcap;
    const float tref = (ORDERED && (p.flags & MPC_F_SCALE_BY_DT)) ? t_ref[0] : 0.f;
    // record r of this bucket: from the forward's list, or rebuilt from event row r of the ordered tensor
    auto fetch = [&](int r) -> float4 {
        if (!ORDERED) return rec[MPC_IDX(r, L.bcap)];
        const int pol = r >= n_pos ? 1 : 0;
        const int row = pol ? o_neg + (r - n_pos) : o_pos + r;
        float e[6];
        load_event(events, (size_t)b * p.M + row, e);
        Warped o;
        warp_cell(p, e, b, 0, o);
        // (an ordered tensor puts every row into the bucket of its own cell; a row that does not belong here -- a
        // foreign tensor passed with offsets -- contributes nothing)
        const bool here = o.it == it && o.iy >= crow0 && o.iy < crow1;
        const int cell = here ? (o.iy - crow0) * p.wq + o.ix : 0;
        const bool mine = warp_event_with(p, e, s_lut[MPC_IDX(cell, ncell)], tref, o) && here;
        const unsigned aux = ((unsigned)((p.P == 2) ? pol : 0) << 31) | (unsigned)cell;
        return make_float4(o.y, o.x, mine ? o.w : 0.f, __uint_as_float(aux));
    };
    // four records per thread in flight: their adjoint-image gathers (the latency of this kernel) overlap
    for (int r0 = tid; r0 < n; r0 += NIF * NT) {
        float4 e[NIF];
#pragma unroll
        for (int u = 0; u < NIF; ++u) e[u] = fetch(min(r0 + u * NT, n - 1));
        LA_STAMP(3);
        float gy[NIF], gx[NIF];
#pragma unroll
        for (int u = 0; u < NIF; ++u) {
            const int pol = (int)(__float_as_uint(e[u].w) >> 31);
            record_grad(e[u].x, e[u].y, e[u].z, gimg + (size_t)(b * p.P + pol) * p.H * p.W, p.H, p.W, gy[u], gx[u]);
        }
        LA_STAMP(4);
#pragma unroll
        for (int u = 0; u < NIF; ++u) {
            if (r0 + u * NT < n && (!ORDERED || e[u].z != 0.f)) {
                const int cell = (int)(__float_as_uint(e[u].w) & 0x7fffffffu);
                atomicAdd(&s_acc[MPC_IDX(2 * cell, 2 * ncell)], (unsigned long long)ev_to_fixed(gy[u]));
                atomicAdd(&s_acc[MPC_IDX(2 * cell + 1, 2 * ncell)], (unsigned long long)ev_to_fixed(gx[u]));
            }
        }
    }
    __syncthreads();
    LA_STAMP(5);
    const float gout = grad_out ? grad_out[0] : 1.f;
    const float coef = valid ? scal[MPC_SCAL_GCOEF] * gout : __int_as_float(0x7fc00000);
    float2 *dst = reinterpret_cast<float2 *>(glut) + ((size_t)bt * p.hq + crow0) * p.wq;
    const float2 *add = add_term ? reinterpret_cast<const float2 *>(add_term) + ((size_t)bt * p.hq + crow0) * p.wq : nullptr;
    for (int i = tid; i < ncell; i += NT) {
        float2 v = make_float2(coef * ev_from_fixed((long long)s_acc[2 * i]), coef * ev_from_fixed((long long)s_acc[2 * i + 1]));
        if (add) { const float2 o = add[i]; v.x += gout * o.x; v.y += gout * o.y; }
        dst[i] = v;
    }
    LA_STAMP_WRITE(tid, dst, n);
}

// the counting pass of ev_count_device.h as a launch of its own (stage entry points; the fused forward lets spare
// workgroups of the KNN strip kernel do it).  grid ev_count_blocks(a), 256 threads, dynamic LDS nb * NCS ints
__global__ __launch_bounds__(256) void k_ev_count(const EvCountArgs a) {
    extern __shared__ int s_ec[];
    ev_count_block(a, blockIdx.x, s_ec);
}
// grid B, 256 threads
__global__ __launch_bounds__(256) void k_ev_prefix(const EvCountArgs a) {
    __shared__ int s_tmp[4];
    ev_prefix_block(a, blockIdx.x, s_tmp);
}

static BinLayout bin_layout(const mpc_shape *s, const mpc_ws_layout &L, void *ws) {
    BinLayout B;
    B.SR = L.strip_rows; B.NS = L.n_strips; B.CSR = L.cstrip_rows; B.NCS = L.n_cstrips;
    B.NF = L.nfb; B.NBk = L.nbb; B.fcap = L.fcap; B.bcap = L.bcap; B.P = L.P; B.exact = L.b_exact;
    B.gcount = (int *)((char *)ws + L.off_fcount);
    B.bcapcnt = B.gcount + L.nfb + L.nbb + 8;
    B.frec = (float4 *)((char *)ws + L.off_frec);
    B.brec = (float4 *)((char *)ws + L.off_brec);
    return B;
}

// what the counting pass needs (null events: nothing to count -- no tiled path, no backward records wanted)
EvCountArgs mpc_event_count_args(const mpc_shape *s, const float *events, void *ws) {
    EvCountArgs a{};
    const mpc_ws_layout L = mpc_layout(s);
    const bool tiled = !(s->flags & MPC_F_ATOMIC_PATH) && s->T == 1 && L.strip_rows > 0 && L.cstrip_rows > 0;
    if (!tiled || !L.b_exact || (s->flags & (MPC_F_NO_WARP | MPC_F_NO_BWD_RECORDS)) || s->B <= 0 || s->M <= 0 || !events) return a;
    a.events = events;
    a.cap = (int *)((char *)ws + L.off_fcount) + L.nfb + L.nbb + 8;
    a.B = s->B; a.M = s->M; a.nb = s->nb; a.sp = s->sp; a.hq = s->hq; a.CSR = L.cstrip_rows; a.NCS = L.n_cstrips;
    return a;
}

static bool use_tiled(const mpc_shape *s, const mpc_ws_layout &L) { return !(s->flags & MPC_F_ATOMIC_PATH) && s->T == 1 && L.strip_rows > 0 && L.cstrip_rows > 0; }

static int set_max_lds_ev(const void *fn, const char *who) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) { mpc_set_error("%s: %s", who, hipGetErrorString(e)); return (int)e; }
    return 0;
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
extern "C" int mpc_event_splat_fwd(const mpc_shape *s, const float *events, const float *flow_lut,
                                   const float *t_ref, float *iwe_raw, void *ws, void *stream) {
    return mpc_event_splat_fwd_ex(s, events, flow_lut, t_ref, iwe_raw, ws, stream, 0, nullptr);
}

static int splat_fwd_impl(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                          float *iwe_raw, void *ws, void *stream, int counters_zeroed, bool fixed, const int32_t *offsets = nullptr);

// Event-axis sharding ("optional finer split" for batches smaller than the number of ranks): the raw IWE of
// THIS rank's events as the Q33.30 accumulators themselves; integer partial images sum exactly, so an all-reduce(SUM) of
// them followed by mpc_iwe_from_fixed gives, bit for bit, the image a single rank computes from all the events.
extern "C" int mpc_event_splat_fwd_fixed(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                                         int64_t *iwe_fixed, void *ws, void *stream) {
    MPC_CHECK_ARG(s && iwe_fixed && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(!(s->flags & MPC_F_ATOMIC_PATH) && s->T == 1, MPC_E_UNSUPPORTED, "fixed-point images come from the LDS-tiled path (num_tref == 1)");
    return splat_fwd_impl(s, events, flow_lut, t_ref, reinterpret_cast<float *>(iwe_fixed), ws, stream, 0, true);
}

extern "C" int mpc_iwe_from_fixed(const int64_t *iwe_fixed, float *iwe_raw, int64_t count, void *stream) {
    MPC_CHECK_ARG((iwe_fixed && iwe_raw) || count == 0, MPC_E_NULL, "null argument");
    if (count <= 0) return 0;
    const int64_t blocks = (count + 255) / 256 < 8192 ? (count + 255) / 256 : 8192;
    MPC_LAUNCH(k_iwe_from_fixed, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long *>(iwe_fixed), iwe_raw, (size_t)count);
    MPC_CHECK_LAUNCH();
    return 0;
}

int mpc_event_splat_fwd_ex(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                           float *iwe_raw, void *ws, void *stream, int counters_zeroed, const int32_t *offsets) {
    return splat_fwd_impl(s, events, flow_lut, t_ref, iwe_raw, ws, stream, counters_zeroed, false, offsets);
}

static int splat_fwd_impl(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                          float *iwe_raw, void *ws, void *stream, int counters_zeroed, bool fixed, const int32_t *offsets) {
    MPC_CHECK_ARG(s && iwe_raw && ws && (events || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG((s->flags & MPC_F_NO_WARP) || flow_lut, MPC_E_NULL, "flow_lut is null");
    MPC_CHECK_ARG(!(s->flags & MPC_F_SCALE_BY_DT) || t_ref, MPC_E_NULL, "t_ref is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    if (use_tiled(s, L)) {
        static mpc_device_once attr_once;   // raising the dynamic-LDS cap: idempotent, once per device
        if (attr_once.need()) {
            if ((rc = set_max_lds_ev((const void *)k_iwe_accum<false>, __func__))) return rc;
            if ((rc = set_max_lds_ev((const void *)k_iwe_accum<true>, __func__))) return rc;
            if ((rc = set_max_lds_ev((const void *)k_lut_accum<false>, __func__))) return rc;
            if ((rc = set_max_lds_ev((const void *)k_lut_accum<true>, __func__))) return rc;
            attr_once.mark();
        }
        const BinLayout BL = bin_layout(s, L, ws);
        const int want_bwd = (s->flags & (MPC_F_NO_WARP | MPC_F_NO_BWD_RECORDS)) ? 0 : 1;
        // counters_zeroed: bit 0 = the counters were zeroed, bit 1 = the backward capacities were counted (mpc_focus_fwd: by the
        // KNN forward's kernels)
        if (!(counters_zeroed & 1) && (rc = mpc_zero_async(BL.gcount, (size_t)(L.nfb + 2 * L.nbb + 8) * sizeof(int), st))) return rc;
        if (s->B > 0 && s->M > 0) {
            const int nblk = mpc_cdiv(s->M, 256 * EV_PER_THREAD) * s->B;
            const dim3 grid(((nblk + 7) / 8) * 8);
            if (want_bwd && L.b_exact && !(counters_zeroed & 2)) {
                const EvCountArgs ca = mpc_event_count_args(s, events, ws);
                MPC_LAUNCH(k_ev_count, dim3(ev_count_blocks(ca)), dim3(256), (size_t)s->nb * L.n_cstrips * sizeof(int), st, ca);
                MPC_LAUNCH(k_ev_prefix, dim3(s->B), dim3(256), 0, st, ca);
                MPC_CHECK_LAUNCH();
            }
            const int nloc = L.P * L.n_strips + s->nb * L.n_cstrips;
            const size_t lds = (size_t)(((3 * nloc + 1 + EV_STAGE / 2) + 3) & ~3) * sizeof(int) + (size_t)EV_STAGE * 16;
            MPC_LAUNCH(k_ev_bin, grid, dim3(256), lds, st, *s, BL, events, flow_lut, t_ref, want_bwd, want_bwd ? nullptr : offsets);
            MPC_CHECK_LAUNCH();
        }
        if (L.nfb > 0) {
            if (fixed) MPC_LAUNCH(k_iwe_accum<true>, dim3(L.nfb), dim3(1024), (size_t)L.strip_rows * s->W * 8, st, BL, iwe_raw, s->H, s->W);
            else MPC_LAUNCH(k_iwe_accum<false>, dim3(L.nfb), dim3(1024), (size_t)L.strip_rows * s->W * 8, st, BL, iwe_raw, s->H, s->W);
            MPC_CHECK_LAUNCH();
        }
        return 0;
    }
    MPC_CHECK_ARG(!fixed, MPC_E_UNSUPPORTED, "fixed-point images need the LDS-tiled path");
    const size_t img_bytes = (size_t)L.nimg * s->H * s->W * sizeof(float);
    const int e = mpc_zero_async(iwe_raw, img_bytes, st);
    if (e) return e;
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    MPC_LAUNCH(k_splat_fwd_atomic, dim3(grid), dim3(256), 0, st, *s, events, flow_lut, t_ref, iwe_raw);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_event_splat_bwd(const mpc_shape *s, const float *events, const float *flow_lut,
                                   const float *t_ref, const float *grad_iwe, const float *scal,
                                   const float *grad_out, float *grad_flow_lut, const float *add_term,
                                   void *ws, void *stream) {
    return mpc_event_splat_bwd_ordered(s, events, nullptr, flow_lut, t_ref, grad_iwe, scal, grad_out, grad_flow_lut, add_term, ws, stream);
}

extern "C" int mpc_event_splat_bwd_ordered(const mpc_shape *s, const float *events, const int32_t *offsets, const float *flow_lut,
                                           const float *t_ref, const float *grad_iwe, const float *scal,
                                           const float *grad_out, float *grad_flow_lut, const float *add_term,
                                           void *ws, void *stream) {
    return mpc_event_splat_bwd_job(s, events, offsets, flow_lut, t_ref, grad_iwe, scal, grad_out, grad_flow_lut, add_term, ws, stream, nullptr, nullptr);
}

// knn_state != nullptr (mpc_focus_bwd): k_lut_accum also computes the tile reaches of the KNN backward that follows; *reach_done
// tells the caller whether it did
int mpc_event_splat_bwd_job(const mpc_shape *s, const float *events, const int32_t *offsets, const float *flow_lut,
                            const float *t_ref, const float *grad_iwe, const float *scal,
                            const float *grad_out, float *grad_flow_lut, const float *add_term,
                            void *ws, void *stream, const float *knn_state, int *reach_done) {
    if (reach_done) *reach_done = 0;
    MPC_CHECK_ARG(s && flow_lut && grad_iwe && scal && grad_flow_lut && ws && (events || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(!(s->flags & MPC_F_NO_WARP), MPC_E_UNSUPPORTED, "no LUT to differentiate with MPC_F_NO_WARP");
    MPC_CHECK_ARG(!(s->flags & MPC_F_SCALE_BY_DT) || t_ref, MPC_E_NULL, "t_ref is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const mpc_ws_layout L = mpc_layout(s);
    if (use_tiled(s, L)) {
        const BinLayout BL = bin_layout(s, L, ws);
        // the records must come from mpc_event_splat_fwd on this same workspace: the kernel checks
        // the marker that call left behind and poisons the output with NaN if it is missing
        if (L.nbb > 0) {
            MPC_CHECK_ARG(!offsets || (size_t)L.cstrip_rows * s->wq * 24 <= 160 * 1024 - 512, MPC_E_UNSUPPORTED,
                          "LUT too wide for the ordered backward (use mpc_event_splat_bwd)");
            KnnReachJob job{};
            const size_t lds_acc = (size_t)L.cstrip_rows * s->wq * (offsets ? 24 : 16);
            if (knn_state && mpc_knn_reach_job(s, knn_state, ws, &job) &&
                ((size_t)job.gx * job.gy * (KNN_NCLS + 1) + 16) * sizeof(float) > lds_acc) job.on = 0;      // (does not fit this kernel's LDS)
            if (offsets)
                MPC_LAUNCH(k_lut_accum<true>, dim3(((L.nbb + 7) / 8) * 8), dim3(EV_LUT_THREADS_ORD), lds_acc, st, *s, BL,
                                   grad_iwe, scal, grad_out, grad_flow_lut, add_term, events, flow_lut, t_ref, offsets, job);
            else
                MPC_LAUNCH(k_lut_accum<false>, dim3(((L.nbb + 7) / 8) * 8), dim3(EV_LUT_THREADS), lds_acc, st, *s, BL,
                                   grad_iwe, scal, grad_out, grad_flow_lut, add_term, events, flow_lut, t_ref, offsets, job);
            MPC_CHECK_LAUNCH();
            if (reach_done) *reach_done = job.on;
        }
        return 0;
    }
    {
        const int64_t cnt = (int64_t)s->B * s->nb * s->hq * s->wq * s->T * 2;
        if (add_term) {
            rc = grad_out ? mpc_scale(add_term, grad_out, grad_flow_lut, cnt, stream) : (int)hipMemcpyAsync(grad_flow_lut, add_term, cnt * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (rc) return rc;
        } else {
            const int e = mpc_zero_async(grad_flow_lut, cnt * sizeof(float), st);
            if (e) return e;
        }
    }
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    MPC_LAUNCH(k_splat_bwd_atomic, dim3(grid), dim3(256), 0, st, *s, events, flow_lut, t_ref,
                       grad_iwe, scal, grad_out, grad_flow_lut);
    MPC_CHECK_LAUNCH();
    return 0;
}


// ---- UNPINNED EXTENSION, fused form (FocusLoss.calc_per_event_basis): the per-event continuous-time basis warp itself and the
// chain from the position gradient back to the tile coefficients, one pass over the events each.
//   k_pe_warp : rows_out[b][i] = (y + sum_k c[cell][0][k] phi[b][i][k], x + sum_k c[cell][1][k] phi[b][i][k], t, p, cell, valid)
//               cell = LUT cell of the event's own position (focus.py:186-187), kept in column 4 (its bits) for the backward --
//               with MPC_F_NO_WARP nothing else reads that column; coef [B*hq*wq][2][k], phi [B][M][k] (basis_k(t_ref) - basis_k(t))
//   k_pe_grad : grad_coef[cell][d][k] += phi[b][i][k] * d objective / d warped position_d  (float atomics: the gradient of this
//               extension is not bitwise reproducible; the forward is)
#define PE_KMAX 8          // orders held in registers (beyond: the generic instantiation reads phi per use)
// phi[j] = basis_j(t_ref) - basis_j(t), j < k, from the tensor, or -- phi == nullptr: the polynomial basis t^(j+1) (basis.py:26-27) --
// worked out here (one multiply per order instead of 4 k bytes of traffic per event and kernel)
template <int KB>
__device__ __forceinline__ void pe_phi(const float *__restrict__ phi, size_t row, int k, float t, float tref, float *out /* [KB or min(k, 8)] */) {
    const int kk = KB > 0 ? KB : min(k, PE_KMAX);       // (k: the row stride of phi; the first PE_KMAX orders go to registers)
    if (phi != nullptr) { for (int j = 0; j < kk; ++j) out[j] = phi[row * k + j]; return; }
    float a = tref, c = t;
    for (int j = 0; j < kk; ++j) { out[j] = a - c; a *= tref; c *= t; }
}

template <int KB>
__global__ __launch_bounds__(256) void k_pe_warp(const mpc_shape s, const float *__restrict__ events, const float *__restrict__ coef,
                                                 const float *__restrict__ phi, int k, const float *__restrict__ t_ref,
                                                 float *__restrict__ rows_out) {
    const EvParams p = make_params(s);
    const float tref = t_ref ? t_ref[0] : 0.f;
    const size_t total = (size_t)p.B * p.M;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / p.M);
        float e[6];
        load_event(events, row, e);
        const int iy = min(max((int)floorf(mpc_div_sp(e[0], p.sp)), 0), p.hq - 1), ix = min(max((int)floorf(mpc_div_sp(e[1], p.sp)), 0), p.wq - 1);
        const int cell = (b * p.hq + iy) * p.wq + ix;
        const float *c = coef + (size_t)cell * 2 * k;
        float fy = 0.f, fx = 0.f;
        float ph[KB > 0 ? KB : PE_KMAX];
        pe_phi<KB>(phi, row, k, e[2], tref, ph);
        if (KB > 0) {
#pragma unroll
            for (int j = 0; j < KB; ++j) { fy += c[j] * ph[j]; fx += c[KB + j] * ph[j]; }
        } else {
            for (int j = 0; j < k; ++j) { const float f = j < PE_KMAX ? ph[j] : phi[row * k + j]; fy += c[j] * f; fx += c[k + j] * f; }
        }
        float2 *o = reinterpret_cast<float2 *>(rows_out + row * 6);
        o[0] = make_float2(e[0] + fy, e[1] + fx);
        o[1] = make_float2(e[2], e[3]);
        o[2] = make_float2(__int_as_float(cell), e[5]);
    }
}

template <int KB>
__global__ __launch_bounds__(256) void k_pe_grad(const mpc_shape s, const float *__restrict__ rows, const float *__restrict__ phi, int k,
                                                 const float *__restrict__ t_ref, const float *__restrict__ gimg,
                                                 const float *__restrict__ scal, const float *__restrict__ grad_out,
                                                 float *__restrict__ gcoef) {
    const EvParams p = make_params(s);
    const float coef = scal[MPC_SCAL_GCOEF] * (grad_out ? grad_out[0] : 1.f);
    const float tref = t_ref[0];
    const size_t total = (size_t)p.B * p.M;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / p.M), i = (int)(row - (size_t)b * p.M);
        float e[6];
        load_event(rows, row, e);
        const int pol = (p.P == 2 && i >= p.Mp) ? 1 : 0;
        Warped o;
        if (!warp_event_with(p, e, make_float2(0.f, 0.f), tref, o)) continue;
        float gy, gx;
        event_pos_grad(p, o, gimg + ((size_t)b * p.P + pol) * p.H * p.W, gy, gx);
        gy *= coef; gx *= coef;
        if (gy == 0.f && gx == 0.f) continue;
        const int cell = __float_as_int(e[4]);
        float *g = gcoef + (size_t)cell * 2 * k;
        const int kk = KB > 0 ? KB : k;
        float ph[KB > 0 ? KB : PE_KMAX];
        pe_phi<KB>(phi, row, k, e[2], tref, ph);
        for (int j = 0; j < kk; ++j) { const float f = j < PE_KMAX ? ph[j] : phi[row * k + j]; atomicAdd(g + j, f * gy); atomicAdd(g + kk + j, f * gx); }
    }
}

//   k_pe_accum: the same gradient WITHOUT global atomics, for bucket-ordered events (mpc_event_bucket_order / ingest: the rows of a
//               (sample, polarity, bin, LUT strip) are contiguous, `offsets`): one workgroup per (sample, LUT strip) walks the
//               2 * nb row ranges of its strip, gathers the adjoint-image taps of every row and adds phi * gradient to the strip's
//               [cell][2][k] accumulators in LDS -- 64-bit fixed point like k_lut_accum: integer sums, bitwise reproducible -- and
//               writes every cell of the strip once (global float atomics ran at ~20 G/s on this chip: 0.83 ms for the 17 M of a C3 step)
#define PE_NT 1024
#define PE_NIF 4
template <int KB>
__global__ __launch_bounds__(PE_NT) void k_pe_accum(const mpc_shape s, int CSR, int NCS, int SPLIT, const int *__restrict__ offsets,
                                                    const float *__restrict__ rows, const float *__restrict__ phi, int k,
                                                    const float *__restrict__ t_ref, const float *__restrict__ gimg,
                                                    const float *__restrict__ scal, const float *__restrict__ grad_out,
                                                    float *__restrict__ gcoef) {
    extern __shared__ unsigned long long s_pacc[];
    __shared__ int s_r0[2 * 64 + 1], s_pre[2 * 64 + 1];          // first row / running count of the 2 * nb ranges (nb <= 64)
    const EvParams p = make_params(s);
    const int tid = threadIdx.x, kk = KB > 0 ? KB : k;
    // (SPLIT workgroups per (sample, strip), each with every SPLIT-th of the strip's row ranges and a partial output of its own:
    // B * NCS workgroups -- 98 at the DSEC batch -- left most of the CUs idle)
    const int part = blockIdx.x % SPLIT, bs = blockIdx.x / SPLIT;
    const int b = bs / NCS, cst = bs - b * NCS;
    const int crow0 = cst * CSR, crow1 = min(crow0 + CSR, p.hq), ncell = (crow1 - crow0) * p.wq;
    const int cell0 = (b * p.hq + crow0) * p.wq;
    const int NK = p.nb * NCS, nrange = 2 * p.nb;
    for (int i = tid; i < ncell * 2 * kk; i += PE_NT) s_pacc[i] = 0ull;
    if (tid == 0) {
        int run = 0;
        for (int j = 0; j < nrange; ++j) {
            const int pol = j / p.nb, it = j - pol * p.nb;
            const int *o = offsets + (size_t)(b * 2 + pol) * (NK + 1) + it * NCS + cst;
            // (a table that does not belong to the tensor: clamp into the sample, never read outside it)
            const int r0 = min(max(o[0], 0), p.M), r1 = (j % SPLIT == part) ? min(max(o[1], r0), p.M) : r0;
            s_r0[j] = r0; s_pre[j] = run; run += r1 - r0;
        }
        s_pre[nrange] = run;
    }
    __syncthreads();
    const int n = s_pre[nrange];
    const float tref = t_ref[0];
    for (int q0 = tid; q0 < n; q0 += PE_NIF * PE_NT) {
        float e[PE_NIF][6];
        size_t grow[PE_NIF];
        bool on[PE_NIF];
        int pol[PE_NIF];
#pragma unroll
        for (int u = 0; u < PE_NIF; ++u) {
            const int q = q0 + u * PE_NT;
            on[u] = q < n;
            int j = 0;
            if (on[u]) { int lo = 0, hi = nrange; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= q) lo = mid; else hi = mid; } j = lo; }
            const int row = on[u] ? s_r0[j] + (q - s_pre[j]) : 0;
            pol[u] = (p.P == 2) ? (row >= p.Mp ? 1 : 0) : 0;
            grow[u] = (size_t)b * p.M + row;
            load_event(rows, grow[u], e[u]);
        }
        float gy[PE_NIF], gx[PE_NIF];
#pragma unroll
        for (int u = 0; u < PE_NIF; ++u) {
            Warped o;
            gy[u] = gx[u] = 0.f;
            if (on[u] && warp_event_with(p, e[u], make_float2(0.f, 0.f), tref, o))
                event_pos_grad(p, o, gimg + ((size_t)b * p.P + pol[u]) * p.H * p.W, gy[u], gx[u]);
        }
#pragma unroll
        for (int u = 0; u < PE_NIF; ++u) {
            const int lc = __float_as_int(e[u][4]) - cell0;
            MPC_EXPECT(!on[u] || (gy[u] == 0.f && gx[u] == 0.f) || (lc >= 0 && lc < ncell));      // (a row outside its strip: the offsets table is not this tensor's)
            if (!on[u] || lc < 0 || lc >= ncell || (gy[u] == 0.f && gx[u] == 0.f)) continue;
            float ph[KB > 0 ? KB : PE_KMAX];
            pe_phi<KB>(phi, grow[u], k, e[u][2], tref, ph);
            unsigned long long *a = s_pacc + (size_t)lc * 2 * kk;
            for (int j = 0; j < kk; ++j) {
                const float f = j < PE_KMAX ? ph[j] : phi[grow[u] * k + j];
                atomicAdd(a + j, (unsigned long long)ev_to_fixed(f * gy[u]));
                atomicAdd(a + kk + j, (unsigned long long)ev_to_fixed(f * gx[u]));
            }
        }
    }
    __syncthreads();
    const float coef = scal[MPC_SCAL_GCOEF] * (grad_out ? grad_out[0] : 1.f);
    float *dst = gcoef + ((size_t)part * p.B * p.hq * p.wq + cell0) * 2 * kk;
    for (int i = tid; i < ncell * 2 * kk; i += PE_NT) dst[i] = coef * ev_from_fixed((long long)s_pacc[i]);
}

extern "C" int mpc_pe_warp(const mpc_shape *s, const float *events, const float *coef_rows, const float *phi, int32_t k,
                           const float *t_ref, float *rows_out, void *stream) {
    MPC_CHECK_ARG(s && coef_rows && (phi || t_ref) && rows_out && (events || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(k >= 1 && k <= 64 && (phi || k <= PE_KMAX), MPC_E_SHAPE, "1 <= num_basis <= 64 (<= 8 for the built-in polynomial basis)");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t st = (hipStream_t)stream;
    if (k == 1) MPC_LAUNCH(k_pe_warp<1>, dim3(grid), dim3(256), 0, st, *s, events, coef_rows, phi, k, t_ref, rows_out);
    else if (k == 3) MPC_LAUNCH(k_pe_warp<3>, dim3(grid), dim3(256), 0, st, *s, events, coef_rows, phi, k, t_ref, rows_out);
    else MPC_LAUNCH(k_pe_warp<0>, dim3(grid), dim3(256), 0, st, *s, events, coef_rows, phi, k, t_ref, rows_out);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_pe_grad(const mpc_shape *s, const float *rows, const float *phi, int32_t k, const float *t_ref, const float *grad_iwe,
                           const float *scal, const float *grad_out, float *grad_coef_rows, void *stream) {
    MPC_CHECK_ARG(s && (phi || t_ref) && grad_iwe && scal && grad_coef_rows && (rows || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG((s->flags & MPC_F_NO_WARP) && s->T == 1, MPC_E_UNSUPPORTED, "mpc_pe_grad takes the rows of mpc_pe_warp (MPC_F_NO_WARP), num_tref == 1");
    MPC_CHECK_ARG(t_ref, MPC_E_NULL, "t_ref is null");
    MPC_CHECK_ARG(k >= 1 && k <= 64 && (phi || k <= PE_KMAX), MPC_E_SHAPE, "1 <= num_basis <= 64 (<= 8 for the built-in polynomial basis)");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int e0 = mpc_zero_async(grad_coef_rows, (size_t)s->B * s->hq * s->wq * 2 * k * sizeof(float), st);
    if (e0) return e0;
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (k == 1) MPC_LAUNCH(k_pe_grad<1>, dim3(grid), dim3(256), 0, st, *s, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    else if (k == 3) MPC_LAUNCH(k_pe_grad<3>, dim3(grid), dim3(256), 0, st, *s, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    else MPC_LAUNCH(k_pe_grad<0>, dim3(grid), dim3(256), 0, st, *s, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    MPC_CHECK_LAUNCH();
    return 0;
}

// 1 where mpc_pe_grad_ordered serves (shape, k) -- a LUT strip's [cell][2][k] accumulators fit the LDS, num_bins <= 64 --, 0 where
// the caller has to take mpc_pe_grad (the one copy of the rule: the Python side asks instead of restating it)
extern "C" int32_t mpc_pe_grad_ordered_supported(const mpc_shape *s, int32_t k) {
    if (!s || mpc_validate_shape(s)) return 0;
    if (!(s->flags & MPC_F_NO_WARP) || s->T != 1 || k < 1 || k > 64 || s->nb > 64 || s->B <= 0) return 0;
    const mpc_ws_layout L = mpc_layout(s);
    if (L.n_cstrips <= 0) return 0;
    return (size_t)L.cstrip_rows * s->wq * 2 * k * 8 <= 150 * 1024 ? 1 : 0;
}

extern "C" int mpc_pe_grad_ordered(const mpc_shape *s, const float *rows, const int32_t *offsets, const float *phi, int32_t k,
                                   const float *t_ref, const float *grad_iwe, const float *scal, const float *grad_out,
                                   float *grad_coef_rows, int32_t split, void *stream) {
    MPC_CHECK_ARG(s && offsets && (phi || t_ref) && grad_iwe && scal && grad_coef_rows && (rows || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG((s->flags & MPC_F_NO_WARP) && s->T == 1, MPC_E_UNSUPPORTED, "mpc_pe_grad_ordered takes the rows of mpc_pe_warp (MPC_F_NO_WARP), num_tref == 1");
    MPC_CHECK_ARG(t_ref, MPC_E_NULL, "t_ref is null");
    MPC_CHECK_ARG(k >= 1 && k <= 64 && (phi || k <= PE_KMAX) && s->nb <= 64, MPC_E_SHAPE, "1 <= num_basis <= 64 (<= 8 built-in polynomial), num_bins <= 64");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    if (s->B == 0) return 0;
    const mpc_ws_layout L = mpc_layout(s);
    MPC_CHECK_ARG(L.n_cstrips > 0, MPC_E_UNSUPPORTED, "no bucketed event layout for this shape");
    const size_t lds = (size_t)L.cstrip_rows * s->wq * 2 * k * 8;
    MPC_CHECK_ARG(lds <= 150 * 1024, MPC_E_UNSUPPORTED, "a LUT strip's accumulators do not fit the LDS (use mpc_pe_grad)");
    hipStream_t st = (hipStream_t)stream;
    static mpc_device_once attr_once;
    if (attr_once.need()) {
        // (these kernels have 1 KB of static LDS of their own: the cap of set_max_lds_ev would pass the CU's 160 KB)
        const void *fns[3] = {(const void *)k_pe_accum<1>, (const void *)k_pe_accum<3>, (const void *)k_pe_accum<0>};
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
        }
        attr_once.mark();
    }
    MPC_CHECK_ARG(split >= 1 && split <= 16, MPC_E_SHAPE, "1 <= split <= 16");
    const dim3 grid(s->B * L.n_cstrips * split);
    if (k == 1) MPC_LAUNCH(k_pe_accum<1>, grid, dim3(PE_NT), lds, st, *s, L.cstrip_rows, L.n_cstrips, split, offsets, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    else if (k == 3) MPC_LAUNCH(k_pe_accum<3>, grid, dim3(PE_NT), lds, st, *s, L.cstrip_rows, L.n_cstrips, split, offsets, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    else MPC_LAUNCH(k_pe_accum<0>, grid, dim3(PE_NT), lds, st, *s, L.cstrip_rows, L.n_cstrips, split, offsets, rows, phi, k, t_ref, grad_iwe, scal, grad_out, grad_coef_rows);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_event_pos_grad(const mpc_shape *s, const float *events, const float *t_ref, const float *grad_iwe,
                                  const float *scal, const float *grad_out, float *grad_pos, void *stream) {
    MPC_CHECK_ARG(s && grad_iwe && scal && grad_pos && (events || s->M == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG((s->flags & MPC_F_NO_WARP) && s->T == 1, MPC_E_UNSUPPORTED, "mpc_event_pos_grad takes rows with warped positions (MPC_F_NO_WARP), num_tref == 1");
    MPC_CHECK_ARG(!(s->flags & MPC_F_SCALE_BY_DT) || t_ref, MPC_E_NULL, "t_ref is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    MPC_LAUNCH(k_event_pos_grad, dim3(grid), dim3(256), 0, (hipStream_t)stream, *s, events, t_ref, grad_iwe, scal, grad_out,
               reinterpret_cast<float2 *>(grad_pos));
    MPC_CHECK_LAUNCH();
    return 0;
}


// ==========================================================================================
// f-1, layout half (reference loader.py:360-415 builds the padded tensor in time order): order the rows of
// every polarity block by (time bin, LUT strip) -- the key of the BACKWARD buckets, which does not depend on the flow.
// Permuting rows inside a polarity block does not change the loss (bit for bit: the accumulators are integers), so the
// ordered tensor is a valid `events` for the reference too; with its offsets table the forward need not write, and the
// backward not read back, a 16-byte record per event.  Padding rows (valid == 0) keep their place at the end of their block.
//   offsets [B][2][nb*NCS + 1]: first row of every (bin, LUT strip) inside the block; [nb*NCS] = first padding row
// Counting sort: per-chunk counts -> scan over the chunks of every key -> scan over the keys -> scatter (the order inside a
// bucket is not defined and need not be).
// ==========================================================================================
#define EVO_ROWS 2048          // rows of one chunk (256 threads x 8)

struct EvoKey { int NCS, CSR, NK; };

__device__ __forceinline__ int evo_key(const EvParams &p, const EvoKey &k, const float e[6]) {
    if (e[5] == 0.f && !(p.flags & MPC_F_UNIT_WEIGHT)) return k.NK;   // padding row (weight 0: it votes for nothing)
    const int it = min(max((int)e[4], 0), p.nb - 1);
    const int iy = min(max((int)floorf(mpc_div_sp(e[0], p.sp)), 0), p.hq - 1);      // as warp_event
    return it * k.NCS + iy / k.CSR;
}

// grid (chunks, 2 * B): counts[b][pol][key][chunk]
__global__ __launch_bounds__(256) void k_evo_count(const mpc_shape s, const EvoKey k, const float *__restrict__ events,
                                                   int *__restrict__ counts, int chunks) {
    extern __shared__ int s_c[];
    const EvParams p = make_params(s);
    const int b = blockIdx.y >> 1, pol = blockIdx.y & 1, chunk = blockIdx.x;
    const int r0 = pol ? p.Mp : 0, r1 = pol ? p.M : p.Mp;
    for (int i = threadIdx.x; i <= k.NK; i += 256) s_c[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < EVO_ROWS; j += 256) {
        const int row = r0 + chunk * EVO_ROWS + j;
        if (row >= r1) break;
        float e[6];
        load_event(events, (size_t)b * p.M + row, e);
        atomicAdd(&s_c[evo_key(p, k, e)], 1);
    }
    __syncthreads();
    int *dst = counts + (size_t)(b * 2 + pol) * (k.NK + 1) * chunks + chunk;
    for (int i = threadIdx.x; i <= k.NK; i += 256) dst[(size_t)i * chunks] = s_c[i];
}

// counts is [b][pol][key][chunk].  Step 1, grid (ceil((NK + 1) / 4), 2 * B), 256 threads: one wavefront per key scans
// its chunks (lanes = chunks; in place -> the chunk's first row inside the key) and leaves the key's total.
__global__ __launch_bounds__(256) void k_evo_scan_chunks(const EvoKey k, int *__restrict__ counts, int *__restrict__ totals, int chunks) {
    const int lane = threadIdx.x & 63, key = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key > k.NK) return;
    int *ci = counts + ((size_t)blockIdx.y * (k.NK + 1) + key) * chunks;
    int run = 0;
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int ch = c0 + lane;
        const int v = ch < chunks ? ci[ch] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (ch < chunks) ci[ch] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0) totals[(size_t)blockIdx.y * (k.NK + 1) + key] = run;
}

// Step 2, grid 2 * B, 256 threads: exclusive scan of the NK + 1 key totals of a polarity block -> offsets (a contiguous
// segment of keys per thread, the 256 segment sums scanned by the first wavefront)
__global__ __launch_bounds__(256) void k_evo_scan_keys(const mpc_shape s, const EvoKey k, const int *__restrict__ totals,
                                                       int *__restrict__ offsets) {
    __shared__ int s_seg[256];
    const EvParams p = make_params(s);
    const int pol = blockIdx.x & 1, lane = threadIdx.x & 63;
    const int *tot = totals + (size_t)blockIdx.x * (k.NK + 1);
    const int per = (k.NK + 1 + 255) / 256;
    const int k0 = min((int)threadIdx.x * per, k.NK + 1), k1 = min(k0 + per, k.NK + 1);
    int sum = 0;
    for (int i = k0; i < k1; ++i) sum += tot[i];
    s_seg[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {
        int carry = 0;
        for (int c0 = 0; c0 < 256; c0 += 64) {
            const int v = s_seg[c0 + lane];
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            s_seg[c0 + lane] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();
    int run = (pol ? p.Mp : 0) + s_seg[threadIdx.x];
    int *o = offsets + (size_t)blockIdx.x * (k.NK + 1);
    for (int i = k0; i < k1; ++i) { o[i] = run; run += tot[i]; }
}

// grid (chunks, 2 * B)
__global__ __launch_bounds__(256) void k_evo_scatter(const mpc_shape s, const EvoKey k, const float *__restrict__ events,
                                                     const int *__restrict__ counts, const int *__restrict__ offsets,
                                                     float *__restrict__ out, int chunks) {
    extern __shared__ int s_c[];
    const EvParams p = make_params(s);
    const int b = blockIdx.y >> 1, pol = blockIdx.y & 1, chunk = blockIdx.x;
    const int r0 = pol ? p.Mp : 0, r1 = pol ? p.M : p.Mp;
    const int *cb = counts + (size_t)(b * 2 + pol) * (k.NK + 1) * chunks + chunk;
    const int *ob = offsets + (size_t)(b * 2 + pol) * (k.NK + 1);
    for (int i = threadIdx.x; i <= k.NK; i += 256) s_c[i] = ob[i] + cb[(size_t)i * chunks];   // first destination row of the chunk's share
    __syncthreads();
    for (int j = threadIdx.x; j < EVO_ROWS; j += 256) {
        const int row = r0 + chunk * EVO_ROWS + j;
        if (row >= r1) break;
        float e[6];
        load_event(events, (size_t)b * p.M + row, e);
        const int dst = atomicAdd(&s_c[evo_key(p, k, e)], 1);
        float2 *d = reinterpret_cast<float2 *>(out + ((size_t)b * p.M + dst) * 6);
        d[0] = make_float2(e[0], e[1]); d[1] = make_float2(e[2], e[3]); d[2] = make_float2(e[4], e[5]);
    }
}

// the two scans of the counting sort, for a counts array [b][pol][key][chunk] someone else filled (ingest.hip)
int mpc_evo_scans(const mpc_shape *loss, int NCS, int CSR, int *kcounts, int *totals, int32_t *offsets, int chunks, hipStream_t st) {
    const EvoKey k{NCS, CSR, loss->nb * NCS};
    MPC_LAUNCH(k_evo_scan_chunks, dim3(mpc_cdiv(k.NK + 1, 4), 2 * loss->B), dim3(256), 0, st, k, kcounts, totals, chunks);
    MPC_LAUNCH(k_evo_scan_keys, dim3(2 * loss->B), dim3(256), 0, st, *loss, k, totals, (int *)offsets);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t mpc_event_lut_strips(const mpc_shape *s) {
    if (!s || mpc_validate_shape(s)) return MPC_E_SHAPE;
    return mpc_layout(s).n_cstrips;
}

extern "C" int64_t mpc_event_order_workspace_bytes(const mpc_shape *s) {
    if (!s || mpc_validate_shape(s)) return MPC_E_SHAPE;
    const mpc_ws_layout L = mpc_layout(s);
    const int64_t chunks = mpc_cdiv(s->M > 0 ? s->M : 1, EVO_ROWS);
    return mpc_align((int64_t)(s->B > 0 ? s->B : 1) * 2 * (chunks + 1) * ((int64_t)s->nb * L.n_cstrips + 1) * 4);      // counts + key totals
}

extern "C" int mpc_event_bucket_order(const mpc_shape *s, const float *events_in, float *events_out, int32_t *offsets,
                                      void *ws, void *stream) {
    MPC_CHECK_ARG(s && ws && ((events_in && events_out) || s->M == 0 || s->B == 0) && (offsets || s->B == 0), MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    MPC_CHECK_ARG(L.n_cstrips > 0, MPC_E_UNSUPPORTED, "no LDS-tiled event path for this shape (num_tref > 1 or the atomic debugging path)");
    if (s->B == 0) return 0;
    // no rows: every bucket is empty and starts at row 0
    if (s->M == 0) return mpc_zero_async(offsets, (size_t)s->B * 2 * ((size_t)s->nb * L.n_cstrips + 1) * sizeof(int32_t), (hipStream_t)stream);
    MPC_CHECK_ARG(events_in != events_out, MPC_E_SHAPE, "in-place ordering is not supported");
    const EvoKey k{L.n_cstrips, L.cstrip_rows, s->nb * L.n_cstrips};
    const int chunks = mpc_cdiv(s->M, EVO_ROWS);       // of the longer block at most; chunks beyond a block's rows are empty
    hipStream_t st = (hipStream_t)stream;
    int *counts = (int *)ws;
    const size_t lds = (size_t)(k.NK + 1) * 4;
    MPC_LAUNCH(k_evo_count, dim3(chunks, 2 * s->B), dim3(256), lds, st, *s, k, events_in, counts, chunks);
    int *totals = counts + (size_t)2 * s->B * (k.NK + 1) * chunks;
    MPC_LAUNCH(k_evo_scan_chunks, dim3(mpc_cdiv(k.NK + 1, 4), 2 * s->B), dim3(256), 0, st, k, counts, totals, chunks);
    MPC_LAUNCH(k_evo_scan_keys, dim3(2 * s->B), dim3(256), 0, st, *s, k, totals, (int *)offsets);
    MPC_LAUNCH(k_evo_scatter, dim3(chunks, 2 * s->B), dim3(256), lds, st, *s, k, events_in, counts, (const int *)offsets,
                       events_out, chunks);
    MPC_CHECK_LAUNCH();
    return 0;
}

MPC_BOUNDS_UNIT("events.hip")
