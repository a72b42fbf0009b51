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
#include <math.h>

#define KNN_BINS 32
#define KNN_SLACK 0.01f   // px, absorbs fp32 rounding of the cell assignment in the ring bound

struct KnnParams {
    int B, nb, T, n, hq, wq, sp, K, G;
    int l1, iwd, want_next;
    float off;   // sp/2 - 0.5 : centre of cell 0 (focus.py:117)
};

static KnnParams knn_params(const mpc_shape *s) {
    KnnParams p;
    p.B = s->B; p.nb = s->nb; p.T = s->T; p.n = s->n; p.hq = s->hq; p.wq = s->wq; p.sp = s->sp;
    p.K = s->K; p.G = s->hq * s->wq;
    p.l1 = (s->flags & MPC_F_DIST_L1) ? 1 : 0;
    p.iwd = ((s->flags & MPC_F_SCHEME_IWD) && s->K > 1) ? 1 : 0;   // focus.py:145-147: K == 1 is a plain gather
    p.want_next = (s->flags & MPC_F_WANT_NEXT) ? 1 : 0;
    p.off = (float)s->sp / 2.f - 0.5f;
    return p;
}

__device__ __forceinline__ int cell_of(float v, int sp, int ncell) {
    // cells are centred on the query points: cell c covers [c*sp - 0.5, (c+1)*sp - 0.5)
    const float c = floorf((v + 0.5f) / (float)sp);
    return (int)fminf(fmaxf(c, 0.f), (float)(ncell - 1));
}

__device__ __forceinline__ float pair_dist(float qy, float qx, float py, float px, int l1) {
    // focus.py:132-135: (grid - traj) ** 2 summed over (y, x), or abs
    const float dy = qy - py, dx = qx - px;
    return l1 ? (fabsf(dy) + fabsf(dx)) : (dy * dy + dx * dx);
}

// ------------------------------------------------------------------------------------------
// bucket the points of one (sample, bin) by cell: counting sort in LDS
// grid B*nb, 1024 threads, dynamic LDS = G * 4 bytes
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_knn_bucket(const KnnParams p, const float *__restrict__ traj,
                                                     int *__restrict__ cell_start,
                                                     float2 *__restrict__ spos, int *__restrict__ sidx,
                                                     unsigned *__restrict__ rmax) {
    extern __shared__ int s_cnt[];
    __shared__ int s_wave[16];
    const int tid = threadIdx.x;
    const int bt = blockIdx.x, b = bt / p.nb, t = bt - b * p.nb;
    const float2 *pts = reinterpret_cast<const float2 *>(traj) + ((size_t)b * (p.T + p.nb) + p.T + t) * p.n;
    if (tid == 0) rmax[bt] = 0u;
    for (int g = tid; g < p.G; g += 1024) s_cnt[g] = 0;
    __syncthreads();
    for (int i = tid; i < p.n; i += 1024) {
        const float2 q = pts[i];
        atomicAdd(&s_cnt[cell_of(q.x, p.sp, p.hq) * p.wq + cell_of(q.y, p.sp, p.wq)], 1);
    }
    __syncthreads();
    // exclusive scan over the G counters: each thread owns a contiguous chunk
    const int chunk = (p.G + 1023) / 1024;
    const int g0 = tid * chunk, g1 = min(g0 + chunk, p.G);
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
    int *cs = cell_start + (size_t)bt * (p.G + 1);
    for (int g = g0; g < g1; ++g) {
        const int c = s_cnt[g];
        s_cnt[g] = run;        // becomes the fill cursor of the cell
        cs[g] = run;
        run += c;
    }
    if (tid == 1023) cs[p.G] = p.n;
    __syncthreads();
    float2 *sp_ = spos + (size_t)bt * p.n;
    int *si_ = sidx + (size_t)bt * p.n;
    for (int i = tid; i < p.n; i += 1024) {
        const float2 q = pts[i];
        const int pos = atomicAdd(&s_cnt[cell_of(q.x, p.sp, p.hq) * p.wq + cell_of(q.y, p.sp, p.wq)], 1);
        sp_[pos] = q;
        si_[pos] = i;
    }
}

// ------------------------------------------------------------------------------------------
// query: one thread per LUT cell; 16x16 cells per workgroup
// grid (ceil(wq/16), ceil(hq/16), B*nb), 256 threads
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_query(const KnnParams p, const float *__restrict__ traj,
                                                   const int *__restrict__ cell_start,
                                                   const float2 *__restrict__ spos,
                                                   const int *__restrict__ sidx,
                                                   float *__restrict__ flow_lut,
                                                   float *__restrict__ flow_next,
                                                   float *__restrict__ knn_state,
                                                   int *__restrict__ idx_out,
                                                   unsigned *__restrict__ rmax, int r_init) {
    __shared__ unsigned s_hist[KNN_BINS / 2][256];   // 32 bins x u16 per thread
    __shared__ unsigned s_max[4];
    const int tid = threadIdx.x;
    const int bt = blockIdx.z, b = bt / p.nb, t = bt - b * p.nb;
    const int cy = blockIdx.y * 16 + (tid >> 4), cx = blockIdx.x * 16 + (tid & 15);
    const bool active = cy < p.hq && cx < p.wq;
    const int *cs = cell_start + (size_t)bt * (p.G + 1);
    const float2 *sp_ = spos + (size_t)bt * p.n;
    const int *si_ = sidx + (size_t)bt * p.n;
    float dK = 0.f;

    if (active) {
        const float qy = (float)(cy * p.sp) + p.off, qx = (float)(cx * p.sp) + p.off;
        // ---- 1. grow the search square until K candidates are provably the nearest -------
        int r = r_init, y0, y1, x0, x1, cnt;
        float upper, scale;
        bool whole;
        for (;;) {
            y0 = max(cy - r, 0); y1 = min(cy + r, p.hq - 1);
            x0 = max(cx - r, 0); x1 = min(cx + r, p.wq - 1);
            whole = (y0 == 0 && x0 == 0 && y1 == p.hq - 1 && x1 == p.wq - 1);
            if (whole) {
                // every point is a candidate: range of the histogram = largest distance
                float dmax = 0.f;
                for (int j = 0; j < p.n; ++j) {
                    const float2 c = sp_[j];
                    dmax = fmaxf(dmax, pair_dist(qy, qx, c.x, c.y, p.l1));
                }
                upper = INFINITY;
                scale = dmax > 0.f ? (float)KNN_BINS / dmax : 0.f;
            } else {
                // anything outside the square is at least lb away along one axis
                const float lb = ((float)r + 0.5f) * (float)p.sp - KNN_SLACK;
                upper = p.l1 ? lb : lb * lb;
                scale = (float)KNN_BINS / upper;
            }
#pragma unroll
            for (int h = 0; h < KNN_BINS / 2; ++h) s_hist[h][tid] = 0u;
            cnt = 0;
            for (int yy = y0; yy <= y1; ++yy) {
                const int js = cs[yy * p.wq + x0], je = cs[yy * p.wq + x1 + 1];
                for (int j = js; j < je; ++j) {
                    const float2 c = sp_[j];
                    const float d = pair_dist(qy, qx, c.x, c.y, p.l1);
                    if (d < upper) {
                        const int bin = min((int)(d * scale), KNN_BINS - 1);
                        s_hist[bin >> 1][tid] += (bin & 1) ? 0x10000u : 1u;
                        ++cnt;
                    }
                }
            }
            if (cnt >= p.K || whole) break;
            r += 1 + (r >> 2);
        }
        // ---- 2. bin holding the K-th smallest ------------------------------------------------
        int bstar = KNN_BINS - 1, before = 0;
        {
            int cum = 0;
            bool found = false;
#pragma unroll
            for (int h = 0; h < KNN_BINS / 2; ++h) {
                const unsigned wv = s_hist[h][tid];
                const int c0 = (int)(wv & 0xffffu), c1 = (int)(wv >> 16);
                if (!found && cum + c0 >= p.K) { bstar = 2 * h; before = cum; found = true; }
                cum += c0;
                if (!found && cum + c1 >= p.K) { bstar = 2 * h + 1; before = cum; found = true; }
                cum += c1;
            }
        }
        // ---- 3. the (K - before) smallest keys inside that bin, by repeated minimum ----------
        const int need = p.K - before;
        float ld = -1.f; int li = -1;               // last selected key
        for (int it = 0; it < need; ++it) {
            float bd = INFINITY; int bi = 0x7fffffff;
            for (int yy = y0; yy <= y1; ++yy) {
                const int js = cs[yy * p.wq + x0], je = cs[yy * p.wq + x1 + 1];
                for (int j = js; j < je; ++j) {
                    const float2 c = sp_[j];
                    const float d = pair_dist(qy, qx, c.x, c.y, p.l1);
                    if (!(d < upper)) continue;
                    if (min((int)(d * scale), KNN_BINS - 1) != bstar) continue;
                    const int id = si_[j];
                    const bool gt_last = (d > ld) || (d == ld && id > li);
                    const bool lt_best = (d < bd) || (d == bd && id < bi);
                    if (gt_last && lt_best) { bd = d; bi = id; }
                }
            }
            ld = bd; li = bi;
        }
        dK = ld;
        const int iK = li;
        // ---- 4. weighted sum of the neighbours' flows ----------------------------------------
        const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
        const size_t BQ = (size_t)p.B * p.nb * p.G;
        const float2 *tmid = reinterpret_cast<const float2 *>(traj) + ((size_t)b * (p.T + p.nb) + p.T + t) * p.n;
        float norm = 0.f;
        for (int tr = 0; tr < p.T; ++tr) {
            const float2 *tref = reinterpret_cast<const float2 *>(traj) + ((size_t)b * (p.T + p.nb) + tr) * p.n;
            float sy = 0.f, sx = 0.f, sw = 0.f;
            for (int yy = y0; yy <= y1; ++yy) {
                const int js = cs[yy * p.wq + x0], je = cs[yy * p.wq + x1 + 1];
                for (int j = js; j < je; ++j) {
                    const float2 c = sp_[j];
                    const float d = pair_dist(qy, qx, c.x, c.y, p.l1);
                    const int id = si_[j];
                    if ((d < dK) || (d == dK && id <= iK)) {
                        const float2 a = tref[id];
                        const float fy = a.x - c.x, fx = a.y - c.y;     // traj(t_ref) - traj(t_mid)
                        if (p.iwd) {
                            const float w = 1.f / (d + 1e-9f);
                            sy += w * fy; sx += w * fx; sw += w;
                        } else {
                            sy += fy; sx += fx;
                        }
                    }
                }
            }
            float2 o;
            if (p.iwd) { o.x = sy / sw; o.y = sx / sw; norm = sw; }
            else { o.x = sy / (float)p.K; o.y = sx / (float)p.K; }
            reinterpret_cast<float2 *>(flow_lut)[q * p.T + tr] = o;
        }
        if (p.want_next && t < p.nb - 1) {
            const float2 *tnx = tmid + p.n;
            float sy = 0.f, sx = 0.f;
            for (int yy = y0; yy <= y1; ++yy) {
                const int js = cs[yy * p.wq + x0], je = cs[yy * p.wq + x1 + 1];
                for (int j = js; j < je; ++j) {
                    const float2 c = sp_[j];
                    const float d = pair_dist(qy, qx, c.x, c.y, p.l1);
                    const int id = si_[j];
                    if ((d < dK) || (d == dK && id <= iK)) {
                        const float2 a = tnx[id];
                        sy += a.x - c.x; sx += a.y - c.y;
                    }
                }
            }
            float2 o; o.x = sy / (float)p.K; o.y = sx / (float)p.K;
            reinterpret_cast<float2 *>(flow_next)[((size_t)(b * (p.nb - 1) + t)) * p.G + (size_t)cy * p.wq + cx] = o;
        }
        knn_state[q] = dK;
        reinterpret_cast<int *>(knn_state)[BQ + q] = iK;
        knn_state[2 * BQ + q] = norm;
        // ---- 5. optional: the K indices in ascending (distance, index) order -----------------
        if (idx_out != nullptr) {
            float pd = -1.f; int pi = -1;
            for (int k = 0; k < p.K; ++k) {
                float bd = INFINITY; int bi = 0x7fffffff;
                for (int yy = y0; yy <= y1; ++yy) {
                    const int js = cs[yy * p.wq + x0], je = cs[yy * p.wq + x1 + 1];
                    for (int j = js; j < je; ++j) {
                        const float2 c = sp_[j];
                        const float d = pair_dist(qy, qx, c.x, c.y, p.l1);
                        const int id = si_[j];
                        const bool gt_last = (d > pd) || (d == pd && id > pi);
                        const bool lt_best = (d < bd) || (d == bd && id < bi);
                        if (gt_last && lt_best) { bd = d; bi = id; }
                    }
                }
                pd = bd; pi = bi;
                idx_out[q * p.K + k] = bi;
            }
        }
    }
    // largest K-th distance of this (sample, bin): bounds the backward's search window
    unsigned m = __float_as_uint(dK);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
    if ((tid & 63) == 0) s_max[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) atomicMax(&rmax[bt], max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
}

// ------------------------------------------------------------------------------------------
// backward: one thread per (sample, trajectory point); loops over the bins and gathers the
// gradient of every query cell that has this point among its K nearest.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_bwd(const KnnParams p, const float *__restrict__ traj,
                                                 const float *__restrict__ glut,
                                                 const float *__restrict__ gnext,
                                                 const float *__restrict__ knn_state,
                                                 const unsigned *__restrict__ rmax,
                                                 float *__restrict__ gtraj) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (size_t)p.B * p.n) return;
    const int b = (int)(gi / p.n), i = (int)(gi - (size_t)b * p.n);
    const size_t BQ = (size_t)p.B * p.nb * p.G;
    const float2 *tr2 = reinterpret_cast<const float2 *>(traj) + (size_t)b * (p.T + p.nb) * p.n;
    float2 *g2 = reinterpret_cast<float2 *>(gtraj) + (size_t)b * (p.T + p.nb) * p.n;
    for (int tr = 0; tr < p.T; ++tr) g2[(size_t)tr * p.n + i] = make_float2(0.f, 0.f);
    float2 carry = make_float2(0.f, 0.f);   // flow_to_next term arriving from bin t-1
    const float invK = 1.f / (float)p.K;
    for (int t = 0; t < p.nb; ++t) {
        const int bt = b * p.nb + t;
        const float2 pt = tr2[(size_t)(p.T + t) * p.n + i];
        const float rm = __uint_as_float(rmax[bt]);
        const float R = (p.l1 ? rm : sqrtf(rm)) * 1.0001f + 0.01f;
        int y0 = (int)ceilf((pt.x - R - p.off) / (float)p.sp) - 1;
        int y1 = (int)floorf((pt.x + R - p.off) / (float)p.sp) + 1;
        int x0 = (int)ceilf((pt.y - R - p.off) / (float)p.sp) - 1;
        int x1 = (int)floorf((pt.y + R - p.off) / (float)p.sp) + 1;
        y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, p.hq - 1); x1 = min(x1, p.wq - 1);
        float2 gmid = make_float2(0.f, 0.f);      // d loss / d traj(t_mid)[b,t,i]
        for (int tr = 0; tr < p.T; ++tr) {
            float ay = 0.f, ax = 0.f;
            for (int cy = y0; cy <= y1; ++cy) {
                const float qy = (float)(cy * p.sp) + p.off;
                for (int cx = x0; cx <= x1; ++cx) {
                    const float qx = (float)(cx * p.sp) + p.off;
                    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
                    const float d = pair_dist(qy, qx, pt.x, pt.y, p.l1);
                    const float dK = knn_state[q];
                    if (d < dK || (d == dK && i <= reinterpret_cast<const int *>(knn_state)[BQ + q])) {
                        const float w = p.iwd ? (1.f / (d + 1e-9f)) / knn_state[2 * BQ + q] : invK;
                        const float2 g = reinterpret_cast<const float2 *>(glut)[q * p.T + tr];
                        ay += w * g.x; ax += w * g.y;
                    }
                }
            }
            float2 cur = g2[(size_t)tr * p.n + i];
            cur.x += ay; cur.y += ax;
            g2[(size_t)tr * p.n + i] = cur;
            gmid.x -= ay; gmid.y -= ax;
        }
        gmid.x += carry.x; gmid.y += carry.y;
        carry = make_float2(0.f, 0.f);
        if (gnext != nullptr && t < p.nb - 1) {
            float ay = 0.f, ax = 0.f;
            for (int cy = y0; cy <= y1; ++cy) {
                const float qy = (float)(cy * p.sp) + p.off;
                for (int cx = x0; cx <= x1; ++cx) {
                    const float qx = (float)(cx * p.sp) + p.off;
                    const size_t q = (size_t)bt * p.G + (size_t)cy * p.wq + cx;
                    const float d = pair_dist(qy, qx, pt.x, pt.y, p.l1);
                    const float dK = knn_state[q];
                    if (d < dK || (d == dK && i <= reinterpret_cast<const int *>(knn_state)[BQ + q])) {
                        const float2 g = reinterpret_cast<const float2 *>(gnext)[(size_t)(b * (p.nb - 1) + t) * p.G + (size_t)cy * p.wq + cx];
                        ay += invK * g.x; ax += invK * g.y;
                    }
                }
            }
            gmid.x -= ay; gmid.y -= ax;
            carry = make_float2(ay, ax);
        }
        g2[(size_t)(p.T + t) * p.n + i] = gmid;
    }
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
extern "C" int mpc_knn_lut_fwd(const mpc_shape *s, const float *traj, float *flow_lut, float *flow_next,
                               float *knn_state, int32_t *idx_out, void *ws, void *stream) {
    MPC_CHECK_ARG(s && traj && flow_lut && knn_state && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(!(s->flags & MPC_F_WANT_NEXT) || flow_next, MPC_E_NULL, "flow_next is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    MPC_CHECK_ARG(s->n >= s->K && s->K >= 1, MPC_E_SHAPE, "need 1 <= K <= n");
    MPC_CHECK_ARG(s->n < 65536, MPC_E_UNSUPPORTED, "more than 65535 trajectories per sample");
    const KnnParams p = knn_params(s);
    MPC_CHECK_ARG((size_t)p.G * 4 <= 150 * 1024, MPC_E_UNSUPPORTED, "LUT grid too large for the LDS counting sort");
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    int *cell_start = (int *)((char *)ws + L.off_cell_start);
    float2 *spos = (float2 *)((char *)ws + L.off_spos);
    int *sidx = (int *)((char *)ws + L.off_sidx);
    unsigned *rmax = reinterpret_cast<unsigned *>(knn_state) + 3 * (size_t)s->B * s->nb * p.G;
    static bool attr_set = false;   // raising the dynamic-LDS cap is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_knn_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(k_knn_bucket, dim3(s->B * s->nb), dim3(1024), (size_t)p.G * 4, st, p, traj, cell_start, spos, sidx, rmax);
    MPC_CHECK_LAUNCH();
    // smallest square that can hold K points at one point per cell and pass the ring bound
    int r_init = (int)ceil(sqrt((double)s->K / 3.14159265) * ((double)s->n > 0 ? sqrt((double)p.G / (double)s->n) : 1.0) - 0.5);
    if (r_init < 1) r_init = 1;
    const dim3 grid(mpc_cdiv(s->wq, 16), mpc_cdiv(s->hq, 16), s->B * s->nb);
    hipLaunchKernelGGL(k_knn_query, grid, dim3(256), 0, st, p, traj, cell_start, spos, sidx, flow_lut,
                       flow_next, knn_state, idx_out, rmax, r_init);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_knn_lut_bwd(const mpc_shape *s, const float *traj, const float *grad_flow_lut,
                               const float *grad_flow_next, const float *knn_state, float *grad_traj,
                               void *ws, void *stream) {
    MPC_CHECK_ARG(s && traj && grad_flow_lut && knn_state && grad_traj && ws, MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const KnnParams p = knn_params(s);
    const unsigned *rmax = reinterpret_cast<const unsigned *>(knn_state) + 3 * (size_t)s->B * s->nb * p.G;
    const int64_t total = (int64_t)s->B * s->n;
    hipLaunchKernelGGL(k_knn_bwd, dim3(mpc_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p, traj,
                       grad_flow_lut, grad_flow_next, knn_state, rmax, grad_traj);
    MPC_CHECK_LAUNCH();
    return 0;
}
