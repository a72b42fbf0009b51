// Event ingest: raw window (x, y, t_us, p per event, ragged batch) -> the padded
// [B][M][6] event tensor the loss consumes, built on the GPU.
//   per-sample half : reference src/loader/dsec/loader.py:152-167 (time normalisation by min/max in
//                     float64, bin = clip(searchsorted(linspace(0,1,nb+1), t) - 1, 0), in-image filter,
//                     positive / negative split, cast to float32)
//   collate half    : loader.py:360-415 (pad_events + sequence_collate_fn: zero rows with valid = 0,
//                     positive block padded to the batch maximum, then the negative block)
// Three passes: per-chunk counts and time extrema -> per-sample scan (chunk offsets, totals, batch
// maxima) -> stable scatter (order inside each block is the input order, as boolean masking gives).
// The caller reads the two batch maxima to size the output (the only host round trip of ingest).
#include "common.h"
#include "bounds.h"

#define ING_CHUNK 1024          // events per workgroup (256 threads x 4 consecutive events)

struct IngLayout {
    int nchunks;
    int *chunk_cnt;             // [B][nchunks][2]  valid positives / negatives per chunk
    long long *chunk_t;         // [B][nchunks][2]  min / max timestamp per chunk
    int *chunk_off;             // [B][nchunks][2]  exclusive offsets
    long long *tminmax;         // [B][2]
    int *totals;                // [B][2]
};

__device__ __forceinline__ int classify(float x, float y, float p, int H, int W) {
    // 1: positive row, 2: negative row, 0: dropped   (loader.py:160-165; comparisons on the float32 values)
    const bool in = (0.f <= y) && (y < (float)H) && (0.f <= x) && (x < (float)W);
    if (!in) return 0;
    return p == 1.f ? 1 : (p == 0.f ? 2 : 0);
}

// the four consecutive events of a thread: one 16-byte load per array (two for the timestamps) where the sample's first event is
// 16-byte aligned (vec: N a multiple of 4 and aligned base pointers) and all four exist, else element by element
struct IngQuad { float x[4], y[4], p[4]; long long t[4]; };
__device__ __forceinline__ IngQuad ing_load4(const float *__restrict__ x, const float *__restrict__ y, const long long *__restrict__ t,
                                             const float *__restrict__ p, size_t base, int i0, int n, int vec) {
    IngQuad q;
    if (vec && i0 + 3 < n) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + base + i0), yv = *reinterpret_cast<const float4 *>(y + base + i0);
        const float4 pv = *reinterpret_cast<const float4 *>(p + base + i0);
        const longlong2 t0 = *reinterpret_cast<const longlong2 *>(t + base + i0), t1 = *reinterpret_cast<const longlong2 *>(t + base + i0 + 2);
        q.x[0] = xv.x; q.x[1] = xv.y; q.x[2] = xv.z; q.x[3] = xv.w;
        q.y[0] = yv.x; q.y[1] = yv.y; q.y[2] = yv.z; q.y[3] = yv.w;
        q.p[0] = pv.x; q.p[1] = pv.y; q.p[2] = pv.z; q.p[3] = pv.w;
        q.t[0] = t0.x; q.t[1] = t0.y; q.t[2] = t1.x; q.t[3] = t1.y;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = i0 + k < n;
            q.x[k] = on ? x[base + i0 + k] : -1.f; q.y[k] = on ? y[base + i0 + k] : -1.f; q.p[k] = on ? p[base + i0 + k] : -1.f;
            q.t[k] = on ? t[base + i0 + k] : 0;
        }
    }
    return q;
}

// grid (nchunks, B), 256 threads
__global__ __launch_bounds__(256) void k_ingest_count(const mpc_ingest_shape s, const IngLayout L,
                                                      const float *__restrict__ x, const float *__restrict__ y,
                                                      const long long *__restrict__ t, const float *__restrict__ p,
                                                      const int *__restrict__ counts, int *__restrict__ out_max, int vec) {
    __shared__ int s_c[2][4];
    __shared__ long long s_t[2][4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    // (the batch maxima that k_ingest_scan raises with atomicMax start at 0: set here, a launch earlier, instead of by a memset)
    if (b == 0 && chunk == 0 && tid < 2) out_max[tid] = 0;
    const int n = min(counts[b], s.N);
    const size_t base = (size_t)b * s.N;
    int cp = 0, cn = 0;
    long long tmin = 0x7fffffffffffffffLL, tmax = -0x7fffffffffffffffLL - 1;
    const int i0 = chunk * ING_CHUNK + tid * 4;
    if (i0 < n) {
        const IngQuad q = ing_load4(x, y, t, p, base, i0, n, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                const int c = classify(q.x[k], q.y[k], q.p[k], s.H, s.W);
                cp += c == 1; cn += c == 2;
                const long long tv = q.t[k];            // extrema over ALL events of the window (loader.py:152)
                tmin = tv < tmin ? tv : tmin; tmax = tv > tmax ? tv : tmax;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cp += __shfl_down(cp, o, 64); cn += __shfl_down(cn, o, 64);
        const long long a = __shfl_down(tmin, o, 64), c2 = __shfl_down(tmax, o, 64);
        tmin = a < tmin ? a : tmin; tmax = c2 > tmax ? c2 : tmax;
    }
    if ((tid & 63) == 0) { s_c[0][tid >> 6] = cp; s_c[1][tid >> 6] = cn; s_t[0][tid >> 6] = tmin; s_t[1][tid >> 6] = tmax; }
    __syncthreads();
    if (tid == 0) {
        const size_t o = ((size_t)b * L.nchunks + chunk) * 2;
        L.chunk_cnt[o] = s_c[0][0] + s_c[0][1] + s_c[0][2] + s_c[0][3];
        L.chunk_cnt[o + 1] = s_c[1][0] + s_c[1][1] + s_c[1][2] + s_c[1][3];
        long long a = s_t[0][0], c2 = s_t[1][0];
        for (int w = 1; w < 4; ++w) { a = s_t[0][w] < a ? s_t[0][w] : a; c2 = s_t[1][w] > c2 ? s_t[1][w] : c2; }
        L.chunk_t[o] = a; L.chunk_t[o + 1] = c2;
    }
}

// grid B, 256 threads: exclusive scan over the chunks of one sample
__global__ __launch_bounds__(256) void k_ingest_scan(const IngLayout L, int *__restrict__ out_max) {
    __shared__ int s_run[2];
    __shared__ int s_part[2][256];
    __shared__ long long s_t[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { s_run[0] = 0; s_run[1] = 0; }
    long long tmin = 0x7fffffffffffffffLL, tmax = -0x7fffffffffffffffLL - 1;
    __syncthreads();
    for (int c0 = 0; c0 < L.nchunks; c0 += 256) {
        const int c = c0 + tid;
        int vp = 0, vn = 0;
        if (c < L.nchunks) {
            const size_t o = ((size_t)b * L.nchunks + c) * 2;
            vp = L.chunk_cnt[o]; vn = L.chunk_cnt[o + 1];
            const long long a = L.chunk_t[o], c2 = L.chunk_t[o + 1];
            tmin = a < tmin ? a : tmin; tmax = c2 > tmax ? c2 : tmax;
        }
        s_part[0][tid] = vp; s_part[1][tid] = vn;
        __syncthreads();
        if (tid < 2) {                      // serial scan of <= 256 values per polarity: tiny
            int run = s_run[tid];
            for (int k = 0; k < 256; ++k) { const int v = s_part[tid][k]; s_part[tid][k] = run; run += v; }
            s_run[tid] = run;
        }
        __syncthreads();
        if (c < L.nchunks) {
            const size_t o = ((size_t)b * L.nchunks + c) * 2;
            L.chunk_off[o] = s_part[0][tid]; L.chunk_off[o + 1] = s_part[1][tid];
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_down(tmin, o, 64), c2 = __shfl_down(tmax, o, 64);
        tmin = a < tmin ? a : tmin; tmax = c2 > tmax ? c2 : tmax;
    }
    if ((tid & 63) == 0) { s_t[0][tid >> 6] = tmin; s_t[1][tid >> 6] = tmax; }
    __syncthreads();
    if (tid == 0) {
        long long a = s_t[0][0], c2 = s_t[1][0];
        for (int w = 1; w < 4; ++w) { a = s_t[0][w] < a ? s_t[0][w] : a; c2 = s_t[1][w] > c2 ? s_t[1][w] : c2; }
        L.tminmax[b * 2] = a; L.tminmax[b * 2 + 1] = c2;
        L.totals[b * 2] = s_run[0]; L.totals[b * 2 + 1] = s_run[1];
        atomicMax(&out_max[0], s_run[0]);
        atomicMax(&out_max[1], s_run[1]);
    }
}

__device__ __forceinline__ int bin_index(double tn, int nb) {
    // np.searchsorted(np.linspace(0, 1, nb + 1), tn, side='left') - 1, clipped at 0.
    // linspace: e_i = i * (1.0 / nb) in float64, e_nb = 1.0 exactly.
    if (tn != tn) return nb;      // a window of one event: 0/0 time; numpy sorts NaN last (searchsorted -> nb + 1)
    const double step = 1.0 / (double)nb;
    int i = (int)ceil(tn * (double)nb);
    i = i < 0 ? 0 : (i > nb ? nb : i);
    auto edge = [&](int k) { return k >= nb ? 1.0 : (double)k * step; };
    while (i > 0 && edge(i - 1) >= tn) --i;
    while (i < nb && edge(i) < tn) ++i;
    if (i == nb && edge(nb) < tn) i = nb + 1;          // tn > 1 cannot happen after min/max normalisation
    return i - 1 < 0 ? 0 : i - 1;
}

// grid (nchunks, B), 256 threads.  Thread t owns 4 consecutive events, so ranks follow the input order.  The rows of a
// workgroup are two contiguous runs of the output (its positives, its negatives): they are put together in LDS and written as
// runs of 8-byte words (a lane writing its own 24-byte row was six scattered 4-byte stores per event).  The padding rows of
// the sample (loader.py:360-364: zero rows, valid = 0) are zeroed here too, a share per workgroup -- not by a memset of the
// whole tensor in front of the kernel.
__global__ __launch_bounds__(256) void k_ingest_scatter(const mpc_ingest_shape s, const IngLayout L,
                                                        const float *__restrict__ x, const float *__restrict__ y,
                                                        const long long *__restrict__ t, const float *__restrict__ p,
                                                        const int *__restrict__ counts, int max_pos, int max_neg,
                                                        float *__restrict__ events, float *__restrict__ xytp, int vec) {
    __shared__ int s_w[2][4];
    __shared__ float2 s_rows[ING_CHUNK * 3];          // positives from the front, the negatives behind them
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[b], s.N);
    const size_t base = (size_t)b * s.N;
    const long long tmin = L.tminmax[b * 2], tmax = L.tminmax[b * 2 + 1];
    const double span = (double)(tmax - tmin);
    const int M = max_pos + max_neg;
    const int i0 = chunk * ING_CHUNK + tid * 4;
    IngQuad q;
    int cls[4], lp = 0, ln = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cls[k] = 0;
    if (i0 < n) {
        q = ing_load4(x, y, t, p, base, i0, n, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cls[k] = (i0 + k < n) ? classify(q.x[k], q.y[k], q.p[k], s.H, s.W) : 0;
            lp += cls[k] == 1; ln += cls[k] == 2;
        }
    }
    // exclusive scan of (lp, ln) over the workgroup
    int ip = lp, in_ = ln;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int a = __shfl_up(ip, o, 64), c2 = __shfl_up(in_, o, 64);
        if ((tid & 63) >= o) { ip += a; in_ += c2; }
    }
    if ((tid & 63) == 63) { s_w[0][tid >> 6] = ip; s_w[1][tid >> 6] = in_; }
    __syncthreads();
    int wp = 0, wn = 0, np_wg = 0, nn_wg = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < (tid >> 6)) { wp += s_w[0][w]; wn += s_w[1][w]; }
        np_wg += s_w[0][w]; nn_wg += s_w[1][w];
    }
    int rp = wp + ip - lp, rn = np_wg + wn + in_ - ln;          // LDS rows of this thread's first positive / negative
    if (i0 < n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k >= n) continue;
            const long long tv = q.t[k];
            if (xytp != nullptr) {
                // loader.py:135-138: t = (t - t[0]).astype(float32); t = t / t[-1]   (t increasing)
                const float tf = (float)(tv - tmin) / (float)(tmax - tmin);
                reinterpret_cast<float4 *>(xytp)[base + i0 + k] = make_float4(q.x[k], q.y[k], tf, q.p[k]);
            }
            if (cls[k] == 0) continue;
            const double tn = (double)(tv - tmin) / span;                       // loader.py:152 (float64)
            const int row = cls[k] == 1 ? rp++ : rn++;
            s_rows[MPC_IDX(3 * row + 0, ING_CHUNK * 3)] = make_float2(q.y[k], q.x[k]);
            s_rows[MPC_IDX(3 * row + 1, ING_CHUNK * 3)] = make_float2((float)tn, q.p[k]);
            s_rows[MPC_IDX(3 * row + 2, ING_CHUNK * 3)] = make_float2((float)bin_index(tn, s.nb), 1.f);
        }
    }
    __syncthreads();
    const size_t co = ((size_t)b * L.nchunks + chunk) * 2;
    float2 *dstp = reinterpret_cast<float2 *>(events + ((size_t)b * M + L.chunk_off[co]) * 6);
    float2 *dstn = reinterpret_cast<float2 *>(events + ((size_t)b * M + max_pos + L.chunk_off[co + 1]) * 6);
    for (int j = tid; j < 3 * np_wg; j += 256) dstp[j] = s_rows[j];
    for (int j = tid; j < 3 * nn_wg; j += 256) dstn[j] = s_rows[3 * np_wg + j];
    // this workgroup's share of the sample's padding rows
    const int tp = L.totals[b * 2], tn_ = L.totals[b * 2 + 1];
    const int pad_p = 3 * (max_pos - tp), pad_n = 3 * (max_neg - tn_), pad = pad_p + pad_n;
    if (pad > 0) {
        const int per = (pad + (int)gridDim.x - 1) / (int)gridDim.x;
        float2 *zp = reinterpret_cast<float2 *>(events + ((size_t)b * M + tp) * 6);
        float2 *zn = reinterpret_cast<float2 *>(events + ((size_t)b * M + max_pos + tn_) * 6);
        const int j1 = min((chunk + 1) * per, pad);
        for (int j = chunk * per + tid; j < j1; j += 256) {
            if (j < pad_p) zp[j] = make_float2(0.f, 0.f); else zn[j - pad_p] = make_float2(0.f, 0.f);
        }
    }
}

// ---- ingest straight into the bucket-ordered layout (both halves in one) -----------------------------
// The rows of each polarity block ordered by (time bin, LUT strip) -- the key of the loss's backward buckets
// (events.hip: mpc_event_bucket_order) -- as ingest WRITES them: the key of an event is known the moment its row is
// built, so ordering costs no pass over the event tensor of its own (ingest + mpc_event_bucket_order read and wrote the
// tensor once more: 175 + 50 us at 14 x 200k events).  Counting sort: per-chunk counts of (polarity, key) -> the scans of
// the bucket-order kernels (events.hip) -> scatter.  Order inside a bucket is not defined and need not be.
struct IngKey { int NCS, CSR, NK, sp, hq; };

__device__ __forceinline__ int ing_key(const mpc_ingest_shape &s, const IngKey &k, float yv, double tn) {
    const int it = min(max(bin_index(tn, s.nb), 0), s.nb - 1);
    const int iy = min(max((int)floorf(mpc_div_sp(yv, k.sp)), 0), k.hq - 1);          // as warp_cell (events.hip)
    return it * k.NCS + iy / k.CSR;
}

// grid (nchunks, B), 256 threads, dynamic LDS 2 * (NK + 1) ints: counts[b][pol][key][chunk]
__global__ __launch_bounds__(256) void k_ingest_keycount(const mpc_ingest_shape s, const IngLayout L, const IngKey k,
                                                         const float *__restrict__ x, const float *__restrict__ y,
                                                         const long long *__restrict__ t, const float *__restrict__ p,
                                                         const int *__restrict__ counts, int *__restrict__ kcounts, int vec) {
    extern __shared__ int s_k[];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[b], s.N);
    const size_t base = (size_t)b * s.N;
    const long long tmin = L.tminmax[b * 2], tmax = L.tminmax[b * 2 + 1];
    const double span = (double)(tmax - tmin);
    for (int i = tid; i < 2 * (k.NK + 1); i += 256) s_k[i] = 0;
    __syncthreads();
    const int i0 = chunk * ING_CHUNK + tid * 4;
    if (i0 < n) {
        const IngQuad q = ing_load4(x, y, t, p, base, i0, n, vec);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u >= n) continue;
            const int c = classify(q.x[u], q.y[u], q.p[u], s.H, s.W);
            if (c == 0) continue;
            const double tn = (double)(q.t[u] - tmin) / span;
            atomicAdd(&s_k[MPC_IDX((c - 1) * (k.NK + 1) + ing_key(s, k, q.y[u], tn), 2 * (k.NK + 1))], 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * (k.NK + 1); i += 256) {
        const int pol = i / (k.NK + 1), key = i - pol * (k.NK + 1);
        kcounts[((size_t)(b * 2 + pol) * (k.NK + 1) + key) * L.nchunks + chunk] = s_k[i];
    }
}

// grid (nchunks, B), 256 threads, dynamic LDS 2 * (NK + 1) ints.  kcounts (scanned in place) = first row of this chunk's
// share inside every (polarity, key); offsets [B][2][NK + 1] = first row of every key
__global__ __launch_bounds__(256) void k_ingest_scatter_ordered(const mpc_ingest_shape s, const IngLayout L, const IngKey k,
                                                                const float *__restrict__ x, const float *__restrict__ y,
                                                                const long long *__restrict__ t, const float *__restrict__ p,
                                                                const int *__restrict__ counts, int M,
                                                                const int *__restrict__ kcounts, const int *__restrict__ offsets,
                                                                float *__restrict__ events, float *__restrict__ xytp, int vec) {
    extern __shared__ int s_k[];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[b], s.N);
    const size_t base = (size_t)b * s.N;
    const long long tmin = L.tminmax[b * 2], tmax = L.tminmax[b * 2 + 1];
    const double span = (double)(tmax - tmin);
    for (int i = tid; i < 2 * (k.NK + 1); i += 256) {
        const int pol = i / (k.NK + 1), key = i - pol * (k.NK + 1);
        const size_t o = (size_t)(b * 2 + pol) * (k.NK + 1) + key;
        s_k[i] = offsets[o] + kcounts[o * L.nchunks + chunk];
    }
    __syncthreads();
    const int i0 = chunk * ING_CHUNK + tid * 4;
    if (i0 < n) {
        const IngQuad q = ing_load4(x, y, t, p, base, i0, n, vec);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i >= n) continue;
            const long long tv = q.t[u];
            if (xytp != nullptr) {
                const float tf = (float)(tv - tmin) / (float)(tmax - tmin);
                reinterpret_cast<float4 *>(xytp)[base + i] = make_float4(q.x[u], q.y[u], tf, q.p[u]);
            }
            const int c = classify(q.x[u], q.y[u], q.p[u], s.H, s.W);
            if (c == 0) continue;
            const double tn = (double)(tv - tmin) / span;
            const int row = atomicAdd(&s_k[MPC_IDX((c - 1) * (k.NK + 1) + ing_key(s, k, q.y[u], tn), 2 * (k.NK + 1))], 1);
            float2 *e = reinterpret_cast<float2 *>(events + ((size_t)b * M + row) * 6);      // (24-byte rows: 8-byte aligned)
            e[0] = make_float2(q.y[u], q.x[u]); e[1] = make_float2((float)tn, q.p[u]);
            e[2] = make_float2((float)bin_index(tn, s.nb), 1.f);
        }
    }
    // this workgroup's share of the sample's padding rows (they stay last in their block; no memset of the tensor)
    const int Mp = offsets[(size_t)(b * 2 + 1) * (k.NK + 1)];          // first row of the negative block = max_pos
    const int tp = L.totals[b * 2], tn_ = L.totals[b * 2 + 1];
    const int pad_p = 3 * (Mp - tp), pad_n = 3 * (M - Mp - tn_), pad = pad_p + pad_n;
    if (pad > 0) {
        const int per = (pad + (int)gridDim.x - 1) / (int)gridDim.x;
        float2 *zp = reinterpret_cast<float2 *>(events + ((size_t)b * M + tp) * 6);
        float2 *zn = reinterpret_cast<float2 *>(events + ((size_t)b * M + Mp + tn_) * 6);
        const int j1 = min((chunk + 1) * per, pad);
        for (int j = chunk * per + tid; j < j1; j += 256) {
            if (j < pad_p) zp[j] = make_float2(0.f, 0.f); else zn[j - pad_p] = make_float2(0.f, 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------
static int ing_validate(const mpc_ingest_shape *s) {
    MPC_CHECK_ARG(s->B >= 0 && s->N >= 0 && s->H >= 1 && s->W >= 1 && s->nb >= 1, MPC_E_SHAPE, "bad ingest shape");
    MPC_CHECK_ARG((int64_t)s->B * s->N < (1LL << 31), MPC_E_UNSUPPORTED, "too many events");
    return 0;
}

// 16-byte loads of four consecutive events: every sample's first event aligned
static int ing_vec(const mpc_ingest_shape *s, const void *x, const void *y, const void *t, const void *p) {
    return (s->N % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)t | (uintptr_t)p) & 15) == 0 ? 1 : 0;
}

struct IngHost { IngLayout L; int64_t total; };

static IngHost ing_layout(const mpc_ingest_shape *s, void *ws) {
    IngHost h;
    const int B = s->B > 0 ? s->B : 1;
    h.L.nchunks = mpc_cdiv(s->N > 0 ? s->N : 1, ING_CHUNK);
    const int64_t nc = (int64_t)B * h.L.nchunks;
    int64_t off = 0;
    char *w = (char *)ws;
    h.L.chunk_t = (long long *)(w + off);  off += mpc_align(nc * 2 * 8);
    h.L.tminmax = (long long *)(w + off);  off += mpc_align((int64_t)B * 2 * 8);
    h.L.chunk_cnt = (int *)(w + off);      off += mpc_align(nc * 2 * 4);
    h.L.chunk_off = (int *)(w + off);      off += mpc_align(nc * 2 * 4);
    h.L.totals = (int *)(w + off);         off += mpc_align((int64_t)B * 2 * 4);
    h.total = off;
    return h;
}

extern "C" int64_t mpc_ingest_workspace_bytes(const mpc_ingest_shape *s) {
    if (!s) { mpc_set_error("mpc_ingest_workspace_bytes: null shape"); return MPC_E_NULL; }
    int rc = ing_validate(s);
    if (rc) return rc;
    return ing_layout(s, nullptr).total;
}

extern "C" int mpc_ingest_count(const mpc_ingest_shape *s, const float *x, const float *y, const int64_t *t_us,
                                const float *p, const int32_t *counts, int32_t *out_max, void *ws, void *stream) {
    MPC_CHECK_ARG(s && counts && out_max && ws && ((x && y && t_us && p) || s->N == 0 || s->B == 0), MPC_E_NULL, "null argument");
    int rc = ing_validate(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (s->B == 0) return mpc_zero_async(out_max, 2 * sizeof(int32_t), st);
    const IngLayout L = ing_layout(s, ws).L;
    MPC_LAUNCH(k_ingest_count, dim3(L.nchunks, s->B), dim3(256), 0, st, *s, L, x, y,
                       reinterpret_cast<const long long *>(t_us), p, counts, out_max, ing_vec(s, x, y, t_us, p));
    MPC_LAUNCH(k_ingest_scan, dim3(s->B), dim3(256), 0, st, L, out_max);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_ingest_scatter(const mpc_ingest_shape *s, const float *x, const float *y, const int64_t *t_us,
                                  const float *p, const int32_t *counts, int32_t max_pos, int32_t max_neg,
                                  float *events, float *xytp, void *ws, void *stream) {
    MPC_CHECK_ARG(s && counts && ws && ((x && y && t_us && p) || s->N == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(max_pos >= 0 && max_neg >= 0 && (events || max_pos + max_neg == 0 || s->B == 0), MPC_E_NULL, "events is null");
    int rc = ing_validate(s);
    if (rc) return rc;
    if (s->B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)max_pos + max_neg;
    if (s->N == 0) return M > 0 ? mpc_zero_async(events, (size_t)s->B * M * 6 * sizeof(float), st) : 0;     // padding rows only
    const IngLayout L = ing_layout(s, ws).L;
    MPC_LAUNCH(k_ingest_scatter, dim3(L.nchunks, s->B), dim3(256), 0, st, *s, L, x, y,
                       reinterpret_cast<const long long *>(t_us), p, counts, max_pos, max_neg, events, xytp, ing_vec(s, x, y, t_us, p));
    MPC_CHECK_LAUNCH();
    return 0;
}


// the scans of the bucket-order kernels (events.hip)
int mpc_evo_scans(const mpc_shape *loss, int NCS, int CSR, int *kcounts, int *totals, int32_t *offsets, int chunks, hipStream_t st);

extern "C" int64_t mpc_ingest_ordered_workspace_bytes(const mpc_ingest_shape *s, const mpc_shape *loss) {
    if (!s || !loss) { mpc_set_error("mpc_ingest_ordered_workspace_bytes: null shape"); return MPC_E_NULL; }
    int rc = ing_validate(s);
    if (rc) return rc;
    const int32_t ncs = mpc_event_lut_strips(loss);
    if (ncs <= 0) { mpc_set_error("mpc_ingest_ordered_workspace_bytes: no bucketed event layout for this loss shape"); return MPC_E_UNSUPPORTED; }
    const int64_t nk1 = (int64_t)loss->nb * ncs + 1;
    const int64_t nch = mpc_cdiv(s->N > 0 ? s->N : 1, ING_CHUNK);
    return mpc_align((int64_t)(s->B > 0 ? s->B : 1) * 2 * nk1 * (nch + 1) * 4);
}

// Same outputs as mpc_ingest_scatter followed by mpc_event_bucket_order for the loss shape `loss` (B, nb, H, W, sp, hq, wq,
// flags of the FocusLoss; M = max_pos + max_neg, Mp = max_pos): `events` with the rows of every polarity block grouped by
// (time bin, LUT strip) and the table `offsets` [B][2][nb * strips + 1].  `ws` = the workspace of mpc_ingest_count
// (unchanged since that call), `ws_order` = mpc_ingest_ordered_workspace_bytes.
extern "C" int mpc_ingest_scatter_ordered(const mpc_ingest_shape *s, const mpc_shape *loss, const float *x, const float *y,
                                          const int64_t *t_us, const float *p, const int32_t *counts, int32_t max_pos,
                                          int32_t max_neg, float *events, int32_t *offsets, float *xytp, void *ws, void *ws_order,
                                          void *stream) {
    MPC_CHECK_ARG(s && loss && counts && ws && ws_order && offsets && ((x && y && t_us && p) || s->N == 0 || s->B == 0), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(max_pos >= 0 && max_neg >= 0 && (events || max_pos + max_neg == 0 || s->B == 0), MPC_E_NULL, "events is null");
    int rc = ing_validate(s);
    if (rc) return rc;
    if ((rc = mpc_validate_shape(loss))) return rc;
    MPC_CHECK_ARG(loss->B == s->B && loss->nb == s->nb && loss->H == s->H && loss->W == s->W && loss->M == max_pos + max_neg && loss->Mp == max_pos,
                  MPC_E_SHAPE, "loss shape does not match the ingest shape / the block sizes");
    const mpc_ws_layout LL = mpc_layout(loss);
    MPC_CHECK_ARG(LL.n_cstrips > 0, MPC_E_UNSUPPORTED, "no bucketed event layout for this loss shape");
    if (s->B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)max_pos + max_neg;
    const IngKey k{LL.n_cstrips, LL.cstrip_rows, loss->nb * LL.n_cstrips, loss->sp, loss->hq};
    if (M > 0 && s->N == 0) {
        const int e = mpc_zero_async(events, (size_t)s->B * M * 6 * sizeof(float), st);     // padding rows only (else: the scatter kernel zeroes them)
        if (e) return e;
    }
    const IngLayout L = ing_layout(s, ws).L;
    int *kcounts = (int *)ws_order;
    int *totals = kcounts + (size_t)2 * s->B * (k.NK + 1) * L.nchunks;
    const size_t lds = (size_t)2 * (k.NK + 1) * sizeof(int);
    if (s->N == 0) return mpc_zero_async(offsets, (size_t)s->B * 2 * (k.NK + 1) * sizeof(int32_t), st);
    MPC_LAUNCH(k_ingest_keycount, dim3(L.nchunks, s->B), dim3(256), lds, st, *s, L, k, x, y, reinterpret_cast<const long long *>(t_us), p, counts, kcounts, ing_vec(s, x, y, t_us, p));
    MPC_CHECK_LAUNCH();
    if ((rc = mpc_evo_scans(loss, k.NCS, k.CSR, kcounts, totals, offsets, L.nchunks, st))) return rc;
    MPC_LAUNCH(k_ingest_scatter_ordered, dim3(L.nchunks, s->B), dim3(256), lds, st, *s, L, k, x, y, reinterpret_cast<const long long *>(t_us), p,
               counts, (int)M, kcounts, (const int *)offsets, events, xytp, ing_vec(s, x, y, t_us, p));
    MPC_CHECK_LAUNCH();
    return 0;
}

MPC_BOUNDS_UNIT("ingest.hip")
