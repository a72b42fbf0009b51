// Voxel-grid builder: the network input that the reference's DataLoader workers
// build on the CPU from the same event window (src/loader/dsec/utils.py:29-77, VoxelGrid.convert):
// trilinear accumulation of +-1 polarity votes into [C][H][W], then normalisation of the non-zero
// entries.  Same machinery as the IWE (events.hip): one binning pass appends 16-byte records to
// per-(sample, channel, row-strip) buckets, one workgroup per bucket accumulates its strip in LDS as
// Q33.30 fixed point (ds_add_u64) and writes it with plain stores; an overflowing bucket spills to its sample's
// spill region in CHUNKS (one contiguous run per binning workgroup and bucket, named in a per-sample chunk list), and only
// the workgroup of a bucket that did overflow looks at its sample's chunk list and reads its own runs.  With mean_std / max
// normalisation the strips are accumulated twice -- statistics first, then written normalised -- so the
// grid is written once and never read back (k_vox_accum).
//
// Arithmetic follows the reference op for op: x0 = int(x) (truncation), tap weight
// value * (1-|xl-x|) * (1-|yl-y|) * (1-|tl-t_norm|) (left to right; value = 2p-1 is +-1, so its sign
// commutes exactly), t_norm = (C-1) * (t - t[0]) / (t[-1] - t[0]).
#include "common.h"
#include "bounds.h"

#define VOX_FIX_SHIFT 30
#define VOX_PER_THREAD 2

struct VoxLayout {
    int SR, NS, NBk, cap;
    int spcap, chcap;     // spill records / chunk descriptors per sample
    int *gcount;          // [NBk + 2 B]   fill of every bucket; then per sample: spilled records, chunks
    float4 *rec, *ovf;    // ovf: [B][spcap]
    int4 *chunk;          // [B][chcap]  {bucket within the sample, first spill record, records, -}
    double *part;         // [B][nblk][4]   partial statistics of k_vox_stats (quantile clipping: the entries change after the strips)
    double *spart;        // [NBk][4]       partial statistics of the strips (k_vox_accum<1>)
    float *stat;          // [B][4]  mean, 1/std (or 1/max), flag
    int nstat_blocks;
};

__device__ __forceinline__ long long vox_to_fixed(float v) {         // |v| < 2^31
    const float hi = truncf(v);
    return ((long long)(int)hi << VOX_FIX_SHIFT) + (long long)(int)((v - hi) * (float)(1 << VOX_FIX_SHIFT));
}
__device__ __forceinline__ float vox_from_fixed(long long a) {
    return (float)((double)a * (1.0 / (double)(1 << VOX_FIX_SHIFT)));
}

// taps of one record restricted to rows [row0, row1): f(yy, xx, value)
template <typename F>
__device__ __forceinline__ void vox_taps(float y, float x, float wt, int H, int W, int row0, int row1, F f) {
    const int y0 = (int)fminf(fmaxf(y, -8.f), (float)H + 8.f), x0 = (int)fminf(fmaxf(x, -8.f), (float)W + 8.f);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
        const int xx = x0 + dx;
        if (xx < 0 || xx >= W) continue;
        const float wx = 1.f - fabsf((float)xx - x);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = y0 + dy;
            if (yy < row0 || yy >= row1) continue;
            const float wy = 1.f - fabsf((float)yy - y);
            f(yy, xx, (wx * wy) * wt);
        }
    }
}

// grid (chunks * B rounded up to 8), 256 threads, dynamic LDS = C*NS*3 ints
__global__ __launch_bounds__(256) void k_vox_bin(const mpc_vox_shape s, const VoxLayout L,
                                                 const float4 *__restrict__ ev, const int *__restrict__ counts) {
    extern __shared__ int s_cnt[];
    const int chunks = (s.N + 256 * VOX_PER_THREAD - 1) / (256 * VOX_PER_THREAD);
    const int per = (chunks * s.B + 7) >> 3;
    const int lblk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-contiguous order
    if (lblk >= chunks * s.B) return;
    const int tid = threadIdx.x, b = lblk / chunks, chunk = lblk - b * chunks;
    const int nloc = s.C * L.NS;
    int *s_base = s_cnt + nloc;
    int *s_spill = s_base + nloc;         // slot - s_spill[lb] = place in the sample's spill region, for the slots >= cap
    for (int i = tid; i < nloc; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const int n = min(counts[b], s.N);
    const float4 *e = ev + (size_t)b * s.N;
    float t_first = 0.f, t_span = 1.f;
    if (n > 0) { t_first = e[0].z; t_span = e[n - 1].z - t_first; }
    const float inv_SR = 1.f / (float)L.SR;
    float ry[VOX_PER_THREAD], rx[VOX_PER_THREAD], rw[VOX_PER_THREAD][2];
    int bk[VOX_PER_THREAD][4], rk[VOX_PER_THREAD][4];
#pragma unroll
    for (int k = 0; k < VOX_PER_THREAD; ++k) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { bk[k][u] = -1; rk[k][u] = 0; }
        ry[k] = rx[k] = rw[k][0] = rw[k][1] = 0.f;
        const int i = (chunk * VOX_PER_THREAD + k) * 256 + tid;
        if (i >= n) continue;
        const float4 v = e[i];                       // x, y, t, p
        const float tn = (float)(s.C - 1) * (v.z - t_first) / t_span;       // utils.py:35-36
        const int t0 = (int)tn;
        const int y0 = (int)fminf(fmaxf(v.y, -8.f), (float)s.H + 8.f);
        const float val = 2.f * v.w - 1.f;
        ry[k] = v.y; rx[k] = v.x;
        const int x0 = (int)fminf(fmaxf(v.x, -8.f), (float)s.W + 8.f);
        if (x0 + 1 < 0 || x0 >= s.W) continue;        // no column inside the sensor
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int tl = t0 + dt;
            if (tl < 0 || tl >= s.C) continue;
            rw[k][dt] = val * (1.f - fabsf((float)tl - tn));
            int prev = -1;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yl = y0 + dy;
                if (yl < 0 || yl >= s.H) continue;
                const int st = (int)(((float)yl + 0.5f) * inv_SR);     // exact, see k_ev_bin
                if (st == prev) continue;
                prev = st;
                const int lb = tl * L.NS + st;
                bk[k][dt * 2 + dy] = lb;
                rk[k][dt * 2 + dy] = atomicAdd(&s_cnt[MPC_IDX(lb, nloc)], 1);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nloc; i += 256) {
        const int c = s_cnt[i];
        const int base = c > 0 ? atomicAdd(&L.gcount[b * nloc + i], c) : 0;
        s_base[i] = base;
        // the slots [max(base, cap), base + c) of this workgroup lie beyond the bucket: ONE run of the sample's spill region,
        // named in the sample's chunk list (the bucket's own workgroup reads the list and then only its runs)
        const int first = max(base, L.cap), nsp = base + c - first;
        if (nsp > 0) {
            const int sp0 = atomicAdd(&L.gcount[L.NBk + 2 * b], nsp);
            const int ci = atomicAdd(&L.gcount[L.NBk + 2 * b + 1], 1);
            if (ci < L.chcap) L.chunk[(size_t)b * L.chcap + MPC_IDX(ci, L.chcap)] = make_int4(i, sp0, nsp, 0);
            s_spill[i] = first - sp0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VOX_PER_THREAD; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int lb = bk[k][u];
            if (lb < 0) continue;
            const int g = b * nloc + lb;
            const int slot = s_base[lb] + rk[k][u];
            const float4 rec = make_float4(ry[k], rx[k], rw[k][u >> 1], __int_as_float(g));
            if (slot < L.cap) L.rec[MPC_IDX((size_t)g * L.cap + slot, (long long)L.NBk * L.cap)] = rec;
            else { const int k = slot - s_spill[lb]; MPC_EXPECT(k >= 0 && k < L.spcap); if (k >= 0 && k < L.spcap) L.ovf[(size_t)b * L.spcap + k] = rec; }
        }
}

// grid NBk, 1024 threads, dynamic LDS = SR * W * 8.  One strip of one channel image accumulated in LDS (64-bit fixed point:
// integer sums, any order, bitwise reproducible) from its bucket of records and -- only a bucket beyond its capacity (events
// piled up in a few rows) -- from its runs of the sample's spill region: the workgroup reads the sample's chunk list once
// (16 bytes per (binning workgroup, overflowed bucket) pair) and then its own runs, nothing of the other buckets' spills.
//   MODE 0: the strip written as it is (no normalisation, or quantile clipping follows)
//   MODE 1: nothing written -- the strip's share of the per-sample statistics of the non-zero entries (count, sum, sum of
//           squares, largest magnitude) to spart[g]
//   MODE 2: the strip written NORMALISED with the sample's (mean, 1 / std) or 1 / max from k_vox_finalize
// mean_std / max normalisation = MODE 1, k_vox_finalize, MODE 2: the records (90 MB at the DSEC batch shape) are read twice and
// the grid (258 MB) is written once; the three passes over the grid this replaces (write, statistics, read-modify-write) moved
// 1.2 GB.
template <int MODE>
__global__ __launch_bounds__(1024) void k_vox_accum(const VoxLayout L, float *__restrict__ grid, int H, int W, int C, int norm) {
    extern __shared__ unsigned long long s_acc[];
    __shared__ double s_red[4][16];
    const int tid = threadIdx.x;
    const int g = blockIdx.x, img = g / L.NS, strip = g - img * L.NS;
    const int row0 = strip * L.SR, row1 = min(row0 + L.SR, H);
    const int npix = (row1 - row0) * W;
    for (int i = tid; i < npix; i += 1024) s_acc[i] = 0ull;
    __syncthreads();
    const int filled = L.gcount[g], n = min(filled, L.cap);
    const float4 *rec = L.rec + (size_t)g * L.cap;
    for (int r = tid; r < n; r += 1024) {
        const float4 e = rec[r];
        vox_taps(e.x, e.y, e.z, H, W, row0, row1, [&](int yy, int xx, float v) {
            atomicAdd(&s_acc[MPC_IDX((yy - row0) * W + xx, npix)], (unsigned long long)vox_to_fixed(v));
        });
    }
    if (filled > L.cap) {                                 // (workgroup-uniform) this bucket spilled
        const int nloc = C * L.NS, b = g / nloc, lb = g - b * nloc;
        const int nch = min(L.gcount[L.NBk + 2 * b + 1], L.chcap);
        const int4 *ch = L.chunk + (size_t)b * L.chcap;
        const float4 *ovf = L.ovf + (size_t)b * L.spcap;
        for (int c0 = 0; c0 < nch; c0 += 1024) {          // the chunk list, a descriptor per thread; a wavefront takes the runs its lanes found
            int4 d = make_int4(-1, 0, 0, 0);
            if (c0 + tid < nch) d = ch[c0 + tid];
            unsigned long long mm = __ballot(d.x == lb);
            while (mm != 0ull) {
                const int l = __ffsll((long long)mm) - 1;
                mm &= mm - 1ull;
                const int sp0 = __shfl(d.y, l, 64);
                const int cnt = min(__shfl(d.z, l, 64), max(L.spcap - sp0, 0));
                for (int r = (tid & 63); r < cnt; r += 64) {
                    const float4 e = ovf[MPC_IDX(sp0 + r, L.spcap)];
                    vox_taps(e.x, e.y, e.z, H, W, row0, row1, [&](int yy, int xx, float v) {
                        atomicAdd(&s_acc[MPC_IDX((yy - row0) * W + xx, npix)], (unsigned long long)vox_to_fixed(v));
                    });
                }
            }
        }
    }
    __syncthreads();
    if (MODE == 1) {
        // per-thread partials in fp32 (a thread sees ~20 entries), everything above them in fp64
        int cnt = 0;
        float sum = 0.f, sq = 0.f, mx = 0.f;
        for (int i = tid; i < npix; i += 1024) {
            const float v = vox_from_fixed((long long)s_acc[i]);
            if (v != 0.f) { ++cnt; sum += v; sq = fmaf(v, v, sq); mx = fmaxf(mx, fabsf(v)); }
        }
        const double r0 = block_sum_d<1024>((double)cnt, s_red[0]);
        const double r1 = block_sum_d<1024>((double)sum, s_red[1]);
        const double r2 = block_sum_d<1024>((double)sq, s_red[2]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
        if ((tid & 63) == 0) s_red[3][tid >> 6] = (double)mx;
        __syncthreads();
        if (tid == 0) {
            double m = 0.0;
            for (int w = 0; w < 16; ++w) m = fmax(m, s_red[3][w]);
            double *p = L.spart + (size_t)g * 4;
            p[0] = r0; p[1] = r1; p[2] = r2; p[3] = m;
        }
        return;
    }
    float sub = 0.f, mul = 1.f;
    if (MODE == 2) { const int b = img / C; sub = L.stat[b * 4 + 0]; mul = L.stat[b * 4 + 1]; }
    float *dst = grid + ((size_t)img * H + row0) * W;
    for (int i = tid; i < npix; i += 1024) {
        float v = vox_from_fixed((long long)s_acc[i]);
        if (MODE == 2) {                                 // (k_vox_norm's arithmetic)
            if (norm == 1) { if (v != 0.f) v = (v - sub) * mul; }
            else v = v * mul;
        }
        dst[i] = v;
    }
}

// per-sample statistics of the non-zero entries: grid (nblk, B), 256 threads -> part[b][blk][4]
__global__ __launch_bounds__(256) void k_vox_stats(const float *__restrict__ grid, double *__restrict__ part, int64_t per_sample) {
    __shared__ double s_red[4][4];
    const float *g = grid + (size_t)blockIdx.y * per_sample;
    double cnt = 0.0, sum = 0.0, sq = 0.0, mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        if (v != 0.f) { cnt += 1.0; sum += (double)v; sq += (double)v * (double)v; mx = fmax(mx, fabs((double)v)); }
    }
    const double r0 = block_sum_d<256>(cnt, s_red[0]);
    const double r1 = block_sum_d<256>(sum, s_red[1]);
    const double r2 = block_sum_d<256>(sq, s_red[2]);
    // max via the same reduction shape
    double m = mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_red[3][threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double *p = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        p[0] = r0; p[1] = r1; p[2] = r2;
        p[3] = fmax(fmax(s_red[3][0], s_red[3][1]), fmax(s_red[3][2], s_red[3][3]));
    }
}

// one workgroup per sample: mean / std of the non-zero entries (unbiased std, torch.std) or max
// (part: nblk partials per sample -- those of the strips, or of k_vox_stats)
__global__ __launch_bounds__(256) void k_vox_finalize(const double *__restrict__ part, float *__restrict__ stat, int nblk, int norm) {
    __shared__ double s_red[4][4];
    const int b = blockIdx.x;
    double cnt = 0.0, sum = 0.0, sq = 0.0, mx = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const double *p = part + ((size_t)b * nblk + i) * 4;
        cnt += p[0]; sum += p[1]; sq += p[2]; mx = fmax(mx, p[3]);
    }
    const double n = block_sum_d<256>(cnt, s_red[0]);
    const double s1 = block_sum_d<256>(sum, s_red[1]);
    const double s2 = block_sum_d<256>(sq, s_red[2]);
    double m = mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_red[3][threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(s_red[3][0], s_red[3][1]), fmax(s_red[3][2], s_red[3][3]));
        float sub = 0.f, mul = 1.f;
        if (norm == 1 && n > 0.0) {                      // utils.py:61-69
            const double mean = s1 / n;
            const double var = n > 1.0 ? (s2 - n * mean * mean) / (n - 1.0) : NAN;
            const double sd = sqrt(var);
            sub = (float)mean;
            mul = (sd > 0.0) ? (float)(1.0 / sd) : 1.f;   // std == 0, or NaN for a single entry: only subtract (utils.py:66-69)
        } else if (norm == 2 && m > 0.0) {               // utils.py:70-73
            mul = (float)(1.0 / m);
        }
        stat[b * 4 + 0] = sub;
        stat[b * 4 + 1] = mul;
    }
}

__global__ __launch_bounds__(256) void k_vox_norm(float *__restrict__ grid, const float *__restrict__ stat, int64_t per_sample, int norm) {
    const int b = blockIdx.y;
    const float sub = stat[b * 4 + 0], mul = stat[b * 4 + 1];
    float *g = grid + (size_t)b * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        if (norm == 1) { if (v != 0.f) g[i] = (v - sub) * mul; }
        else g[i] = v * mul;
    }
}

// ------------------------------------------------------------------------------------------
// quantile clipping (utils.py:57-61): threshold = torch.quantile(|grid|.view(-1), 1 - quantile) per sample, entries
// beyond it are set to sign * threshold.  The order statistics come from a radix select over the bit patterns of
// |v| (non-negative floats order like their bits): three histogram passes (11 + 11 + 10 bits) find the element of rank
// k = floor(pos), a fourth pass the next larger value (rank k + 1 unless it repeats), then torch's own interpolation:
// pos = float(1 - quantile) * float(n - 1) in fp32, weight = pos - floor(pos), lerp(below, above, weight).
//   qstate[b][8]: 0 prefix of the key found so far, 1 rank still to go inside the prefix, 2 smallest key above the
//                 selected one (atomicMin), 3 elements <= selected key, 4 threshold (float bits)
// ------------------------------------------------------------------------------------------
#define VOX_QBINS 2048
__device__ __forceinline__ void vox_qpass(int pass, int &shift, int &bits) {
    shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    bits = pass == 2 ? 10 : 11;
}

// grid (nblk, B), 256 threads
__global__ __launch_bounds__(256) void k_vox_qhist(const float *__restrict__ grid, unsigned *__restrict__ hist,
                                                   const unsigned *__restrict__ qstate, int64_t per_sample, int pass) {
    __shared__ unsigned s_h[VOX_QBINS];
    int shift, bits;
    vox_qpass(pass, shift, bits);
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < VOX_QBINS; i += 256) s_h[i] = 0u;
    __syncthreads();
    const unsigned prefix = qstate[b * 8 + 0];
    const float *g = grid + (size_t)b * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const unsigned key = __float_as_uint(fabsf(g[i]));
        const bool match = pass == 0 || (key >> (shift + bits)) == prefix;
        if (match) atomicAdd(&s_h[(key >> shift) & ((1u << bits) - 1u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < VOX_QBINS; i += 256)
        if (s_h[i]) atomicAdd(&hist[(size_t)b * VOX_QBINS + i], s_h[i]);
}

// grid B, 256 threads: the bin holding the wanted rank; zeroes the histogram for the next pass
__global__ __launch_bounds__(256) void k_vox_qscan(unsigned *__restrict__ hist, unsigned *__restrict__ qstate, int64_t per_sample,
                                                   float qf, int pass) {
    __shared__ unsigned s_c[VOX_QBINS];
    __shared__ unsigned s_part[256];
    int shift, bits;
    vox_qpass(pass, shift, bits);
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned *h = hist + (size_t)b * VOX_QBINS;
    if (pass == 0 && tid == 0) {
        const float pos = qf * (float)(per_sample - 1);          // torch.quantile: q and the ranks in the input dtype
        qstate[b * 8 + 0] = 0u;
        qstate[b * 8 + 1] = (unsigned)floorf(pos);
        qstate[b * 8 + 2] = 0xffffffffu;
        qstate[b * 8 + 3] = 0u;
    }
    unsigned loc = 0u;
    for (int i = 0; i < VOX_QBINS / 256; ++i) { s_c[tid * (VOX_QBINS / 256) + i] = h[tid * (VOX_QBINS / 256) + i]; loc += s_c[tid * (VOX_QBINS / 256) + i]; }
    s_part[tid] = loc;
    __syncthreads();
    if (tid == 0) {
        unsigned k = qstate[b * 8 + 1], below = qstate[b * 8 + 3], run = 0u;
        int t = 0;
        while (t < 255 && run + s_part[t] <= k) { run += s_part[t]; ++t; }
        int bin = t * (VOX_QBINS / 256);
        while (bin < VOX_QBINS - 1 && run + s_c[bin] <= k) { run += s_c[bin]; ++bin; }
        qstate[b * 8 + 0] = (qstate[b * 8 + 0] << bits) | (unsigned)bin;
        qstate[b * 8 + 1] = k - run;
        // elements <= the selected key once the last pass is done: everything before the bin at every level, + the bin
        qstate[b * 8 + 3] = below + run + (pass == 2 ? s_c[bin] : 0u);
    }
    __syncthreads();
    for (int i = tid; i < VOX_QBINS; i += 256) h[i] = 0u;
}

// grid (nblk, B): smallest key above the selected one
__global__ __launch_bounds__(256) void k_vox_qnext(const float *__restrict__ grid, unsigned *__restrict__ qstate, int64_t per_sample) {
    const int b = blockIdx.y;
    const unsigned sel = qstate[b * 8 + 0];
    const float *g = grid + (size_t)b * per_sample;
    unsigned m = 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const unsigned key = __float_as_uint(fabsf(g[i]));
        if (key > sel) m = min(m, key);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_down((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m != 0xffffffffu) atomicMin(&qstate[b * 8 + 2], m);
}

// grid (nblk, B): threshold (torch's lerp) and the clip itself, in place
__global__ __launch_bounds__(256) void k_vox_qclip(float *__restrict__ grid, unsigned *__restrict__ qstate, int64_t per_sample, float qf) {
    const int b = blockIdx.y;
    const float pos = qf * (float)(per_sample - 1);
    const float fl = floorf(pos), w = pos - fl;
    const unsigned k0 = (unsigned)fl, k1 = (unsigned)ceilf(pos);
    const float below = __uint_as_float(qstate[b * 8 + 0]);
    // rank k1 == k0, or the selected value repeats beyond rank k0, or the next larger value
    const float above = (k1 == k0 || qstate[b * 8 + 3] > k1) ? below : __uint_as_float(qstate[b * 8 + 2]);
    const float d = above - below;
    const float thr = (w < 0.5f) ? below + w * d : above - d * (1.f - w);       // at::lerp
    float *g = grid + (size_t)b * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        if (fabsf(v) > thr) g[i] = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * thr;        // sign(v) * threshold, utils.py:59-61
    }
}

// ------------------------------------------------------------------------------------------
static int vox_validate(const mpc_vox_shape *s) {
    MPC_CHECK_ARG(s->B >= 0 && s->N >= 0 && s->C >= 1 && s->H >= 1 && s->W >= 1, MPC_E_SHAPE, "bad voxel-grid shape");
    MPC_CHECK_ARG(s->norm >= 0 && s->norm <= 2, MPC_E_SHAPE, "norm must be 0 (none), 1 (mean_std) or 2 (max)");
    MPC_CHECK_ARG(s->quantile >= 0.f && s->quantile < 0.15f, MPC_E_SHAPE, "quantile must lie in [0, 0.15) (utils.py:27)");
    MPC_CHECK_ARG((int64_t)s->W * 8 <= 150 * 1024, MPC_E_UNSUPPORTED, "sensor too wide for one LDS strip row");
    MPC_CHECK_ARG((int64_t)s->B * s->C * s->H * s->W < (1LL << 31), MPC_E_UNSUPPORTED, "voxel grid too large");
    return 0;
}

struct VoxHostLayout { VoxLayout L; int64_t off_count, off_rec, off_ovf, off_chunk, off_part, off_spart, off_stat, off_qhist, off_qstate, total; unsigned *qhist, *qstate; };

static VoxHostLayout vox_layout(const mpc_vox_shape *s, void *ws) {
    VoxHostLayout h;
    VoxLayout &L = h.L;
    L.SR = (int)(((int64_t)VOX_STRIP_KB * 1024) / ((int64_t)s->W * 8));
    if (L.SR < 1) L.SR = (int)((150 * 1024) / ((int64_t)s->W * 8));
    if (L.SR > s->H) L.SR = s->H;
    L.NS = mpc_cdiv(s->H, L.SR);
    L.SR = mpc_cdiv(s->H, L.NS);
    L.NBk = s->B * s->C * L.NS;
    int64_t cap = 4 * ((2 * (int64_t)s->N + (int64_t)s->C * L.NS - 1) / ((int64_t)s->C * L.NS));
    if (cap < 4096) cap = 4096;
    if (cap > 2 * (int64_t)s->N) cap = 2 * (int64_t)s->N;
    L.cap = (int)(cap > 0 ? cap : 1);
    L.nstat_blocks = 256;
    int64_t off = 0;
    // spill region of a sample: every record it can produce (an event votes into two channels x up to two strips); chunk
    // list of a sample: one descriptor per (binning workgroup, bucket it overflowed) -- a binning workgroup holds
    // 256 * VOX_PER_THREAD events, i.e. at most that many x 4 records, in at most C * NS buckets
    L.spcap = (int)(4 * (int64_t)s->N > 0 ? 4 * (int64_t)s->N : 1);
    {
        const int64_t wgs = mpc_cdiv(s->N > 0 ? s->N : 1, 256 * VOX_PER_THREAD);
        const int64_t per_wg = (int64_t)s->C * L.NS < 256 * VOX_PER_THREAD * 4 ? (int64_t)s->C * L.NS : 256 * VOX_PER_THREAD * 4;
        L.chcap = (int)(wgs * per_wg);
    }
    h.off_count = off; off += mpc_align((int64_t)(L.NBk + 2 * (s->B > 0 ? s->B : 1) + 8) * 4);
    h.off_rec = off;   off += mpc_align((int64_t)L.NBk * L.cap * 16);
    h.off_ovf = off;   off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * L.spcap * 16 + 16);
    h.off_chunk = off; off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * L.chcap * 16 + 16);
    h.off_part = off;  off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * L.nstat_blocks * 4 * 8);
    h.off_spart = off; off += mpc_align((int64_t)(L.NBk > 0 ? L.NBk : 1) * 4 * 8);
    h.off_stat = off;  off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * 4 * 4);
    h.off_qhist = off; off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * VOX_QBINS * 4);
    h.off_qstate = off; off += mpc_align((int64_t)(s->B > 0 ? s->B : 1) * 8 * 4);
    h.total = off;
    char *w = (char *)ws;
    L.gcount = (int *)(w + h.off_count);
    L.rec = (float4 *)(w + h.off_rec);
    L.ovf = (float4 *)(w + h.off_ovf);
    L.chunk = (int4 *)(w + h.off_chunk);
    L.part = (double *)(w + h.off_part);
    L.spart = (double *)(w + h.off_spart);
    L.stat = (float *)(w + h.off_stat);
    h.qhist = (unsigned *)(w + h.off_qhist);
    h.qstate = (unsigned *)(w + h.off_qstate);
    return h;
}

extern "C" int64_t mpc_voxel_workspace_bytes(const mpc_vox_shape *s) {
    if (!s) { mpc_set_error("mpc_voxel_workspace_bytes: null shape"); return MPC_E_NULL; }
    int rc = vox_validate(s);
    if (rc) return rc;
    return vox_layout(s, nullptr).total;
}

extern "C" int mpc_voxel_grid(const mpc_vox_shape *s, const float *xytp, const int32_t *counts, float *grid,
                              void *ws, void *stream) {
    MPC_CHECK_ARG(s && counts && grid && ws && (xytp || s->N == 0 || s->B == 0), MPC_E_NULL, "null argument");
    int rc = vox_validate(s);
    if (rc) return rc;
    if (s->B == 0) return 0;
    const VoxHostLayout h = vox_layout(s, ws);
    const VoxLayout &L = h.L;
    hipStream_t st = (hipStream_t)stream;
    static mpc_device_once attr_once;   // raising the dynamic-LDS cap: idempotent, once per device
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_vox_accum<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_vox_accum<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_vox_accum<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
        attr_once.mark();
    }
    const int e0 = mpc_zero_async(L.gcount, (size_t)(L.NBk + 2 * s->B + 8) * 4, st);
    if (e0) return e0;
    if (s->N > 0) {
        const int nblk = mpc_cdiv(s->N, 256 * VOX_PER_THREAD) * s->B;
        MPC_LAUNCH(k_vox_bin, dim3(((nblk + 7) / 8) * 8), dim3(256), (size_t)s->C * L.NS * 3 * sizeof(int), st,
                           *s, L, reinterpret_cast<const float4 *>(xytp), counts);
        MPC_CHECK_LAUNCH();
    }
    const size_t strip_lds = (size_t)L.SR * s->W * 8;
    const int64_t per_sample = (int64_t)s->C * s->H * s->W;
    if (s->norm != 0 && !(s->quantile > 0.f)) {
        // statistics from the strips in LDS, then the strips again, written normalised: the grid is written once and never read
        MPC_LAUNCH(k_vox_accum<1>, dim3(L.NBk), dim3(1024), strip_lds, st, L, grid, s->H, s->W, s->C, s->norm);
        MPC_LAUNCH(k_vox_finalize, dim3(s->B), dim3(256), 0, st, L.spart, L.stat, s->C * L.NS, s->norm);
        MPC_LAUNCH(k_vox_accum<2>, dim3(L.NBk), dim3(1024), strip_lds, st, L, grid, s->H, s->W, s->C, s->norm);
        MPC_CHECK_LAUNCH();
        return 0;
    }
    MPC_LAUNCH(k_vox_accum<0>, dim3(L.NBk), dim3(1024), strip_lds, st, L, grid, s->H, s->W, s->C, s->norm);
    MPC_CHECK_LAUNCH();
    if (s->quantile > 0.f) {
        // torch.quantile(x, 1 - q) gets 1 - q computed in double and rounds it to fp32: the caller passes that value
        const float qf = s->keep > 0.f ? s->keep : (float)(1.0 - (double)s->quantile);
        const int e1 = mpc_zero_async(h.qhist, (size_t)s->B * VOX_QBINS * 4, st);
        if (e1) return e1;
        const dim3 gq(L.nstat_blocks, s->B);
        for (int pass = 0; pass < 3; ++pass) {
            MPC_LAUNCH(k_vox_qhist, gq, dim3(256), 0, st, grid, h.qhist, h.qstate, per_sample, pass);
            MPC_LAUNCH(k_vox_qscan, dim3(s->B), dim3(256), 0, st, h.qhist, h.qstate, per_sample, qf, pass);
        }
        MPC_LAUNCH(k_vox_qnext, gq, dim3(256), 0, st, grid, h.qstate, per_sample);
        MPC_LAUNCH(k_vox_qclip, gq, dim3(256), 0, st, grid, h.qstate, per_sample, qf);
        MPC_CHECK_LAUNCH();
    }
    if (s->norm != 0) {
        MPC_LAUNCH(k_vox_stats, dim3(L.nstat_blocks, s->B), dim3(256), 0, st, grid, L.part, per_sample);
        MPC_LAUNCH(k_vox_finalize, dim3(s->B), dim3(256), 0, st, L.part, L.stat, L.nstat_blocks, s->norm);
        MPC_LAUNCH(k_vox_norm, dim3(L.nstat_blocks, s->B), dim3(256), 0, st, grid, L.stat, per_sample, s->norm);
        MPC_CHECK_LAUNCH();
    }
    return 0;
}

MPC_BOUNDS_UNIT("voxel.hip")
