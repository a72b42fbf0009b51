// Dense flow from tile trajectories + flow error metrics: the validation / inference
// tail of the reference (src/modules/trajectory_net.py:124-140, scripts/dsec_inference.py:84-92,
// src/utils/metrics.py:50-56).
//
//  k_list_to_grid   src/utils/trajectories.py:54-76  [B][n][C] at pixel_positions // patch -> [B][C][hp][wp]
//  k_resize_aa      src/utils/flow.py:9-10: torchvision resize(BICUBIC, antialias=True) == torch's separable
//                   anti-aliased bicubic filter (a = -0.5, align_corners=False): per output index
//                   center = scale*(i+0.5), first tap max(int(center-support+0.5), 0), taps renormalised to
//                   sum 1; the width pass is applied first, then the height pass, both accumulated left to right.
//  k_flow_err_*     src/utils/flow.py:18-70: masked EPE / N-pixel-error / angular error.
//
// Both are pure streaming kernels (HBM roofline): the resize reads the patch grid through L1/L2 (1/16 of the
// output at patch 4) and writes each output element once; the metrics read 17 B per pixel once.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_list_to_grid(const mpc_flow_shape s, const float *__restrict__ list,
                                                      const long long *__restrict__ pix, float *__restrict__ grid) {
    const int hp = s.H / s.patch, wp = s.W / s.patch;
    const long long total = (long long)s.B * s.n * s.C;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % s.C);
    const long long r = i / s.C;
    const int k = (int)(r % s.n), b = (int)(r / s.n);
    const long long py = pix[2 * k] / s.patch, px = pix[2 * k + 1] / s.patch;
    if (py < 0 || py >= hp || px < 0 || px >= wp) return;              // torch would raise; nothing is written
    grid[(((long long)b * s.C + c) * hp + py) * wp + px] = list[i];
}

__device__ __forceinline__ float aa_cubic(float x) {                    // HelperInterpCubic::aa_filter, a = -0.5
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}

struct AaTaps {
    int first, count;
    float center, invscale, inv_total;
    __device__ __forceinline__ float weight(int j) const {
        return aa_cubic(((float)(j + first) - center + 0.5f) * invscale) * inv_total;
    }
};

__device__ __forceinline__ AaTaps aa_taps(int i, int n_in, float scale) {
    AaTaps t;
    const float support = scale >= 1.f ? 2.f * scale : 2.f;
    t.invscale = scale >= 1.f ? 1.f / scale : 1.f;
    t.center = scale * ((float)i + 0.5f);
    t.first = max((int)(t.center - support + 0.5f), 0);
    t.count = min((int)(t.center + support + 0.5f), n_in) - t.first;
    float tot = 0.f;
    t.inv_total = 1.f;
    for (int j = 0; j < t.count; ++j) tot += t.weight(j);
    t.inv_total = 1.f / tot;
    return t;
}

// grid (ceil(W/64), ceil(H/4), B*C), block (64, 4): one output element per thread
__global__ __launch_bounds__(256) void k_resize_aa(const mpc_flow_shape s, const float *__restrict__ src,
                                                   float *__restrict__ dst) {
    const int hp = s.H / s.patch, wp = s.W / s.patch;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= s.W || y >= s.H) return;
    const AaTaps tx = aa_taps(x, wp, (float)wp / (float)s.W), ty = aa_taps(y, hp, (float)hp / (float)s.H);
    const float *plane = src + (long long)blockIdx.z * hp * wp;
    float acc = 0.f;
    if (tx.count == 4) {                                     // interior of an upscale: weights kept in registers
        const float w0 = tx.weight(0), w1 = tx.weight(1), w2 = tx.weight(2), w3 = tx.weight(3);
        for (int j = 0; j < ty.count; ++j) {
            const float *row = plane + (long long)(ty.first + j) * wp + tx.first;
            float h = row[0] * w0;
            h += row[1] * w1;
            h += row[2] * w2;
            h += row[3] * w3;
            acc = j == 0 ? h * ty.weight(0) : acc + h * ty.weight(j);
        }
    } else {
        for (int j = 0; j < ty.count; ++j) {
            const float *row = plane + (long long)(ty.first + j) * wp + tx.first;
            float h = row[0] * tx.weight(0);
            for (int i = 1; i < tx.count; ++i) h += row[i] * tx.weight(i);
            acc = j == 0 ? h * ty.weight(0) : acc + h * ty.weight(j);
        }
    }
    dst[((long long)blockIdx.z * s.H + y) * s.W + x] = acc;
}

extern "C" int mpc_dense_flow(const mpc_flow_shape *s, const float *traj_flow, const int64_t *pixel_positions,
                              float *patch_flow, float *dense, void *stream) {
    MPC_CHECK_ARG(s && patch_flow && dense, -1, "null argument");
    MPC_CHECK_ARG(s->B >= 1 && s->C >= 1 && s->n >= 0 && s->patch >= 1 && s->H >= s->patch && s->W >= s->patch, -2,
                  "bad shape");
    MPC_CHECK_ARG((long long)s->B * s->C <= 65535, -2, "B*C exceeds 65535 planes");
    MPC_CHECK_ARG(s->n == 0 || (traj_flow && pixel_positions), -1, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const int hp = s->H / s->patch, wp = s->W / s->patch;
    const int e = mpc_zero_async(patch_flow, sizeof(float) * (size_t)s->B * s->C * hp * wp, st);
    if (e) return e;
    const long long total = (long long)s->B * s->n * s->C;
    if (total > 0)
        MPC_LAUNCH(k_list_to_grid, dim3(mpc_cdiv(total, 256)), dim3(256), 0, st, *s, traj_flow,
                           (const long long *)pixel_positions, patch_flow);
    MPC_LAUNCH(k_resize_aa, dim3(mpc_cdiv(s->W, 64), mpc_cdiv(s->H, 4), s->B * s->C), dim3(64, 4), 0, st, *s,
                       patch_flow, dense);
    MPC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
#define FE_BLOCKS 64          // partial-sum workgroups per sample
#define FE_NACC 6             // epe, >1, >2, >3, acos, n_points

struct FeAcc {
    double v[FE_NACC];
};

__device__ __forceinline__ void fe_pixel(float g0, float g1, float p0, float p1, bool em, float ts, bool has_ts,
                                         FeAcc &a) {
    const bool fm = !isinf(g0) && !isinf(g1) && fabsf(g0) > 0.f && fabsf(g1) > 0.f;
    const float m = (fm && em) ? 1.f : 0.f;
    g0 *= m; g1 *= m; p0 *= m; p1 *= m;                       // a product as in flow.py:48-49 (inf * 0 = nan)
    if (has_ts) { g0 *= ts; g1 *= ts; p0 *= ts; p1 *= ts; }
    const float d0 = g0 - p0, d1 = g1 - p1;
    const float epe = sqrtf(d0 * d0 + d1 * d1);
    a.v[0] += (double)epe;
    a.v[1] += epe > 1.f ? 1.0 : 0.0;
    a.v[2] += epe > 2.f ? 1.0 : 0.0;
    a.v[3] += epe > 3.f ? 1.0 : 0.0;
    // channel 0 is "u", channel 1 is "v" (flow.py:63-64)
    float cs = (1.0f + p0 * g0 + p1 * g1) / (sqrtf(1.f + p0 * p0 + p1 * p1) * sqrtf(1.f + g0 * g0 + g1 * g1));
    cs = cs < -1.f ? -1.f : (cs > 1.f ? 1.f : cs);            // torch.clamp keeps nan
    a.v[4] += (double)acosf(cs);
    a.v[5] += (double)m;
}

// grid (FE_BLOCKS, B), 256 threads
template <bool VEC>
__global__ __launch_bounds__(256) void k_flow_err_partial(const mpc_err_shape s, const float *__restrict__ gt,
                                                          const float *__restrict__ pred,
                                                          const unsigned char *__restrict__ mask,
                                                          const float *__restrict__ time_scale,
                                                          double *__restrict__ part) {
    __shared__ double s_red[4];
    const int b = blockIdx.y;
    const long long HW = (long long)s.H * s.W;
    const float *g0 = gt + (long long)b * 2 * HW, *g1 = g0 + HW;
    const float *p0 = pred + (long long)b * 2 * HW, *p1 = p0 + HW;
    const unsigned char *mk = mask ? mask + (long long)b * HW : nullptr;
    const bool has_ts = time_scale != nullptr;
    const float ts = has_ts ? time_scale[b] : 1.f;
    FeAcc a;
#pragma unroll
    for (int k = 0; k < FE_NACC; ++k) a.v[k] = 0.0;
    if (VEC) {
        const long long n4 = HW >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)FE_BLOCKS * 256) {
            const float4 a0 = reinterpret_cast<const float4 *>(g0)[i], a1 = reinterpret_cast<const float4 *>(g1)[i];
            const float4 b0 = reinterpret_cast<const float4 *>(p0)[i], b1 = reinterpret_cast<const float4 *>(p1)[i];
            uchar4 m = make_uchar4(1, 1, 1, 1);
            if (mk) m = reinterpret_cast<const uchar4 *>(mk)[i];
            fe_pixel(a0.x, a1.x, b0.x, b1.x, m.x != 0, ts, has_ts, a);
            fe_pixel(a0.y, a1.y, b0.y, b1.y, m.y != 0, ts, has_ts, a);
            fe_pixel(a0.z, a1.z, b0.z, b1.z, m.z != 0, ts, has_ts, a);
            fe_pixel(a0.w, a1.w, b0.w, b1.w, m.w != 0, ts, has_ts, a);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)FE_BLOCKS * 256)
            fe_pixel(g0[i], g1[i], p0[i], p1[i], mk ? mk[i] != 0 : true, ts, has_ts, a);
    }
#pragma unroll
    for (int k = 0; k < FE_NACC; ++k) {
        const double r = block_sum_d<256>(a.v[k], s_red);
        if (threadIdx.x == 0) part[((long long)b * FE_BLOCKS + blockIdx.x) * FE_NACC + k] = r;
    }
}

// 1 workgroup of 64 threads: per-sample ratios in fp32 (flow.py:58-61,69), then the mean over the batch
__global__ __launch_bounds__(64) void k_flow_err_final(const mpc_err_shape s, const double *__restrict__ part,
                                                       float *__restrict__ out) {
    __shared__ double s_red[1];
    double acc[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < s.B; b += 64) {
        double t[FE_NACC];
        for (int k = 0; k < FE_NACC; ++k) {
            t[k] = 0.0;
            for (int j = 0; j < FE_BLOCKS; ++j) t[k] += part[((long long)b * FE_BLOCKS + j) * FE_NACC + k];
        }
        const float npts = (float)t[5] + 1e-5f;
        for (int k = 0; k < 5; ++k) acc[k] += (double)((float)t[k] / npts);
    }
    for (int k = 0; k < 5; ++k) {
        const double r = block_sum_d<64>(acc[k], s_red);
        if (threadIdx.x == 0) {
            float v = (float)(r / (double)s.B);
            if (k == 4) v *= (float)(180.0 / 3.141592653589793);
            out[k] = v;
        }
    }
}

extern "C" int64_t mpc_flow_error_workspace_bytes(const mpc_err_shape *s) {
    if (!s || s->B < 1 || s->H < 1 || s->W < 1) return -2;
    return mpc_align((int64_t)s->B * FE_BLOCKS * FE_NACC * sizeof(double));
}

extern "C" int mpc_flow_error(const mpc_err_shape *s, const float *flow_gt, const float *flow_pred,
                              const uint8_t *event_mask, const float *time_scale, float *out, void *ws,
                              void *stream) {
    MPC_CHECK_ARG(s && flow_gt && flow_pred && out && ws, -1, "null argument");
    MPC_CHECK_ARG(s->B >= 1 && s->B <= 65535 && s->H >= 1 && s->W >= 1, -2, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)s->H * s->W;
    const bool vec = (HW & 3) == 0 && (((uintptr_t)flow_gt | (uintptr_t)flow_pred) & 15) == 0 &&
                     ((uintptr_t)event_mask & 3) == 0;
    if (vec)
        MPC_LAUNCH(k_flow_err_partial<true>, dim3(FE_BLOCKS, s->B), dim3(256), 0, st, *s, flow_gt, flow_pred,
                           event_mask, time_scale, (double *)ws);
    else
        MPC_LAUNCH(k_flow_err_partial<false>, dim3(FE_BLOCKS, s->B), dim3(256), 0, st, *s, flow_gt, flow_pred,
                           event_mask, time_scale, (double *)ws);
    MPC_LAUNCH(k_flow_err_final, dim3(1), dim3(64), 0, st, *s, (const double *)ws, out);
    MPC_CHECK_LAUNCH();
    return 0;
}
