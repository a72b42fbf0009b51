// Per-event part of the CMax loss: LUT warp, event weight, bilinear vote into the Image of
// Warped Events, and its hand-derived backward into the flow look-up table.
//   forward : reference src/losses/focus.py:182-230 + src/utils/event_image_converter.py:333-391
//   backward: SURVEY.md 8a row A11
// Arithmetic follows SURVEY.md Appendix A op for op (fp32, no FMA contraction: this file is
// compiled with -ffp-contract=off).
#include "common.h"

struct EvParams {
    int B, M, Mp, nb, T, H, W, sp, hq, wq, P;
    unsigned flags;
};

struct Warped {
    float y, x, w;     // warped position and event weight
    int lut;           // index of the LUT cell (b,bin,iy,ix) -> element offset / (2T) ... see below
    float fy, fx;      // fractional parts
    int y0, x0;        // top-left tap
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
__device__ __forceinline__ bool warp_event(const EvParams &p, const float e[6], int b, int tr,
                                           const float *__restrict__ lut, float t_ref, Warped &o) {
    float w = (p.flags & MPC_F_UNIT_WEIGHT) ? 1.0f : e[5];
    float y = e[0], x = e[1];
    o.lut = -1;
    if (!(p.flags & MPC_F_NO_WARP)) {
        // focus.py:184-191: it = int(bin), iy = int(y // sp), ix = int(x // sp); pos = lut + event
        int it = (int)e[4];
        int iy = (int)floorf(e[0] / (float)p.sp);
        int ix = (int)floorf(e[1] / (float)p.sp);
        // torch indexing would raise on out-of-range indices; clamp instead of faulting
        it = min(max(it, 0), p.nb - 1);
        iy = min(max(iy, 0), p.hq - 1);
        ix = min(max(ix, 0), p.wq - 1);
        o.lut = (((b * p.nb + it) * p.hq + iy) * p.wq + ix) * p.T + tr;
        const float2 f = reinterpret_cast<const float2 *>(lut)[o.lut];
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
// host entry points
// ------------------------------------------------------------------------------------------
extern "C" int mpc_event_splat_fwd(const mpc_shape *s, const float *events, const float *flow_lut,
                                   const float *t_ref, float *iwe_raw, void *ws, void *stream) {
    MPC_CHECK_ARG(s && events && iwe_raw && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG((s->flags & MPC_F_NO_WARP) || flow_lut, MPC_E_NULL, "flow_lut is null");
    MPC_CHECK_ARG(!(s->flags & MPC_F_SCALE_BY_DT) || t_ref, MPC_E_NULL, "t_ref is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    const size_t img_bytes = (size_t)L.nimg * s->H * s->W * sizeof(float);
    hipError_t e = hipMemsetAsync(iwe_raw, 0, img_bytes, st);
    if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_splat_fwd_atomic, dim3(grid), dim3(256), 0, st, *s, events, flow_lut, t_ref, iwe_raw);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_event_splat_bwd(const mpc_shape *s, const float *events, const float *flow_lut,
                                   const float *t_ref, const float *grad_iwe, const float *scal,
                                   const float *grad_out, float *grad_flow_lut, int32_t accumulate,
                                   void *ws, void *stream) {
    MPC_CHECK_ARG(s && events && flow_lut && grad_iwe && scal && grad_flow_lut && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(!(s->flags & MPC_F_NO_WARP), MPC_E_UNSUPPORTED, "no LUT to differentiate with MPC_F_NO_WARP");
    MPC_CHECK_ARG(!(s->flags & MPC_F_SCALE_BY_DT) || t_ref, MPC_E_NULL, "t_ref is null");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        const size_t bytes = (size_t)s->B * s->nb * s->hq * s->wq * s->T * 2 * sizeof(float);
        hipError_t e = hipMemsetAsync(grad_flow_lut, 0, bytes, st);
        if (e != hipSuccess) { mpc_set_error("%s: %s", __func__, hipGetErrorString(e)); return (int)e; }
    }
    const int64_t total = (int64_t)s->B * s->M;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_splat_bwd_atomic, dim3(grid), dim3(256), 0, st, *s, events, flow_lut, t_ref,
                       grad_iwe, scal, grad_out, grad_flow_lut);
    MPC_CHECK_LAUNCH();
    return 0;
}
