// Flow curves sampled at the tile centres as `trajectories` for FocusLoss.calc (BASELINE.json configs[3]) -- the product
// of a curve's control points with a basis matrix evaluated on the host, and its adjoint.
//   reference: src/models/raft_spline/curves/base.py:88-123 (CurveBase.get_flow_from_reference), bezier.py:92-113: the flow at time t
//   is sum_k B_k(t) P_k with P_0 == 0; dim 1 of the parameter tensor is in (x, y) order (polynomial.py:60-61), trajectories are (y, x).
// In plain torch (utils/basis.py, the path of CPU tensors) this is an einsum, a stack, a reshape and an add, and in the backward their
// four adjoints: a dozen launches of a few microseconds each around a loss step of 0.26 ms that is bound by the host (round 5: the
// cubic B-spline step 0.53-0.61 ms against 0.26 for precomputed trajectories).  Here: one kernel each way.
//   traj[b][t][i] = (pos[i].y + scale * sum_k basis[t][k] params[b][1][k][i],  pos[i].x + scale * sum_k basis[t][k] params[b][0][k][i])
//   grad_params[b][c][k][i] = scale * sum_t basis[t][k] grad_traj[b][t][i][1 - c]
// One thread per (sample, tile): the 2 d control values (forward) or the 2 d sums (backward) live in registers, the basis matrix in
// LDS; every global access is coalesced over the tiles.  Sums run over k (forward) and t (backward) in index order, one rounding per
// multiply and per add (-ffp-contract=off): reproducible; they differ from a BLAS einsum by the order of a 10-term fp32 sum.
#include "common.h"

#define CURVE_DMAX 16          // control points per axis held in registers (the reference's curves: degree <= 10)

template <int D>
__global__ __launch_bounds__(256) void k_curve_traj_fwd(const float *__restrict__ params, const float *__restrict__ basis,
                                                        const float *__restrict__ pos, float scale, float *__restrict__ traj,
                                                        int B, int d, int T, int n) {
    extern __shared__ float s_basis[];          // [T][d]
    for (int i = threadIdx.x; i < T * d; i += 256) s_basis[i] = basis[i];
    __syncthreads();
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (long long)B * n) return;
    const int b = (int)(gi / n), i = (int)(gi - (long long)b * n);
    float px[D], py[D];
    const float *pb = params + (size_t)b * 2 * d * n;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        px[k] = k < d ? pb[(size_t)k * n + i] : 0.f;                    // channel 0: x
        py[k] = k < d ? pb[(size_t)(d + k) * n + i] : 0.f;              // channel 1: y
    }
    const float2 p0 = reinterpret_cast<const float2 *>(pos)[i];
    float2 *out = reinterpret_cast<float2 *>(traj) + (size_t)b * T * n + i;
    for (int t = 0; t < T; ++t) {
        float fy = 0.f, fx = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < d) { const float w = s_basis[t * d + k]; fy = fy + w * py[k]; fx = fx + w * px[k]; }
        }
        out[(size_t)t * n] = make_float2(p0.x + fy * scale, p0.y + fx * scale);
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_curve_traj_bwd(const float *__restrict__ grad_traj, const float *__restrict__ basis, float scale,
                                                        float *__restrict__ grad_params, int B, int d, int T, int n) {
    extern __shared__ float s_basis[];
    for (int i = threadIdx.x; i < T * d; i += 256) s_basis[i] = basis[i];
    __syncthreads();
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (long long)B * n) return;
    const int b = (int)(gi / n), i = (int)(gi - (long long)b * n);
    float gx[D], gy[D];
#pragma unroll
    for (int k = 0; k < D; ++k) gx[k] = gy[k] = 0.f;
    const float2 *g = reinterpret_cast<const float2 *>(grad_traj) + (size_t)b * T * n + i;
    for (int t = 0; t < T; ++t) {
        const float2 gt = g[(size_t)t * n];                               // (d/dy, d/dx)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < d) { const float w = s_basis[t * d + k]; gy[k] = gy[k] + w * gt.x; gx[k] = gx[k] + w * gt.y; }
        }
    }
    float *gp = grad_params + (size_t)b * 2 * d * n;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < d) { gp[(size_t)k * n + i] = gx[k] * scale; gp[(size_t)(d + k) * n + i] = gy[k] * scale; }
    }
}

static int curve_check(const void *a, const void *b, const void *c, int B, int d, int T, int n, const char *who) {
    if (!a || !b || !c) { mpc_set_error("%s: null argument", who); return MPC_E_NULL; }
    if (B < 0 || n < 0 || d < 1 || T < 1) { mpc_set_error("%s: bad B / d / T / n", who); return MPC_E_SHAPE; }
    if (d > CURVE_DMAX || (size_t)T * d * sizeof(float) > 48 * 1024) { mpc_set_error("%s: more than %d control points per axis (or a basis matrix beyond 48 KB)", who, CURVE_DMAX); return MPC_E_UNSUPPORTED; }
    return 0;
}

extern "C" int mpc_curve_traj_fwd(const float *params, const float *basis, const float *pos, float scale, float *traj,
                                  int32_t B, int32_t d, int32_t T, int32_t n, void *stream) {
    int rc = curve_check(params, basis, pos, B, d, T, n, __func__);
    if (rc) return rc;
    MPC_CHECK_ARG(traj, MPC_E_NULL, "null argument");
    if ((long long)B * n == 0) return 0;
    const dim3 grid((unsigned)(((long long)B * n + 255) / 256));
    const size_t lds = (size_t)T * d * sizeof(float);
    if (d <= 4) MPC_LAUNCH(k_curve_traj_fwd<4>, grid, dim3(256), lds, (hipStream_t)stream, params, basis, pos, scale, traj, B, d, T, n);
    else if (d <= 10) MPC_LAUNCH(k_curve_traj_fwd<10>, grid, dim3(256), lds, (hipStream_t)stream, params, basis, pos, scale, traj, B, d, T, n);
    else MPC_LAUNCH(k_curve_traj_fwd<CURVE_DMAX>, grid, dim3(256), lds, (hipStream_t)stream, params, basis, pos, scale, traj, B, d, T, n);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_curve_traj_bwd(const float *grad_traj, const float *basis, float scale, float *grad_params,
                                  int32_t B, int32_t d, int32_t T, int32_t n, void *stream) {
    int rc = curve_check(grad_traj, basis, grad_params, B, d, T, n, __func__);
    if (rc) return rc;
    if ((long long)B * n == 0) return 0;
    const dim3 grid((unsigned)(((long long)B * n + 255) / 256));
    const size_t lds = (size_t)T * d * sizeof(float);
    if (d <= 4) MPC_LAUNCH(k_curve_traj_bwd<4>, grid, dim3(256), lds, (hipStream_t)stream, grad_traj, basis, scale, grad_params, B, d, T, n);
    else if (d <= 10) MPC_LAUNCH(k_curve_traj_bwd<10>, grid, dim3(256), lds, (hipStream_t)stream, grad_traj, basis, scale, grad_params, B, d, T, n);
    else MPC_LAUNCH(k_curve_traj_bwd<CURVE_DMAX>, grid, dim3(256), lds, (hipStream_t)stream, grad_traj, basis, scale, grad_params, B, d, T, n);
    MPC_CHECK_LAUNCH();
    return 0;
}
