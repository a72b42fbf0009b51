// UNPINNED EXTENSION (FocusLoss.calc_per_event_basis; BASELINE.json's north_star names a per-event continuous-time warp, the reference
// has none: focus.py:182-195 gathers a binned flow LUT): the small dense <-> per-tile operators around mpc_pe_warp / mpc_pe_grad_ordered,
// as kernels.  In plain torch (round 4-5) they were two dozen operators -- slicing the tile centres out of the network's dense
// coefficient grid (src/utils/trajectories.py:3-52), a 103 MB zero fill + strided copy for its adjoint, two small GEMMs for the
// smoothness field (src/utils/basis.py:18-31 at the bin mid-times) and their glue -- around 0.30 ms of library kernels: the step was
// bound by the host (0.74-0.97 ms).  Four kernels; every global access coalesced over the tiles / pixels.
//   k_tile_rows        rows[(b, iy, ix)][c]      = sum_s grid[b][s][c][iy * tile + tile / 2][ix * tile + tile / 2]
//   k_tile_rows_bwd    ggrid[b][s][c][y][x]      = grows[(b, y / tile, x / tile)][c] at a tile centre, else 0  (EVERY element written)
//   k_basis_field      field[(b, t)][cell][d]    = sum_j rows[(b, cell)][d][j] phim[t][j]            (the layout mpc_lut_smooth takes)
//   k_rows_grad_finish out[(b, cell)][d][j]      = sum_split gp[split][(b, cell)][d][j] + gout * sum_t gfield[(b, t)][cell][d] phim[t][j]
#include "common.h"

#define TILE_KMAX 8            // basis orders per axis (mpc_pe_warp's register variant holds as many)

__global__ __launch_bounds__(256) void k_tile_rows(const float *__restrict__ grid, float *__restrict__ rows, int B, int S, int c2, int H, int W,
                                                   int tile, int hq, int wq) {
    const long long total = (long long)B * hq * wq * c2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // (thread -> (b, c, iy, ix): neighbouring lanes read neighbouring tiles of one channel plane; the row write is strided by 2k floats)
    const int ix = (int)(i % wq);
    long long r = i / wq;
    const int iy = (int)(r % hq); r /= hq;
    const int c = (int)(r % c2), b = (int)(r / c2);
    const int y = iy * tile + tile / 2, x = ix * tile + tile / 2;
    float v = 0.f;
    if (y < H && x < W)
        for (int s = 0; s < S; ++s) v += grid[((((size_t)b * S + s) * c2 + c) * H + y) * W + x];
    rows[(((size_t)b * hq + iy) * wq + ix) * c2 + c] = v;
}

__global__ __launch_bounds__(256) void k_tile_rows_bwd(const float *__restrict__ grows, float *__restrict__ ggrid, int B, int S, int c2, int H, int W,
                                                       int tile, int hq, int wq) {
    // one thread per four consecutive pixels of a row (16-byte stores where W is a multiple of 4; the zero fill IS the kernel: the
    // dense gradient is 103 MB at the DSEC batch shape)
    const int W4 = (W + 3) >> 2;
    const long long total = (long long)B * S * c2 * H * W4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x4 = (int)(i % W4);
        long long r = i / W4;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % c2); r /= c2;
        const int b = (int)(r / S);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int iy = y / tile;
        if (y - iy * tile == tile / 2 && iy < hq) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = 4 * x4 + u, ix = x / tile;
                if (x < W && x - ix * tile == tile / 2 && ix < wq) v[u] = grows[(((size_t)b * hq + iy) * wq + ix) * c2 + c];
            }
        }
        float *dst = ggrid + (size_t)(i / W4) * W + 4 * x4;
        if ((W & 3) == 0) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int u = 0; u < 4 && 4 * x4 + u < W; ++u) dst[u] = v[u];
    }
}

__global__ __launch_bounds__(256) void k_basis_field(const float *__restrict__ rows, const float *__restrict__ phim, float *__restrict__ field,
                                                     int B, int G, int k, int nb) {
    __shared__ float s_phi[64 * TILE_KMAX];
    for (int i = threadIdx.x; i < nb * k; i += 256) s_phi[i] = phim[i];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (b, cell)
    if (i >= (long long)B * G) return;
    const int b = (int)(i / G), cell = (int)(i - (long long)b * G);
    float cy[TILE_KMAX], cx[TILE_KMAX];
#pragma unroll
    for (int j = 0; j < TILE_KMAX; ++j) {
        cy[j] = j < k ? rows[(size_t)i * 2 * k + j] : 0.f;
        cx[j] = j < k ? rows[(size_t)i * 2 * k + k + j] : 0.f;
    }
    for (int t = 0; t < nb; ++t) {
        float fy = 0.f, fx = 0.f;
#pragma unroll
        for (int j = 0; j < TILE_KMAX; ++j)
            if (j < k) { fy += cy[j] * s_phi[t * k + j]; fx += cx[j] * s_phi[t * k + j]; }
        reinterpret_cast<float2 *>(field)[((size_t)b * nb + t) * G + cell] = make_float2(fy, fx);
    }
}

__global__ __launch_bounds__(256) void k_rows_grad_finish(const float *__restrict__ gp, int split, const float *__restrict__ gfield,
                                                          const float *__restrict__ phim, const float *__restrict__ gout,
                                                          float *__restrict__ out, int B, int G, int k, int nb) {
    __shared__ float s_phi[64 * TILE_KMAX];
    for (int i = threadIdx.x; i < nb * k; i += 256) s_phi[i] = phim ? phim[i] : 0.f;
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (b, cell)
    const long long n = (long long)B * G;
    if (i >= n) return;
    const int b = (int)(i / G), cell = (int)(i - (long long)b * G);
    float gy[TILE_KMAX], gx[TILE_KMAX];
#pragma unroll
    for (int j = 0; j < TILE_KMAX; ++j) gy[j] = gx[j] = 0.f;
    if (gfield != nullptr) {
        const float go = gout ? gout[0] : 1.f;
        for (int t = 0; t < nb; ++t) {
            const float2 g = reinterpret_cast<const float2 *>(gfield)[((size_t)b * nb + t) * G + cell];
#pragma unroll
            for (int j = 0; j < TILE_KMAX; ++j)
                if (j < k) { gy[j] += g.x * s_phi[t * k + j]; gx[j] += g.y * s_phi[t * k + j]; }
        }
#pragma unroll
        for (int j = 0; j < TILE_KMAX; ++j) { gy[j] *= go; gx[j] *= go; }
    }
    for (int s = 0; s < split; ++s) {                       // (the partial sums of mpc_pe_grad_ordered's `split` workgroups per strip: in order)
        const float *src = gp + ((size_t)s * n + i) * 2 * k;
#pragma unroll
        for (int j = 0; j < TILE_KMAX; ++j)
            if (j < k) { gy[j] += src[j]; gx[j] += src[k + j]; }
    }
#pragma unroll
    for (int j = 0; j < TILE_KMAX; ++j)
        if (j < k) { out[(size_t)i * 2 * k + j] = gy[j]; out[(size_t)i * 2 * k + k + j] = gx[j]; }
}

static int tile_args(int B, int S, int c2, int H, int W, int tile, const char *who) {
    if (B < 0 || S < 1 || c2 < 2 || (c2 & 1) || H < 1 || W < 1 || tile < 1) { mpc_set_error("%s: bad shape", who); return MPC_E_SHAPE; }
    return 0;
}

extern "C" int mpc_pe_tile_rows(const float *grid, float *rows, int32_t B, int32_t S, int32_t c2, int32_t H, int32_t W, int32_t tile, void *stream) {
    MPC_CHECK_ARG(grid && rows, MPC_E_NULL, "null argument");
    int rc = tile_args(B, S, c2, H, W, tile, __func__);
    if (rc) return rc;
    const int hq = (H + tile - 1) / tile, wq = (W + tile - 1) / tile;
    const long long total = (long long)B * hq * wq * c2;
    if (total == 0) return 0;
    MPC_LAUNCH(k_tile_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grid, rows, B, S, c2, H, W, tile, hq, wq);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_pe_tile_rows_bwd(const float *grad_rows, float *grad_grid, int32_t B, int32_t S, int32_t c2, int32_t H, int32_t W, int32_t tile, void *stream) {
    MPC_CHECK_ARG(grad_rows && grad_grid, MPC_E_NULL, "null argument");
    int rc = tile_args(B, S, c2, H, W, tile, __func__);
    if (rc) return rc;
    const int hq = (H + tile - 1) / tile, wq = (W + tile - 1) / tile;
    const long long total = (long long)B * S * c2 * H * ((W + 3) / 4);
    if (total == 0) return 0;
    const long long blocks = (total + 255) / 256;
    MPC_LAUNCH(k_tile_rows_bwd, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, grad_rows, grad_grid, B, S, c2, H, W, tile, hq, wq);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_pe_basis_field(const float *rows, const float *phim, float *field, int32_t B, int32_t G, int32_t k, int32_t nb, void *stream) {
    MPC_CHECK_ARG(rows && phim && field, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(B >= 0 && G >= 0 && k >= 1 && nb >= 1, MPC_E_SHAPE, "bad shape");
    MPC_CHECK_ARG(k <= TILE_KMAX && nb <= 64, MPC_E_UNSUPPORTED, "more than 8 basis orders or 64 time bins");
    if ((long long)B * G == 0) return 0;
    MPC_LAUNCH(k_basis_field, dim3((unsigned)(((long long)B * G + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, phim, field, B, G, k, nb);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_pe_rows_grad_finish(const float *grad_parts, int32_t split, const float *grad_field, const float *phim, const float *grad_out,
                                       float *grad_rows, int32_t B, int32_t G, int32_t k, int32_t nb, void *stream) {
    MPC_CHECK_ARG(grad_rows && (grad_parts || split == 0) && (!grad_field || phim), MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(B >= 0 && G >= 0 && k >= 1 && split >= 0 && (!grad_field || nb >= 1), MPC_E_SHAPE, "bad shape");
    MPC_CHECK_ARG(k <= TILE_KMAX && nb <= 64, MPC_E_UNSUPPORTED, "more than 8 basis orders or 64 time bins");
    if ((long long)B * G == 0) return 0;
    MPC_LAUNCH(k_rows_grad_finish, dim3((unsigned)(((long long)B * G + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_parts, split, grad_field,
               phim, grad_out, grad_rows, B, G, k, nb);
    MPC_CHECK_LAUNCH();
    return 0;
}
