// Contrast objective on the Image of Warped Events: 3x3 Gaussian blur (reflect), Sobel (zero pad),
// gradient-magnitude / variance reductions, and the hand-derived adjoint image.
//   forward  : reference src/utils/event_image_converter.py:170-175 + src/utils/loss.py:4-27,58-87
//   backward : SURVEY.md 8a row A11 (closed form of what autograd does in the reference)
// HBM-bound stencils: every image is read once and written once per kernel, tiles are staged
// in LDS with their halo, reductions use wavefront shuffles and fp64 block partials.
#include "common.h"
#include "bounds.h"
#include <stdlib.h>

// torchvision gaussian_blur(kernel_size=3, sigma=1): [a, c, a] = exp(-x^2/2)/sum, fp32
__device__ __forceinline__ void blur_taps(float &a, float &c) {
    const float e = 0.60653066f;              // exp(-0.5) in fp32
    const float s = (e + 1.0f) + e;           // pdf.sum() : e + 1 + e
    a = e / s;
    c = 1.0f / s;
}

__device__ __forceinline__ int reflect1(int i, int n) {
    // reflect padding of width 1: -1 -> 1, n -> n-2
    return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

// ------------------------------------------------------------------------------------------
// forward: raw -> blurred (+ per-block partial sums)
// grid (ceil(W/64), ceil(H/32), nimg), 256 threads
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_contrast_fwd(const float *__restrict__ raw,
                                                      float *__restrict__ blur,
                                                      double *__restrict__ part, int H, int W,
                                                      int norm_l2, int variance) {
    constexpr int TH = MPC_CT_H, TW = MPC_CT_W;
    __shared__ float s_raw[TH + 4][TW + 4 + 1];
    __shared__ float s_hb[TH + 4][TW + 2 + 1];
    __shared__ float s_bl[TH + 2][TW + 2 + 1];
    __shared__ double s_red[2][4];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const size_t img_off = (size_t)blockIdx.z * H * W;
    const float *src = raw + img_off;
    float ka, kc;
    blur_taps(ka, kc);

    // stage raw tile with a 2-px halo; coordinates -1 and H (W) are the reflect-padding ring
    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
        const int ly = i / (TW + 4), lx = i - ly * (TW + 4);
        const int y = ty0 - 2 + ly, x = tx0 - 2 + lx;
        float v = 0.f;
        if (y >= -1 && y <= H && x >= -1 && x <= W) v = src[(size_t)reflect1(y, H) * W + reflect1(x, W)];
        s_raw[ly][lx] = v;
    }
    __syncthreads();
    // horizontal pass: columns tx0-1 .. tx0+TW
    for (int i = tid; i < (TH + 4) * (TW + 2); i += 256) {
        const int ly = i / (TW + 2), lx = i - ly * (TW + 2);
        s_hb[ly][lx] = ka * s_raw[ly][lx] + kc * s_raw[ly][lx + 1] + ka * s_raw[ly][lx + 2];
    }
    __syncthreads();
    // vertical pass: rows ty0-1 .. ty0+TH ; zero outside the image (Sobel uses zero padding)
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
        const int ly = i / (TW + 2), lx = i - ly * (TW + 2);
        const int y = ty0 - 1 + ly, x = tx0 - 1 + lx;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W)
            v = ka * s_hb[ly][lx] + kc * s_hb[ly + 1][lx] + ka * s_hb[ly + 2][lx];
        s_bl[ly][lx] = v;
    }
    __syncthreads();

    double acc0 = 0.0, acc1 = 0.0;
    const int cx = tid & 63;
    for (int ry = tid >> 6; ry < TH; ry += 4) {
        const int y = ty0 + ry, x = tx0 + cx;
        if (y < H && x < W) {
            const int ly = ry + 1, lx = cx + 1;
            const float b = s_bl[ly][lx];
            blur[img_off + (size_t)y * W + x] = b;
            if (variance) {
                acc0 += (double)b;
                acc1 += (double)b * (double)b;
            } else {
                const float tl = s_bl[ly - 1][lx - 1], tc = s_bl[ly - 1][lx], tr = s_bl[ly - 1][lx + 1];
                const float ml = s_bl[ly][lx - 1], mr = s_bl[ly][lx + 1];
                const float bl_ = s_bl[ly + 1][lx - 1], bc = s_bl[ly + 1][lx], br = s_bl[ly + 1][lx + 1];
                const float dx = (tr - tl) + 2.f * (mr - ml) + (br - bl_);
                const float dy = (bl_ - tl) + 2.f * (bc - tc) + (br - tr);
                acc0 += norm_l2 ? (double)(dx * dx + dy * dy) : (double)(fabsf(dx) + fabsf(dy));
            }
        }
    }
    const double r0 = block_sum_d<256>(acc0, s_red[0]);
    const double r1 = block_sum_d<256>(acc1, s_red[1]);
    if (tid == 0) {
        const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[2 * bid] = r0;
        part[2 * bid + 1] = r1;
    }
}

// weight of a raw coordinate in a blurred one (adjoint of the reflect-padded 3-tap blur)
__device__ __forceinline__ float blurT_w(int q, int y, int n, float ka, float kc) {
    // weight of raw coordinate y in blurred coordinate q (both inside [0,n)), reflect padding
    const int d = y - q;
    float w = (d == 0) ? kc : ((d == 1 || d == -1) ? ka : 0.f);
    if (q == 0 && y == 1) w += ka;
    if (q == n - 1 && y == n - 2) w += ka;
    return w;
}

// ------------------------------------------------------------------------------------------
// fused forward + adjoint image (gradient magnitude), MARCHING form: raw -> blurred, partial sums, Blur^T Sobel^T u
// with no LDS and no barrier.  One wavefront owns 56 columns (+ 4 halo columns each side: lane = column) and a band of
// MPC_CT_H rows (+ 4 halo rows each side) and walks down the rows; every stage of the chain keeps the two or three
// previous rows it needs in registers (the row loop is fully unrolled, so the rolling windows are register renaming,
// not moves), horizontal neighbours come from the adjacent lanes (DPP wave shifts, no LDS round trip):
//   row y      : raw a(y)                      -> hb(y)   = ka a[c-1] + kc a[c] + ka a[c+1]           (reflect ring staged)
//   row y - 1  : B = ka hb(y-2) + kc hb(y-1) + ka hb(y)   (0 outside the image: Sobel pads with zeros)
//   row y - 2  : dx = hd(y-3) + 2 hd(y-2) + hd(y-1), hd = B[c+1] - B[c-1];  dy = vd[c-1] + 2 vd[c] + vd[c+1], vd = B(y-1) - B(y-3)
//                u = sign / 2x;  blurred output and objective of the own pixels
//   row y - 3  : gB = Sobel^T u  (hx = ux[c-1] - ux[c+1] per row, vy = uy(y-4) - uy(y-2) per column)
//   row y - 4  : adjoint of the blur (border weights fold the reflect ring back) -> grad image of the own pixels
// (Round 1's LDS-tiled kernel spent 190 vector instructions per pixel on LDS traffic and index arithmetic; this one ~70 per
// lane-row.  It was removed in round 4: profiles/README.md has its numbers.)
// grid (ceil(W/56), ceil(H/MPC_CT_H), nimg), 64 threads
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_left(float v) {      // value of lane - 1 (0 for lane 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));     // wave_shr:1
}
__device__ __forceinline__ float lane_right(float v) {     // value of lane + 1 (0 for lane 63)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));     // wave_shl:1
}

// TH rows per band: MPC_CT_H (32) for batches, 16 when that leaves the chip short of wavefronts (B = 1: 360 bands of 32
// rows on 256 CUs; measured at C2 18.2 -> 12.0 us, neutral at C3); `nslots` entries of the partial-sum array exist and
// the finalize kernel adds them all, so the bands of the coarser tiling clear the entries they do not use
template <bool L2N, int TH>
__global__ __launch_bounds__(64) void k_contrast_march(const float *__restrict__ raw, float *__restrict__ blur,
                                                       float *__restrict__ gimg, double *__restrict__ part, int H, int W,
                                                       int nslots) {
    constexpr int TW = MPC_CF_TW;
    const int c = threadIdx.x;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const size_t img_off = (size_t)blockIdx.z * H * W;
    const float *src = raw + img_off;
    float ka, kc;
    blur_taps(ka, kc);
    const int x = tx0 - 4 + c;
    const bool xin = x >= 0 && x < W;
    const int xr = (x >= -1 && x <= W) ? reflect1(x, W) : -1;
    const bool own_col = c >= 4 && c < 4 + TW && x < W;
    float wx0 = 0.f, wx1 = 0.f, wx2 = 0.f;                   // column weights of the blur adjoint for this lane's pixel
    if (xin) {
        wx0 = (x - 1 >= 0) ? blurT_w(x - 1, x, W, ka, kc) : 0.f;
        wx1 = blurT_w(x, x, W, ka, kc);
        wx2 = (x + 1 < W) ? blurT_w(x + 1, x, W, ka, kc) : 0.f;
    }
    float hb1 = 0.f, hb2 = 0.f;                               // hb(y-1), hb(y-2)
    float B2 = 0.f, B3 = 0.f, hd2 = 0.f, hd3 = 0.f;           // B and hd at rows y-2, y-3
    float hx3 = 0.f, hx4 = 0.f, uy3 = 0.f, uy4 = 0.f;         // hx and uy at rows y-3, y-4
    float hT4 = 0.f, hT5 = 0.f;                               // horizontal part of the blur adjoint at rows y-4, y-5
    double acc = 0.0;
    // the raw rows are requested CM_PF iterations ahead
    float a_next[CM_PF];
#pragma unroll
    for (int k = 0; k < CM_PF; ++k) {
        const int y = ty0 - 4 + k;
        a_next[k] = (xr >= 0 && y >= -1 && y <= H) ? src[(size_t)reflect1(y, H) * W + xr] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < TH + 8; ++it) {
        const int y = ty0 - 4 + it;
        const float a = a_next[it % CM_PF];
        {
            const int yn = y + CM_PF;
            a_next[it % CM_PF] = (it + CM_PF < TH + 8 && xr >= 0 && yn >= -1 && yn <= H) ? src[(size_t)reflect1(yn, H) * W + xr] : 0.f;
        }
        // row y: horizontal blur
        const float hb0 = ka * lane_left(a) + kc * a + ka * lane_right(a);
        // row y - 1: vertical blur; zero outside the image
        const int yb = y - 1;
        const float B1 = (xin && yb >= 0 && yb < H) ? ka * hb2 + kc * hb1 + ka * hb0 : 0.f;
        const float hd1 = lane_right(B1) - lane_left(B1);
        // row y - 2: Sobel, u, blurred output, objective
        const int ys = y - 2;
        float ux = 0.f, uy2 = 0.f;
        {
            const float vd = B1 - B3;
            const float dxv = hd3 + 2.f * hd2 + hd1;
            const float dyv = lane_left(vd) + 2.f * vd + lane_right(vd);
            if (xin && ys >= 0 && ys < H) {
                if (L2N) { ux = 2.f * dxv; uy2 = 2.f * dyv; }
                else {
                    ux = (dxv > 0.f) ? 1.f : ((dxv < 0.f) ? -1.f : 0.f);
                    uy2 = (dyv > 0.f) ? 1.f : ((dyv < 0.f) ? -1.f : 0.f);
                }
                if (own_col && ys >= ty0 && ys < ty0 + TH) {
                    blur[img_off + (size_t)ys * W + x] = B2;
                    acc += L2N ? (double)(dxv * dxv + dyv * dyv) : (double)(fabsf(dxv) + fabsf(dyv));
                }
            }
        }
        const float hx2 = lane_left(ux) - lane_right(ux);
        // row y - 3: gB = Sobel^T u; zero outside the image
        const int yg = y - 3;
        float g = 0.f;
        {
            const float vy = uy4 - uy2;
            const float gx = hx2 + 2.f * hx3 + hx4;
            const float gy = lane_right(vy) + 2.f * vy + lane_left(vy);
            if (xin && yg >= 0 && yg < H) g = gx + gy;
        }
        const float hT3 = wx0 * lane_left(g) + wx1 * g + wx2 * lane_right(g);
        // row y - 4: adjoint of the vertical blur (reflect ring folded into the border weights)
        const int yo = y - 4;
        if (own_col && yo >= ty0 && yo < ty0 + TH && yo < H) {
            float o = 0.f;
            if (yo - 1 >= 0) o += blurT_w(yo - 1, yo, H, ka, kc) * hT5;
            o += blurT_w(yo, yo, H, ka, kc) * hT4;
            if (yo + 1 < H) o += blurT_w(yo + 1, yo, H, ka, kc) * hT3;
            gimg[img_off + (size_t)yo * W + x] = o;
        }
        // roll the windows
        hb2 = hb1; hb1 = hb0;
        B3 = B2; B2 = B1; hd3 = hd2; hd2 = hd1;
        hx4 = hx3; hx3 = hx2; uy4 = uy3; uy3 = uy2;
        hT5 = hT4; hT4 = hT3;
    }
    acc = wave_sum_d(acc);
    if (c == 0) {
        const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const size_t nb_ = (size_t)gridDim.x * gridDim.y * gridDim.z;
        part[2 * MPC_IDX(bid, nslots)] = acc;
        part[2 * bid + 1] = 0.0;
        for (size_t e = bid + nb_; e < (size_t)nslots; e += nb_) { part[2 * e] = 0.0; part[2 * e + 1] = 0.0; }
    }
}

// ------------------------------------------------------------------------------------------
// backward (variance objective): blurred -> Blur^T (x - mean_img)   (unscaled)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_image_means(const double *__restrict__ part,
                                                     float *__restrict__ means, int tiles_per_img,
                                                     int HW) {
    __shared__ double s_red[4];
    const int img = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < tiles_per_img; i += 256) a += part[2 * ((size_t)img * tiles_per_img + i)];
    const double r = block_sum_d<256>(a, s_red);
    if (threadIdx.x == 0) means[img] = (float)(r / (double)HW);
}

__global__ __launch_bounds__(256) void k_contrast_bwd_var(const float *__restrict__ blur,
                                                          const float *__restrict__ means,
                                                          float *__restrict__ gimg, int H, int W) {
    constexpr int TH = MPC_CT_H, TW = MPC_CT_W;
    __shared__ float s_gb[TH + 2][TW + 2 + 1];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const size_t img_off = (size_t)blockIdx.z * H * W;
    const float mean = means[blockIdx.z];
    float ka, kc;
    blur_taps(ka, kc);
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
        const int ly = i / (TW + 2), lx = i - ly * (TW + 2);
        const int y = ty0 - 1 + ly, x = tx0 - 1 + lx;
        s_gb[ly][lx] = (y >= 0 && y < H && x >= 0 && x < W) ? blur[img_off + (size_t)y * W + x] - mean : 0.f;
    }
    __syncthreads();
    const int cx = tid & 63;
    for (int ry = tid >> 6; ry < TH; ry += 4) {
        const int y = ty0 + ry, x = tx0 + cx;
        if (y < H && x < W) {
            float acc = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int qy = y + dy;
                if (qy < 0 || qy >= H) continue;
                const float wy = blurT_w(qy, y, H, ka, kc);
                float row = 0.f;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int qx = x + dx;
                    if (qx < 0 || qx >= W) continue;
                    row += blurT_w(qx, x, W, ka, kc) * s_gb[ry + 1 + dy][cx + 1 + dx];
                }
                acc += wy * row;
            }
            gimg[img_off + (size_t)y * W + x] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// finalize: fp64 reduction of the partials -> device scalars
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_finalize(const double *__restrict__ cpart, int n_cblocks,
                                                  int tiles_per_img, int nimg, int HW,
                                                  const double *__restrict__ spart, int n_sblocks,
                                                  double smooth_count, float smooth_weight,
                                                  int variance, float *__restrict__ scal, float *__restrict__ scal_out) {
    __shared__ double s_red[2][16];
    __shared__ double s_var;
    const int tid = threadIdx.x;
    double val = 0.0, gcoef = 0.0;
    // (the thread's share of the smoothness partials is read here, with the contrast partials, not after their reduction:
    // one memory round trip of this single-workgroup kernel instead of two)
    // (eight partials of a thread requested together, then added in index order -- the sums keep their association; one
    // load per loop trip made this kernel a chain of memory round trips: 9 us at C3 for 20k doubles)
    double sm_a = 0.0, sm_b = 0.0;
    for (int i0 = tid; i0 < n_sblocks; i0 += 8 * 1024) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            v[u] = (i < n_sblocks) ? reinterpret_cast<const double2 *>(spart)[i] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < n_sblocks) { sm_a += v[u].x; sm_b += v[u].y; }
    }
    if (!variance) {
        double a = 0.0;
        for (int i0 = tid; i0 < n_cblocks; i0 += 8 * 1024) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024; v[u] = (i < n_cblocks) ? cpart[2 * (size_t)i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 1024 < n_cblocks) a += v[u];
        }
        const double tot = block_sum_d<1024>(a, s_red[0]);
        const double N = (double)nimg * (double)HW;
        val = tot / N;
        gcoef = -1.0 / (val * val) / N;
    } else {
        // mean over images of the unbiased variance over H*W (loss.py:14-16)
        if (tid == 0) s_var = 0.0;
        __syncthreads();
        for (int img = 0; img < nimg; ++img) {
            double a = 0.0, b = 0.0;
            for (int i = tid; i < tiles_per_img; i += 1024) {
                a += cpart[2 * ((size_t)img * tiles_per_img + i)];
                b += cpart[2 * ((size_t)img * tiles_per_img + i) + 1];
            }
            const double s1 = block_sum_d<1024>(a, s_red[0]);
            const double s2 = block_sum_d<1024>(b, s_red[1]);
            if (tid == 0) s_var += (s2 - s1 * s1 / (double)HW) / (double)(HW - 1);
        }
        __syncthreads();
        val = s_var / (double)nimg;
        gcoef = -1.0 / (val * val) * 2.0 / ((double)(HW - 1) * (double)nimg);
    }
    double smooth = 0.0;
    if (n_sblocks > 0) {
        const double sx = block_sum_d<1024>(sm_a, s_red[0]);
        const double sy = block_sum_d<1024>(sm_b, s_red[1]);
        smooth = (double)smooth_weight * ((sx / smooth_count + sy / smooth_count) / 2.0);
    }
    if (tid == 0) {
        const double focus = 1.0 / val;
        scal[MPC_SCAL_LOSS] = (float)(focus + smooth);
        scal[MPC_SCAL_FOCUS] = (float)focus;
        scal[MPC_SCAL_SMOOTH] = (float)smooth;
        scal[MPC_SCAL_VAL] = (float)val;
        scal[MPC_SCAL_GCOEF] = (float)gcoef;
        scal[5] = scal[6] = scal[7] = 0.f;
        if (scal_out != nullptr) { scal_out[0] = (float)(focus + smooth); scal_out[1] = (float)focus; scal_out[2] = (float)smooth; }
    }
}

__global__ void k_scale(const float *__restrict__ x, const float *__restrict__ a,
                        float *__restrict__ y, int64_t n) {
    const float s = a[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = s * x[i];
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// smoothness, MARCHING form (as k_contrast_march: lane = column, rows walked with rolling register windows, DPP lane
// shifts, no LDS, no barrier): one wavefront owns 60 cell columns (+ 2 halo each side) and MPC_SM_H rows (+ 2) of one
// channel pair.  Same sums, in the same association, as round 1's LDS-tiled kernel had (removed in round 4).
// grid (ceil(wq/60), ceil(hq/MPC_SM_H), nimg*C/2), 64 threads
// ------------------------------------------------------------------------------------------
// TH rows per band: MPC_SM_H (16), or half of it for small fields (smooth_band_rows below)
template <int TH>
__global__ __launch_bounds__(64) void k_lut_smooth_march(const float *__restrict__ field, float *__restrict__ gfield,
                                                         double *__restrict__ part, int hq, int wq, int C, float gscale) {
    constexpr int TW = MPC_SM_W;
    const int c = threadIdx.x;
    const int C2 = C >> 1;
    const int img = blockIdx.z / C2, cp = blockIdx.z - img * C2;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t base = (size_t)img * hq * wq * C2 + cp;        // in float2 units
    const float2 *f2 = reinterpret_cast<const float2 *>(field);
    float2 *g2 = reinterpret_cast<float2 *>(gfield);
    const float eps2 = 1e-3f * 1e-3f;   // charbonnier epsilon ** 2 (loss.py:46,55)
    const int x = x0 - 2 + c;
    const bool xin = x >= 0 && x < wq;
    const bool own_col = c >= 2 && c < 2 + TW && x < wq;
    float2 f1 = make_float2(0.f, 0.f), f2r = f1, hd1 = f1, hd2 = f1;          // f and hd at rows y-1, y-2
    float2 hx2 = f1, hx3 = f1, vy2 = f1, vy3 = f1;                              // hx and vy at rows y-2, y-3
    double a0 = 0.0, a1 = 0.0;
    // rows requested MPC_SM_PF iterations ahead of their use (rolling window in registers: the loop is fully unrolled)
    constexpr int PF = MPC_SM_PF;
    float2 f_next[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int y = y0 - 2 + k;
        f_next[k] = (k < TH + 4 && xin && y >= 0 && y < hq) ? f2[base + ((size_t)y * wq + x) * C2] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < TH + 4; ++it) {
        const int y = y0 - 2 + it;
        const float2 f0 = f_next[it % PF];
        {
            const int yn = y + PF;
            f_next[it % PF] = (it + PF < TH + 4 && xin && yn >= 0 && yn < hq) ? f2[base + ((size_t)yn * wq + x) * C2] : make_float2(0.f, 0.f);
        }
        const float2 hd0 = make_float2(lane_right(f0.x) - lane_left(f0.x), lane_right(f0.y) - lane_left(f0.y));
        // row y - 1: Sobel -> charbonnier terms and their derivatives
        const int ys = y - 1;
        float2 vx = make_float2(0.f, 0.f), vy1 = make_float2(0.f, 0.f);
        {
            const float2 vd = make_float2(f0.x - f2r.x, f0.y - f2r.y);
            const float dxx = hd2.x + 2.f * hd1.x + hd0.x, dxy = hd2.y + 2.f * hd1.y + hd0.y;
            const float dyx = lane_left(vd.x) + 2.f * vd.x + lane_right(vd.x), dyy = lane_left(vd.y) + 2.f * vd.y + lane_right(vd.y);
            if (xin && ys >= 0 && ys < hq) {
                // hardware sqrt and reciprocal (1 ulp each)
                const float sxx = __builtin_amdgcn_sqrtf(dxx * dxx + eps2), syx = __builtin_amdgcn_sqrtf(dyx * dyx + eps2);
                const float sxy = __builtin_amdgcn_sqrtf(dxy * dxy + eps2), syy = __builtin_amdgcn_sqrtf(dyy * dyy + eps2);
                vx.x = dxx * __builtin_amdgcn_rcpf(sxx); vy1.x = dyx * __builtin_amdgcn_rcpf(syx);
                vx.y = dxy * __builtin_amdgcn_rcpf(sxy); vy1.y = dyy * __builtin_amdgcn_rcpf(syy);
                if (own_col && ys >= y0 && ys < y0 + TH) {
                    a0 += (double)sxx; a1 += (double)syx;
                    a0 += (double)sxy; a1 += (double)syy;
                }
            }
        }
        const float2 hx1 = make_float2(lane_left(vx.x) - lane_right(vx.x), lane_left(vx.y) - lane_right(vx.y));
        // row y - 2: adjoint of the Sobel pair
        const int yo = y - 2;
        {
            const float2 vyd = make_float2(vy3.x - vy1.x, vy3.y - vy1.y);
            const float gyx = lane_right(vyd.x) + 2.f * vyd.x + lane_left(vyd.x), gyy = lane_right(vyd.y) + 2.f * vyd.y + lane_left(vyd.y);
            if (gfield != nullptr && own_col && yo >= y0 && yo < y0 + TH && yo < hq) {
                float2 o;
                o.x = gscale * ((hx1.x + 2.f * hx2.x + hx3.x) + gyx);
                o.y = gscale * ((hx1.y + 2.f * hx2.y + hx3.y) + gyy);
                g2[base + ((size_t)yo * wq + x) * C2] = o;
            }
        }
        f2r = f1; f1 = f0; hd2 = hd1; hd1 = hd0;
        hx3 = hx2; hx2 = hx1; vy3 = vy2; vy2 = vy1;
    }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1);
    if (c == 0) {
        const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[2 * bid] = a0;
        part[2 * bid + 1] = a1;
    }
}

extern "C" int mpc_contrast_fwd(const mpc_shape *s, const float *iwe_raw, float *iwe_blur,
                                float *grad_iwe, void *ws, void *stream) {
    MPC_CHECK_ARG(s && iwe_raw && iwe_blur && ws, MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    hipStream_t st = (hipStream_t)stream;
    double *cpart = (double *)((char *)ws + L.off_cpart);
    const dim3 grid(mpc_cdiv(s->W, MPC_CT_W), mpc_cdiv(s->H, MPC_CT_H), L.nimg);
    const int variance = (s->flags & MPC_F_OBJ_VARIANCE) ? 1 : 0;
    const int l2 = (s->flags & MPC_F_NORM_L2) ? 1 : 0;
    if (grad_iwe && !variance) {
        const dim3 gridf(mpc_cdiv(s->W, MPC_CF_TW), mpc_cdiv(s->H, MPC_CT_H), L.nimg);
        if ((int64_t)gridf.x * gridf.y * gridf.z < 1536) {
            // few images: bands of 16 rows, twice the wavefronts (the partial-sum array is sized for them)
            const dim3 gridh(gridf.x, mpc_cdiv(s->H, MPC_CT_H / 2), L.nimg);
            if (l2) MPC_LAUNCH((k_contrast_march<true, MPC_CT_H / 2>), gridh, dim3(64), 0, st, iwe_raw, iwe_blur, grad_iwe, cpart, s->H, s->W, L.n_cblocks);
            else MPC_LAUNCH((k_contrast_march<false, MPC_CT_H / 2>), gridh, dim3(64), 0, st, iwe_raw, iwe_blur, grad_iwe, cpart, s->H, s->W, L.n_cblocks);
        } else if (l2) MPC_LAUNCH((k_contrast_march<true, MPC_CT_H>), gridf, dim3(64), 0, st, iwe_raw, iwe_blur, grad_iwe, cpart, s->H, s->W, L.n_cblocks);
        else MPC_LAUNCH((k_contrast_march<false, MPC_CT_H>), gridf, dim3(64), 0, st, iwe_raw, iwe_blur, grad_iwe, cpart, s->H, s->W, L.n_cblocks);
        MPC_CHECK_LAUNCH();
        return 0;
    }
    {   // the unfused tiling has fewer workgroups than the partial-sum array: clear the rest
        const int e = mpc_zero_async(cpart, (size_t)L.n_cblocks * 2 * sizeof(double), st);
        if (e) return e;
    }
    MPC_LAUNCH(k_contrast_fwd, grid, dim3(256), 0, st, iwe_raw, iwe_blur, cpart, s->H, s->W, l2, variance);
    MPC_CHECK_LAUNCH();
    if (grad_iwe) {      // variance objective: the adjoint needs the image means first
        {
            float *means = (float *)((char *)ws + L.off_counts) + 8;   // nimg floats after the counters
            MPC_LAUNCH(k_image_means, dim3(L.nimg), dim3(256), 0, st, cpart, means,
                               (int)(grid.x * grid.y), s->H * s->W);
            MPC_LAUNCH(k_contrast_bwd_var, grid, dim3(256), 0, st, iwe_blur, means, grad_iwe, s->H, s->W);
        }
        MPC_CHECK_LAUNCH();
    }
    return 0;
}

// rows per band of the marching smoothness kernel -- mpc_lut_smooth and mpc_finalize (which adds the partial sums of
// the bands) must agree on it: bands of 8 rows when 16 would leave the chip short of wavefronts (B = 1: 360 bands)
static int smooth_band_rows(const mpc_shape *s, int nimg, int C) {
    const int64_t bands = (int64_t)mpc_cdiv(s->wq, MPC_SM_W) * mpc_cdiv(s->hq, MPC_SM_H) * nimg * (C / 2);
    return bands < 1536 ? MPC_SM_H / 2 : MPC_SM_H;
}

extern "C" int mpc_lut_smooth(const mpc_shape *s, const float *field, int32_t nimg, int32_t C,
                              float smooth_weight, float *grad_field, void *ws, void *stream) {
    MPC_CHECK_ARG(s && field && ws, MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    MPC_CHECK_ARG(nimg > 0 && C > 0 && (C % 2) == 0, MPC_E_SHAPE, "field must have an even number of channels");
    const int band = smooth_band_rows(s, nimg, C);
    const dim3 grid(mpc_cdiv(s->wq, MPC_SM_W), mpc_cdiv(s->hq, band), nimg * (C / 2));
    const int64_t nblk = (int64_t)grid.x * grid.y * grid.z;
    MPC_CHECK_ARG(nblk <= L.n_sblocks_max, MPC_E_SHAPE, "field larger than the LUT of this shape");
    hipStream_t st = (hipStream_t)stream;
    double *spart = (double *)((char *)ws + L.off_spart);
    const double count = (double)nimg * C * s->hq * s->wq;
    const float gscale = (float)((double)smooth_weight / (2.0 * count));
    if (band == MPC_SM_H) MPC_LAUNCH(k_lut_smooth_march<MPC_SM_H>, grid, dim3(64), 0, st, field, grad_field, spart, s->hq, s->wq, C, gscale);
    else MPC_LAUNCH(k_lut_smooth_march<MPC_SM_H / 2>, grid, dim3(64), 0, st, field, grad_field, spart, s->hq, s->wq, C, gscale);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_finalize(const mpc_shape *s, int32_t smooth_nimg, int32_t smooth_C,
                            float smooth_weight, float *scal, void *ws, void *stream) {
    return mpc_finalize_ex(s, smooth_nimg, smooth_C, smooth_weight, scal, nullptr, ws, stream);
}

// scal_out: loss, focus, smooth once more (mpc_focus_fwd: the copy the caller hands out), or null
int mpc_finalize_ex(const mpc_shape *s, int32_t smooth_nimg, int32_t smooth_C, float smooth_weight, float *scal, float *scal_out,
                    void *ws, void *stream) {
    MPC_CHECK_ARG(s && scal && ws, MPC_E_NULL, "null argument");
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    const int tiles = mpc_cdiv(s->W, MPC_CT_W) * mpc_cdiv(s->H, MPC_CT_H);
    int64_t nsblk = 0;
    double count = 1.0;
    if (smooth_nimg > 0) {
        const int band = smooth_band_rows(s, smooth_nimg, smooth_C);
        nsblk = (int64_t)mpc_cdiv(s->wq, MPC_SM_W) * mpc_cdiv(s->hq, band) * smooth_nimg * (smooth_C / 2);
        MPC_CHECK_ARG(nsblk <= L.n_sblocks_max, MPC_E_SHAPE, "field larger than the LUT of this shape");
        count = (double)smooth_nimg * smooth_C * s->hq * s->wq;
    }
    MPC_LAUNCH(k_finalize, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const double *)((char *)ws + L.off_cpart), L.n_cblocks, tiles, L.nimg,
                       s->H * s->W, (const double *)((char *)ws + L.off_spart), (int)nsblk, count,
                       smooth_weight, (s->flags & MPC_F_OBJ_VARIANCE) ? 1 : 0, scal, scal_out);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_scale(const float *x, const float *a, float *y, int64_t count, void *stream) {
    MPC_CHECK_ARG(x && a && y, MPC_E_NULL, "null argument");
    if (count <= 0) return 0;
    const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    MPC_LAUNCH(k_scale, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, a, y, count);
    MPC_CHECK_LAUNCH();
    return 0;
}


// ---- UNPINNED EXTENSION (default off): IWE pyramid (BASELINE.json configs[2] names one; the reference has none, SURVEY.md
// Appendix C).  Level l + 1 = 2x2 average of level l; the objective of every level is the reference's
// calculate_focus_loss on that level (src/utils/loss.py:4-27), the focus term their sum.  Two helpers: the pooling and its
// adjoint, which carries the coarser level's adjoint image (in units of ITS 1 / val^2 coefficient) into the finer one's.
__global__ __launch_bounds__(256) void k_pool2_fwd(const float *__restrict__ in, float *__restrict__ out, int nimg, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1;
    const size_t n = (size_t)nimg * H2 * W2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W2), y = (int)((i / W2) % H2);
        const size_t img = i / ((size_t)W2 * H2);
        const float *p = in + (img * H + 2 * y) * W + 2 * x;
        out[i] = 0.25f * ((p[0] + p[1]) + (p[W] + p[W + 1]));
    }
}
// big[y][x] += (coef_small / coef_big) * 0.25 * small[y / 2][x / 2]     (coefficients: device scalars)
__global__ __launch_bounds__(256) void k_pool2_bwd_add(const float *__restrict__ small, const float *__restrict__ coef_small,
                                                       float *__restrict__ big, const float *__restrict__ coef_big, int nimg, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1;
    // (a level whose objective is flat -- an empty or constant image -- has coefficient 0 or inf: it passes nothing on, and
    // a finer level without a coefficient of its own receives nothing: no inf / NaN into the gradient)
    const float cs_ = coef_small[0], cb_ = coef_big[0];
    const float ratio = cs_ / cb_;
    const float r = (cb_ != 0.f && ratio == ratio && fabsf(ratio) < INFINITY) ? 0.25f * ratio : 0.f;
    const size_t n = (size_t)nimg * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const size_t img = i / ((size_t)W * H);
        if ((y >> 1) < H2 && (x >> 1) < W2) big[i] += r * small[(img * H2 + (y >> 1)) * W2 + (x >> 1)];
    }
}

extern "C" int mpc_pool2_fwd(const float *in, float *out, int32_t nimg, int32_t H, int32_t W, void *stream) {
    MPC_CHECK_ARG(in && out, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(nimg >= 0 && H >= 2 && W >= 2, MPC_E_SHAPE, "image smaller than a pooling window");
    const int64_t n = (int64_t)nimg * (H / 2) * (W / 2);
    if (n == 0) return 0;
    MPC_LAUNCH(k_pool2_fwd, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, in, out, nimg, H, W);
    MPC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mpc_pool2_bwd_add(const float *small, const float *coef_small, float *big, const float *coef_big, int32_t nimg,
                                 int32_t H, int32_t W, void *stream) {
    MPC_CHECK_ARG(small && coef_small && big && coef_big, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(nimg >= 0 && H >= 2 && W >= 2, MPC_E_SHAPE, "image smaller than a pooling window");
    const int64_t n = (int64_t)nimg * H * W;
    if (n == 0) return 0;
    MPC_LAUNCH(k_pool2_bwd_add, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream,
               small, coef_small, big, coef_big, nimg, H, W);
    MPC_CHECK_LAUNCH();
    return 0;
}

MPC_BOUNDS_UNIT("contrast.hip")
