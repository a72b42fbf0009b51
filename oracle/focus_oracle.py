"""CPU ORACLE for the contrast-maximisation (CMax) loss hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32, autograd) restatement of the reference algorithm in
  /root/reference/src/losses/focus.py                      (FocusLoss)
  /root/reference/src/utils/event_image_converter.py       (bilinear vote + 3x3 blur)
  /root/reference/src/utils/loss.py                        (contrast objectives, smoothness)
  /root/reference/src/utils/{basis,trajectories}.py        (trajectory construction)
  /root/reference/src/models/raft_spline/curves/bezier.py  (Bernstein basis)
Each function cites the reference lines it follows.  It is the checker for the HIP path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The
product package (motionpriorcmax_amd) never imports anything from oracle/.

Pinning: the reference ships no tests or golden vectors.  This oracle is
pinned against outputs of the UNMODIFIED reference imported in the build container
(oracle/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).  Two third-party
operations inside the path are absent from /root/reference and from this image and are
restated from their published behaviour: pykeops==2.2.2 LazyTensor.argKmin/Kmin (exact
brute-force K-min; tie order unpinned, we use lowest index first) and
torchvision.transforms.functional.gaussian_blur (kernel_size=3, sigma=1, reflect padding).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IWD_EPS = 1e-9  # focus.py:7


# ----------------------------------------------------------------------------------------------
# trajectory construction (caller side of the path)
# ----------------------------------------------------------------------------------------------
def tile_mask(image_shape, tile_size: int) -> torch.Tensor:
    """One trajectory per tile, at offset tile//2.  trajectories.py:3-13."""
    m = torch.zeros(tuple(image_shape), dtype=torch.bool)
    s = tile_size // 2
    m[s::tile_size, s::tile_size] = True
    return m


def coeff_grid_to_list(coeff_grid: torch.Tensor, mask: torch.Tensor, num_coeffs: int):
    """[b,s,2k,h,w] -> coeffs [b,s,2,n,k] (first k channels are y), pixel positions [n,2] (y,x).
    trajectories.py:15-52."""
    b, s, c2, h, w = coeff_grid.shape
    assert c2 == 2 * num_coeffs
    pos = torch.nonzero(mask)
    flat = coeff_grid.reshape(b, s, c2, h * w)[..., mask.reshape(-1)]
    coeffs = flat.reshape(b, s, 2, num_coeffs, -1).permute(0, 1, 2, 4, 3)
    return coeffs, pos


def basis_matrix(times: torch.Tensor, num_basis: int, basis_type: str) -> torch.Tensor:
    """[n_t] -> [n_t,k].  basis.py:18-31 (dct / polynomial)."""
    k_idx = torch.arange(1, num_basis + 1, device=times.device)
    if basis_type == 'dct':
        a = (2 * times[:, None] + 1) * k_idx[None, :]
        return np.sqrt(2.0) * torch.cos(np.pi / 2.0 * a)
    if basis_type == 'polynomial':
        return times[:, None] ** k_idx[None, :]
    raise ValueError(basis_type)


def eval_basis(coeffs: torch.Tensor, times: torch.Tensor, num_basis: int, basis_type: str):
    """coeffs [b,s,2,n,k], times [n_t] -> [b,n,n_t,2]; sums over scales s.  basis.py:33-46."""
    bm = basis_matrix(times, num_basis, basis_type)               # [n_t,k]
    cy = coeffs[:, :, 0]                                          # [b,s,n,k]
    cx = coeffs[:, :, 1]
    ty = (bm[None, None, None] * cy[..., None, :]).sum(-1)        # [b,s,n,n_t]
    tx = (bm[None, None, None] * cx[..., None, :]).sum(-1)
    return torch.stack([ty, tx], dim=-1).sum(1)


def trajectories_at(coeff_grid, times, mask, num_basis, basis_type, add_offsets=True,
                    anchor_time=0.0):
    """TrajectoryNet.calculate_trajectories_at_t / calculate_coords (trajectory_net.py:101-119).
    Returns [b, n_t, n, 2] (y,x)."""
    if coeff_grid.dim() == 4:
        coeff_grid = coeff_grid[:, None]
    coeffs, pos = coeff_grid_to_list(coeff_grid, mask, num_basis)
    anchor = torch.full((1,), anchor_time, dtype=coeffs.dtype, device=coeffs.device)
    traj = eval_basis(coeffs, times, num_basis, basis_type) - \
        eval_basis(coeffs, anchor, num_basis, basis_type)
    if add_offsets:
        traj = traj + pos[None, :, None, :]
    return traj.permute(0, 2, 1, 3).contiguous()


def bernstein_matrix(times, degree: int) -> torch.Tensor:
    """[n_t] -> [n_t,degree] : C(d,i)(1-t)^(d-i) t^i, i=1..d, computed in float64 then cast to
    fp32.  bezier.py:69-107 (P0 == 0)."""
    t = np.asarray(times, dtype=np.float64)
    out = np.zeros((t.size, degree))
    for ti in range(t.size):
        for di in range(degree):
            i = di + 1
            out[ti, di] = math.comb(degree, i) * (1 - t[ti]) ** (degree - i) * t[ti] ** i
    return torch.from_numpy(out).float()


def bezier_flow(params: torch.Tensor, times, degree: int) -> torch.Tensor:
    """params [b, 2*degree, h, w] viewed [b,2,degree,h,w] with dim-1 = (x,y) -> flow
    [n_t,b,2,h,w].  bezier.py:92-113."""
    b, _, h, w = params.shape
    p = params.view(b, 2, degree, h, w)
    return torch.einsum('bdphw,tp->tbdhw', p, bernstein_matrix(times, degree).to(params.device))


# ----------------------------------------------------------------------------------------------
# A2  reconstruction times
# ----------------------------------------------------------------------------------------------
def bin_mid_times(num_bins: int) -> torch.Tensor:
    """focus.py:61-62."""
    e = torch.linspace(0, 1, num_bins + 1)
    return (e[:-1] + e[1:]) / 2


# ----------------------------------------------------------------------------------------------
# A5  KNN flow look-up table
# ----------------------------------------------------------------------------------------------
def lut_grid_points(image_shape, sp: int) -> Tuple[torch.Tensor, int, int]:
    """Cell centres arange(0,H,sp)+sp/2-0.5.  focus.py:116-126."""
    h, w = image_shape
    off = float(sp) / 2 - 0.5
    y = torch.arange(0, h, sp, dtype=torch.float32) + off
    x = torch.arange(0, w, sp, dtype=torch.float32) + off
    gy, gx = torch.meshgrid(y, x, indexing='ij')
    return torch.stack((gy, gx), -1).reshape(-1, 2), len(y), len(x)


def knn_indices(points: torch.Tensor, queries: torch.Tensor, k: int, dist_norm: str,
                q_chunk: int = 2048, kmin: bool = False):
    """Exact K nearest of `points` [n,2] for each of `queries` [Q,2]; distance formula and
    operand order as focus.py:132-135 ((grid - traj)**2 summed y then x, or abs).  Ties: lowest
    point index first (stable sort).  `kmin=True` selects with torch.topk instead -- a K-min scan like KeOps'
    argKmin, the fair thing to TIME (bench.py's cpu_baseline); its tie order is not defined, so the checker
    never uses it.  Returns (idx [Q,k] int64, dist [Q,k])."""
    pts = points.detach()
    idx_out, d_out = [], []
    for s in range(0, queries.shape[0], q_chunk):
        q = queries[s:s + q_chunk]
        diff = q[:, None, :] - pts[None, :, :]                     # [q,n,2]
        if dist_norm == 'l2':
            d = (diff ** 2).sum(-1)
        elif dist_norm == 'l1':
            d = diff.abs().sum(-1)
        else:
            raise ValueError(dist_norm)
        if kmin:
            dv, di = torch.topk(d, k, dim=1, largest=False, sorted=True)
        else:
            dv, di = torch.sort(d, dim=1, stable=True)
        idx_out.append(di[:, :k])
        d_out.append(dv[:, :k])
    return torch.cat(idx_out), torch.cat(d_out)


def interpolate_flow(traj_tref, traj_tmid, image_shape, sp, num_knn, dist_norm='l2',
                     scheme='mean', want_next=False, return_idx=False):
    """traj_tref [b,T,n,2], traj_tmid [b,nb,n,2] -> flow_lut [b,nb,hq,wq,T,2]
    (+ flow_to_next [b,nb-1,hq,wq,1,2]).  focus.py:115-180."""
    grid, hq, wq = lut_grid_points(image_shape, sp)
    grid = grid.to(traj_tmid.device)
    b, nb, n, _ = traj_tmid.shape
    T = traj_tref.shape[1]
    lut = []
    nxt = []
    idx_all = []
    for ib in range(b):
        lut_b, nxt_b, idx_b = [], [], []
        for it in range(nb):
            idx, dk = knn_indices(traj_tmid[ib, it], grid, num_knn, dist_norm)
            # flow_to_tref[n, T, 2] = traj(t_ref) - traj(t_mid)        focus.py:140-141
            f = traj_tref[ib].permute(1, 0, 2) - traj_tmid[ib, it][:, None, :]
            g = f[idx]                                              # [Q,K,T,2]
            if num_knn == 1 or scheme == 'mean':
                val = g.mean(1)
            elif scheme == 'iwd':
                wgt = 1 / (dk + IWD_EPS)                            # focus.py:158-162 (no grad)
                wgt = (wgt / wgt.sum(1, keepdim=True)).detach()
                val = (wgt[..., None, None] * g).sum(1)
            else:
                raise ValueError(scheme)
            lut_b.append(val)
            idx_b.append(idx)
            if want_next and it < nb - 1:
                fn = traj_tmid[ib, it + 1] - traj_tmid[ib, it]       # focus.py:171
                nxt_b.append(fn[idx].mean(1))                       # always the mean (:175)
        lut.append(torch.stack(lut_b))
        idx_all.append(torch.stack(idx_b))
        if want_next:
            nxt.append(torch.stack(nxt_b))
    flow_lut = torch.stack(lut).reshape(b, nb, hq, wq, T, 2)
    flow_next = torch.stack(nxt).reshape(b, nb - 1, hq, wq, 1, 2) if want_next else None
    if return_idx:
        return flow_lut, flow_next, torch.stack(idx_all)
    return flow_lut, flow_next


# ----------------------------------------------------------------------------------------------
# A6  warp, A7 weights, A8 bilinear vote + blur
# ----------------------------------------------------------------------------------------------
def warp_events(events: torch.Tensor, flow_lut: torch.Tensor, sp: int) -> torch.Tensor:
    """events [b,m,6], flow_lut [b,nb,hq,wq,T,2] -> warped (y,x) [b,T,m,2].  focus.py:182-191
    (lut + event, single fp32 add)."""
    b, m, _ = events.shape
    ib = torch.arange(b)[:, None].expand(b, m)
    it = events[..., 4].to(torch.int64)
    iy = torch.div(events[..., 0], sp, rounding_mode='floor').to(torch.int64)
    ix = torch.div(events[..., 1], sp, rounding_mode='floor').to(torch.int64)
    d = flow_lut[ib, it, iy, ix]                                    # [b,m,T,2]
    return d.permute(0, 2, 1, 3) + events[:, None, :, :2]


def event_weights(events, warped, t_ref, image_shape, scale_by_dt, mask_border):
    """[b,T,m].  focus.py:201-214 (no gradient; strict > on the upper border)."""
    with torch.no_grad():
        w = events[:, None, :, 5].expand(warped.shape[:3])
        if scale_by_dt:
            dt = torch.clamp(torch.abs(events[:, None, :, 2] - t_ref.reshape(1, -1, 1)), 0, 1)
            w = (1 - dt) * w
        if mask_border:
            y, x = warped[..., 0], warped[..., 1]
            oob = (y > image_shape[0]) | (x > image_shape[1]) | (y < 0) | (x < 0)
            w = torch.where(oob, torch.zeros_like(w), w)
    return w


def bilinear_vote(pos: torch.Tensor, weight, image_shape) -> torch.Tensor:
    """pos [nimg,m,2] (y,x), weight [nimg,m] or scalar -> [nimg,H,W].
    event_image_converter.py:333-391 (floor(pos+1e-6); masked taps add 0 to pixel 0)."""
    h, w = image_shape
    nimg = pos.shape[0]
    fl = torch.floor(pos + 1e-6)
    fr = pos - fl
    fl = fl.long()
    y0, x0 = fl[..., 0], fl[..., 1]
    fy, fx = fr[..., 0], fr[..., 1]
    img = pos.new_zeros((nimg, h * w))
    taps = ((y0, x0, (1 - fy) * (1 - fx) * weight),
            (y0 + 1, x0, fy * (1 - fx) * weight),
            (y0, x0 + 1, (1 - fy) * fx * weight),
            (y0 + 1, x0 + 1, fy * fx * weight))
    for yy, xx, v in taps:
        ok = (0 <= xx) & (xx < w) & (0 <= yy) & (yy < h)
        ind = ((xx + yy * w) * ok).long()
        img.scatter_add_(1, ind, v * ok)
    return img.reshape(nimg, h, w)


def blur_kernel_1d(dtype=torch.float32) -> torch.Tensor:
    """[a,c,a], a=exp(-.5)/(1+2exp(-.5)).  torchvision gaussian_blur(kernel_size=3, sigma=1)."""
    x = torch.linspace(-1.0, 1.0, steps=3, dtype=dtype)
    pdf = torch.exp(-0.5 * x.pow(2))
    return pdf / pdf.sum()


def gaussian_blur3(img: torch.Tensor) -> torch.Tensor:
    """[n,c,H,W] 3x3 sigma=1 blur with reflect padding.  event_image_converter.py:170-175."""
    k1 = blur_kernel_1d(img.dtype).to(img.device)
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = img.shape[1]
    return F.conv2d(F.pad(img, [1, 1, 1, 1], mode='reflect'), k2.expand(c, 1, 3, 3), groups=c)


def make_iwes(events, warped, t_ref, image_shape, scale_by_dt, mask_border, polarity_split,
              num_pos):
    """-> [b*T,2,H,W] (polarity split by ROW INDEX < num_pos) or [b*T,H,W].  focus.py:197-230."""
    b, T, m, _ = warped.shape
    w = event_weights(events, warped, t_ref, image_shape, scale_by_dt, mask_border)
    pos = warped.reshape(b * T, m, 2)
    w = w.reshape(b * T, m)
    if polarity_split:
        raw = torch.stack((bilinear_vote(pos[:, :num_pos], w[:, :num_pos], image_shape),
                           bilinear_vote(pos[:, num_pos:], w[:, num_pos:], image_shape)), 1)
        return gaussian_blur3(raw), raw
    raw = bilinear_vote(pos, w, image_shape)[:, None]
    return gaussian_blur3(raw)[:, 0], raw[:, 0]


# ----------------------------------------------------------------------------------------------
# A9 contrast objectives, A10 smoothness
# ----------------------------------------------------------------------------------------------
def sobel(a: torch.Tensor):
    """[n,c,H,W] -> dx, dy; zero padding.  loss.py:58-87."""
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=a.dtype, device=a.device)
    ky = kx.t()
    c = a.shape[1]
    return (F.conv2d(a, kx.expand(c, 1, 3, 3).contiguous(), padding=1, groups=c),
            F.conv2d(a, ky.expand(c, 1, 3, 3).contiguous(), padding=1, groups=c))


def contrast_value(iwes, loss_type='gradient_magnitude', norm='l1'):
    """loss.py:4-27.  iwes [n,H,W] or [n,c,H,W]."""
    if loss_type == 'variance':
        return torch.var(iwes, dim=(-2, -1)).mean()
    if iwes.dim() == 3:
        iwes = iwes[:, None]
    dx, dy = sobel(iwes)
    if norm == 'l2':
        return (dx.square() + dy.square()).mean()
    if norm == 'l1':
        return (dx.abs() + dy.abs()).mean()
    raise ValueError(norm)


def smoothness(field: torch.Tensor, eps=1e-3):
    """field [n,2,hq,wq]: (mean sqrt(dx^2+eps^2) + mean sqrt(dy^2+eps^2))/2.  loss.py:29-56."""
    dx, dy = sobel(field)
    return (torch.sqrt(dx ** 2 + eps ** 2).mean() + torch.sqrt(dy ** 2 + eps ** 2).mean()) / 2.


def contrast_grad_image(iwes_raw: torch.Tensor, norm='l1'):
    """Hand-derived d(1/val)/d(raw IWE) for the gradient-magnitude objective,
    written WITHOUT autograd so the HIP backward can be checked stage-wise.
    iwes_raw [n,c,H,W].  Returns (val, grad_raw)."""
    n, c, H, W = iwes_raw.shape
    blur = gaussian_blur3(iwes_raw)
    dx, dy = sobel(blur)
    if norm == 'l1':
        val = (dx.abs() + dy.abs()).mean()
        ux, uy = torch.sign(dx), torch.sign(dy)
    else:
        val = (dx.square() + dy.square()).mean()
        ux, uy = 2 * dx, 2 * dy
    N = iwes_raw.numel()
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=blur.dtype)
    ky = kx.t()
    # adjoint of zero-padded correlation = zero-padded correlation with the flipped kernel
    gB = F.conv2d(ux, kx.flip(0, 1).expand(c, 1, 3, 3).contiguous(), padding=1, groups=c) + \
        F.conv2d(uy, ky.flip(0, 1).expand(c, 1, 3, 3).contiguous(), padding=1, groups=c)
    gB = gB * (-(1.0 / (val * val)) / N)
    # adjoint of reflect-pad + 3x3 correlation: full correlation onto (H+2)x(W+2), fold ring back
    k1 = blur_kernel_1d(blur.dtype)
    k2 = torch.mm(k1[:, None], k1[None, :])
    full = F.conv2d(F.pad(gB, [2, 2, 2, 2]), k2.flip(0, 1).expand(c, 1, 3, 3).contiguous(),
                    groups=c)                                       # [n,c,H+2,W+2]
    g = full[..., 1:-1, 1:-1].clone()
    g[..., 1, :] += full[..., 0, 1:-1]
    g[..., H - 2, :] += full[..., H + 1, 1:-1]
    g[..., :, 1] += full[..., 1:-1, 0]
    g[..., :, W - 2] += full[..., 1:-1, W + 1]
    g[..., 1, 1] += full[..., 0, 0]
    g[..., 1, W - 2] += full[..., 0, W + 1]
    g[..., H - 2, 1] += full[..., H + 1, 0]
    g[..., H - 2, W - 2] += full[..., H + 1, W + 1]
    return val, g


# ----------------------------------------------------------------------------------------------
# A1/A4  the loss object (same constructor keys and calc() contract as the reference FocusLoss)
# ----------------------------------------------------------------------------------------------
class FocusLossOracle:
    """CPU twin of reference FocusLoss (focus.py:9-113).  `loss_type` is a build-side switch
    (default = the reference's hard-coded 'gradient_magnitude', focus.py:90-91)."""

    def __init__(self, image_shape, num_tref, num_bins, num_knn, smooth_weight,
                 lut_superpixel_size, focus_loss_norm, dist_norm, scale_iwe_by_dt,
                 mask_image_border, polarity_aware_batching, interpolation_scheme, smooth_type,
                 loss_type='gradient_magnitude', **kwargs):
        self.image_shape = tuple(image_shape)
        self.num_tref = num_tref
        self.num_bins = num_bins
        self.num_knn = num_knn
        self.smooth_weight = smooth_weight
        self.sp = lut_superpixel_size
        self.focus_loss_norm = focus_loss_norm
        self.dist_norm = dist_norm
        self.scale_iwe_by_dt = scale_iwe_by_dt
        self.mask_image_border = mask_image_border
        self.polarity_aware_batching = polarity_aware_batching
        self.interpolation_scheme = interpolation_scheme
        self.smooth_type = smooth_type
        self.loss_type = loss_type
        self.is_needing_offsets = True
        assert not scale_iwe_by_dt or num_tref == 1
        assert not polarity_aware_batching or num_tref == 1
        assert not smooth_type == 'on_flow_to_next' or num_tref == 1

    def get_reconstruction_times(self, device='cpu', generator=None):
        """focus.py:53-64."""
        if self.num_tref > 1:
            t_ref = torch.linspace(0, 1, self.num_tref)
        elif self.num_tref == 1:
            t_ref = torch.rand(1, generator=generator)
        else:
            raise ValueError("Invalid value for num_tref. Must be >= 1.")
        return torch.cat((t_ref, bin_mid_times(self.num_bins)))

    def event_path(self, events, flow_lut, t_ref, num_pos):
        """A6-A9 with the LUT given: returns (focus_loss, iwes_blurred, iwes_raw)."""
        warped = warp_events(events, flow_lut, self.sp)
        iwes, raw = make_iwes(events, warped, t_ref, self.image_shape, self.scale_iwe_by_dt,
                              self.mask_image_border, self.polarity_aware_batching, num_pos)
        val = contrast_value(iwes, self.loss_type, self.focus_loss_norm)
        return 1 / val, iwes, raw

    def smooth_loss(self, flow_lut, flow_next):
        """focus.py:232-246."""
        if self.smooth_weight == 0:
            return torch.tensor(0.)
        field = flow_lut if self.smooth_type == 'on_flow_to_tref' else flow_next
        if self.smooth_type not in ('on_flow_to_tref', 'on_flow_to_next'):
            raise ValueError(self.smooth_type)
        f = field.permute(0, 1, 4, 5, 2, 3)
        f = f.reshape(-1, f.shape[3], f.shape[4], f.shape[5])
        return self.smooth_weight * smoothness(f)

    def calc_per_event_basis(self, coeff_grid, t_ref, batch, num_basis, basis_type='polynomial'):
        """PARITY UNPINNED: the DEFINITION of the per-event continuous-time basis warp of
        motionpriorcmax_amd.FocusLoss.calc_per_event_basis, written from the reference's building blocks -- the reference
        itself has no such path (focus.py:182-195 gathers a binned KNN look-up table), so there is nothing to pin it to.
            warped = (y, x) + sum_k c_k[tile(y, x)] * (basis_k(t_ref) - basis_k(t_event))
        with the tile coefficients of trajectories.py:15-52 and the basis of basis.py:18-31 (the flow to t_ref of the trajectory
        that starts at the event's tile), then focus.py:197-230 / loss.py unchanged; smoothness (loss.py:29-56) on the same
        flow at the bin mid-times."""
        events = batch['events']
        num_pos = batch['num_pos_events'] if 'num_pos_events' in batch else -1
        h, w = self.image_shape
        mask = tile_mask((h, w), self.sp)
        if coeff_grid.dim() == 4:
            coeff_grid = coeff_grid[:, None]
        coeffs, _ = coeff_grid_to_list(coeff_grid, mask, num_basis)              # [b,s,2,n,k]
        hq, wq = -(-h // self.sp), -(-w // self.sp)
        b, m, _ = events.shape
        c = coeffs.sum(1).reshape(b, 2, hq, wq, num_basis)
        t_ref = torch.as_tensor(t_ref, dtype=torch.float32).reshape(1)
        ib = torch.arange(b)[:, None].expand(b, m)
        iy = torch.div(events[..., 0], self.sp, rounding_mode='floor').to(torch.int64).clamp(0, hq - 1)
        ix = torch.div(events[..., 1], self.sp, rounding_mode='floor').to(torch.int64).clamp(0, wq - 1)
        phi = basis_matrix(t_ref, num_basis, basis_type)[None] - \
            basis_matrix(events[..., 2].reshape(-1), num_basis, basis_type).reshape(b, m, num_basis)        # [b,m,k]
        flow = (c[ib, :, iy, ix] * phi[:, :, None, :]).sum(-1)                  # [b,m,2]
        warped = (events[..., :2] + flow)[:, None]                               # [b,T=1,m,2]
        iwes, raw = make_iwes(events, warped, t_ref, self.image_shape, self.scale_iwe_by_dt,
                              self.mask_image_border, self.polarity_aware_batching, num_pos)
        focus = 1 / contrast_value(iwes, self.loss_type, self.focus_loss_norm)
        smooth = torch.tensor(0.)
        if self.smooth_weight > 0:
            phim = basis_matrix(t_ref, num_basis, basis_type) - basis_matrix(bin_mid_times(self.num_bins), num_basis, basis_type)
            field = torch.einsum('bdhwk,tk->btdhw', c, phim).reshape(-1, 2, hq, wq)
            smooth = self.smooth_weight * smoothness(field)
        return focus + smooth, {'focus_loss': focus.detach(), 'smoothness_loss': smooth.detach()}, {'iwes': iwes.detach()}

    def calc(self, trajectories, times, batch):
        """focus.py:66-113.  Returns (loss, log_metadata, misc_metadata)."""
        events = batch['events']
        num_pos = batch['num_pos_events'] if 'num_pos_events' in batch else -1
        assert not self.polarity_aware_batching or num_pos > -1
        T = self.num_tref
        t_ref = times[:T]
        want_next = self.smooth_weight > 0 and self.smooth_type == 'on_flow_to_next'
        flow_lut, flow_next = interpolate_flow(
            trajectories[:, :T], trajectories[:, T:], self.image_shape, self.sp, self.num_knn,
            self.dist_norm, self.interpolation_scheme, want_next)
        focus, iwes, _ = self.event_path(events, flow_lut, t_ref, num_pos)
        smooth = self.smooth_loss(flow_lut, flow_next)
        loss = focus + smooth
        b = events.shape[0]
        h, w = self.image_shape
        iw = iwes.reshape(b, T, 2, h, w) if self.polarity_aware_batching else \
            iwes.reshape(b, T, h, w)
        return (loss,
                {'focus_loss': focus.detach(), 'smoothness_loss': smooth.detach()},
                {'iwes': iw.detach()})


# ----------------------------------------------------------------------------------------------
# seeded synthetic inputs shared by tests, smoke() and bench.py
# ----------------------------------------------------------------------------------------------
def synth_events(b, m, image_shape, num_bins, seed=0, pad_frac=0.0, time_sorted=False,
                 num_pos=None):
    """Event tensor [b,m,6] with columns (y,x,t,p,bin,valid) as built by the DSEC loader
    (loader.py:152-167,360-395): positive block then negative block, zero padding rows at the
    end of each block."""
    g = torch.Generator().manual_seed(seed)
    h, w = image_shape
    num_pos = m // 2 if num_pos is None else num_pos
    ev = torch.zeros(b, m, 6)
    ev[..., 0] = torch.rand(b, m, generator=g) * (h - 1)
    ev[..., 1] = torch.rand(b, m, generator=g) * (w - 1)
    t = torch.rand(b, m, generator=g)
    if time_sorted:
        t = torch.cat((torch.sort(t[:, :num_pos], 1).values, torch.sort(t[:, num_pos:], 1).values), 1)
    ev[..., 2] = t
    ev[:, :num_pos, 3] = 1
    ev[..., 4] = torch.clamp(torch.floor(t * num_bins), 0, num_bins - 1)
    ev[..., 5] = 1
    if pad_frac > 0:
        for blk in ((0, num_pos), (num_pos, m)):
            n_pad = int((blk[1] - blk[0]) * pad_frac)
            if n_pad:
                ev[:, blk[1] - n_pad:blk[1]] = 0
    return ev, num_pos
