"""CPU ORACLE for dense flow + flow metrics -- TEST INFRASTRUCTURE ONLY.

Restates reference src/utils/flow.py:9-16 (`dense_flow_from_traj`: `list_to_grid`
(src/utils/trajectories.py:54-76) then torchvision `resize(..., BICUBIC, antialias=True)`) and
src/utils/flow.py:18-70 (`calculate_flow_error`).

Third-party arithmetic: torchvision (unpinned in requirements.txt) `transforms.functional.resize` on a float
tensor is `torch.nn.functional.interpolate(img, size, mode='bicubic', align_corners=False, antialias=True)`
without clamping; `resize_bicubic_aa` below restates torch's separable anti-aliased bicubic filter
(aten UpSampleKernel.cpp `_compute_indices_min_size_weights_aa` + `HelperInterpCubic::aa_filter`, a = -0.5;
width pass first, then height).  Pinned by tests/golden/g10_*.npz, produced by the unmodified reference
flow.py with the torchvision stand-in forwarding to torch's own interpolate (oracle/gen_golden_flow.py)."""
import numpy as np
import torch


def _aa_cubic(x):
    a = np.float32(-0.5)
    x = np.abs(x)
    return np.where(x < 1, ((a + 2) * x - (a + 3)) * x * x + 1,
                    np.where(x < 2, (((x - 5) * x + 8) * x - 4) * a, 0)).astype(np.float32)


def _aa_weights(n_in, n_out):
    """Per output index: (first input index, normalised weights)."""
    scale = np.float32(n_in) / np.float32(n_out)
    support = np.float32(2.0) * scale if scale >= 1 else np.float32(2.0)
    invscale = np.float32(1.0) / scale if scale >= 1 else np.float32(1.0)
    out = []
    for i in range(n_out):
        center = scale * np.float32(i + 0.5)
        xmin = max(int(center - support + np.float32(0.5)), 0)
        xsize = min(int(center + support + np.float32(0.5)), n_in) - xmin
        j = np.arange(xsize, dtype=np.float32)
        w = _aa_cubic((j + np.float32(xmin) - center + np.float32(0.5)) * invscale)
        tot = np.float32(0)
        for v in w:
            tot = np.float32(tot + v)
        out.append((xmin, (w / tot).astype(np.float32)))
    return out


def resize_bicubic_aa(img, size):
    """img [..., h, w] float32 numpy -> [..., H, W]."""
    img = np.asarray(img, dtype=np.float32)
    H, W = size
    h, w = img.shape[-2:]
    tmp = np.zeros(img.shape[:-1] + (W,), np.float32)
    for i, (x0, wt) in enumerate(_aa_weights(w, W)):
        acc = img[..., x0] * wt[0]
        for j in range(1, len(wt)):
            acc = (acc + img[..., x0 + j] * wt[j]).astype(np.float32)
        tmp[..., i] = acc
    out = np.zeros(img.shape[:-2] + (H, W), np.float32)
    for i, (y0, wt) in enumerate(_aa_weights(h, H)):
        acc = tmp[..., y0, :] * wt[0]
        for j in range(1, len(wt)):
            acc = (acc + tmp[..., y0 + j, :] * wt[j]).astype(np.float32)
        out[..., i, :] = acc
    return out


def list_to_grid(feature_list, pixel_positions, image_shape):
    """trajectories.py:54-76: [b, n, c] at integer (y, x) -> [b, c, h, w], zeros elsewhere."""
    f = np.asarray(feature_list, np.float32)
    b, n, c = f.shape
    g = np.zeros((b, c) + tuple(image_shape), np.float32)
    pp = np.asarray(pixel_positions)
    g[:, :, pp[:, 0], pp[:, 1]] = f.transpose(0, 2, 1)
    return g


def dense_flow_from_traj(traj_flow, pixel_positions, patch_size, image_shape):
    """flow.py:12-16."""
    h, w = image_shape
    patch = list_to_grid(traj_flow, np.asarray(pixel_positions) // patch_size, (h // patch_size, w // patch_size))
    return resize_bicubic_aa(patch, image_shape), patch


def calculate_flow_error(flow_gt, flow_pred, event_mask=None, time_scale=None):
    """flow.py:18-70, same op order, fp32 torch on the CPU.  Returns {'EPE','1PE','2PE','3PE','AE'}."""
    gt, pr = torch.as_tensor(flow_gt, dtype=torch.float32), torch.as_tensor(flow_pred, dtype=torch.float32)
    fm = (~torch.isinf(gt[:, [0]])) & (~torch.isinf(gt[:, [1]])) & (gt[:, [0]].abs() > 0) & (gt[:, [1]].abs() > 0)
    if event_mask is None:
        tm = fm
    else:
        em = torch.as_tensor(event_mask)
        if em.dim() == 3:
            em = em[:, None]
        tm = torch.logical_and(em, fm)
    gm, pm = gt * tm, pr * tm                      # a product, not a select: inf * 0 = nan as in the reference
    npts = tm.sum(dim=(1, 2, 3)) + 1e-5
    if time_scale is not None:
        ts = torch.as_tensor(time_scale, dtype=torch.float32).reshape(len(gm), 1, 1, 1)
        gm, pm = gm * ts, pm * ts
    epe = torch.linalg.norm(gm - pm, dim=1)
    out = {'EPE': torch.mean(epe.sum(dim=(1, 2)) / npts)}
    for k in (1, 2, 3):
        out['%dPE' % k] = torch.mean((epe > k).sum(dim=(1, 2)) / npts)
    u, v, ug, vg = pm[:, 0], pm[:, 1], gm[:, 0], gm[:, 1]
    cs = (1.0 + u * ug + v * vg) / (torch.sqrt(1 + u * u + v * v) * torch.sqrt(1 + ug * ug + vg * vg))
    out['AE'] = torch.mean(torch.acos(cs.clamp(-1, 1)).sum(dim=(1, 2)) / npts) * (180.0 / torch.pi)
    return out


def synth_flow_case(B, H, W, seed=0, with_mask=True, with_scale=True, zeros=0.1):
    """Seeded ground-truth/predicted flow pair with invalid (zero) ground-truth pixels and an event mask."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.randn(B, 2, H, W, generator=g) * 4
    pr = gt + torch.randn(B, 2, H, W, generator=g) * 1.5
    gt[torch.rand(B, 2, H, W, generator=g) < zeros] = 0.0
    em = (torch.rand(B, 1, H, W, generator=g) < 0.7) if with_mask else None
    ts = (0.5 + torch.rand(B, 1, generator=g)) if with_scale else None
    return gt, pr, em, ts
