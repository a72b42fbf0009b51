"""End-to-end sanity beyond parity: contrast maximisation with the HIP loss recovers a known flow.

Events are emitted by random scene points that move with a constant velocity v over the window; a
constant polynomial-k1 coefficient field c is optimised with Adam from c = 0.  The loss gradient
(hand-derived backward through LUT, warp, vote, blur, Sobel) must drive c towards v."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionpriorcmax_amd import LossFactory, utils


def run(v=(6.0, -9.0), shape=(128, 160), n_pts=1500, ev_per_pt=12, steps=120, lr=0.4, seed=0, verbose=True):
    H, W = shape
    nb = 15
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed)
    p0 = torch.rand(n_pts, 2, generator=g) * torch.tensor([H - 40.0, W - 40.0]) + 20.0
    t = torch.rand(n_pts, ev_per_pt, generator=g)
    pos = p0[:, None, :] + torch.tensor(v)[None, None, :] * t[..., None]
    M = n_pts * ev_per_pt
    ev = torch.zeros(1, M, 6)
    ev[0, :, :2] = pos.reshape(-1, 2)
    ev[0, :, 2] = t.reshape(-1)
    ev[0, :M // 2, 3] = 1
    ev[0, :, 4] = torch.clamp(torch.floor(t.reshape(-1) * nb), 0, nb - 1)
    ev[0, :, 5] = 1
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=8, smooth_weight=0.0, lut_superpixel_size=4,
               focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
               polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    mask = utils.get_optical_flow_tile_mask(shape, 4)
    c = torch.zeros(2, device=dev, requires_grad=True)          # (cy, cx): one constant flow for the whole image
    opt = torch.optim.Adam([c], lr=lr)
    batch = {'events': ev.to(dev), 'num_pos_events': M // 2}
    pos_grid = torch.nonzero(mask).float().to(dev)
    for it in range(steps):
        times = L.get_reconstruction_times(dev)
        traj = pos_grid[None, None] + c[None, None, None, :] * times[None, :, None, None]   # poly-k1, anchor t=0
        loss, log, _ = L.calc(traj, times, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if verbose and it % 20 == 0:
            print(f'it {it:3d}  loss {loss.item():.5f}  c = ({c[0].item():+.3f}, {c[1].item():+.3f})')
    return c.detach().cpu(), torch.tensor(v)


if __name__ == '__main__':
    c, v = run()
    print('recovered', c.tolist(), 'true', v.tolist(), 'error', (c - v).abs().max().item())
