"""The 'learned' motion basis (reference trajectory_net.py:35-47,79-80: an MLP 1 -> 64 -> 64 -> 64 -> k of the time) through the
package's compute_basis and, on the GPU, through calc: golden `g11_learned_basis` holds the reference's MLP weights, its
trajectories, loss and the gradients w.r.t. the coefficient grid and the MLP weights (oracle/gen_golden_learned.py)."""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import load_golden


def _net(g, device='cpu'):
    net = nn.Sequential(nn.Linear(1, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(),
                        nn.Linear(64, int(g['num_basis'])))
    net.load_state_dict({k: torch.from_numpy(g['net_' + k.replace('.', '_')]) for k in net.state_dict()})
    return net.to(device)


def _trajectories(g, net, device):
    from motionpriorcmax_amd import utils
    cfg = g['cfg']
    k, patch = int(g['num_basis']), int(g['patch'])
    cg = torch.from_numpy(g['coeff_grid']).to(device).requires_grad_(True)
    times = torch.from_numpy(g['times']).to(device)
    mask = utils.get_optical_flow_tile_mask(cfg['image_shape'], patch).to(device)
    coeffs, pos, _ = utils.coeffs_grid_to_list(cg, mask, num_coeffs=k)
    traj = utils.compute_basis(coeffs, times, k, 'learned', net) - utils.compute_basis(coeffs, torch.zeros(1, device=device), k, 'learned', net)
    return cg, times, (traj + pos[None, :, None, :]).permute(0, 2, 1, 3).contiguous(), mask


def test_learned_basis_trajectories_match_reference():
    g = load_golden('g11_learned_basis')
    net = _net(g)
    np.testing.assert_allclose(net(torch.from_numpy(g['times'])[..., None]).detach().numpy(), g['basis_at_times'], rtol=0, atol=1e-6)
    _, _, traj, _ = _trajectories(g, net, 'cpu')
    np.testing.assert_allclose(traj.detach().numpy(), g['trajectories'], rtol=0, atol=1e-5)


def test_oracle_on_learned_basis_matches_reference():
    from oracle import focus_oracle as O
    g = load_golden('g11_learned_basis')
    net = _net(g)
    cg, times, traj, mask = _trajectories(g, net, 'cpu')
    loss, _, _ = O.FocusLossOracle(**g['cfg']).calc(traj, times, {'events': torch.from_numpy(g['events']), 'num_pos_events': int(g['num_pos'])})
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-6 * abs(float(g['loss']))
    ref = g['grad_coeff_grid_at_tiles']
    got = cg.grad[..., mask].numpy()
    assert np.linalg.norm(got - ref) <= 1e-4 * np.linalg.norm(ref)
    # (some of these gradients cancel to rounding noise -- the anchor subtracts the basis at t = 0: absolute floor from the largest)
    top = max(np.linalg.norm(g[k]) for k in g if k.startswith('grad_net_'))
    for name, prm in net.named_parameters():
        r = g['grad_net_' + name.replace('.', '_')]
        assert np.linalg.norm(prm.grad.numpy() - r) <= 1e-4 * np.linalg.norm(r) + 1e-4 * top, name


@pytest.mark.gpu
def test_calc_on_learned_basis_matches_reference_on_the_device():
    from motionpriorcmax_amd import LossFactory
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda', 0)
    g = load_golden('g11_learned_basis')
    net = _net(g, dev)
    cg, times, traj, mask = _trajectories(g, net, dev)
    assert traj.is_cuda
    L = LossFactory.get_loss_calculator('FOCUS', g['cfg'])
    loss, log, _ = L.calc(traj, times, {'events': torch.from_numpy(g['events']).to(dev), 'num_pos_events': int(g['num_pos'])})
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    ref = g['grad_coeff_grid_at_tiles']
    got = cg.grad[..., mask].cpu().numpy()
    assert np.linalg.norm(got - ref) <= 1e-3 * np.linalg.norm(ref)
    top = max(np.linalg.norm(g[k]) for k in g if k.startswith('grad_net_'))
    for name, prm in net.named_parameters():
        r = g['grad_net_' + name.replace('.', '_')]
        assert np.linalg.norm(prm.grad.cpu().numpy() - r) <= 1e-3 * np.linalg.norm(r) + 1e-4 * top, name
