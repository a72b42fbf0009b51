"""Flow curves -> trajectories on the device (csrc/curves.hip: mpc_curve_traj_fwd / _bwd; SURVEY.md 8f-4, BASELINE.json configs[3]) against
the same product in plain torch -- the path CPU tensors take, which tests/test_abi_and_host.py pins to the reference's Bezier curves
(golden G6) -- forward and adjoint, for the Bernstein and the clamped B-spline basis, several degrees and batch sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kind,d,B,hw,scale', [('bezier', 10, 1, (120, 160), 1.0), ('bezier', 3, 3, (24, 32), 8.0), ('bspline', 10, 2, (120, 160), 1.0),
                                             ('bspline', 4, 1, (12, 16), 0.5), ('bezier', 16, 1, (12, 16), 1.0)])
def test_curve_trajectories_match_plain_torch(kind, d, B, hw, scale):
    from motionpriorcmax_amd import utils
    from motionpriorcmax_amd.utils.synth import bin_mid_times
    dev = torch.device('cuda:0')
    h, w = hw
    H, W, tile = 4 * h, 4 * w, 4
    g = torch.Generator().manual_seed(5)
    params_c = torch.randn(B, 2 * d, h, w, generator=g) * 2.0
    times = torch.cat((torch.tensor([0.41]), bin_mid_times(15), torch.tensor([0.0, 1.0])))
    fn = utils.trajectories_from_bezier if kind == 'bezier' else utils.trajectories_from_bspline
    pc = params_c.clone().requires_grad_(True)
    ref, pos_c = fn(pc, times, tile, (H, W), scale)                     # plain torch on the host
    pd = params_c.to(dev).requires_grad_(True)
    got, pos_d = fn(pd, times.to(dev), tile, (H, W), scale)             # one kernel on the device
    assert got.is_cuda and got.shape == ref.shape and torch.equal(pos_c, pos_d)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-5)
    # t = 0: the flow from the reference time vanishes, the trajectory is at its tile centre (curves/base.py:102-106)
    assert torch.equal(got[:, -2].detach().cpu(), pos_c.to(torch.float32)[None].expand(B, -1, -1))
    # the adjoint against autograd through the torch product
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    got.backward(go.to(dev))
    num = (pd.grad.cpu() - pc.grad).norm().item()
    assert num <= 2e-6 * pc.grad.norm().item(), (num, pc.grad.norm().item())
    # reproducible: the kernels sum in index order
    pd2 = params_c.to(dev).requires_grad_(True)
    got2, _ = fn(pd2, times.to(dev), tile, (H, W), scale)
    got2.backward(go.to(dev))
    assert torch.equal(got2, got) and torch.equal(pd2.grad, pd.grad)


def test_curve_entry_points_refuse_bad_arguments():
    import ctypes
    from motionpriorcmax_amd import _lib as C
    dev = torch.device('cuda:0')
    x = torch.zeros(64, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    L = C.lib()
    assert L.mpc_curve_traj_fwd(None, vp(x), vp(x), 1.0, vp(x), 1, 2, 2, 4, None) == C.E_NULL
    assert L.mpc_curve_traj_fwd(vp(x), vp(x), vp(x), 1.0, vp(x), 1, 0, 2, 4, None) == C.E_SHAPE
    assert L.mpc_curve_traj_fwd(vp(x), vp(x), vp(x), 1.0, vp(x), 1, 17, 2, 1, None) == C.E_UNSUPPORTED
    assert L.mpc_curve_traj_bwd(vp(x), vp(x), 1.0, None, 1, 2, 2, 4, None) == C.E_NULL
    assert b'control points' in L.mpc_last_error_string() or b'null' in L.mpc_last_error_string()
