"""Error behaviour of the C ABI on the device: argument errors come back as negative codes with a message
(surfaced as RuntimeError by the ctypes layer), never as a crash or a silent wrong answer; the Python mirror
keeps the reference's assertions (focus.py:49-51,80)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cfg(**kw):
    cfg = dict(image_shape=(48, 64), num_tref=1, num_bins=3, num_knn=4, smooth_weight=0.003, lut_superpixel_size=4,
               focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
               polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    cfg.update(kw)
    return cfg


def test_c_abi_argument_errors():
    from motionpriorcmax_amd import LossFactory, _lib as C, ops
    L = LossFactory.get_loss_calculator('FOCUS', _cfg())
    lib = C.lib()
    # inconsistent LUT grid -> MPC_E_SHAPE from the size query already
    bad = ops.make_shape(L._cfg, 1, 0, 0, 192)
    bad.hq += 1
    assert lib.mpc_workspace_bytes(ctypes.byref(bad)) == -2
    assert b'hq/wq' in lib.mpc_last_error_string()
    # K > n -> MPC_E_SHAPE; 65536 trajectories per sample -> MPC_E_UNSUPPORTED; null output -> MPC_E_NULL
    shape = ops.make_shape(L._cfg, 1, 0, 0, 3)          # n = 3 < K = 4
    ws = ops.alloc_workspace(shape, DEV)
    traj = torch.zeros(1, 4, 3, 2, device=DEV)
    out = torch.zeros(1, 3, 12, 16, 1, 2, device=DEV)
    state = torch.zeros(3 * 3 * 192 + 3, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mpc_knn_lut_fwd(ctypes.byref(shape), P(traj), P(out), None, P(state), None, P(ws), st) == -2
    big = ops.make_shape(L._cfg, 1, 0, 0, 70000)
    wsb = ops.alloc_workspace(big, DEV)
    trajb = torch.zeros(1, 4, 70000, 2, device=DEV)
    assert lib.mpc_knn_lut_fwd(ctypes.byref(big), P(trajb), P(out), None, P(state), None, P(wsb), st) == -4
    assert b'65535' in lib.mpc_last_error_string()
    ok = ops.make_shape(L._cfg, 1, 0, 0, 192)
    assert lib.mpc_knn_lut_fwd(ctypes.byref(ok), P(torch.zeros(1, 4, 192, 2, device=DEV)), None, None, P(state), None,
                               P(ops.alloc_workspace(ok, DEV)), st) == -1
    torch.cuda.synchronize()


def test_python_mirror_keeps_the_reference_assertions():
    from motionpriorcmax_amd import LossFactory
    with pytest.raises(ValueError, match='Unsupported loss type'):
        LossFactory.get_loss_calculator('PHOTOMETRIC', _cfg())
    with pytest.raises(AssertionError):                                 # focus.py:49-51
        LossFactory.get_loss_calculator('FOCUS', _cfg(num_tref=2))
    L = LossFactory.get_loss_calculator('FOCUS', _cfg())
    traj = torch.zeros(1, 4, 192, 2, device=DEV)
    times = torch.tensor([0.5, 1 / 6, 0.5, 5 / 6], device=DEV)
    with pytest.raises((AssertionError, KeyError)):                     # focus.py:78-80: num_pos_events is required
        L.calc(traj, times, {'events': torch.zeros(1, 10, 6, device=DEV)})
    with pytest.raises(RuntimeError, match='rc=-2'):                    # fewer trajectories than num_knn
        L.calc(torch.zeros(1, 4, 3, 2, device=DEV), times, {'events': torch.zeros(1, 10, 6, device=DEV), 'num_pos_events': 5})
