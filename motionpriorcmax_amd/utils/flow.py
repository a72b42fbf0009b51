"""Dense optical flow from tile trajectories and the flow error metrics on the GPU.

Mirrors reference src/utils/flow.py (`dense_flow_from_traj` :12-16, `calculate_flow_error` :18-70) and
src/utils/metrics.py:36-56 (`ErrorCalculatorFactory`, `OpticalFlowError.run`); the numerics run in
libmpcmax.so (csrc/flow.hip).  Both are evaluation-time operators: no gradient is provided."""
import ctypes

import torch

from .. import _lib as C
from ..ops import _ptr, _require_gpu, _stage, _stream


def dense_flow_from_traj(traj_flow, pixel_positions, patch_size, image_shape):
    """traj_flow [B, n, C], pixel_positions [n, 2] (y, x) -> (dense [B, C, H, W], patch_flow [B, C, H//p, W//p])."""
    _require_gpu(traj_flow, 'traj_flow')
    h, w = (int(v) for v in image_shape)
    tf = traj_flow.detach().float().contiguous()
    B, n, Cn = tf.shape
    pix = pixel_positions.to(device=tf.device, dtype=torch.int64).contiguous()
    assert pix.shape == (n, 2)
    shape = C.FlowShape(B=B, C=Cn, n=n, patch=int(patch_size), H=h, W=w)
    patch = torch.empty((B, Cn, h // int(patch_size), w // int(patch_size)), dtype=torch.float32, device=tf.device)
    dense = torch.empty((B, Cn, h, w), dtype=torch.float32, device=tf.device)
    with _stage('mpc_dense_flow', tf.device):
        C.check(C.lib().mpc_dense_flow(ctypes.byref(shape), _ptr(tf), _ptr(pix), _ptr(patch), _ptr(dense),
                                       _stream(tf.device)), 'mpc_dense_flow')
    return dense, patch


def calculate_flow_error(flow_gt, flow_pred, event_mask=None, time_scale=None) -> dict:
    """flow_gt, flow_pred [B, 2, H, W]; event_mask [B, 1, H, W] or [B, H, W]; time_scale [B, 1].
    Returns {'EPE', '1PE', '2PE', '3PE', 'AE'} as 0-dim tensors on the device."""
    _require_gpu(flow_gt, 'flow_gt')
    gt = flow_gt.detach().float().contiguous()
    pr = flow_pred.detach().to(device=gt.device, dtype=torch.float32).contiguous()
    B, two, H, W = gt.shape
    assert two == 2 and pr.shape == gt.shape
    em = None
    if event_mask is not None:
        em = event_mask.to(gt.device)
        em = (em if em.dtype == torch.bool else em != 0).reshape(B, H, W).contiguous()
    ts = None if time_scale is None else time_scale.to(device=gt.device, dtype=torch.float32).reshape(B).contiguous()
    shape = C.ErrShape(B=B, H=H, W=W)
    nbytes = C.lib().mpc_flow_error_workspace_bytes(ctypes.byref(shape))
    if nbytes < 0:
        C.check(int(nbytes), 'mpc_flow_error_workspace_bytes')
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=gt.device)
    out = torch.empty(5, dtype=torch.float32, device=gt.device)
    with _stage('mpc_flow_error', gt.device):
        C.check(C.lib().mpc_flow_error(ctypes.byref(shape), _ptr(gt), _ptr(pr), _ptr(em), _ptr(ts), _ptr(out), _ptr(ws),
                                       _stream(gt.device)), 'mpc_flow_error')
    return {k: out[i] for i, k in enumerate(('EPE', '1PE', '2PE', '3PE', 'AE'))}


class OpticalFlowError:
    """metrics.py:50-56."""

    @staticmethod
    def run(predictions, batch):
        return calculate_flow_error(batch['forward_flow'], predictions['flow'], batch['flow_valid'])


class ErrorCalculatorFactory:
    """metrics.py:36-42."""

    @staticmethod
    def get_error_calculator(data_type):
        if data_type == 'DSEC' or data_type == 'MVSEC':
            return OpticalFlowError()
        raise ValueError("Unsupported dataset type")
