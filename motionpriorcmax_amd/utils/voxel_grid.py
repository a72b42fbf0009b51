"""Voxel-grid event representation on the GPU.

Mirrors reference src/loader/dsec/utils.py:19-77 (`VoxelGrid(input_size, norm_type, quantile).convert(
{'p','t','x','y'})`) and adds a batched entry point; the numerics run in libmpcmax.so (csrc/voxel.hip)."""
import ctypes

import torch

from .. import _lib as C
from ..ops import _ptr, _require_gpu, _stream, _stage


def voxel_grids(xytp: torch.Tensor, counts: torch.Tensor, input_size, norm_type='mean_std', quantile=0.0) -> torch.Tensor:
    """xytp [B, N, 4] (x, y, t, p), counts [B] int32 (valid rows per sample) -> [B, C, H, W]."""
    _require_gpu(xytp, 'events')
    if norm_type not in ('mean_std', 'max', None):
        raise AssertionError(norm_type)
    Cn, H, W = (int(v) for v in input_size)
    ev = xytp.float().contiguous()
    B, N, _ = ev.shape
    cnt = counts.to(device=ev.device, dtype=torch.int32).contiguous()
    shape = C.VoxShape(B=B, N=N, C=Cn, H=H, W=W, norm={None: 0, 'mean_std': 1, 'max': 2}[norm_type], quantile=float(quantile),
                       keep=float(1.0 - float(quantile)) if quantile > 0 else 0.0)      # 1 - q in double, rounded once (torch.quantile's argument)
    nbytes = C.lib().mpc_voxel_workspace_bytes(ctypes.byref(shape))
    if nbytes < 0:
        C.check(int(nbytes), 'mpc_voxel_workspace_bytes')
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ev.device)
    grid = torch.empty((B, Cn, H, W), dtype=torch.float32, device=ev.device)
    with _stage('mpc_voxel_grid', ev.device):
        C.check(C.lib().mpc_voxel_grid(ctypes.byref(shape), _ptr(ev), _ptr(cnt), _ptr(grid), _ptr(ws), _stream(ev.device)),
                'mpc_voxel_grid')
    return grid


class VoxelGrid:
    """Same constructor and `convert` contract as the reference class (utils.py:19-77)."""

    def __init__(self, input_size: tuple, norm_type: str, quantile):
        assert len(input_size) == 3
        self.nb_channels = input_size[0]
        self.input_size = tuple(int(v) for v in input_size)
        self.norm_type = norm_type
        assert self.norm_type in ['mean_std', 'max', None]
        self.quantile = quantile
        assert 0 <= self.quantile < 0.15

    def convert(self, events):
        """events: dict of [N] tensors 'p', 't', 'x', 'y' on the GPU -> [C, H, W]."""
        x = events['x']
        ev = torch.stack((x.float(), events['y'].float(), events['t'].float(), events['p'].float()), dim=-1)[None]
        cnt = torch.tensor([ev.shape[1]], dtype=torch.int32, device=ev.device)
        with torch.no_grad():
            return voxel_grids(ev, cnt, self.input_size, self.norm_type, self.quantile)[0]
