"""Event ingest on the GPU: raw windows -> the padded [B, M, 6] event tensor and
`num_pos_events` that `FocusLoss.calc` consumes, and (optionally) the (x, y, t, p) rows for the
voxel-grid builder.  Mirrors reference src/loader/dsec/loader.py:152-167 + 360-415; numerics in
libmpcmax.so (csrc/ingest.hip)."""
import ctypes

import torch

from .. import _lib as C
from ..ops import _ptr, _require_gpu, _stream, _stage


def ingest_events(x, y, t_us, p, counts, image_shape, num_bins, want_voxel_input=False, order_for=None):
    """x, y, p: [B, N] float32; t_us: [B, N] int64 (increasing per sample); counts: [B] valid lengths.
    Returns {'events': [B, M, 6], 'num_pos_events': int, 'xytp': [B, N, 4] or None}.
    One host round trip (two integers) sizes the output, as the reference's CPU collate does.
    order_for: a FocusLoss -- the rows of each polarity block are then ordered by (time bin, LUT strip) for that loss
    (FocusLoss.order_events) and the dict carries 'event_offsets'; `calc` gives the same loss and gradient bit for bit."""
    _require_gpu(x, 'x')
    dev = x.device
    B, N = x.shape
    x = x.float().contiguous(); y = y.float().contiguous(); p = p.float().contiguous()
    t_us = t_us.to(torch.int64).contiguous()
    cnt = counts.to(device=dev, dtype=torch.int32).contiguous()
    shape = C.IngestShape(B=B, N=N, H=int(image_shape[0]), W=int(image_shape[1]), nb=int(num_bins))
    nbytes = C.lib().mpc_ingest_workspace_bytes(ctypes.byref(shape))
    if nbytes < 0:
        C.check(int(nbytes), 'mpc_ingest_workspace_bytes')
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
    out_max = torch.empty(2, dtype=torch.int32, device=dev)
    st = _stream(dev)
    with _stage('mpc_ingest_count', dev):
        C.check(C.lib().mpc_ingest_count(ctypes.byref(shape), _ptr(x), _ptr(y), _ptr(t_us), _ptr(p), _ptr(cnt),
                                         _ptr(out_max), _ptr(ws), st), 'mpc_ingest_count')
    max_pos, max_neg = (int(v) for v in out_max.tolist())          # the collate's host decision
    events = torch.empty((B, max_pos + max_neg, 6), dtype=torch.float32, device=dev)
    xytp = torch.empty((B, N, 4), dtype=torch.float32, device=dev) if want_voxel_input else None
    if order_for is not None:
        # the rows are written in bucket order right away (mpc_ingest_scatter_ordered): no ordering pass over the tensor
        from ..ops import make_shape
        cfg = order_for._cfg
        lshape = make_shape(cfg, B, max_pos + max_neg, max_pos if cfg.polarity_split else max_pos + max_neg, 1)
        ncs = int(C.lib().mpc_event_lut_strips(ctypes.byref(lshape)))
        if ncs > 0 and cfg.polarity_split and tuple(cfg.image_shape) == (int(image_shape[0]), int(image_shape[1])) and cfg.num_bins == int(num_bins):
            nb2 = int(C.lib().mpc_ingest_ordered_workspace_bytes(ctypes.byref(shape), ctypes.byref(lshape)))
            ws2 = torch.empty(max(nb2, 256), dtype=torch.uint8, device=dev)
            offs = torch.empty((B, 2, cfg.num_bins * ncs + 1), dtype=torch.int32, device=dev)
            with _stage('mpc_ingest_scatter_ordered', dev):
                C.check(C.lib().mpc_ingest_scatter_ordered(ctypes.byref(shape), ctypes.byref(lshape), _ptr(x), _ptr(y), _ptr(t_us), _ptr(p),
                                                           _ptr(cnt), max_pos, max_neg, _ptr(events), _ptr(offs), _ptr(xytp), _ptr(ws),
                                                           _ptr(ws2), st), 'mpc_ingest_scatter_ordered')
            return {'events': events, 'num_pos_events': max_pos, 'xytp': xytp, 'event_offsets': offs}
    with _stage('mpc_ingest_scatter', dev):
        C.check(C.lib().mpc_ingest_scatter(ctypes.byref(shape), _ptr(x), _ptr(y), _ptr(t_us), _ptr(p), _ptr(cnt),
                                           max_pos, max_neg, _ptr(events), _ptr(xytp), _ptr(ws), st), 'mpc_ingest_scatter')
    out = {'events': events, 'num_pos_events': max_pos, 'xytp': xytp}
    return out if order_for is None else order_for.order_events(out)
