"""torch-facing wrappers of the HIP kernels behind include/mpcmax.h.

PyTorch is used only for device memory, the current HIP stream and autograd plumbing; every
numerical step of the path runs in libmpcmax.so.  Tensors must live on a GPU: there is no CPU
or eager fallback (a CPU tensor raises)."""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass

import torch

from . import _lib as C


@dataclass(frozen=True)
class PathConfig:
    """Static configuration of one FocusLoss instance (reference focus.py:28-45)."""
    image_shape: tuple
    num_tref: int
    num_bins: int
    num_knn: int
    smooth_weight: float
    sp: int
    norm_l2: bool
    dist_l1: bool
    scale_by_dt: bool
    mask_border: bool
    polarity_split: bool
    scheme_iwd: bool
    smooth_on_next: bool
    variance: bool = False
    atomic_path: bool = False

    def flags(self):
        f = 0
        f |= C.F_SCALE_BY_DT if self.scale_by_dt else 0
        f |= C.F_MASK_BORDER if self.mask_border else 0
        f |= C.F_POLARITY_SPLIT if self.polarity_split else 0
        f |= C.F_NORM_L2 if self.norm_l2 else 0
        f |= C.F_OBJ_VARIANCE if self.variance else 0
        f |= C.F_DIST_L1 if self.dist_l1 else 0
        f |= C.F_SCHEME_IWD if self.scheme_iwd else 0
        f |= C.F_WANT_NEXT if (self.smooth_on_next and self.smooth_weight > 0) else 0
        f |= C.F_ATOMIC_PATH if self.atomic_path else 0
        return f

    @property
    def lut_grid(self):
        h, w = self.image_shape
        return (h + self.sp - 1) // self.sp, (w + self.sp - 1) // self.sp


class StageTimer:
    """Optional per-stage HIP-event timing (bench.py): events are recorded on the stream the
    kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = {}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {'launches': len(ms), 'avg_us': 1e3 * sum(ms) / max(len(ms), 1)}
        return out


STAGE_TIMER = None   # set to a StageTimer to time every C-ABI call


_KERNEL_TIMERS = [0]


def kernel_timer_on():
    """True inside `with KernelTimer()`: the library brackets every launch with HIP events then, which must not fall into a graph
    capture -- and a replayed graph would show the timer nothing."""
    return _KERNEL_TIMERS[0] > 0


class KernelTimer:
    """Per-KERNEL HIP-event timing by the library itself (mpc_profile_start / mpc_profile_stop: two events around every
    launch, on the launch stream).  `with KernelTimer() as kt: ...steps...` then kt.summary() ->
    {kernel name: {'launches': n, 'avg_us': mean duration of a launch, 'total_us': sum}}.  Diagnostics (bench.py)."""

    def __enter__(self):
        C.lib().mpc_profile_start()
        _KERNEL_TIMERS[0] += 1
        self.records = None
        return self

    def __exit__(self, *exc):
        _KERNEL_TIMERS[0] -= 1
        torch.cuda.synchronize()
        cap = 1 << 16
        names = ctypes.create_string_buffer(cap * 48)
        ms = (ctypes.c_float * cap)()
        n = int(C.lib().mpc_profile_stop(names, len(names), ms, cap))
        out = {}
        for nm, t in zip(names.value.decode().split('\n')[:n], ms[:n]):
            k = nm.strip().lstrip('(').split('<')[0].rstrip(')').strip()
            r = out.setdefault(k, {'launches': 0, 'total_us': 0.0})
            r['launches'] += 1
            r['total_us'] += 1e3 * t
        for r in out.values():
            r['avg_us'] = r['total_us'] / r['launches']
        self.records = out
        return False

    def summary(self):
        return self.records or {}


class _stage:
    def __init__(self, name, device):
        self.name, self.device = name, device

    def __enter__(self):
        # the library launches on the CURRENT HIP device: make the tensors' device current for the call (tensors on
        # cuda:1 while cuda:0 is current would otherwise launch in the wrong device context).  The usual case -- it is
        # current already -- skips the guard object (~10 us of host time per call, a B = 1 step is host bound)
        idx = self.device.index
        if idx is None or idx == torch.cuda.current_device():
            self.guard = None
        else:
            self.guard = torch.cuda.device(self.device)
            self.guard.__enter__()
        if STAGE_TIMER is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if STAGE_TIMER is not None:
            self.b.record(torch.cuda.current_stream(self.device))
            STAGE_TIMER.records.setdefault(self.name, []).append((self.a, self.b))
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a GPU tensor: the CMax path runs in libmpcmax.so (HIP, '
                           f'gfx950) and has no CPU fallback (got device {t.device})')


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(device):
    """Handle of torch's CURRENT stream on `device` (the library launches on it)."""
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))        # ~1 us; the Stream object below costs ~15
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_shape(cfg: PathConfig, B, M, Mp, n, extra_flags=0, K=None):
    hq, wq = cfg.lut_grid
    return C.Shape(B=B, M=M, Mp=Mp, nb=cfg.num_bins, T=cfg.num_tref, H=cfg.image_shape[0],
                   W=cfg.image_shape[1], sp=cfg.sp, hq=hq, wq=wq, n=n,
                   K=cfg.num_knn if K is None else K, flags=cfg.flags() | extra_flags)


def alloc_workspace(shape: C.Shape, device) -> torch.Tensor:
    # (sized under the device the kernels will run on: the layout follows its CU count, api.hip: mpc_layout)
    with torch.cuda.device(device):
        nbytes = C.lib().mpc_workspace_bytes(ctypes.byref(shape))
    if nbytes < 0:
        C.check(int(nbytes), 'mpc_workspace_bytes')
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------
# stage wrappers (no autograd)
# ------------------------------------------------------------------------------------------
def knn_lut_fwd(cfg, shape, traj, ws, want_idx=False):
    dev = traj.device
    B, nb, T, K = shape.B, shape.nb, shape.T, shape.K
    hq, wq = shape.hq, shape.wq
    Q = hq * wq
    flow_lut = torch.empty((B, nb, hq, wq, T, 2), dtype=torch.float32, device=dev)
    flow_next = None
    if shape.flags & C.F_WANT_NEXT:
        flow_next = torch.empty((B, max(nb - 1, 0), hq, wq, 1, 2), dtype=torch.float32, device=dev)
    state = torch.empty(int(C.lib().mpc_knn_state_floats(ctypes.byref(shape))), dtype=torch.float32, device=dev)
    idx = torch.empty((B, nb, Q, K), dtype=torch.int32, device=dev) if want_idx else None
    with _stage('mpc_knn_lut_fwd', dev):
        C.check(C.lib().mpc_knn_lut_fwd(ctypes.byref(shape), _ptr(traj), _ptr(flow_lut), _ptr(flow_next),
                                        _ptr(state), _ptr(idx), _ptr(ws), _stream(dev)), 'mpc_knn_lut_fwd')
    return flow_lut, flow_next, state, idx


def knn_lut_bwd(shape, traj, g_lut, g_next, state, ws):
    g_traj = torch.empty_like(traj)
    with _stage('mpc_knn_lut_bwd', traj.device):
        C.check(C.lib().mpc_knn_lut_bwd(ctypes.byref(shape), _ptr(traj), _ptr(g_lut), _ptr(g_next),
                                        _ptr(state), _ptr(g_traj), _ptr(ws), _stream(traj.device)),
                'mpc_knn_lut_bwd')
    return g_traj


def event_splat_fwd(shape, events, flow_lut, t_ref, ws):
    dev = events.device
    P = 2 if shape.flags & C.F_POLARITY_SPLIT else 1
    raw = torch.empty((shape.B * shape.T, P, shape.H, shape.W), dtype=torch.float32, device=dev)
    with _stage('mpc_event_splat_fwd', dev):
        C.check(C.lib().mpc_event_splat_fwd(ctypes.byref(shape), _ptr(events), _ptr(flow_lut), _ptr(t_ref),
                                            _ptr(raw), _ptr(ws), _stream(dev)), 'mpc_event_splat_fwd')
    return raw


def event_splat_fwd_fixed(shape, events, flow_lut, t_ref, ws):
    """mpc_event_splat_fwd_fixed: the raw IWE of these event rows as Q33.30 accumulators (int64), for an exact sum over shards."""
    dev = events.device
    P = 2 if shape.flags & C.F_POLARITY_SPLIT else 1
    fixed = torch.empty((shape.B * shape.T, P, shape.H, shape.W), dtype=torch.int64, device=dev)
    with _stage('mpc_event_splat_fwd_fixed', dev):
        C.check(C.lib().mpc_event_splat_fwd_fixed(ctypes.byref(shape), _ptr(events), _ptr(flow_lut), _ptr(t_ref),
                                                  _ptr(fixed), _ptr(ws), _stream(dev)), 'mpc_event_splat_fwd_fixed')
    return fixed


def iwe_from_fixed(fixed):
    raw = torch.empty(fixed.shape, dtype=torch.float32, device=fixed.device)
    with _stage('mpc_iwe_from_fixed', fixed.device):
        C.check(C.lib().mpc_iwe_from_fixed(_ptr(fixed), _ptr(raw), fixed.numel(), _stream(fixed.device)), 'mpc_iwe_from_fixed')
    return raw


def contrast_fwd(shape, raw, ws, want_grad):
    blur = torch.empty_like(raw)
    gimg = torch.empty_like(raw) if want_grad else None
    with _stage('mpc_contrast_fwd', raw.device):
        C.check(C.lib().mpc_contrast_fwd(ctypes.byref(shape), _ptr(raw), _ptr(blur), _ptr(gimg), _ptr(ws),
                                         _stream(raw.device)), 'mpc_contrast_fwd')
    return blur, gimg


def lut_smooth(shape, field, nimg, Cch, weight, ws, want_grad):
    g = torch.empty_like(field) if want_grad else None
    with _stage('mpc_lut_smooth', field.device):
        C.check(C.lib().mpc_lut_smooth(ctypes.byref(shape), _ptr(field), nimg, Cch, float(weight), _ptr(g),
                                       _ptr(ws), _stream(field.device)), 'mpc_lut_smooth')
    return g


def finalize(shape, smooth_nimg, smooth_C, weight, ws, device):
    scal = torch.empty(C.SCAL_COUNT, dtype=torch.float32, device=device)
    with _stage('mpc_finalize', device):
        C.check(C.lib().mpc_finalize(ctypes.byref(shape), smooth_nimg, smooth_C, float(weight), _ptr(scal),
                                     _ptr(ws), _stream(device)), 'mpc_finalize')
    return scal


def event_splat_bwd(shape, events, flow_lut, t_ref, gimg, scal, grad_out, g_lut, add_term, ws, offs=None):
    with _stage('mpc_event_splat_bwd', events.device):
        C.check(C.lib().mpc_event_splat_bwd_ordered(ctypes.byref(shape), _ptr(events), _ptr(offs), _ptr(flow_lut), _ptr(t_ref),
                                            _ptr(gimg), _ptr(scal), _ptr(grad_out), _ptr(g_lut),
                                            _ptr(add_term), _ptr(ws), _stream(events.device)),
                'mpc_event_splat_bwd')
    return g_lut


def scale(x, a):
    y = torch.empty_like(x)
    with _stage('mpc_scale', x.device):
        C.check(C.lib().mpc_scale(_ptr(x), _ptr(a), _ptr(y), x.numel(), _stream(x.device)), 'mpc_scale')
    return y


FUSED_CALLS = True    # FocusCalcFn: mpc_focus_fwd / mpc_focus_bwd (two C-ABI calls per step) instead of one call per stage


def _vp(t):
    return None if t is None else t.data_ptr()


class _Plan:
    """What one (configuration, shape) needs on the host and does not change from step to step: the shape struct, the
    workspace size and the layout of ONE float buffer that holds everything the forward leaves for the backward (LUT,
    flow_next, KNN state, smoothness gradient, raw IWE, adjoint image, scalars).  A B = 1 step is host bound: building these
    per step (two ctypes size queries, ~10 torch.empty, two struct fills) cost ~100 us of Python per direction."""

    def __init__(self, cfg, B, M, Mp, n, need_grad, has_offs):
        self.cfg = cfg
        # (bucket-ordered events: the backward reads the rows themselves -- no record region in the workspace either)
        self.shape = make_shape(cfg, B, M, Mp, n, extra_flags=0 if (need_grad and not has_offs) else C.F_NO_BWD_RECORDS)
        self.shape_ref = ctypes.byref(self.shape)
        self.need_grad = need_grad
        nbytes = C.lib().mpc_workspace_bytes(self.shape_ref)
        if nbytes < 0:
            C.check(int(nbytes), 'mpc_workspace_bytes')
        self.ws_bytes = max(int(nbytes), 256)
        s = self.shape
        self.P = 2 if s.flags & C.F_POLARITY_SPLIT else 1
        self.lut_shape = (B, s.nb, s.hq, s.wq, s.T, 2)
        n_lut = B * s.nb * s.hq * s.wq * s.T * 2
        n_nxt = B * max(s.nb - 1, 0) * s.hq * s.wq * 2 if s.flags & C.F_WANT_NEXT else 0
        n_state = int(C.lib().mpc_knn_state_floats(self.shape_ref))
        n_gf = 0
        if cfg.smooth_weight > 0 and need_grad:
            n_gf = n_nxt if cfg.smooth_on_next else n_lut
        n_img = B * s.T * self.P * s.H * s.W
        self.img_shape = (B * s.T, self.P, s.H, s.W)
        off = 0
        self.o = {}
        for name, cnt in (('lut', n_lut), ('nxt', n_nxt), ('state', n_state), ('gf', n_gf), ('raw', n_img),
                          ('gimg', n_img if need_grad else 0), ('scal', C.SCAL_COUNT)):
            self.o[name] = (off, cnt)
            off += (cnt + 63) // 64 * 64          # 256-byte aligned pieces
        self.buf_floats = max(off, 64)
        # backward scratch: dL/dLUT and (smoothness on flow_to_next) the scaled dL/dflow_next
        self.n_glut = n_lut
        self.n_gnext = n_gf if (cfg.smooth_on_next and n_gf) else 0
        self.smooth_weight = float(cfg.smooth_weight)

    def io(self, base, traj, ev, tr, blur, offs, scal_out=None):
        def at(name):
            off, cnt = self.o[name]
            return base + 4 * off if cnt else None
        return C.FocusBuffers(traj=traj, events=ev, t_ref=tr, flow_lut=at('lut'), flow_next=at('nxt'), knn_state=at('state'),
                              smooth_grad=at('gf'), iwe_raw=at('raw'), iwe_blur=blur, grad_iwe=at('gimg'), scal=at('scal'),
                              smooth_weight=self.smooth_weight, event_offsets=offs, scal_out=scal_out)

    def view(self, buf, name):
        off, cnt = self.o[name]
        return buf[off:off + cnt] if cnt else None


_PLANS = {}


def _plan(cfg, B, M, Mp, n, need_grad, has_offs, dev):
    # (per device: the workspace layout follows the CU count of the device the step runs on, api.hip: mpc_layout)
    key = (cfg, B, M, Mp, n, need_grad, has_offs, dev.index)
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) > 256:
            _PLANS.clear()
        with torch.cuda.device(dev):
            p = _PLANS[key] = _Plan(cfg, B, M, Mp, n, need_grad, has_offs)
    return p


def _focus_fwd_call(p, dev, traj, ev, tr, buf, blur, ws, offs, out3=None):
    io = p.io(buf.data_ptr(), traj.data_ptr(), ev.data_ptr(), tr.data_ptr(), blur.data_ptr(), _vp(offs), _vp(out3))
    with _stage('mpc_focus_fwd', dev):
        C.check(C.lib().mpc_focus_fwd(p.shape_ref, ctypes.byref(io), ctypes.c_void_p(ws.data_ptr()), _stream(dev)), 'mpc_focus_fwd')


def _focus_bwd_call(p, dev, traj, ev, tr, buf, ws, offs, grad_out, scratch, g_traj):
    io = p.io(buf.data_ptr(), traj.data_ptr(), ev.data_ptr(), tr.data_ptr(), None, _vp(offs))
    io.flow_next = None
    io.iwe_raw = None
    g_lut = scratch.data_ptr()
    g_next = g_lut + 4 * ((p.n_glut + 63) // 64 * 64) if p.n_gnext else None
    with _stage('mpc_focus_bwd', dev):
        C.check(C.lib().mpc_focus_bwd(p.shape_ref, ctypes.byref(io), ctypes.c_void_p(grad_out.data_ptr()), ctypes.c_void_p(g_lut),
                                      ctypes.c_void_p(g_next) if g_next else None, ctypes.c_void_p(g_traj.data_ptr()),
                                      ctypes.c_void_p(ws.data_ptr()), _stream(dev)), 'mpc_focus_bwd')


def _bwd_scratch_floats(p):
    return (p.n_glut + 63) // 64 * 64 + p.n_gnext + 64


def event_bucket_order(cfg: PathConfig, events, num_pos):
    """SURVEY.md 8f-1, layout half (mpc_event_bucket_order): the rows of each polarity block of `events` [B, M, 6]
    ordered by (time bin, LUT strip), and the offsets table [B, 2, nb * strips + 1] int32.  The ordered tensor gives
    the same loss and gradient bit for bit; handed to FocusCalcFn together with the table, the step skips one 16-byte
    record per event in each direction.  Done once per batch, outside the training step (it belongs to ingest)."""
    B, M, Mp = _check_events(events, cfg, num_pos)
    dev = events.device
    ev = _f32c(events.detach())
    shape = make_shape(cfg, B, M, Mp, 1)
    ncs = _lut_strips(shape, dev)
    if ncs <= 0:
        raise ValueError('no bucketed event layout for this configuration (num_tref > 1 or the atomic debugging path)')
    out = torch.empty_like(ev)
    offs = torch.empty((B, 2, cfg.num_bins * ncs + 1), dtype=torch.int32, device=dev)
    nbytes = int(C.lib().mpc_event_order_workspace_bytes(ctypes.byref(shape)))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    with _stage('mpc_event_bucket_order', dev):
        C.check(C.lib().mpc_event_bucket_order(ctypes.byref(shape), _ptr(ev), _ptr(out), _ptr(offs), _ptr(ws), _stream(dev)),
                'mpc_event_bucket_order')
    return out, offs


_NCS_CACHE = {}


def _lut_strips(shape, device):
    """LUT strips of the backward buckets for `shape` on `device` (the count follows the device's layout, so the query runs
    under that device and the cache is keyed on it)."""
    key = (shape.B, shape.M, shape.Mp, shape.nb, shape.T, shape.H, shape.W, shape.sp, shape.flags, device.index)
    ncs = _NCS_CACHE.get(key)
    if ncs is None:
        with torch.cuda.device(device):
            ncs = _NCS_CACHE[key] = int(C.lib().mpc_event_lut_strips(ctypes.byref(shape)))
    return ncs


def _check_offsets(offs, cfg, shape, device):
    if offs is None:
        return None
    _require_gpu(offs, "batch['event_offsets']")
    ncs = _lut_strips(shape, device)
    want = (shape.B, 2, cfg.num_bins * ncs + 1)
    if ncs <= 0 or offs.dtype != torch.int32 or tuple(offs.shape) != want or offs.device != device:
        raise ValueError(f"event_offsets must be int32 {want} on {device} (from event_bucket_order with this configuration), "
                         f"got {offs.dtype} {tuple(offs.shape)} on {offs.device}")
    return offs.contiguous()


def _check_events(events, cfg, num_pos):
    _require_gpu(events, "batch['events']")
    if events.dim() != 3 or events.shape[-1] != 6:
        raise ValueError(f"events must be [B, M, 6], got {tuple(events.shape)}")
    B, M, _ = events.shape
    if cfg.polarity_split:
        if not (0 <= num_pos <= M):
            raise ValueError(f'num_pos_events={num_pos} outside [0, {M}]')
        Mp = int(num_pos)
    else:
        Mp = M
    return B, M, Mp


# ------------------------------------------------------------------------------------------
# autograd: the whole loss  trajectories -> (loss, focus, smooth, iwes)
# ------------------------------------------------------------------------------------------
def _calc_inputs(trajectories, events, t_ref, cfg, num_pos):
    _require_gpu(trajectories, 'trajectories')
    B, M, Mp = _check_events(events, cfg, num_pos)
    T, nb = cfg.num_tref, cfg.num_bins
    if trajectories.dim() != 4 or trajectories.shape[0] != B or trajectories.shape[1] != T + nb \
            or trajectories.shape[3] != 2:
        raise ValueError(f'trajectories must be [B={B}, {T + nb}, n, 2], got {tuple(trajectories.shape)}')
    dev = trajectories.device
    traj = _f32c(trajectories.detach())
    ev = _f32c(events.detach())
    tr = _f32c(t_ref.detach().to(dev))
    return B, M, Mp, traj.shape[2], dev, traj, ev, tr


class FocusCalcFn(torch.autograd.Function):
    """FocusLoss.calc (reference focus.py:66-113) as one differentiable op: KNN LUT -> smoothness
    -> warp + vote -> blur + objective in forward, hand-derived backward to `trajectories`."""

    @staticmethod
    def forward(ctx, trajectories, events, t_ref, cfg: PathConfig, num_pos: int, event_offsets=None):
        B, M, Mp, n, dev, traj, ev, tr = _calc_inputs(trajectories, events, t_ref, cfg, num_pos)
        T, nb = cfg.num_tref, cfg.num_bins
        need_grad = trajectories.requires_grad
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)      # no zero-filled [B,P,H,W] gradient for the detached outputs

        if STAGE_TIMER is None and FUSED_CALLS:
            # one C-ABI call for the whole forward (mpc_focus_fwd issues the same launches); per-shape host state is cached
            p = _plan(cfg, B, M, Mp, n, need_grad, event_offsets is not None, dev)
            offs = _check_offsets(event_offsets, cfg, p.shape, dev)
            ws = torch.empty(p.ws_bytes, dtype=torch.uint8, device=dev)
            buf = torch.empty(p.buf_floats, dtype=torch.float32, device=dev)
            blur = torch.empty(p.img_shape, dtype=torch.float32, device=dev)
            # (the outputs must not alias the saved scalars: the library writes loss / focus / smooth a second time into `out` --
            # io.scal_out -- instead of the host cloning them with a kernel of its own)
            out = torch.empty(3, dtype=torch.float32, device=dev)
            _focus_fwd_call(p, dev, traj, ev, tr, buf, blur, ws, offs, out)
            ctx.plan, ctx.ws, ctx.offs = p, ws, offs
            ctx.save_for_backward(traj, ev, tr, buf)
            loss, focus, smooth = out[C.SCAL_LOSS], out[C.SCAL_FOCUS], out[C.SCAL_SMOOTH]
            ctx.mark_non_differentiable(focus, smooth, blur)
            return loss, focus, smooth, blur

        shape = make_shape(cfg, B, M, Mp, n, extra_flags=0 if need_grad else C.F_NO_BWD_RECORDS)
        ws = alloc_workspace(shape, dev)
        offs = _check_offsets(event_offsets, cfg, shape, dev)
        flow_lut, flow_next, state, _ = knn_lut_fwd(cfg, shape, traj, ws)
        g_field = None
        s_nimg = s_C = 0
        if cfg.smooth_weight > 0:
            if cfg.smooth_on_next:
                field, s_nimg, s_C = flow_next, B * (nb - 1), 2
            else:
                field, s_nimg, s_C = flow_lut, B * nb, 2 * T
            if s_nimg > 0:
                g_field = lut_smooth(shape, field, s_nimg, s_C, cfg.smooth_weight, ws, need_grad)
        fshape = shape if offs is None else make_shape(cfg, B, M, Mp, n, extra_flags=C.F_NO_BWD_RECORDS)
        raw = event_splat_fwd(fshape, ev, flow_lut, tr, ws)
        blur, gimg = contrast_fwd(shape, raw, ws, need_grad)
        scal = finalize(shape, s_nimg, s_C, cfg.smooth_weight, ws, dev)
        ctx.plan = None
        ctx.shape = shape
        ctx.offs = offs
        ctx.ws = ws
        ctx.save_for_backward(traj, ev, tr, flow_lut, state, gimg, scal, g_field)
        out = scal[:3].clone()                # one tiny copy: the outputs must not alias the saved scalars
        loss, focus, smooth = out[C.SCAL_LOSS], out[C.SCAL_FOCUS], out[C.SCAL_SMOOTH]
        ctx.mark_non_differentiable(focus, smooth, blur)
        return loss, focus, smooth, blur

    @staticmethod
    def backward(ctx, g_loss, g_focus, g_smooth, g_iwes):
        if g_loss is None:
            return None, None, None, None, None, None
        g = _f32c(g_loss.reshape(1))
        cfg, ws, offs = ctx.cfg, ctx.ws, ctx.offs
        if ctx.plan is not None:
            p = ctx.plan
            traj, ev, tr, buf = ctx.saved_tensors
            scratch = torch.empty(_bwd_scratch_floats(p), dtype=torch.float32, device=traj.device)
            g_traj = torch.empty_like(traj)
            _focus_bwd_call(p, traj.device, traj, ev, tr, buf, ws, offs, g, scratch, g_traj)
            return g_traj, None, None, None, None, None
        shape = ctx.shape
        traj, ev, tr, flow_lut, state, gimg, scal, g_field = ctx.saved_tensors
        g_next = None
        g_lut = torch.empty_like(flow_lut)
        if g_field is not None and not cfg.smooth_on_next:
            event_splat_bwd(shape, ev, flow_lut, tr, gimg, scal, g, g_lut, g_field, ws, offs)   # smoothness folded in
        else:
            event_splat_bwd(shape, ev, flow_lut, tr, gimg, scal, g, g_lut, None, ws, offs)
            if g_field is not None:
                g_next = scale(g_field, g)
        g_traj = knn_lut_bwd(shape, traj, g_lut, g_next, state, ws)
        return g_traj, None, None, None, None, None


class PyramidFocusFn(torch.autograd.Function):
    """UNPINNED EXTENSION, default off (FocusLoss(pyramid_levels=L > 1)): FocusLoss.calc with an IWE pyramid -- BASELINE.json's
    configs[2] names one, the reference has none (focus.py:90-91 evaluates one scale), so this is a definition, not a port:
    level l + 1 is the 2x2 average of the RAW level-l IWE, every level goes through the reference's own blur + gradient-magnitude
    objective (mpc_contrast_fwd on a shape of that level's size), and the focus term is the sum of the levels' 1 / val.
    Stage calls (no fused path); `iwes` are the blurred level-0 images."""

    @staticmethod
    def forward(ctx, trajectories, events, t_ref, cfg: PathConfig, num_pos: int, levels: int):
        B, M, Mp, n, dev, traj, ev, tr = _calc_inputs(trajectories, events, t_ref, cfg, num_pos)
        H, W = cfg.image_shape
        if H % (1 << (levels - 1)) or W % (1 << (levels - 1)) or min(H, W) >> (levels - 1) < 3:
            raise ValueError(f'pyramid_levels={levels}: image shape {H}x{W} must be divisible by {1 << (levels - 1)} and stay >= 3 pixels')
        need_grad = trajectories.requires_grad
        shape = make_shape(cfg, B, M, Mp, n, extra_flags=0 if need_grad else C.F_NO_BWD_RECORDS)
        ws = alloc_workspace(shape, dev)
        flow_lut, flow_next, state, _ = knn_lut_fwd(cfg, shape, traj, ws)
        g_field, s_nimg, s_C = None, 0, 0
        if cfg.smooth_weight > 0:
            field, s_nimg, s_C = (flow_next, B * (cfg.num_bins - 1), 2) if cfg.smooth_on_next else (flow_lut, B * cfg.num_bins, 2 * cfg.num_tref)
            if s_nimg > 0:
                g_field = lut_smooth(shape, field, s_nimg, s_C, cfg.smooth_weight, ws, need_grad)
        raw = event_splat_fwd(shape, ev, flow_lut, tr, ws)
        nimg = raw.shape[0] * raw.shape[1]
        blur0, gimg0 = contrast_fwd(shape, raw, ws, need_grad)
        scal0 = finalize(shape, s_nimg, s_C, cfg.smooth_weight, ws, dev)
        gimgs, scals, sizes = [gimg0], [scal0], [(H, W)]
        focus_total = scal0[C.SCAL_FOCUS].clone()
        cur, h, w = raw, H, W
        for lv in range(1, levels):
            nxt = torch.empty((raw.shape[0], raw.shape[1], h // 2, w // 2), dtype=torch.float32, device=dev)
            with _stage('mpc_pool2_fwd', dev):
                C.check(C.lib().mpc_pool2_fwd(_ptr(cur), _ptr(nxt), nimg, h, w, _stream(dev)), 'mpc_pool2_fwd')
            h, w = h // 2, w // 2
            cfg_l = PathConfig(**{**cfg.__dict__, 'image_shape': (h, w), 'smooth_weight': 0.0})
            sh_l = make_shape(cfg_l, B, 0, 0, 0, K=0)
            ws_l = alloc_workspace(sh_l, dev)
            _, g_l = contrast_fwd(sh_l, nxt, ws_l, need_grad)
            sc_l = finalize(sh_l, 0, 0, 0.0, ws_l, dev)
            gimgs.append(g_l); scals.append(sc_l); sizes.append((h, w))
            focus_total = focus_total + sc_l[C.SCAL_FOCUS]
            cur = nxt
        ctx.cfg, ctx.shape, ctx.ws, ctx.nimg, ctx.sizes = cfg, shape, ws, nimg, sizes
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(traj, ev, tr, flow_lut, state, g_field, *scals, *([g for g in gimgs] if need_grad else []))
        ctx.levels = levels
        smooth = scal0[C.SCAL_SMOOTH].clone()
        loss = focus_total + smooth
        ctx.mark_non_differentiable(focus_total, smooth, blur0)
        return loss, focus_total, smooth, blur0

    @staticmethod
    def backward(ctx, g_loss, g_focus, g_smooth, g_iwes):
        if g_loss is None:
            return None, None, None, None, None, None
        cfg, shape, ws, L = ctx.cfg, ctx.shape, ctx.ws, ctx.levels
        saved = ctx.saved_tensors
        traj, ev, tr, flow_lut, state, g_field = saved[:6]
        scals, gimgs = saved[6:6 + L], list(saved[6 + L:6 + 2 * L])
        # (the coarser levels are ADDED into the finer adjoint images below: on copies, so that a second backward of a retained
        # graph starts from the saved images again)
        gimgs = [gi.clone() for gi in gimgs[:-1]] + gimgs[-1:]
        dev = traj.device
        g = _f32c(g_loss.reshape(1))
        # the coarser levels' adjoint images into the finer ones, each in units of its own level's 1 / val^2 coefficient
        for lv in range(L - 1, 0, -1):
            h, w = ctx.sizes[lv - 1]
            with _stage('mpc_pool2_bwd_add', dev):
                C.check(C.lib().mpc_pool2_bwd_add(_ptr(gimgs[lv]), ctypes.c_void_p(scals[lv].data_ptr() + 4 * C.SCAL_GCOEF), _ptr(gimgs[lv - 1]),
                                                  ctypes.c_void_p(scals[lv - 1].data_ptr() + 4 * C.SCAL_GCOEF), ctx.nimg, h, w, _stream(dev)),
                        'mpc_pool2_bwd_add')
        g_lut = torch.empty_like(flow_lut)
        g_next = None
        if g_field is not None and not cfg.smooth_on_next:
            event_splat_bwd(shape, ev, flow_lut, tr, gimgs[0], scals[0], g, g_lut, g_field, ws)
        else:
            event_splat_bwd(shape, ev, flow_lut, tr, gimgs[0], scals[0], g, g_lut, None, ws)
            if g_field is not None:
                g_next = scale(g_field, g)
        return knn_lut_bwd(shape, traj, g_lut, g_next, state, ws), None, None, None, None, None


class _Marker:
    """Lives as long as the autograd context it is attached to (StaticFocusCalcFn, automatic mode)."""
    __slots__ = ('__weakref__',)


class StaticFocusPlan:
    """FocusLoss(static_shapes=True): `calc` + backward of ONE shape captured once into two HIP graphs (forward; backward)
    over buffers that never move, and replayed from then on -- what a B = 1 step costs on the host drops from two eager
    C-ABI calls with their allocations to two graph launches and three small copies.  The reference has no counterpart
    (PyTorch eager); the caller is unchanged (src/modules/trajectory_net.py:152-158).  Consequences of buffers that never
    move, all checked or documented: `misc_metadata['iwes']` is valid until the next `calc` of this shape; a backward must
    belong to the latest `calc` (checked); inputs are COPIED into the captured buffers every step (events only when the
    caller hands over a different tensor than last time)."""

    def __init__(self, cfg, B, M, Mp, n, dev, need_grad, traj, ev, tr, offs):
        p = self.plan = _plan(cfg, B, M, Mp, n, need_grad, offs is not None, dev)
        self.dev = dev
        # the offsets table of bucket-ordered events is an INPUT like the events: ingest hands over a fresh tensor with every
        # batch, so the plan owns a buffer of its own and the table is copied in every step (a few KB)
        self.offs = None
        if offs is not None:
            self.offs = torch.empty_like(_check_offsets(offs, cfg, p.shape, dev))
            self.offs.copy_(offs)
        self.traj, self.ev, self.tr = torch.empty_like(traj), torch.empty_like(ev), torch.empty_like(tr)
        self.ev_src, self.ev_version = None, -1
        self.ws = torch.empty(p.ws_bytes, dtype=torch.uint8, device=dev)
        self.buf = torch.empty(p.buf_floats, dtype=torch.float32, device=dev)
        self.blur = torch.empty(p.img_shape, dtype=torch.float32, device=dev)
        self.gout = torch.ones(1, dtype=torch.float32, device=dev)
        self.scratch = torch.empty(_bwd_scratch_floats(p), dtype=torch.float32, device=dev) if need_grad else None
        self.g_traj = torch.empty_like(traj) if need_grad else None
        self.generation = 0
        self.pending = None              # weak reference to the marker of the latest calc that still waits for its backward (automatic mode)
        self.traj.copy_(traj); self.ev.copy_(ev); self.tr.copy_(tr)
        # one eager run on a side stream first: the library's one-time set-up must not fall into a capture
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._fwd()
            if need_grad:
                self._bwd()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # capture_error_mode='thread_local': only THIS thread's calls are held to the capture rules.  In the reference's training
        # process other threads use the device meanwhile (the DataLoader's pin_memory thread, src/modules/data_loading.py:141-142;
        # DDP's side streams and the RCCL watchdog, scripts/flow_training.py:125-130): in the default global mode an allocation of
        # theirs inside this window would invalidate the capture.
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, capture_error_mode='thread_local'):
            self._fwd()
        self.g_bwd = None
        if need_grad:
            self.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_bwd, capture_error_mode='thread_local'):
                self._bwd()

    def _fwd(self):
        _focus_fwd_call(self.plan, self.dev, self.traj, self.ev, self.tr, self.buf, self.blur, self.ws, self.offs)

    def _bwd(self):
        _focus_bwd_call(self.plan, self.dev, self.traj, self.ev, self.tr, self.buf, self.ws, self.offs, self.gout, self.scratch, self.g_traj)


class AutoPlanFailed(RuntimeError):
    """Automatic static shapes: the plan of a shape could not be captured (FocusLoss.calc then stays on the eager path)."""


class StaticFocusCalcFn(torch.autograd.Function):
    """FocusCalcFn replayed from the HIP graphs of a StaticFocusPlan."""

    @staticmethod
    def plan_key(trajectories, events, cfg, num_pos, event_offsets):
        """The key of the plan a calc of these inputs would use (shape only: every input is copied in)."""
        B, M = int(events.shape[0]), int(events.shape[1])
        Mp = int(num_pos) if cfg.polarity_split else M
        return (B, M, Mp, int(trajectories.shape[2]), bool(trajectories.requires_grad), trajectories.device.index, event_offsets is not None)

    @staticmethod
    def plan_busy(plans, key):
        """True while the latest calc of this plan still waits for its backward (its autograd context is alive and has not run):
        the captured buffers hold that step, another calc of the same shape must not replay over them."""
        sp = plans.get(key)
        return sp is not None and sp.pending is not None and sp.pending() is not None

    @staticmethod
    def forward(ctx, trajectories, events, t_ref, cfg: PathConfig, num_pos: int, event_offsets, plans: dict, auto: bool = False):
        B, M, Mp, n, dev, traj, ev, tr = _calc_inputs(trajectories, events, t_ref, cfg, num_pos)
        need_grad = trajectories.requires_grad
        key = (B, M, Mp, n, need_grad, dev.index, event_offsets is not None)      # shape only: every input is copied in
        sp = plans.get(key)
        if sp is None:
            if len(plans) >= 8:              # every shape holds its buffers: keep the set small
                plans.pop(next(iter(plans)))
            if auto:
                # nobody asked for a capture: whatever goes wrong while the plan is built (a capture error, no memory for the
                # plan's buffers) must not reach the training step -- the caller takes the eager path for this shape from now on
                try:
                    sp = StaticFocusPlan(cfg, B, M, Mp, n, dev, need_grad, traj, ev, tr, event_offsets)
                except Exception as e:          # noqa: BLE001
                    raise AutoPlanFailed(f'{type(e).__name__}: {e}') from e
                plans[key] = sp
            else:
                sp = plans[key] = StaticFocusPlan(cfg, B, M, Mp, n, dev, need_grad, traj, ev, tr, event_offsets)
        sp.traj.copy_(traj)
        if sp.ev_src is None or sp.ev_src() is not events or events._version != sp.ev_version:
            sp.ev.copy_(ev)                 # a new batch (or one modified in place): copied once
            sp.ev_src, sp.ev_version = weakref.ref(events), events._version
        sp.tr.copy_(tr)
        if event_offsets is not None:
            sp.offs.copy_(_check_offsets(event_offsets, cfg, sp.plan.shape, dev))
        sp.generation += 1
        ctx.sp, ctx.gen = sp, sp.generation
        ctx.set_materialize_grads(False)
        sp.g_fwd.replay()
        o = sp.plan.o['scal'][0]
        out = sp.buf[o:o + 3].clone()
        loss, focus, smooth = out[C.SCAL_LOSS], out[C.SCAL_FOCUS], out[C.SCAL_SMOOTH]
        ctx.mark_non_differentiable(focus, smooth)
        # (the automatic mode keeps the reference's semantics: the images are the caller's own copy, and the plan is marked busy until
        # this step's backward has run or its graph is dropped -- FocusLoss.calc then takes the eager path for a second calc)
        sp.pending = None
        ctx.redo = None
        if auto and need_grad:
            ctx.marker = _Marker()
            sp.pending = weakref.ref(ctx.marker)
            # (what an eager step needs, should this backward come after the plan has moved on: references, no copies)
            ctx.redo = (traj, ev, tr, cfg, num_pos, event_offsets)
        return loss, focus, smooth, (sp.blur.clone() if auto else sp.blur.detach())

    @staticmethod
    def backward(ctx, g_loss, g_focus, g_smooth, g_iwes):
        if g_loss is None:
            return None, None, None, None, None, None, None, None
        sp = ctx.sp
        if ctx.gen == sp.generation:
            sp.pending = None
        elif ctx.redo is not None:
            # automatic mode, and the captured buffers hold a later calc of this shape (a second backward of a retained graph
            # after the next step): the caller never asked for static shapes, so it gets what the eager path gives -- the
            # step once more, eagerly, from the inputs this context kept
            traj, ev, tr, cfg, num_pos, offs = ctx.redo
            with torch.enable_grad():
                t = traj.detach().requires_grad_(True)
                out = FocusCalcFn.apply(t, ev, tr, cfg, num_pos, offs)
                (g,) = torch.autograd.grad(out[0], t, g_loss.reshape(()).to(out[0].dtype))
            return g, None, None, None, None, None, None, None
        if ctx.gen != sp.generation:
            raise RuntimeError('FocusLoss(static_shapes=True): this backward belongs to an earlier calc() of the same shape; the '
                               'captured buffers hold the latest one (call backward before the next calc, or use static_shapes=False)')
        sp.gout.copy_(g_loss.reshape(1))
        sp.g_bwd.replay()
        return sp.g_traj.clone(), None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------
# autograd: stage-level ops (used by tests, the bench's event-path figure and other callers)
# ------------------------------------------------------------------------------------------
class EventFocusFn(torch.autograd.Function):
    """A6-A9 with the LUT given: flow_lut -> (focus_loss, iwes_blurred, iwes_raw)."""

    @staticmethod
    def forward(ctx, flow_lut, events, t_ref, cfg: PathConfig, num_pos: int):
        _require_gpu(flow_lut, 'flow_lut')
        B, M, Mp = _check_events(events, cfg, num_pos)
        dev = flow_lut.device
        lut = _f32c(flow_lut.detach())
        ev = _f32c(events.detach())
        tr = _f32c(t_ref.detach().to(dev))
        need_grad = flow_lut.requires_grad
        shape = make_shape(cfg, B, M, Mp, 0, K=0, extra_flags=0 if need_grad else C.F_NO_BWD_RECORDS)
        ws = alloc_workspace(shape, dev)
        raw = event_splat_fwd(shape, ev, lut, tr, ws)
        blur, gimg = contrast_fwd(shape, raw, ws, need_grad)
        scal = finalize(shape, 0, 0, 0.0, ws, dev)
        ctx.shape, ctx.ws = shape, ws
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(ev, tr, lut, gimg, scal)
        focus = scal[C.SCAL_FOCUS].clone()
        ctx.mark_non_differentiable(blur, raw)
        return focus, blur, raw

    @staticmethod
    def backward(ctx, g_focus, g_blur, g_raw):
        ev, tr, lut, gimg, scal = ctx.saved_tensors
        if g_focus is None:
            return None, None, None, None, None
        g = _f32c(g_focus.reshape(1))
        g_lut = torch.empty_like(lut)
        event_splat_bwd(ctx.shape, ev, lut, tr, gimg, scal, g, g_lut, None, ctx.ws)
        return g_lut, None, None, None, None


class PerEventBasisFocusFn(torch.autograd.Function):
    """UNPINNED extension (FocusLoss.calc_per_event_basis), fused form: tile coefficients coef_rows [B*hq*wq, 2k] (differentiable),
    events [B, M, 6], phi [B, M, k] = basis(t_ref) - basis(t_event) -> (focus_loss, iwes_blurred).
    Forward: mpc_pe_warp -> mpc_event_splat_fwd (MPC_F_NO_WARP) -> mpc_contrast_fwd -> mpc_finalize; backward: mpc_pe_grad."""

    @staticmethod
    def forward(ctx, coef_rows, events, phi, t_ref, cfg: PathConfig, num_pos: int, offsets=None):
        _require_gpu(coef_rows, 'coef_rows')
        B, M, Mp = _check_events(events, cfg, num_pos)
        dev = coef_rows.device
        cr = _f32c(coef_rows.detach())
        ev = _f32c(events.detach())
        ph = _f32c(phi.detach()) if phi is not None else None       # None: the polynomial basis, worked out in the kernels
        k = int(ph.shape[-1]) if ph is not None else int(coef_rows.shape[1]) // 2
        tr = _f32c(t_ref.detach().to(dev))
        need_grad = coef_rows.requires_grad
        shape = make_shape(cfg, B, M, Mp, 0, K=0, extra_flags=C.F_NO_WARP | C.F_NO_BWD_RECORDS)
        if tuple(cr.shape) != (B * shape.hq * shape.wq, 2 * k) or (ph is not None and tuple(ph.shape) != (B, M, k)) or (ph is None and k > 8):
            raise ValueError(f'coef_rows {tuple(cr.shape)} / phi do not match [B*hq*wq, 2k] / [B, M, k] (k <= 8 without phi)')
        ws = alloc_workspace(shape, dev)
        rows = torch.empty_like(ev)
        if B * M > 0:                 # (an empty tensor has no pointer to hand over)
            with _stage('mpc_pe_warp', dev):
                C.check(C.lib().mpc_pe_warp(ctypes.byref(shape), _ptr(ev), _ptr(cr), _ptr(ph), k, _ptr(tr), _ptr(rows), _stream(dev)), 'mpc_pe_warp')
        raw = event_splat_fwd(shape, rows, None, tr, ws)
        blur, gimg = contrast_fwd(shape, raw, ws, need_grad)
        scal = finalize(shape, 0, 0, 0.0, ws, dev)
        ctx.shape, ctx.k, ctx.n = shape, k, cr.shape[0]
        # bucket-ordered events: the backward accumulates per LUT strip in LDS (no global atomics) where a strip's accumulators fit
        # (the library says whether the ordered backward serves this shape -- one copy of the rule; where it does not, an offsets
        # table is simply not used: the atomic backward needs none)
        with torch.cuda.device(dev):
            ordered = offsets is not None and int(C.lib().mpc_pe_grad_ordered_supported(ctypes.byref(shape), k)) == 1
        offs = _check_offsets(offsets, cfg, shape, dev) if ordered else None
        ctx.has_offs = offs is not None
        ctx.set_materialize_grads(False)
        ctx.has_phi = ph is not None
        # (the table goes through save_for_backward: autograd's version check then catches a refill between forward and backward)
        ctx.save_for_backward(rows, ph if ph is not None else tr, tr, gimg, scal, offs if offs is not None else tr)
        ctx.mark_non_differentiable(blur)
        return scal[C.SCAL_FOCUS].clone(), blur

    @staticmethod
    def backward(ctx, g_focus, g_blur):
        rows, ph, tr, gimg, scal, offs = ctx.saved_tensors
        if not ctx.has_phi:
            ph = None
        if g_focus is None:
            return None, None, None, None, None, None, None
        go = _f32c(g_focus.reshape(1))
        if ctx.has_offs:
            # (a few workgroups per (sample, LUT strip), each with a share of the strip's row ranges and a partial result)
            split = max(1, min(16, 512 // max(1, ctx.shape.B * _lut_strips(ctx.shape, rows.device))))
            gp = torch.empty((split, ctx.n, 2 * ctx.k), dtype=torch.float32, device=rows.device)
            with _stage('mpc_pe_grad_ordered', rows.device):
                rc = C.lib().mpc_pe_grad_ordered(ctypes.byref(ctx.shape), _ptr(rows), _ptr(offs), _ptr(ph), ctx.k, _ptr(tr), _ptr(gimg),
                                                 _ptr(scal), _ptr(go), _ptr(gp), split, _stream(rows.device))
            if rc != C.E_UNSUPPORTED:         # (unsupported after all: the atomic backward below serves every shape)
                C.check(rc, 'mpc_pe_grad_ordered')
                return (gp.sum(0) if split > 1 else gp[0]), None, None, None, None, None, None
        g = torch.empty((ctx.n, 2 * ctx.k), dtype=torch.float32, device=rows.device)
        with _stage('mpc_pe_grad', rows.device):
            C.check(C.lib().mpc_pe_grad(ctypes.byref(ctx.shape), _ptr(rows), _ptr(ph), ctx.k, _ptr(tr), _ptr(gimg), _ptr(scal),
                                        _ptr(go), _ptr(g), _stream(rows.device)), 'mpc_pe_grad')
        return g, None, None, None, None, None, None


class PerEventBasisCalcFn(torch.autograd.Function):
    """UNPINNED extension (FocusLoss.calc_per_event_basis), the whole step as ONE autograd node of library calls (round 6):
    dense coefficient grid -> tile rows -> per-event warp -> vote -> blur + objective [-> smoothness field -> Charbonnier term]
    -> scalars, and back: position gradient per LUT strip -> (+ smoothness adjoint) -> dense gradient.  Rounds 4-5 spelt the dense <->
    per-tile operators and the smoothness field in torch (TileCoeffRowsFn, BasisFieldFn, LutSmoothFn below: kept for the unfused
    cross-check): two dozen operators around 0.30 ms of kernels, and the step was bound by the host (0.74-0.97 ms at the DSEC batch
    shape).  Returns (loss, focus, smooth, iwes_blurred); differentiable w.r.t. coeff_grid through `loss`."""

    @staticmethod
    def forward(ctx, coeff_grid, events, phi, phim, t_ref, cfg: PathConfig, num_pos: int, offsets, k: int, tile: int):
        _require_gpu(coeff_grid, 'coeff_grid')
        B, M, Mp = _check_events(events, cfg, num_pos)
        dev = coeff_grid.device
        cg = _f32c(coeff_grid.detach())
        ev = _f32c(events.detach())
        ph = _f32c(phi.detach()) if phi is not None else None       # None: the polynomial basis, worked out in the kernels
        pm = _f32c(phim.detach()) if phim is not None else None     # [nb, k]: basis(t_ref) - basis(bin mid-times), or None: no smoothness term
        tr = _f32c(t_ref.detach().to(dev))
        need_grad = coeff_grid.requires_grad
        Bc, S, c2, H, W = cg.shape
        shape = make_shape(cfg, B, M, Mp, 0, K=0, extra_flags=C.F_NO_WARP | C.F_NO_BWD_RECORDS)
        hq, wq = shape.hq, shape.wq
        G = hq * wq
        if Bc != B or c2 != 2 * k or (ph is not None and tuple(ph.shape) != (B, M, k)) or (ph is None and k > 8):
            raise ValueError(f'coeff_grid {tuple(cg.shape)} / phi do not match [B, S, 2k, H, W] / [B, M, k] (k <= 8 without phi)')
        st = _stream(dev)
        L = C.lib()
        c_rows = torch.empty((B * G, 2 * k), dtype=torch.float32, device=dev)
        with _stage('mpc_pe_tile_rows', dev):
            C.check(L.mpc_pe_tile_rows(_ptr(cg), _ptr(c_rows), B, S, c2, H, W, tile, st), 'mpc_pe_tile_rows')
        ws = alloc_workspace(shape, dev)
        rows = torch.empty_like(ev)
        if B * M > 0:
            with _stage('mpc_pe_warp', dev):
                C.check(L.mpc_pe_warp(ctypes.byref(shape), _ptr(ev), _ptr(c_rows), _ptr(ph), k, _ptr(tr), _ptr(rows), st), 'mpc_pe_warp')
        raw = event_splat_fwd(shape, rows, None, tr, ws)
        blur, gimg = contrast_fwd(shape, raw, ws, need_grad)
        g_field, s_nimg = None, 0
        if pm is not None and cfg.smooth_weight > 0:
            nb = pm.shape[0]
            field = torch.empty((B * nb, hq, wq, 2), dtype=torch.float32, device=dev)
            with _stage('mpc_pe_basis_field', dev):
                C.check(L.mpc_pe_basis_field(_ptr(c_rows), _ptr(pm), _ptr(field), B, G, k, nb, st), 'mpc_pe_basis_field')
            s_nimg = B * nb
            g_field = lut_smooth(shape, field, s_nimg, 2, cfg.smooth_weight, ws, need_grad)
        scal = finalize(shape, s_nimg, 2 if s_nimg else 0, cfg.smooth_weight if s_nimg else 0.0, ws, dev)
        with torch.cuda.device(dev):
            ordered = offsets is not None and int(L.mpc_pe_grad_ordered_supported(ctypes.byref(shape), k)) == 1
        offs = _check_offsets(offsets, cfg, shape, dev) if ordered else None
        ctx.shape, ctx.k, ctx.tile, ctx.grid_shape = shape, k, tile, (B, S, c2, H, W)
        ctx.has = (ph is not None, pm is not None and g_field is not None, offs is not None)
        ctx.set_materialize_grads(False)
        none = tr
        ctx.save_for_backward(rows, ph if ph is not None else none, tr, gimg if gimg is not None else none, scal,
                              offs if offs is not None else none, g_field if g_field is not None else none, pm if pm is not None else none)
        out = scal[:3].clone()
        loss, focus, smooth = out[C.SCAL_LOSS], out[C.SCAL_FOCUS], out[C.SCAL_SMOOTH]
        ctx.mark_non_differentiable(focus, smooth, blur)
        return loss, focus, smooth, blur

    @staticmethod
    def backward(ctx, g_loss, g_focus, g_smooth, g_blur):
        if g_loss is None:
            return (None,) * 10
        rows, ph, tr, gimg, scal, offs, g_field, pm = ctx.saved_tensors
        has_phi, has_smooth, has_offs = ctx.has
        ph = ph if has_phi else None
        shape, k = ctx.shape, ctx.k
        B, S, c2, H, W = ctx.grid_shape
        G = shape.hq * shape.wq
        dev = rows.device
        st = _stream(dev)
        L = C.lib()
        go = _f32c(g_loss.reshape(1))
        split, gp = 1, None
        if has_offs:
            split = max(1, min(16, 512 // max(1, B * _lut_strips(shape, dev))))
            gp = torch.empty((split, B * G, 2 * k), dtype=torch.float32, device=dev)
            with _stage('mpc_pe_grad_ordered', dev):
                rc = L.mpc_pe_grad_ordered(ctypes.byref(shape), _ptr(rows), _ptr(offs), _ptr(ph), k, _ptr(tr), _ptr(gimg), _ptr(scal), _ptr(go),
                                           _ptr(gp), split, st)
            if rc == C.E_UNSUPPORTED:
                gp = None
            else:
                C.check(rc, 'mpc_pe_grad_ordered')
        if gp is None:
            split = 1
            gp = torch.empty((1, B * G, 2 * k), dtype=torch.float32, device=dev)
            with _stage('mpc_pe_grad', dev):
                C.check(L.mpc_pe_grad(ctypes.byref(shape), _ptr(rows), _ptr(ph), k, _ptr(tr), _ptr(gimg), _ptr(scal), _ptr(go), _ptr(gp), st), 'mpc_pe_grad')
        g_rows = torch.empty((B * G, 2 * k), dtype=torch.float32, device=dev)
        nb = pm.shape[0] if has_smooth else 0
        with _stage('mpc_pe_rows_grad_finish', dev):
            C.check(L.mpc_pe_rows_grad_finish(_ptr(gp), split, _ptr(g_field) if has_smooth else None, _ptr(pm) if has_smooth else None, _ptr(go),
                                              _ptr(g_rows), B, G, k, max(nb, 1), st), 'mpc_pe_rows_grad_finish')
        g_grid = torch.empty(ctx.grid_shape, dtype=torch.float32, device=dev)
        with _stage('mpc_pe_tile_rows_bwd', dev):
            C.check(L.mpc_pe_tile_rows_bwd(_ptr(g_rows), _ptr(g_grid), B, S, c2, H, W, ctx.tile, st), 'mpc_pe_tile_rows_bwd')
        return (g_grid,) + (None,) * 9


class TileCoeffRowsFn(torch.autograd.Function):
    """coeff_grid [B, S, 2k, H, W] -> the coefficients at the tile centres (offset tile // 2, row-major: trajectories.py:3-52),
    scales summed, one row per tile: [B*hq*wq, 2k].  One autograd node with a strided copy each way instead of the half dozen
    view / sum / permute nodes of the plain-torch spelling (the per-event step is bound by the host)."""

    @staticmethod
    def forward(ctx, coeff_grid, tile):
        ctx.shape_in, ctx.tile = tuple(coeff_grid.shape), int(tile)
        cs = coeff_grid[:, :, :, tile // 2::tile, tile // 2::tile]
        cs = cs[:, 0] if cs.shape[1] == 1 else cs.sum(1)                      # [B, 2k, hq, wq]
        B, c2, hq, wq = cs.shape
        return cs.permute(0, 2, 3, 1).reshape(B * hq * wq, c2)

    @staticmethod
    def backward(ctx, g):
        B, S, c2, H, W = ctx.shape_in
        t = ctx.tile
        out = torch.zeros(ctx.shape_in, dtype=g.dtype, device=g.device)
        view = out[:, :, :, t // 2::t, t // 2::t]
        view.copy_(g.view(B, view.shape[3], view.shape[4], c2).permute(0, 3, 1, 2)[:, None].expand_as(view))
        return out, None


class BasisFieldFn(torch.autograd.Function):
    """c_rows [B*G, 2k] (per tile: y orders, x orders), phim [nb, k] -> flow field at nb times [B*nb, hq, wq, 2] (the layout
    of LutSmoothFn): field[b, t, cell, d] = sum_j c[b, cell, d, j] phim[t, j].  One node, a small GEMM and a copy each way."""

    @staticmethod
    def forward(ctx, c_rows, phim, B, hq, wq):
        k = phim.shape[1]
        ctx.save_for_backward(phim)
        ctx.dims = (B, hq, wq, k)
        f = c_rows.reshape(B * hq * wq * 2, k) @ phim.t()                     # [B*G*2, nb]
        return f.view(B, hq * wq, 2, phim.shape[0]).permute(0, 3, 1, 2).reshape(B * phim.shape[0], hq, wq, 2)

    @staticmethod
    def backward(ctx, g):
        (phim,) = ctx.saved_tensors
        B, hq, wq, k = ctx.dims
        nb = phim.shape[0]
        gf = g.reshape(B, nb, hq * wq, 2).permute(0, 2, 3, 1).reshape(B * hq * wq * 2, nb)
        return (gf @ phim).view(B * hq * wq, 2 * k), None, None, None, None


class GatherRowsFn(torch.autograd.Function):
    """rows[idx] whose backward is index_add_ (float atomics) -- torch's own advanced-indexing backward sorts the indices
    (index_put_ with accumulate): tens of milliseconds for the 2.8 M events of a DSEC batch.  UNPINNED extension
    (FocusLoss.calc_per_event_basis); the gradient it returns is not bitwise reproducible."""

    @staticmethod
    def forward(ctx, rows, idx):
        ctx.save_for_backward(idx)
        ctx.n = rows.shape[0]
        return rows.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return torch.zeros((ctx.n, g.shape[1]), dtype=g.dtype, device=g.device).index_add_(0, idx, g.contiguous()), None


def event_pos_grad(shape, rows, t_ref, gimg, scal, grad_out):
    """UNPINNED extension: d objective / d warped position per event row -> [B, M, 2] (mpc_event_pos_grad)."""
    B, M = rows.shape[0], rows.shape[1]
    g = torch.empty((B, M, 2), dtype=torch.float32, device=rows.device)
    with _stage('mpc_event_pos_grad', rows.device):
        C.check(C.lib().mpc_event_pos_grad(ctypes.byref(shape), _ptr(rows), _ptr(t_ref), _ptr(gimg), _ptr(scal), _ptr(grad_out),
                                           _ptr(g), _stream(rows.device)), 'mpc_event_pos_grad')
    return g


class PrewarpedFocusFn(torch.autograd.Function):
    """UNPINNED extension (FocusLoss.calc_per_event_basis): A7-A9 on events whose positions are warped already.
    warped_pos [B, M, 2] (y, x; the differentiable input), events [B, M, 6] (columns 2.. = t, p, bin, valid as the loader
    writes them) -> (focus_loss, iwes_blurred).  Forward: the LDS-tiled vote of mpc_event_splat_fwd with MPC_F_NO_WARP;
    backward: mpc_event_pos_grad (one thread per row, no atomics)."""

    @staticmethod
    def forward(ctx, warped_pos, events, t_ref, cfg: PathConfig, num_pos: int):
        _require_gpu(warped_pos, 'warped_pos')
        B, M, Mp = _check_events(events, cfg, num_pos)
        dev = warped_pos.device
        rows = torch.cat((_f32c(warped_pos.detach()), _f32c(events.detach())[..., 2:]), dim=-1).contiguous()
        tr = _f32c(t_ref.detach().to(dev))
        need_grad = warped_pos.requires_grad
        shape = make_shape(cfg, B, M, Mp, 0, K=0, extra_flags=C.F_NO_WARP | C.F_NO_BWD_RECORDS)
        ws = alloc_workspace(shape, dev)
        raw = event_splat_fwd(shape, rows, None, tr, ws)
        blur, gimg = contrast_fwd(shape, raw, ws, need_grad)
        scal = finalize(shape, 0, 0, 0.0, ws, dev)
        ctx.shape = shape
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(rows, tr, gimg, scal)
        ctx.mark_non_differentiable(blur)
        return scal[C.SCAL_FOCUS].clone(), blur

    @staticmethod
    def backward(ctx, g_focus, g_blur):
        rows, tr, gimg, scal = ctx.saved_tensors
        if g_focus is None:
            return None, None, None, None, None
        return event_pos_grad(ctx.shape, rows, tr, gimg, scal, _f32c(g_focus.reshape(1))), None, None, None, None


class KnnLutFn(torch.autograd.Function):
    """A5: trajectories -> (flow_lut, flow_next or empty)."""

    @staticmethod
    def forward(ctx, trajectories, cfg: PathConfig):
        _require_gpu(trajectories, 'trajectories')
        traj = _f32c(trajectories.detach())
        B, _, n, _ = traj.shape
        shape = make_shape(cfg, B, 0, 0, n)
        ws = alloc_workspace(shape, traj.device)
        flow_lut, flow_next, state, _ = knn_lut_fwd(cfg, shape, traj, ws)
        ctx.shape, ctx.ws = shape, ws
        ctx.has_next = flow_next is not None
        ctx.save_for_backward(traj, state)
        if flow_next is None:
            flow_next = traj.new_zeros(0)
        return flow_lut, flow_next

    @staticmethod
    def backward(ctx, g_lut, g_next):
        traj, state = ctx.saved_tensors
        g_lut = _f32c(g_lut)
        g_next = _f32c(g_next) if ctx.has_next else None
        return knn_lut_bwd(ctx.shape, traj, g_lut, g_next, state, ctx.ws), None


class LutSmoothFn(torch.autograd.Function):
    """A10: field [nimg, hq, wq, C] -> smooth_weight * smoothness."""

    @staticmethod
    def forward(ctx, field, cfg: PathConfig, weight: float):
        _require_gpu(field, 'field')
        f = _f32c(field.detach())
        nimg, hq, wq, Cch = f.shape
        if (hq, wq) != cfg.lut_grid:
            raise ValueError(f'field grid {(hq, wq)} does not match the LUT grid {cfg.lut_grid}')
        B = max(1, -(-nimg * Cch // (cfg.num_bins * cfg.num_tref * 2)))
        shape = make_shape(cfg, B, 0, 0, 0, K=0)
        ws = alloc_workspace(shape, f.device)
        g = lut_smooth(shape, f, nimg, Cch, weight, ws, field.requires_grad)
        # only the smoothness scalar of finalize is meaningful here (no contrast pass was run)
        scal = finalize(shape, nimg, Cch, weight, ws, f.device)
        ctx.save_for_backward(g)
        return scal[C.SCAL_SMOOTH].clone()

    @staticmethod
    def backward(ctx, g_out):
        (g,) = ctx.saved_tensors
        return scale(g, _f32c(g_out.reshape(1))), None, None


class CurveTrajFn(torch.autograd.Function):
    """Flow curves -> `trajectories` (BASELINE.json configs[3]): params [B, 2d, h, w] ((x, y) channel order,
    reference curves/polynomial.py:60-61), basis [T, d] on the device, tile centres [n, 2] (y, x) -> [B, T, n, 2], one kernel each
    way (csrc/curves.hip) instead of the einsum / stack / add chain of plain torch and its adjoints."""

    @staticmethod
    def forward(ctx, params, basis, pos, scale: float):
        _require_gpu(params, 'params')
        B, c2, h, w = params.shape
        d, n, T = c2 // 2, h * w, basis.shape[0]
        dev = params.device
        p = _f32c(params.detach())
        bm = _f32c(basis.detach())
        ps = _f32c(pos.detach())
        traj = torch.empty((B, T, n, 2), dtype=torch.float32, device=dev)
        C.check(C.lib().mpc_curve_traj_fwd(_ptr(p), _ptr(bm), _ptr(ps), float(scale), _ptr(traj), B, d, T, n, _stream(dev)), 'mpc_curve_traj_fwd')
        ctx.save_for_backward(bm)
        ctx.dims = (B, d, T, n, h, w, float(scale))
        return traj

    @staticmethod
    def backward(ctx, g):
        (bm,) = ctx.saved_tensors
        B, d, T, n, h, w, scale = ctx.dims
        g = _f32c(g)
        gp = torch.empty((B, 2 * d, h, w), dtype=torch.float32, device=g.device)
        C.check(C.lib().mpc_curve_traj_bwd(_ptr(g), _ptr(bm), scale, _ptr(gp), B, d, T, n, _stream(g.device)), 'mpc_curve_traj_bwd')
        return gp, None, None, None


def knn_indices(cfg: PathConfig, trajectories):
    """Debug/test helper: the K neighbour indices [B, nb, Q, K] (ascending distance, index)."""
    _require_gpu(trajectories, 'trajectories')
    traj = _f32c(trajectories.detach())
    B, _, n, _ = traj.shape
    shape = make_shape(cfg, B, 0, 0, n)
    ws = alloc_workspace(shape, traj.device)
    _, _, _, idx = knn_lut_fwd(cfg, shape, traj, ws, want_idx=True)
    return idx


def splat_events(image_shape, pos_weight_rows, unit_weight, blur):
    """imager path (reference event_image_converter.py:45-74,134-176 'bilinear_vote'):
    rows [nimg, m, 6] with (y, x, -, -, -, weight) -> [nimg, H, W], optionally blurred."""
    _require_gpu(pos_weight_rows, 'events')
    rows = _f32c(pos_weight_rows)
    nimg, m, _ = rows.shape
    cfg = PathConfig(image_shape=tuple(image_shape), num_tref=1, num_bins=1, num_knn=0, smooth_weight=0.0,
                     sp=max(image_shape), norm_l2=False, dist_l1=False, scale_by_dt=False, mask_border=False,
                     polarity_split=False, scheme_iwd=False, smooth_on_next=False)
    extra = C.F_NO_WARP | (C.F_UNIT_WEIGHT if unit_weight else 0)
    shape = make_shape(cfg, nimg, m, m, 0, extra_flags=extra, K=0)
    ws = alloc_workspace(shape, rows.device)
    raw = event_splat_fwd(shape, rows, None, None, ws)
    if blur:
        raw, _ = contrast_fwd(shape, raw, ws, False)
    return raw[:, 0]
