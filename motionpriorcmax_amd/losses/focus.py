"""FocusLoss behind the reference's loss plugin API, computed by the HIP kernels of libmpcmax.

Same constructor keys, attributes and `calc` contract as reference src/losses/focus.py:9-113, so
`TrajectoryNet.step` (src/modules/trajectory_net.py:142-161) and the image-logging callback
(src/utils/logging.py:53-120) can use it unchanged."""
import torch

from . import base
from .. import ops
from ..utils.event_image_converter import EventImageConverter


class FocusLoss(base.TrajectoryLossBase):
    """
    Args (reference focus.py:13-27):
        image_shape (tuple): (height, width).
        num_tref (int): number of reference times; 1 = one random reference time.
        num_bins (int): number of time bins of the flow look-up table.
        num_knn (int): nearest trajectories averaged per look-up-table cell.
        smooth_weight (float): weight of the smoothness term.
        lut_superpixel_size (int): pixel size of one look-up-table cell.
        focus_loss_norm (str): 'l1' or 'l2' gradient-magnitude norm.
        dist_norm (str): 'l1' or 'l2' neighbour distance.
        scale_iwe_by_dt, mask_image_border, polarity_aware_batching (bool)
        interpolation_scheme (str): 'mean' or 'iwd'.
        smooth_type (str): 'on_flow_to_tref' or 'on_flow_to_next'.
        loss_type (str, build-side extension): 'gradient_magnitude' (what the reference hard-codes,
            focus.py:90-91) or 'variance' (reference src/utils/loss.py:14-16).
        pyramid_levels (int, UNPINNED extension, default 1 = the reference): IWE pyramid of that many levels (2x2 averages of the
            raw IWE), the focus term summed over the levels (`ops.PyramidFocusFn`); BASELINE.json names one, the reference has none.
        trefs_as_samples (bool, build-side switch, default True): with num_tref > 1 the T reference times of a sample run as T samples
            of the num_tref == 1 kernels (`_calc_trefs_as_samples`); False: the general kernels (every stage in one launch for all T).
        static_shapes (bool, build-side extension, default False): capture `calc` + backward once per input shape into HIP
            graphs and replay them (ops.StaticFocusPlan): for small batches, whose eager step is bound by the host.
            `misc_metadata['iwes']` is then only valid until the next `calc`.
    """

    def __init__(self, image_shape, num_tref, num_bins, num_knn, smooth_weight,
                 lut_superpixel_size, focus_loss_norm, dist_norm,
                 scale_iwe_by_dt, mask_image_border, polarity_aware_batching,
                 interpolation_scheme, smooth_type, loss_type='gradient_magnitude', profiler=None,
                 static_shapes=False, pyramid_levels=1, auto_static_shapes=True, trefs_as_samples=True, **kwargs):
        super().__init__()
        self.image_shape = image_shape
        self.num_tref = num_tref
        self.num_bins = num_bins
        self.num_knn = num_knn
        self.smooth_weight = smooth_weight
        self.lut_superpixel_size = lut_superpixel_size
        self.focus_loss_norm = focus_loss_norm
        self.dist_norm = dist_norm
        self.scale_iwe_by_dt = scale_iwe_by_dt
        self.mask_image_border = mask_image_border
        self.polarity_aware_batching = polarity_aware_batching
        self.interpolation_scheme = interpolation_scheme
        self.smooth_type = smooth_type
        self.loss_type = loss_type
        self.profiler = profiler
        self.static_shapes = bool(static_shapes)
        self.pyramid_levels = int(pyramid_levels)      # UNPINNED extension (ops.PyramidFocusFn); 1 = the reference's single scale
        if self.pyramid_levels < 1 or (self.pyramid_levels > 1 and (num_tref != 1 or loss_type != 'gradient_magnitude')):
            raise ValueError('pyramid_levels > 1 needs num_tref == 1 and the gradient-magnitude objective')
        self._static_plans = {}
        # Small steps are bound by the HOST (a B = 1 step: ~13 kernel boundaries of one wave of workgroups each behind two eager
        # C-ABI calls): when the same small shape comes `AUTO_STATIC_AFTER` times in a row, calc switches to the captured
        # plan of that shape by itself (ops.StaticFocusCalcFn, automatic mode: images copied out, the eager path again while an
        # earlier calc of the shape still waits for its backward, and for any new shape -- a loader whose batches differ in
        # length never leaves the eager path).  The caller (src/modules/trajectory_net.py:152-158) writes no capture code.
        self.auto_static_shapes = bool(auto_static_shapes)
        self._auto_key, self._auto_run = None, 0
        self._auto_never = set()              # shapes whose plan could not be captured: eager from then on
        self._tmid_dev = {}                   # bin mid-times per device (calc_per_event_basis)
        # num_tref > 1 (not in the shipped yaml files): the T reference times of a sample as T samples of the num_tref == 1 path
        # (_calc_trefs_as_samples); False: the general kernels (one launch for all reference times, global-atomic event path)
        self.trefs_as_samples = bool(trefs_as_samples)
        self.is_needing_offsets = True
        self.imager = EventImageConverter(self.image_shape)

        assert not scale_iwe_by_dt or num_tref == 1
        assert not polarity_aware_batching or num_tref == 1
        assert not smooth_type == 'on_flow_to_next' or num_tref == 1
        if focus_loss_norm not in ('l1', 'l2') or dist_norm not in ('l1', 'l2'):
            raise ValueError
        if loss_type not in ('gradient_magnitude', 'variance'):
            raise ValueError
        if num_knn > 1 and interpolation_scheme not in ('mean', 'iwd'):
            raise ValueError
        if smooth_weight != 0 and smooth_type not in ('on_flow_to_tref', 'on_flow_to_next'):
            raise ValueError

        self._cfg = ops.PathConfig(
            image_shape=(int(image_shape[0]), int(image_shape[1])), num_tref=int(num_tref),
            num_bins=int(num_bins), num_knn=int(num_knn), smooth_weight=float(smooth_weight),
            sp=int(lut_superpixel_size), norm_l2=(focus_loss_norm == 'l2'), dist_l1=(dist_norm == 'l1'),
            scale_by_dt=bool(scale_iwe_by_dt), mask_border=bool(mask_image_border),
            polarity_split=bool(polarity_aware_batching),
            scheme_iwd=(interpolation_scheme == 'iwd'), smooth_on_next=(smooth_type == 'on_flow_to_next'),
            variance=(loss_type == 'variance'), atomic_path=bool(kwargs.get('debug_atomic_path', False)))
        import dataclasses
        self._cfg_t1 = dataclasses.replace(self._cfg, num_tref=1)

    def _calc_trefs_as_samples(self, trajectories, events, t_ref, offsets):
        """num_tref = T > 1 (focus.py:53-57, 66-113) on the kernels of the shipped num_tref == 1 configurations: sample b with its T
        reference times IS T samples (b, 0) .. (b, T - 1) that share b's events and mid-time trajectories and differ in the
        reference-time trajectory -- the same B * T images, the same B * T * num_bins look-up tables, hence the same objective
        (loss.py: a mean over all images) and smoothness (a mean over all tables); the gradient of the shared rows is summed
        by autograd through `repeat_interleave`.  The neighbour search runs T times over the same mid-time points -- twice the
        work of an ideal T-flow kernel for T = 2, a third of the time of the general kernels (C3 shape: 3.73 -> ~1.3 ms).
        Requires what focus.py:49-50 requires of num_tref > 1 anyway (no scale_iwe_by_dt, no polarity_aware_batching)."""
        T = self.num_tref
        b = events.shape[0]
        n = trajectories.shape[2]
        ref = trajectories[:, :T].reshape(b * T, 1, n, 2)
        mid = trajectories[:, T:].repeat_interleave(T, dim=0)
        traj_v = torch.cat((ref, mid), dim=1)
        events_v = events.repeat_interleave(T, dim=0)
        offs_v = offsets.repeat_interleave(T, dim=0) if torch.is_tensor(offsets) else offsets
        return ops.FocusCalcFn.apply(traj_v, events_v, t_ref[:1], self._cfg_t1, -1, offs_v)

    def get_reconstruction_times(self, device):
        """Reference focus.py:53-64: [t_ref..., bin mid-times]."""
        if self.num_tref > 1:
            t_ref = torch.linspace(0, 1, self.num_tref, device=device)
        elif self.num_tref == 1:
            t_ref = torch.rand(1, device=device)  # random reference time
        else:
            raise ValueError("Invalid value for num_tref. Must be >= 1.")

        t_bins = torch.linspace(0, 1, self.num_bins + 1, device=device)
        t_mid = (t_bins[:-1] + t_bins[1:]) / 2
        return torch.concat((t_ref, t_mid), dim=0)

    def order_events(self, batch):
        """Not in the reference (layout half): `batch` with its event rows ordered by (time bin, LUT
        strip) inside each polarity block and the table `event_offsets` added.  `calc` returns the same loss and
        gradient for the ordered batch bit for bit, and with the table skips one record per event in each direction.
        Meant for the data pipeline: once per batch, next to `utils.ingest_events`."""
        num_pos = batch['num_pos_events'] if 'num_pos_events' in batch else -1
        ev, offs = ops.event_bucket_order(self._cfg, batch['events'], int(num_pos))
        out = dict(batch)
        out['events'], out['event_offsets'] = ev, offs
        return out

    def calc_per_event_basis(self, coeff_grid, t_ref, batch, num_basis, basis_type='polynomial', basis_network=None, tile_size=None,
                             fused=True):
        """UNPINNED EXTENSION -- no reference code exists for it (the reference warps through a binned, KNN-inverted flow LUT,
        focus.py:115-195); BASELINE.json's north_star names it: the per-event continuous-time warp.

        Every event is warped with the motion basis evaluated at ITS OWN timestamp and the coefficients of the tile it lies in:
            warped = (y, x) + sum_k c_k[tile(y, x)] * (basis_k(t_ref) - basis_k(t_event))
        i.e. trajectory_at(t_ref) - trajectory_at(t_event) of the trajectory that STARTS at the event's tile (utils/trajectories.py,
        utils/basis.py) -- no time bins, no nearest-neighbour look-up table; then the reference's weights, bilinear vote, blur and
        objective (focus.py:197-230, event_image_converter.py:333-391, loss.py:4-27) on the HIP kernels.  The smoothness term is
        the reference's Charbonnier term on the same flow sampled at the num_bins bin mid-times.

        coeff_grid [B, 1, 2k, H, W] (first k channels y, next k x: the network's dense output, trajectory_net.py:142-161),
        t_ref scalar tensor / float in [0, 1], batch as for `calc`.  Returns the triple of `calc`.  Differentiable w.r.t. coeff_grid
        (not w.r.t. the weights of a 'learned' basis: the basis values enter as constants).  With a bucket-ordered batch
        (`order_events` / `ingest_events(order_for=...)`: `event_offsets`) the backward accumulates per LUT strip in LDS fixed point
        and is bitwise reproducible; without the table it uses global float atomics (slow: ~20 G atomics/s, and not
        reproducible).  fused=False: the same in plain torch around the vote / objective kernels (cross-check)."""
        from ..utils import basis_values
        if self.num_tref != 1:
            raise ValueError('calc_per_event_basis needs num_tref == 1')
        events = batch['events']
        num_pos_events = batch['num_pos_events'] if 'num_pos_events' in batch else -1
        assert not self.polarity_aware_batching or num_pos_events > -1
        dev = events.device
        h, w = self._cfg.image_shape
        sp = self.lut_superpixel_size
        tile = sp if tile_size is None else int(tile_size)
        if tile != sp:
            raise ValueError('the coefficient tiles must be the look-up-table cells (tile_size == lut_superpixel_size)')
        hq, wq = self._cfg.lut_grid
        B, M = events.shape[0], events.shape[1]
        G = hq * wq
        # the coefficients at the tile centres (get_optical_flow_tile_mask + coeffs_grid_to_list, trajectories.py:3-52: offset
        # tile // 2, row-major -- as a strided view, whose backward is a strided copy), scales summed as compute_basis does
        if coeff_grid.dim() != 5 or coeff_grid.shape[2] != 2 * num_basis or -(-coeff_grid.shape[3] // tile) != hq or -(-coeff_grid.shape[4] // tile) != wq:
            raise ValueError(f'coeff_grid {tuple(coeff_grid.shape)} is not [B, scales, {2 * num_basis}, H, W] of this image shape')
        # (no host -> device copy inside the step: a copy from pageable memory waits for the stream, and the host then no longer runs
        # ahead of the device -- round 6: 0.86 ms per step of which 0.36 were kernels.  A float t_ref becomes a fill kernel, the bin
        # mid-times are kept on the device.)
        if torch.is_tensor(t_ref):
            t_ref = t_ref.to(device=dev, dtype=torch.float32).reshape(1)
        else:
            t_ref = torch.full((1,), float(t_ref), dtype=torch.float32, device=dev)
        offsets = batch['event_offsets'] if 'event_offsets' in batch else None     # from order_events / ingest (optional)
        phi = None           # (fused + polynomial basis of up to 8 orders: worked out inside the kernels)
        if not (fused and basis_type == 'polynomial' and num_basis <= 8):
            with torch.no_grad():
                phi = basis_values(t_ref, num_basis, basis_type, basis_network) - basis_values(events[..., 2], num_basis, basis_type, basis_network)
        phim = None
        if self.smooth_weight > 0:
            from ..utils.synth import bin_mid_times
            tm = self._tmid_dev.get(dev)
            if tm is None:
                tm = self._tmid_dev[dev] = bin_mid_times(self.num_bins).to(dev)
            with torch.no_grad():
                phim = basis_values(t_ref, num_basis, basis_type, basis_network) - basis_values(tm, num_basis, basis_type, basis_network)   # [nb, k]
        if fused and num_basis <= 8 and self.num_bins <= 64:
            # one autograd node of library calls (ops.PerEventBasisCalcFn): the step is no longer bound by the host
            loss, focus, smooth, iwes = ops.PerEventBasisCalcFn.apply(coeff_grid, events, phi, phim, t_ref, self._cfg, int(num_pos_events), offsets,
                                                                      int(num_basis), tile)
            iwes = iwes.reshape(B, 1, 2, h, w) if self.polarity_aware_batching else iwes.reshape(B, 1, h, w)
            return loss, {'focus_loss': focus.detach(), 'smoothness_loss': smooth.detach()}, {'iwes': iwes.detach()}
        c_rows = ops.TileCoeffRowsFn.apply(coeff_grid, tile)                                       # [B*G, 2k]: per tile (y: k, x: k)
        if fused:
            focus, iwes = ops.PerEventBasisFocusFn.apply(c_rows, events, phi, t_ref, self._cfg, int(num_pos_events), offsets)
        else:
            # the same in plain torch around the vote / objective kernels (cross-check of the fused kernels)
            with torch.no_grad():
                iy = torch.div(events[..., 0], sp, rounding_mode='floor').long().clamp_(0, hq - 1)      # focus.py:186-187
                ix = torch.div(events[..., 1], sp, rounding_mode='floor').long().clamp_(0, wq - 1)
                idx = (torch.arange(B, device=dev).view(-1, 1) * G + iy * wq + ix).reshape(-1)
            ce = ops.GatherRowsFn.apply(c_rows, idx).view(B, M, 2, num_basis)
            warped = events[..., :2] + (ce * phi[:, :, None, :]).sum(-1)                              # [B, M, 2]  (y, x)
            focus, iwes = ops.PrewarpedFocusFn.apply(warped, events, t_ref, self._cfg, int(num_pos_events))
        smooth = torch.zeros((), device=dev)
        if phim is not None:
            field = ops.BasisFieldFn.apply(c_rows, phim, B, hq, wq)                                # [B*nb, hq, wq, 2]
            smooth = ops.LutSmoothFn.apply(field, self._cfg, float(self.smooth_weight))
        loss = focus + smooth
        iwes = iwes.reshape(B, 1, 2, h, w) if self.polarity_aware_batching else iwes.reshape(B, 1, h, w)
        return loss, {'focus_loss': focus.detach(), 'smoothness_loss': smooth.detach()}, {'iwes': iwes.detach()}

    AUTO_STATIC_MAX_EVENTS = 150_000      # B * M up to which the captured plan wins (C2: 0.16 against 0.20 ms; C4, 500k events: it loses)
    AUTO_STATIC_AFTER = 3                 # consecutive calcs of one shape before it is captured

    def _auto_static(self, trajectories, events, num_pos_events, offsets):
        """Should this calc replay the captured plan of its shape?  (automatic static shapes, see __init__)"""
        if not self.auto_static_shapes or self.profiler is not None or ops.STAGE_TIMER is not None or ops.kernel_timer_on():
            return False
        if not (torch.is_tensor(events) and events.is_cuda and events.dim() == 3 and torch.is_tensor(trajectories) and trajectories.is_cuda
                and trajectories.dim() == 4):
            return False                  # (the eager path raises the proper error)
        if events.shape[0] * events.shape[1] > self.AUTO_STATIC_MAX_EVENTS or torch.cuda.is_current_stream_capturing():
            return False
        key = ops.StaticFocusCalcFn.plan_key(trajectories, events, self._cfg, int(num_pos_events), offsets)
        if key in self._auto_never:
            return False
        if key == self._auto_key:
            self._auto_run += 1
        else:
            self._auto_key, self._auto_run = key, 1
        if self._auto_run < self.AUTO_STATIC_AFTER:
            return False
        return not ops.StaticFocusCalcFn.plan_busy(self._static_plans, key)

    def calc(self, trajectories, times, batch):
        """Reference focus.py:66-113.

        trajectories [B, num_tref + num_bins, n, 2] (y, x), times [num_tref + num_bins],
        batch['events'] [B, M, 6], batch['num_pos_events'] (int, with polarity_aware_batching).
        Returns (loss with grad, {'focus_loss', 'smoothness_loss'} detached,
                 {'iwes': [B, T, 2, H, W] or [B, T, H, W]} detached)."""
        events = batch['events']
        num_pos_events = batch['num_pos_events'] if 'num_pos_events' in batch else -1
        assert not self.polarity_aware_batching or num_pos_events > -1

        t_ref = times[:self.num_tref]
        offsets = batch['event_offsets'] if 'event_offsets' in batch else None     # from order_events (optional)
        if self.num_tref > 1 and self.trefs_as_samples and not self._cfg.atomic_path and torch.is_tensor(events) and events.dim() == 3 \
                and torch.is_tensor(trajectories) and trajectories.dim() == 4:
            out = self._calc_trefs_as_samples(trajectories, events, t_ref, offsets)
        elif self.pyramid_levels > 1:
            out = ops.PyramidFocusFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), self.pyramid_levels)
        elif self.static_shapes and ops.STAGE_TIMER is None and not torch.cuda.is_current_stream_capturing():
            out = ops.StaticFocusCalcFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), offsets, self._static_plans)
        elif self._auto_static(trajectories, events, num_pos_events, offsets):
            try:
                out = ops.StaticFocusCalcFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), offsets, self._static_plans, True)
            except ops.AutoPlanFailed:
                # the plan of this shape could not be captured (another thread of the training process used the device in the
                # capture window, no memory for the plan's buffers, ...): this step and every later one of the shape run eagerly
                self._auto_never.add(ops.StaticFocusCalcFn.plan_key(trajectories, events, self._cfg, int(num_pos_events), offsets))
                out = ops.FocusCalcFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), offsets)
        elif self.profiler is not None:
            with torch.profiler.record_function('mpcmax::FocusLoss.calc'):
                out = ops.FocusCalcFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), offsets)
        else:
            out = ops.FocusCalcFn.apply(trajectories, events, t_ref, self._cfg, int(num_pos_events), offsets)
        loss, focus_loss, smooth_loss, iwes = out

        h, w = self._cfg.image_shape
        b = events.shape[0]
        if self.polarity_aware_batching:
            iwes = iwes.reshape(b, self.num_tref, 2, h, w)
        else:
            iwes = iwes.reshape(b, self.num_tref, h, w)

        log_metadata = {
            'focus_loss': focus_loss.detach(),
            'smoothness_loss': smooth_loss.detach(),
        }
        misc_metadata = {
            'iwes': iwes.detach()
        }
        return loss, log_metadata, misc_metadata
