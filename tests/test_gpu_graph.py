"""calc + backward captured into a HIP graph through torch.cuda.CUDAGraph and replayed many times: the library
makes no synchronising or allocating call and issues kernels only (memset NODES made replays fault on ROCm 7.2,
so the counters are zeroed by a kernel).  Replays must reproduce the eager loss and gradient bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_calc_and_backward_replay_as_a_hip_graph():
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl = dict(bench.WORKLOADS['C2'])
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=4)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    t = traj.to(dev).requires_grad_(True)
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    td = times.to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                  # warm-up outside the capture (one-time set-up calls)
        for _ in range(3):
            loss, _, _ = L.calc(t, td, batch)
            loss.backward()
            ref_loss, ref_grad = loss.detach().clone(), t.grad.clone()
            t.grad = None
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g, _, misc_g = L.calc(t, td, batch)
        loss_g.backward()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_g.detach(), ref_loss)
    assert torch.equal(t.grad, ref_grad)
    assert torch.isfinite(misc_g['iwes']).all()


def _c2(seed):
    import bench
    wl = dict(bench.WORKLOADS['C2'])
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=seed)
    return wl, ev, num_pos, traj, times


def test_static_shapes_replays_calc_and_backward_bit_for_bit():
    """FocusLoss(static_shapes=True): the caller writes no capture code; loss, IWEs and gradient equal the eager ones bit for
    bit, step after step, also when the trajectories, the reference time or the event tensor change between steps."""
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(5)
    Le = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    Ls = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), static_shapes=True))
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    _, ev2, _, _, _ = _c2(6)
    batch2 = {'events': ev2.to(dev), 'num_pos_events': num_pos}
    td = times.to(dev)
    for it in range(6):
        b = batch if it < 3 else batch2                                   # a new batch from step 3 on
        tt = (traj + 0.25 * it).to(dev)
        tdi = td.clone(); tdi[0] = 0.1 + 0.13 * it                          # a new reference time every step
        te = tt.clone().requires_grad_(True)
        ts = tt.clone().requires_grad_(True)
        le, loge, misce = Le.calc(te, tdi, b)
        (2.5 * le).backward()
        ls, logs, miscs = Ls.calc(ts, tdi, b)
        (2.5 * ls).backward()
        assert torch.equal(ls.detach(), le.detach()), it
        assert torch.equal(logs['focus_loss'], loge['focus_loss']) and torch.equal(logs['smoothness_loss'], loge['smoothness_loss'])
        assert torch.equal(miscs['iwes'], misce['iwes'])
        assert torch.equal(ts.grad, te.grad), it
    assert len(Ls._static_plans) == 1
    # an in-place change of the event tensor is noticed as well (version counter)
    batch2['events'][0, :100, 0] += 1.0
    te = traj.to(dev).requires_grad_(True); ts = traj.to(dev).requires_grad_(True)
    le, _, _ = Le.calc(te, td, batch2); le.backward()
    ls, _, _ = Ls.calc(ts, td, batch2); ls.backward()
    assert torch.equal(ls.detach(), le.detach()) and torch.equal(ts.grad, te.grad)


def test_static_shapes_refuses_a_stale_backward_and_handles_new_shapes_and_no_grad():
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(7)
    Ls = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), static_shapes=True))
    Le = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    td = times.to(dev)
    t1 = traj.to(dev).requires_grad_(True)
    l1, _, _ = Ls.calc(t1, td, batch)
    t2 = (traj + 1).to(dev).requires_grad_(True)
    l2, _, _ = Ls.calc(t2, td, batch)
    with pytest.raises(RuntimeError, match='earlier calc'):
        l1.backward()
    l2.backward()
    assert torch.isfinite(t2.grad).all()
    # another shape (fewer events): its own plan; the first keeps working
    b3 = {'events': ev[:, :30000].contiguous().to(dev), 'num_pos_events': 15000}
    t3 = traj.to(dev).requires_grad_(True); t3e = traj.to(dev).requires_grad_(True)
    l3, _, _ = Ls.calc(t3, td, b3); l3.backward()
    l3e, _, _ = Le.calc(t3e, td, b3); l3e.backward()
    assert torch.equal(l3.detach(), l3e.detach()) and torch.equal(t3.grad, t3e.grad)
    assert len(Ls._static_plans) == 2
    # no gradient wanted (logging callback): forward-only plan
    with torch.no_grad():
        l4, _, m4 = Ls.calc(traj.to(dev), td, batch)
        l4e, _, m4e = Le.calc(traj.to(dev), td, batch)
    assert torch.equal(l4, l4e) and torch.equal(m4['iwes'], m4e['iwes'])


def test_static_shapes_replays_with_fresh_ordered_batches():
    """static_shapes=True with the library's own bucket-ordered ingest: every batch brings a NEW event tensor and a NEW offsets
    table of the same shape.  The plan is keyed on the shape only and copies the table in like the events: one plan, replayed,
    bit-identical to eager for both batches (the plans used to be keyed on the table's address: a capture per step)."""
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(8)
    _, ev2, _, _, _ = _c2(9)
    Le = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    Ls = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), static_shapes=True))
    td = times.to(dev)
    batches = [Le.order_events({'events': e.to(dev), 'num_pos_events': num_pos}) for e in (ev, ev2, ev, ev2)]
    assert batches[0]['event_offsets'].data_ptr() != batches[1]['event_offsets'].data_ptr()
    for it, b in enumerate(batches):
        te = (traj + 0.1 * it).to(dev).requires_grad_(True)
        ts = (traj + 0.1 * it).to(dev).requires_grad_(True)
        le, _, me = Le.calc(te, td, b); le.backward()
        ls, _, ms = Ls.calc(ts, td, b); ls.backward()
        assert torch.equal(ls.detach(), le.detach()), it
        assert torch.equal(ms['iwes'], me['iwes']) and torch.equal(ts.grad, te.grad), it
    assert len(Ls._static_plans) == 1


def test_small_steps_switch_to_the_captured_plan_by_themselves():
    """Automatic static shapes (FocusLoss.auto_static_shapes, default on): the third calc in a row of one small shape replays the
    captured plan -- same loss, images and gradient as the eager path bit for bit; the images are the caller's own copy; a
    second calc before the first one's backward takes the eager path (both backwards stay right); a shape that changes every step
    never leaves the eager path."""
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(7)
    cfg = bench.loss_config(wl)
    Le = LossFactory.get_loss_calculator('FOCUS', dict(cfg, auto_static_shapes=False))
    La = LossFactory.get_loss_calculator('FOCUS', cfg)
    assert La.auto_static_shapes and not Le.auto_static_shapes
    td = times.to(dev)
    _, ev2, _, _, _ = _c2(8)
    batches = [{'events': ev.to(dev), 'num_pos_events': num_pos}, {'events': ev2.to(dev), 'num_pos_events': num_pos}]
    keep = []
    for step in range(6):
        tj = (traj + 0.01 * step).to(dev)
        b = batches[step % 2]
        te, ta = tj.clone().requires_grad_(True), tj.clone().requires_grad_(True)
        le, loge, me = Le.calc(te, td, b)
        la, loga, ma = La.calc(ta, td, b)
        le.backward(); la.backward()
        assert torch.equal(le.detach(), la.detach()) and torch.equal(te.grad, ta.grad), step
        assert torch.equal(me['iwes'], ma['iwes']) and torch.equal(loge['focus_loss'], loga['focus_loss'])
        keep.append((ma['iwes'], me['iwes'].clone()))
    assert len(La._static_plans) == 1 and len(Le._static_plans) == 0        # captured once, from the third step on
    for got, want in keep:                                                   # the images of earlier steps are still theirs
        assert torch.equal(got, want)
    # two calcs of the same shape before any backward: the second one cannot replay over the first one's buffers
    t1, t2 = traj.to(dev).requires_grad_(True), (traj + 0.05).to(dev).requires_grad_(True)
    l1, _, _ = La.calc(t1, td, batches[0])
    l2, _, _ = La.calc(t2, td, batches[0])
    l2.backward(); l1.backward()
    r1, r2 = traj.to(dev).requires_grad_(True), (traj + 0.05).to(dev).requires_grad_(True)
    k1, _, _ = Le.calc(r1, td, batches[0]); k1.backward()
    k2, _, _ = Le.calc(r2, td, batches[0]); k2.backward()
    assert torch.equal(l1.detach(), k1.detach()) and torch.equal(t1.grad, r1.grad)
    assert torch.equal(l2.detach(), k2.detach()) and torch.equal(t2.grad, r2.grad)
    # batches of different lengths (what a loader delivers): no plan is ever built
    Lb = LossFactory.get_loss_calculator('FOCUS', cfg)
    for step in range(6):
        m = 40000 + 1000 * step
        tb = traj.to(dev).requires_grad_(True)
        lb, _, _ = Lb.calc(tb, td, {'events': ev[:, :m].contiguous().to(dev), 'num_pos_events': min(num_pos, m)})
        lb.backward()
    assert len(Lb._static_plans) == 0


def test_automatic_capture_survives_other_threads_and_streams():
    """The plan of the automatic mode is captured inside the caller's training step: in the reference's training process a
    DataLoader thread allocates pinned memory (src/modules/data_loading.py:141-142) and DDP runs side streams
    (scripts/flow_training.py:125-130) meanwhile.  The capture is thread-local, so neither may break the step; losses and gradients
    stay those of the eager path bit for bit."""
    import threading
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(11)
    cfg = bench.loss_config(wl)
    Le = LossFactory.get_loss_calculator('FOCUS', dict(cfg, auto_static_shapes=False))
    La = LossFactory.get_loss_calculator('FOCUS', cfg)
    td = times.to(dev)
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    stop = threading.Event()
    errors = []

    def pinner():                        # what a pin_memory thread does: pinned allocations + async copies on a stream of its own
        try:
            s = torch.cuda.Stream(dev)
            while not stop.is_set():
                h = torch.empty(1 << 18, dtype=torch.float32, pin_memory=True)
                with torch.cuda.stream(s):
                    d = h.to(dev, non_blocking=True)
                    d.add_(1.0)
                s.synchronize()
        except Exception as e:           # noqa: BLE001
            errors.append(e)

    def cruncher():                      # a side stream busy with matrix products (DDP / the network)
        try:
            s = torch.cuda.Stream(dev)
            a = torch.randn(1024, 1024, device=dev)
            while not stop.is_set():
                with torch.cuda.stream(s):
                    for _ in range(8):
                        a = (a @ a).clamp_(-1, 1)
                s.synchronize()
        except Exception as e:           # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=pinner), threading.Thread(target=cruncher)]
    for t in ths:
        t.start()
    try:
        for step in range(8):
            tj = (traj + 0.01 * step).to(dev)
            te, ta = tj.clone().requires_grad_(True), tj.clone().requires_grad_(True)
            le, _, me = Le.calc(te, td, batch); le.backward()
            la, _, ma = La.calc(ta, td, batch); la.backward()
            assert torch.equal(le.detach(), la.detach()) and torch.equal(te.grad, ta.grad), step
            assert torch.equal(me['iwes'], ma['iwes']), step
    finally:
        stop.set()
        for t in ths:
            t.join()
    assert not errors, errors
    # either the plan was captured (the usual outcome) or the shape was marked "never": both are fine, an exception is not
    assert len(La._static_plans) + len(La._auto_never) == 1


def test_automatic_capture_failure_falls_back_to_eager(monkeypatch):
    """If the plan cannot be built, calc() stays eager for that shape instead of raising out of the training step."""
    import bench
    from motionpriorcmax_amd import LossFactory, ops
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(12)
    cfg = bench.loss_config(wl)
    Le = LossFactory.get_loss_calculator('FOCUS', dict(cfg, auto_static_shapes=False))
    La = LossFactory.get_loss_calculator('FOCUS', cfg)
    calls = []

    def boom(*a, **k):
        calls.append(1)
        raise RuntimeError('capture failed (injected)')
    monkeypatch.setattr(ops, 'StaticFocusPlan', boom)
    td = times.to(dev)
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    for step in range(6):
        tj = (traj + 0.01 * step).to(dev)
        te, ta = tj.clone().requires_grad_(True), tj.clone().requires_grad_(True)
        le, _, _ = Le.calc(te, td, batch); le.backward()
        la, _, _ = La.calc(ta, td, batch); la.backward()
        assert torch.equal(le.detach(), la.detach()) and torch.equal(te.grad, ta.grad), step
    assert len(calls) == 1 and len(La._auto_never) == 1 and len(La._static_plans) == 0


def test_automatic_mode_second_backward_of_a_retained_graph():
    """backward(retain_graph=True), another calc of the same shape, then the first graph's backward again: the caller never asked
    for static shapes, so this must give what the eager path gives."""
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    wl, ev, num_pos, traj, times = _c2(13)
    cfg = bench.loss_config(wl)
    Le = LossFactory.get_loss_calculator('FOCUS', dict(cfg, auto_static_shapes=False))
    La = LossFactory.get_loss_calculator('FOCUS', cfg)
    td = times.to(dev)
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    for _ in range(3):                                       # get the shape onto the captured plan
        t0 = traj.to(dev).requires_grad_(True)
        l0, _, _ = La.calc(t0, td, batch); l0.backward()
    assert len(La._static_plans) == 1
    t1 = (traj + 0.02).to(dev).requires_grad_(True)
    l1, _, _ = La.calc(t1, td, batch)
    l1.backward(retain_graph=True)
    g_first = t1.grad.clone()
    t2 = (traj + 0.04).to(dev).requires_grad_(True)
    l2, _, _ = La.calc(t2, td, batch); l2.backward()         # the plan moves on
    t1.grad = None
    l1.backward()                                            # ... and the retained graph is walked again
    assert torch.equal(t1.grad, g_first)
    r1 = (traj + 0.02).to(dev).requires_grad_(True)
    k1, _, _ = Le.calc(r1, td, batch); k1.backward()
    assert torch.equal(g_first, r1.grad) and torch.equal(l1.detach(), k1.detach())
