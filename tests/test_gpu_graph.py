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
