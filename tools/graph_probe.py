"""Can calc + backward be captured into a HIP graph through torch.cuda.CUDAGraph? (diagnostics)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory
name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
wl = bench.WORKLOADS[name]
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev).requires_grad_(True)
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
td = times.to(dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        loss, _, _ = L.calc(t, td, batch); loss.backward(); 
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref_grad = t.grad.clone(); ref_loss = float(loss)
print('eager ok', ref_loss, flush=True)
from motionpriorcmax_amd import ops
if os.environ.get('STAGE'):
    ops.STAGE_TIMER = ops.StageTimer()
    for _ in range(3):
        loss, _, _ = L.calc(t, td, batch); loss.backward(); t.grad = None
    st = ops.STAGE_TIMER.summary(); ops.STAGE_TIMER = None
    print('stage timer pass done', flush=True)
if os.environ.get('NOWARN'):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
g = torch.cuda.CUDAGraph()
t.grad = None
with torch.cuda.graph(g):
    loss_g, _, _ = L.calc(t, td, batch)
    loss_g.backward()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('replay ok', float(loss_g), float((t.grad - ref_grad).abs().max()), flush=True)
for n in (20, 20, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); print('graph  %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / n))
for n in (20, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        loss, _, _ = L.calc(t, td, batch); loss.backward(); t.grad = None
    torch.cuda.synchronize(); print('eager  %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / n))
