"""Which stage breaks repeated HIP-graph replay? (diagnostics)  usage: graph_probe2.py <what>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory, ops
what = sys.argv[1]
wl = bench.WORKLOADS['C2']
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev).requires_grad_(True)
evd, td = ev.to(dev), times.to(dev)
batch = {'events': evd, 'num_pos_events': num_pos}
lut0 = (torch.randn(1, wl['nb'], 120, 160, 1, 2, device=dev) * 2)

def body():
    if what == 'knn_fwd':
        with torch.no_grad():
            return ops.KnnLutFn.apply(t.detach(), L._cfg)[0]
    if what == 'knn_fwd_bwd':
        lut, _ = ops.KnnLutFn.apply(t, L._cfg); lut.sum().backward(); return lut
    if what == 'event_fwd':
        with torch.no_grad():
            return ops.EventFocusFn.apply(lut0, evd, td[:1], L._cfg, num_pos)[0]
    if what == 'event_fwd_bwd':
        lt = lut0.requires_grad_(True)
        f = ops.EventFocusFn.apply(lt, evd, td[:1], L._cfg, num_pos)[0]; f.backward(); return f
    if what == 'calc_fwd':
        with torch.no_grad():
            return L.calc(t.detach(), td, batch)[0]
    loss = L.calc(t, td, batch)[0]; loss.backward(); return loss

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
if t.grad is not None: t.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
for i in range(int(os.environ.get("REPLAYS", 30))):
    g.replay()
    torch.cuda.synchronize()
print(what, 'ok after replays', float(out.detach().float().abs().sum()))
import time
for mode in ('synced', 'back-to-back'):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        g.replay()
        if mode == 'synced':
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(what, mode, '%.4f ms per replay' % (1e3 * (time.perf_counter() - t0) / 40), flush=True)
