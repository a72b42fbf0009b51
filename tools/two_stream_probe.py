"""Would splitting the batch over two streams pay? Two independent B=7 losses on two streams vs one B=14 (diagnostics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory
dev = torch.device('cuda:0')
def setup(B, seed):
    wl = dict(bench.WORKLOADS['C3']); wl['B'] = B
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=seed)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    return L, traj.to(dev).requires_grad_(True), times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos}
full = setup(14, 1)
parts = [setup(int(sys.argv[1]) if len(sys.argv) > 1 else 7, 2), setup(int(sys.argv[1]) if len(sys.argv) > 1 else 7, 3)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def step_full():
    L, t, td, b = full
    loss, _, _ = L.calc(t, td, b); loss.backward(); t.grad = None
def step_split():
    cur = torch.cuda.current_stream()
    for s_, (L, t, td, b) in zip(streams, parts):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            loss, _, _ = L.calc(t, td, b); loss.backward(); t.grad = None
    for s_ in streams:
        cur.wait_stream(s_)
def step_seq():
    for (L, t, td, b) in parts:
        loss, _, _ = L.calc(t, td, b); loss.backward(); t.grad = None
for name, fn in (('one stream, B=14', step_full), ('two streams, 2 x B/2', step_split), ('one stream, 2 x B/2', step_seq), ('one stream, B=14', step_full), ('two streams, 2 x B/2', step_split)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); print('%-24s %.4f ms' % (name, 1e3 * (time.perf_counter() - t0) / 30), flush=True)
