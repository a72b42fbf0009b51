"""The RCCL leg of the data-parallel plumbing on the 1-GPU box: a process group of ONE rank with backend
'nccl' (= RCCL) runs the same calls `bench.py --gpus N` makes -- group init bound to the device, bucketed
asynchronous all-reduce on the side stream, the fp64 MAX / SUM reductions of the timing -- and overlaps them
with a loss step.  (Groups of more than one rank are covered on CPU by tests/test_dp_gloo.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bench
from motionpriorcmax_amd import LossFactory, dp
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
red = dp.GradAllReducer(numel=3_000_001, n_buckets=4, device=dev, force_collective=True)
red.flat.copy_(torch.arange(red.flat.numel(), device=dev, dtype=torch.float32) % 1024)
want = red.flat.clone()
wl = dict(bench.WORKLOADS['C2'])
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=5)
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev).requires_grad_(True)
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
for _ in range(3):
    red.wait(); red.start()                       # overlaps the loss step below
    loss, _, _ = L.calc(t, times.to(dev), batch)
    loss.backward(); t.grad = None
red.wait()
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(red.flat, want), 'sum over one rank / 1 must be the identity'
assert dp.max_over_ranks(1.25, dev, force_collective=True) == 1.25
assert dp.sum_over_ranks(3.5, dev, force_collective=True) == 3.5
assert torch.isfinite(loss).item()
dist.destroy_process_group()
print('rccl-ok')
'''


def test_rccl_group_of_one_runs_the_bench_collectives():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29653', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'rccl-ok' in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


def test_bench_two_ranks_control_flow_over_gloo():
    """`bench.py --gpus 2` under torch.distributed.run with the debugging backend (both ranks on cuda:0, collectives
    over gloo): barriers, MAX/SUM timing reductions, rank-0-only JSON line, the second loop with the gradient
    all-reduce.  (The measured configuration uses RCCL; this covers the multi-rank control flow on a 1-GPU box.)"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPC_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    # plain `python bench.py --gpus 2`: the parent launches the two ranks itself and relays rank 0's line
    env = {k: v for k, v in env.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--workload', 'C2']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 0 and d['scaling'] == 'weak' and d['config']['global_batch'] == 2 and d['value'] > 0
    assert len(d['blocks_ms_per_step']) == 3
    assert d['dp_with_grad_allreduce']['grad_allreduce_MB'] > 100 and d['dp_with_grad_allreduce']['value'] > 0
