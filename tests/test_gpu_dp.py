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


CHILD_PRODUCER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bench
from motionpriorcmax_amd import LossFactory, dp
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
red = dp.GradAllReducer(device=dev, force_collective=True)
fired = []
sb = red.start_bucket
red.start_bucket = lambda i: (fired.append(i), sb(i))
prod = dp.OverlappedGradProducer(red, rows=256)
assert sum(p.numel() for p in prod.net.parameters()) == dp.UNET_GRAD_NUMEL == red.flat.numel() == 31044610
# the gradient of one step without the exchange
red.skip = True
prod.step(); torch.cuda.synchronize()
own = red.flat.clone()
red.skip = False
wl = dict(bench.WORKLOADS['C2'])
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=5)
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev).requires_grad_(True)
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
ref = None
for step in range(3):
    loss, _, _ = L.calc(t, times.to(dev), batch)       # the next loss overlaps the buckets of the step before
    loss.backward(); t.grad = None
    prod.wait()
    if step > 0:
        assert torch.allclose(red.flat, own, rtol=1e-5, atol=1e-8), float((red.flat - own).abs().max())      # all-reduce over one rank / 1
    fired.clear()
    prod.step()
    assert fired == [0, 1, 2, 3], fired
    ref = loss.detach().clone() if ref is None else ref
    assert torch.equal(loss.detach(), ref)               # the loss is untouched by what runs beside it
prod.wait(); torch.cuda.synchronize()
assert torch.allclose(red.flat, own, rtol=1e-5, atol=1e-8)
dist.destroy_process_group()
print('PRODUCER_OK')
'''


def test_gradient_producer_fills_and_exchanges_the_buckets_beside_the_loss():
    """dp.OverlappedGradProducer on the device with an RCCL group of one rank: the stand-in network of 31 044 610 parameters writes its
    gradients into the reducer's flat buffer (views), every bucket is all-reduced once per step in production order on the side
    stream, and the loss steps that run beside it are bitwise what they are alone."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29657', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', CHILD_PRODUCER], capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and 'PRODUCER_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


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


def test_event_shards_sum_to_the_single_rank_image_bit_for_bit():
    """SURVEY.md 8e, optional finer split (dp.event_sharded_calc): on the device, the Q33.30 images of the row shards of a
    batch add up -- as integers, in any order -- to the image of all rows, and mpc_iwe_from_fixed of the sum is the fp32 raw
    IWE of mpc_event_splat_fwd bit for bit; the sharded loss with no process group equals FocusLoss.calc."""
    import torch
    import bench
    from motionpriorcmax_amd import LossFactory, dp, ops
    dev = torch.device('cuda:0')
    wl = dict(bench.WORKLOADS['C2'])
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=9)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    cfg = L._cfg
    evd, trajd, td = ev.to(dev), traj.to(dev), times.to(dev)
    shape = ops.make_shape(cfg, 1, ev.shape[1], num_pos, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    lut, _, _, _ = ops.knn_lut_fwd(cfg, shape, trajd, ws)
    raw = ops.event_splat_fwd(shape, evd, lut, td[:1], ws)
    full = ops.event_splat_fwd_fixed(shape, evd, lut, td[:1], ws)
    assert torch.equal(ops.iwe_from_fixed(full), raw)
    for world in (2, 3):
        acc = torch.zeros_like(full)
        for r in reversed(range(world)):
            e, npos = dp.shard_event_rows(evd, num_pos, r, world)
            sh = ops.make_shape(cfg, 1, e.shape[1], npos, traj.shape[2])
            acc += ops.event_splat_fwd_fixed(sh, e, lut, td[:1], ops.alloc_workspace(sh, dev))
        assert torch.equal(acc, full), world
    # the sharded loss without a process group = the plain loss
    t1 = trajd.clone().requires_grad_(True); t2 = trajd.clone().requires_grad_(True)
    l1, log1, m1 = L.calc(t1, td, {'events': evd, 'num_pos_events': num_pos}); l1.backward()
    l2, log2, m2 = dp.event_sharded_calc(L, t2, td, {'events': evd, 'num_pos_events': num_pos}); l2.backward()
    assert torch.equal(l1.detach(), l2.detach()) and torch.equal(m1['iwes'], m2['iwes'])
    assert float((t1.grad - t2.grad).norm() / t1.grad.norm()) < 1e-6


SHARD_CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bench
from motionpriorcmax_amd import LossFactory, dp
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('gloo', rank=rank, world_size=world)
wl = dict(bench.WORKLOADS['C2'])
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=11)
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
evd, td = ev.to(dev), times.to(dev)
e, npos = dp.shard_event_rows(evd, num_pos, rank, world)
t = traj.to(dev).requires_grad_(True)
loss, log, misc = dp.event_sharded_calc(L, t, td, {'events': e, 'num_pos_events': npos})
loss.backward()
t1 = traj.to(dev).requires_grad_(True)
l1, _, m1 = L.calc(t1, td, {'events': evd, 'num_pos_events': num_pos})
l1.backward()
torch.cuda.synchronize()
assert torch.equal(loss.detach(), l1.detach()), (loss.item(), l1.item())
assert torch.equal(misc['iwes'], m1['iwes'])
rel = float((t.grad - t1.grad).norm() / t1.grad.norm())
assert rel < 1e-5, rel
dist.barrier()
dist.destroy_process_group()
print('shard-ok', rank)
'''


def test_event_axis_split_two_ranks_on_one_gpu():
    """B = 1 on two ranks (configs[1] cannot use a second GPU by batch sharding): both ranks share cuda:0 here, the two
    exchange steps run over gloo; loss and IWEs equal the single-rank ones bit for bit, the gradient to fp32 rounding."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29671', RANK=str(r), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', SHARD_CHILD], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f'shard-ok {r}' in so, (so[-500:], se[-1500:])
