"""`python bench.py --gpus N` launches its own ranks (reference: scripts/flow_training.py:125-128 fans out from one
command with `devices=args.gpus`).  Checked here without a GPU through the dry-run mode (MPC_BENCH_DRYRUN=1: the same
fan-out, process group, barriers and MAX/SUM reductions over gloo with an empty step)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra)
    return subprocess.run([sys.executable, 'bench.py'] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_rank_env_launches_two_ranks():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1'], {'MPC_BENCH_DRYRUN': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_counted'] == 2 and d['config']['parallelism'] == 'dp2'
    assert d['dry_run'] is True and d['value'] == 0.0        # a dry run can never pass for a measurement


def test_world_size_must_match_gpus():
    # a rank environment whose WORLD_SIZE disagrees with --gpus is an error, not a silent 1-GPU run
    r = _run(['--gpus', '4', '--steps', '1', '--warmup', '0'],
             {'MPC_BENCH_DRYRUN': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_more_gpus_than_visible_is_refused_by_the_parent():
    # no dry run: the parent counts the devices and refuses (it starts no rank)
    import torch
    if torch.cuda.device_count() >= 64:
        return
    r = _run(['--gpus', '64', '--steps', '1', '--warmup', '0'], {})
    assert r.returncode != 0 and 'GPU(s) visible' in (r.stderr + r.stdout)


def test_failing_rank_fails_the_parent():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--workload', 'C2'],
             {'MPC_BENCH_BACKEND': 'gloo'}, timeout=600)           # ranks need a GPU: here they exit non-zero
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
