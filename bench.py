#!/usr/bin/env python3
"""Benchmark of the CMax loss hot path (BASELINE.json metric: Mevents/s through the CMax loss
forward + backward, DSEC 480x640).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C4]

A step = FocusLoss.calc(trajectories, times, batch) + backward to `trajectories` on one synthetic
batch that is already resident in HBM.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank processes its own shard of the batch (weak scaling, no
data-path collective) and the averaged network gradient (124 MB fp32, the reference UNet) is
all-reduced over RCCL on a side stream, overlapped with the next step's loss.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
`roofline` (dominant kernel, algorithmic bytes / HIP-event time / 8 TB/s) and `cpu_baseline`
(the CPU oracle timed on the host cores on a bounded sample)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)

WORKLOADS = {
    # name: (B per GPU, events per sample, num_bins, num_basis, smooth_type, smooth_weight, description)
    'C2': dict(B=1, M=50_000, nb=15, k=1, smooth_type='on_flow_to_tref', smooth_weight=0.003,
               desc='DSEC 480x640, 50k events/window, poly-k1, batch=1 (BASELINE configs[1])'),
    'C3': dict(B=14, M=200_000, nb=15, k=3, smooth_type='on_flow_to_tref', smooth_weight=0.003,
               desc='DSEC 480x640, 200k events/window, poly-k3, batch=14 per GPU (BASELINE configs[2]; '
                    'the per-GPU shard of configs[4] = batch 112 on 8 GPUs)'),
    'C4': dict(B=1, M=500_000, nb=41, k=10, smooth_type='on_flow_to_next', smooth_weight=0.06,
               desc='EVIMO2 480x640, 500k events/window, Bezier degree 10, 41 bins, batch=1 (BASELINE configs[3])'),
    'C4b6': dict(B=6, M=500_000, nb=41, k=10, smooth_type='on_flow_to_next', smooth_weight=0.06,
                 desc='EVIMO2 480x640, 500k events/window, Bezier degree 10, 41 bins, batch=6 (configs[3] at the batch size of '
                      'its yaml: ...Tab2L5.yaml training.batch_size)'),
}
H, W, SP, PATCH, KNN = 480, 640, 4, 4, 32


def loss_config(wl):
    return dict(image_shape=(H, W), num_tref=1, num_bins=wl['nb'], num_knn=KNN,
                smooth_weight=wl['smooth_weight'], lut_superpixel_size=SP, focus_loss_norm='l1',
                dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
                polarity_aware_batching=True, interpolation_scheme='mean', smooth_type=wl['smooth_type'])


def synth_inputs(wl, seed, B=None, trefs=(0.41,)):
    """Seeded synthetic batch: events [B,M,6] + trajectories [B,len(trefs)+nb,n,2] on CPU."""
    from motionpriorcmax_amd import utils
    from motionpriorcmax_amd.utils.synth import synth_events, bin_mid_times
    B = wl['B'] if B is None else B
    nb, k = wl['nb'], wl['k']
    ev, num_pos = synth_events(B, wl['M'], (H, W), nb, seed=seed, pad_frac=0.02, time_sorted=True)
    g = torch.Generator().manual_seed(seed + 7)
    times = torch.cat((torch.tensor(list(trefs)), bin_mid_times(nb)))
    mask = utils.get_optical_flow_tile_mask((H, W), PATCH)
    if wl['k'] == 10:
        # Bezier control points N(0, 2^2) per tile, (x, y) channel order (bezier.py / polynomial.py:60-61)
        params = torch.randn(B, 2 * k, H // PATCH, W // PATCH, generator=g) * 2.0
        traj, _ = utils.trajectories_from_bezier(params, times, PATCH, (H, W))
    else:
        sigma = 3.0 if k == 1 else 1.0
        coeff = torch.randn(B, 1, 2 * k, H, W, generator=g) * sigma
        coeffs, pos, _ = utils.coeffs_grid_to_list(coeff, mask, num_coeffs=k)
        traj = utils.compute_basis(coeffs, times, k, 'polynomial') - \
            utils.compute_basis(coeffs, torch.zeros(1), k, 'polynomial')
        traj = (traj + pos[None, :, None, :]).permute(0, 2, 1, 3).contiguous()
    return ev, num_pos, traj.contiguous(), times


def algorithmic_bytes(B, M, nb, T=1, P=2):
    """events read once per pass, IWE written + read once, LUT read + dLUT written."""
    hq, wq = H // SP, W // SP
    return 48 * B * M + 8 * B * T * P * H * W + 16 * B * nb * hq * wq * T


def stage_bytes(B, M, nb, n, T=1, P=2):
    """Algorithmic HBM bytes per C-ABI stage ('Kernels and their rooflines')."""
    Q = (H // SP) * (W // SP)
    lut = 8 * B * nb * Q * T
    img = 4 * B * T * P * H * W
    traj = 8 * B * (T + nb) * n
    return {
        'mpc_knn_lut_fwd': traj + lut,
        'mpc_knn_lut_bwd': lut + traj,
        'mpc_lut_smooth': lut,
        'mpc_event_splat_fwd': 24 * B * M + lut + img,
        'mpc_contrast_fwd': 2 * img,
        'mpc_event_splat_bwd': 24 * B * M + lut,
        'mpc_finalize': 0, 'mpc_scale': 0,
    }


# Kernels behind each C-ABI stage.  Not a hand-kept table: the kernels are the ones the committed rocprofv3 kernel-trace
# summary of the workload lists (profiles/r<NN>_rocprofv3_kernel_stats_<workload>.csv, newest round present), attributed to
# a stage by the naming rule of csrc/ (one prefix per stage), and the dominant kernel of a stage is the one with the
# largest total time in that summary.
STAGE_PREFIXES = (          # first match wins
    ('mpc_knn_lut_bwd', ('k_knn_bwd', 'k_knn_reach')),
    ('mpc_knn_lut_fwd', ('k_knn_',)),
    ('mpc_event_splat_bwd', ('k_lut_accum', 'k_splat_bwd', 'k_lut_overflow')),
    ('mpc_event_splat_fwd', ('k_ev_bin', 'k_iwe_', 'k_splat_fwd')),
    ('mpc_lut_smooth', ('k_lut_smooth',)),
    ('mpc_contrast_fwd', ('k_contrast', 'k_image_means')),
    ('mpc_finalize', ('k_finalize',)),
    ('mpc_scale', ('k_scale',)),
)


def stage_of_kernel(name):
    for stage, prefixes in STAGE_PREFIXES:
        if any(name.startswith(p) for p in prefixes):
            return stage
    return None


def _short_kernel_name(full):
    return full.split('(')[0].replace('void ', '').split('<')[0].strip()


def profile_kernels(workload):
    """{stage: [(kernel, total_ns, avg_ns, calls), ...] sorted by total time} from the newest committed kernel-trace summary of
    `workload`, and the file it came from; ({}, None) if there is none."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_rocprofv3_kernel_stats_{workload}.csv')))
    if not files:
        return {}, None
    agg = {}
    for r in csv.DictReader(open(files[-1])):
        k = _short_kernel_name(r['Name'])
        st = stage_of_kernel(k)
        if st is None:
            continue
        t = agg.setdefault(st, {}).setdefault(k, [0.0, 0])
        t[0] += float(r['TotalDurationNs']); t[1] += int(r['Calls'])
    out = {st: sorted(((k, v[0], v[0] / max(v[1], 1), v[1]) for k, v in ks.items()), key=lambda x: -x[1]) for st, ks in agg.items()}
    return out, os.path.relpath(files[-1], ROOT)


def pmc_traffic(workload, stage, launches=None):
    """HBM-side bytes per step of `stage` from the committed rocprofv3 PMC summary of this workload
    (profiles/traffic_<workload>.json, tools/summarize_pmc.py: separate FETCH_SIZE and WRITE_SIZE passes), summed over
    the kernels the kernel-trace summary attributes to the stage.  Returns (upper, lower): the read counter doubled as the
    calibration in the same file prescribes for reads served from HBM (gfx950 tallies 128-byte requests at 64), and the raw
    counters -- a kernel whose reads hit the Infinity Cache lies between the two.  (None, None) without a summary."""
    f = os.path.join(ROOT, 'profiles', f'traffic_{workload}.json')
    ks, _ = profile_kernels(workload)
    if not os.path.exists(f) or stage not in ks:
        return None, None
    d = json.load(open(f))
    up = lo = 0.0
    for k, _, _, calls in ks[stage]:
        if k in d:
            per_step = (launches or {}).get(k, 1)         # launches of the kernel per step, from the live kernel timer
            up += per_step * (d[k]['hbm_bytes_upper'] if 'hbm_bytes_upper' in d[k] else d[k]['hbm_bytes_per_launch'])
            lo += per_step * (d[k]['hbm_bytes_lower'] if 'hbm_bytes_lower' in d[k] else d[k]['hbm_bytes_per_launch_raw'])
    return (round(up), round(lo)) if up > 0 else (None, None)


def knn_ceiling(workload, stages, B, nb):
    """The KNN stages against the ceiling that applies to them (VALU issue, not HBM): queries per second from the LIVE stage
    times; and, clearly marked as FROM THE COMMITTED PROFILE (not re-measured by this run, and only valid for the library
    that profile was taken with), the share of the chip's vector-instruction issue slots their dominant kernels used
    (profiles/r<NN>_sq_<workload>.json: SIMD cycles available per SQ_INSTS_VALU = duration x SIMDs x clock / instructions; SIMD
    count and clock from the device properties when a GPU is present)."""
    import glob
    Q = (H // SP) * (W // SP)
    out = {}
    fwd = stages.get('mpc_knn_lut_fwd'); bwd = stages.get('mpc_knn_lut_bwd')
    if fwd and fwd['us_per_step'] > 0:
        out['queries_per_s'] = round(B * nb * Q / (fwd['us_per_step'] * 1e-6), 1)
        out['fwd_us'] = round(fwd['us_per_step'], 1)
    if bwd and bwd['us_per_step'] > 0:
        out['bwd_us'] = round(bwd['us_per_step'], 1)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_sq_{workload}.json')))
    if files:
        d = json.load(open(files[-1]))
        simds, ghz = 1024, 2.4
        try:
            pr = torch.cuda.get_device_properties(0)
            simds = pr.multi_processor_count * 4
            ghz = getattr(pr, 'clock_rate', 2400000) / 1e6
        except Exception:
            pass
        prof = {}
        for key, k in d.items():
            if isinstance(k, dict) and stage_of_kernel(key) in ('mpc_knn_lut_fwd', 'mpc_knn_lut_bwd') and k.get('kernel_us', 0) > 20:
                # SIMD cycles (at the nominal clock) the launch had per vector instruction it issued.  Measured issue
                # costs on this chip in shader-clock cycles (profiles/r03_ubench_op_cycles.txt): 1.65 for plain fp32 / integer
                # add, sub, mul, and, xor; 2.9 for nearly everything else (fma with three registers 2.5); 5.5 for rcp / sqrt --
                # a kernel near 3-4 nominal cycles per instruction (the clock under load is below nominal) is bound by what it issues
                slots = k['kernel_us'] * 1e-6 * ghz * 1e9 * simds
                prof[key] = {'valu_wave_instr_per_launch': k['valu_insts'], 'kernel_us_in_profile': k['kernel_us'],
                             'simd_cycles_per_valu_instr': round(slots / k['valu_insts'], 2),
                             'valu_issue_frac_at_4_cycles': round(k['valu_insts'] * 4.0 / slots, 3),
                             'lds_active_frac': k.get('lds_active_frac'), 'wait_any_frac': k.get('wait_any_frac')}
        # the counters belong to the library they were taken with: say so when the loaded one differs (stale figures)
        from motionpriorcmax_amd import build as _build
        cur = 'libmpcmax.so of sources ' + _build.source_hash()
        if d.get('_library') != cur:
            out['from_committed_profile'] = {'file': os.path.relpath(files[-1], ROOT), 'library': d.get('_library'), 'loaded_library': cur,
                                             'stale': True, 'note': 'taken with another build of the library: counters omitted'}
            return out
        out['from_committed_profile'] = {'file': os.path.relpath(files[-1], ROOT), 'library': d.get('_library'),
                                         'assumes': f'{simds} SIMDs at {ghz:.2f} GHz (nominal); valu_issue_frac_at_4_cycles can exceed 1: fp32 '
                                                    'instructions issue in fewer than 4 cycles',
                                         'kernels': prof}
        # (round 6) the INSTRUCTION-MIX floor of the two issue-bound kernels: their own dynamic mix (SQ class counters of the same
        # committed profile) priced with the measured issue cost of every instruction class (tools/valu_floor.py, which also
        # disassembles the loaded library for the mix inside a class) -- the ceiling the 2-cycle figure is not
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import valu_floor as _vf
            fl = {}
            for kname in ('k_knn_strip', 'k_knn_bwd_tile'):
                if kname in d:
                    r = _vf.floor_of(kname, 'false, false', d[kname])
                    fl[kname] = {k2: r[k2] for k2 in ('floor_us', 'two_cycle_ideal_us', 'kernel_us_in_profile', 'achieved_over_floor', 'mean_issue_cycles',
                                                    'dynamic_valu_wave_instructions', 'mix_source')}
            out['valu_floor'] = dict(fl, note='floor_us = sum over instruction classes (dynamic count x measured issue cycles, profiles/r03_ubench_op_cycles.txt) '
                                              '/ (SIMDs x clock); achieved_over_floor = floor_us / measured kernel time; tools/valu_floor.py')
        except Exception as e:          # noqa: BLE001
            out['valu_floor'] = {'error': repr(e)[:200]}
    return out


def cpu_baseline(wl, budget_s=20.0):
    """The CPU oracle (a restatement of the reference's PyTorch CPU path, `kind: port`) timed on
    the host cores on a bounded sample of the same workload: the event path (warp -> IWE ->
    objective -> autograd backward to the LUT) on ONE sample of the batch, plus the brute-force
    KNN LUT on a slice of the query cells of one (sample, bin), extrapolated linearly to the
    whole batch (the KNN cost does not depend on the event count)."""
    from oracle import focus_oracle as O
    ncpu = os.cpu_count() or 1
    ev, num_pos, traj, times = synth_inputs(wl, seed=0, B=1)
    cfg = loss_config(wl)
    Lo = O.FocusLossOracle(**cfg)
    nb = wl['nb']
    hq, wq = H // SP, W // SP
    g = torch.Generator().manual_seed(5)
    lut = torch.randn(1, nb, hq, wq, 1, 2, generator=g)

    def event_step():
        lt = lut.clone().requires_grad_(True)
        f, _, _ = Lo.event_path(ev, lt, times[:1], num_pos)
        (f + Lo.smooth_loss(lt, None) if cfg['smooth_type'] == 'on_flow_to_tref' else f).backward()
    # torch's CPU scatter/conv ops scale badly past a few dozen threads (36 s per step with 256
    # threads on the GPU host vs 0.24 s with 8): give the CPU its best thread count
    best = None
    tried = {}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        event_step()
        ts_ = []
        for _ in range(3):                       # median of three timed steps per candidate (one step is at the mercy of a host stall)
            t0 = time.perf_counter()
            event_step()
            ts_.append(time.perf_counter() - t0)
            if ts_[-1] > 5.0:
                break
        dt1 = sorted(ts_)[len(ts_) // 2]
        tried[nt] = round(1e3 * dt1, 1)
        if best is None or dt1 < best[0]:
            best = (dt1, nt)
        if dt1 > 5.0:
            break
    cores = best[1]
    torch.set_num_threads(cores)
    event_step()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < budget_s * 0.4 and reps < 20):
        event_step()
        reps += 1
    t_event = (time.perf_counter() - t0) / reps
    # KNN: ONE WHOLE (sample, bin) -- every query cell against every trajectory point -- selected with a K-min scan
    # (torch.topk; what KeOps' argKmin does), then x num_bins for the sample.  The KNN cost does not depend on the
    # event count.
    grid, _, _ = O.lut_grid_points((H, W), SP)
    pts = traj[0, 1]
    O.knn_indices(pts, grid[:512], KNN, 'l2', kmin=True)
    t0 = time.perf_counter()
    O.knn_indices(pts, grid, KNN, 'l2', q_chunk=1024, kmin=True)
    t_knn_bin = time.perf_counter() - t0
    t_knn_sample = t_knn_bin * nb
    valid = float(ev[..., 5].sum())
    return {
        # `value` = the WHOLE path on the CPU (KNN LUT + event path), the counterpart of the GPU `value`; the event path alone beside it
        'value': valid / (t_event + t_knn_sample) / 1e6, 'unit': 'Mevents/s', 'cores': cores, 'host_cpu_count': ncpu, 'kind': 'port',
        'cores_note': f'`cores` = the torch thread count that ran the timed steps (the fastest of those tried below); the host has {ncpu} logical CPUs',
        'sample': (f'1 sample of the batch ({int(valid)} valid events): brute-force K-min KNN (torch.topk; what KeOps argKmin computes) timed '
                   f'on one whole (sample, bin) = {t_knn_bin:.2f} s, x{nb} bins = {t_knn_sample:.1f} s per sample, + the event path fwd+bwd (LUT '
                   f'given) timed {reps}x = {t_event * 1e3:.1f} ms per step'),
        'event_path_value': valid / t_event / 1e6,
        'event_path_note': 'warp -> IWE -> objective -> autograd backward to the LUT only (no KNN): NOT the counterpart of the GPU `value`',
        'threads_tried_ms_per_event_step': tried,
    }


def _in_rank_env():
    """True inside a rank started by torch.distributed.run (which exports RANK and WORLD_SIZE)."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process becomes the PARENT of the run,
    as `scripts/flow_training.py:125-128` fans out from one command (`devices=args.gpus`).  It starts N fresh ranks
    (`python -m torch.distributed.run --nproc-per-node N bench.py ...`, one per GPU, rendezvous on 127.0.0.1),
    relays rank 0's JSON line (the line that carries the "metric" key) and exits non-zero if any rank fails.  The parent
    only COUNTS the devices (torch.cuda.device_count(); on a ROCm build without amdsmi that call may initialise the HIP
    runtime in the parent -- harmless, because the ranks are fresh child processes started with subprocess, never an exec of
    this one) and launches nothing on a GPU itself."""
    import subprocess
    dry = os.environ.get('MPC_BENCH_DRYRUN') == '1'
    shared_gpu = os.environ.get('MPC_BENCH_BACKEND', 'nccl') != 'nccl'
    if not dry and not shared_gpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node')
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    if r.returncode != 0 or len(lines) != 1:
        sys.stderr.write(r.stdout[-4000:])
        raise SystemExit(r.returncode if r.returncode != 0 else 1)
    print(lines[0])
    raise SystemExit(0)


def dry_run(args, rank, world):
    """MPC_BENCH_DRYRUN=1: the launch and timing CONTROL FLOW only (rank fan-out, process group, barriers, MAX/SUM
    over ranks, rank-0-only line) on CPU over gloo with an empty step -- what tests/test_bench_launch.py checks in
    the GPU-less container.  No kernel runs: `value` is 0 and the line says so."""
    from motionpriorcmax_amd import dp
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    blocks = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(1e-4)
        if world > 1:
            dist.barrier()
        blocks.append(dp.max_over_ranks(time.perf_counter() - t0))
    ranks = dp.sum_over_ranks(1.0)
    if rank == 0:
        wl = WORKLOADS[args.workload]
        print(json.dumps({'metric': 'Mevents/s through CMax loss fwd+bwd, DSEC 480x640', 'value': 0.0, 'unit': 'Mevents/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(1e3 * sorted(blocks)[1] / args.steps, 4), 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'none (dry run)', 'dry_run': True,
                          'ranks_counted': int(ranks), 'rccl_ranks': 0,
                          'config': {'workload': args.workload, 'global_batch': wl['B'] * world, 'parallelism': f'dp{world}'}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='C3', choices=sorted(WORKLOADS))
    ap.add_argument('--grad-allreduce', action='store_true',
                    help='N>1: put the overlapped 124 MB network-gradient all-reduce inside the main timed loop '
                         '(default: the loss path alone is timed -- it has no data-path collective -- and the '
                         'variant with the all-reduce is timed in a second loop and reported beside it)')
    ap.add_argument('--events-layout', default='time', choices=['bucket', 'time'],
                    help='row order of the event tensor the timed steps run on: the reference loader\'s time order (default: the layout '
                         'whose cost is counted is the one that is timed) or as the library\'s ingest can deliver it (rows of a polarity '
                         'block grouped by (time bin, LUT strip) + offsets table; the ordering is then NOT inside the timed step)')
    ap.add_argument('--batches', type=int, default=4,
                    help='distinct resident batches (events + trajectories) the timed steps rotate over: a training loop sees a new batch '
                         'every step; 1 = every step on the same batch, as rounds 1-4 timed it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-realistic', action='store_true', help='skip the realistic-input variants reported in also.realistic_inputs')
    ap.add_argument('--no-hip-graph', action='store_true', help='skip the HIP-graph replay timing reported beside the eager one')
    ap.add_argument('--also', default='C2,C4,C4b6', help='extra workloads reported in the "also" field (N=1 only)')
    ap.add_argument('--no-nondefault', action='store_true', help='skip the timing of the non-default loss settings (profiling runs: '
                    'their kernels share names with the headline\'s)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and not _in_rank_env():
        launch_ranks(args, sys.argv[1:])          # does not return
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` '
                         f'(it launches its own ranks) or under torch.distributed.run with --nproc-per-node N')
    if os.environ.get('MPC_BENCH_DRYRUN') == '1':
        return dry_run(args, rank, world)
    # ONE line on stdout, whatever the libraries print: file descriptor 1 points at stderr for the whole run (RCCL's version banner
    # -- C stdio, flushed when the process exits, i.e. BEHIND anything printed here -- landed behind the JSON line of a round-5 run),
    # and the line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the CMax path has no CPU fallback')
    # MPC_BENCH_BACKEND=gloo: debugging aid for boxes with fewer GPUs than ranks -- the ranks share cuda:0 and the
    # collectives (barriers, timing reductions, the gradient all-reduce) run over gloo on host tensors, so the
    # multi-rank control flow can be exercised on a 1-GPU box.  The measured configuration is always RCCL.
    backend = os.environ.get('MPC_BENCH_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    comm_dev = dev if backend == 'nccl' else torch.device('cpu')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    rccl_ranks = (dist.get_world_size() if world > 1 else 1) if backend == 'nccl' else 0     # ranks in the RCCL group

    from motionpriorcmax_amd import LossFactory, ops, dp

    def run_workload(name, steps, warmup, with_comm, instrument=True, force_group=False, net_only=False):
        wl = WORKLOADS[name]
        L = LossFactory.get_loss_calculator('FOCUS', loss_config(wl))
        # The timed steps ROTATE over `args.batches` distinct resident batches (events + trajectories; C3: 4 x 101 MB), as a training
        # loop sees a new batch every step (src/loader/dsec/loader.py:417-427): no step finds its inputs in the 256 MB Infinity
        # Cache because the step before read them.  The same steps on ONE batch are timed beside it (`single_batch`).
        nbat = max(1, args.batches)
        sets = []
        for j in range(nbat):
            ev_j, num_pos, traj_j, times = synth_inputs(wl, seed=1000 * rank + 1 + 17 * j)
            bt_j = {'events': ev_j.to(dev), 'num_pos_events': num_pos}
            sets.append({'batch_time': bt_j, 'traj': traj_j.to(dev).requires_grad_(True), 'valid': float(ev_j[..., 5].sum())})
            if j == 0:
                ev, traj = ev_j, traj_j
        times_d = times.to(dev)
        evd, trajd, batch_time = sets[0]['batch_time']['events'], sets[0]['traj'], sets[0]['batch_time']
        # The timed steps run on the reference loader's time-ordered tensor (the layout whose cost is counted is the layout that
        # is timed).  The layout the library's own ingest can deliver instead (utils.ingest_events(order_for=loss): the rows of
        # each polarity block grouped by (time bin, LUT strip) + the offsets table) is timed beside it (`other_event_layout`,
        # the ordering outside the step); same loss and gradient bit for bit (tests/test_gpu_event_order.py).
        for st_ in sets:
            st_['batch'] = L.order_events(st_['batch_time']) if args.events_layout == 'bucket' else st_['batch_time']
        batch = sets[0]['batch']
        valid_local = sum(sets[i % nbat]['valid'] for i in range(steps)) / steps      # valid events of an average timed step
        # with_comm: the data-parallel training step's one exchange, with a REAL producer -- a stand-in network of the reference UNet's
        # 31 044 610 parameters (dp.StandInNetwork; the UNet itself is out of scope) runs forward + backward behind the loss; its
        # gradients are views into the reducer's flat buffer, and every bucket is all-reduced (RCCL, side stream) the moment its last
        # gradient is written, overlapping the rest of that backward and the next step's loss (dp.OverlappedGradProducer).
        # `force_group`: a process group of ONE rank (N = 1) still goes through the RCCL call path.
        reducer = dp.GradAllReducer(device=comm_dev, force_collective=force_group) if (with_comm and (world > 1 or force_group)) else None
        producer = dp.OverlappedGradProducer(reducer) if (reducer is not None and comm_dev.type == 'cuda') else None
        if net_only:                      # (the same step with the network but without any exchange: what the exchange itself costs)
            producer = dp.OverlappedGradProducer(dp.GradAllReducer(device=comm_dev))
        counter = [0]

        def step(rotate=True):
            st_ = sets[counter[0] % nbat] if rotate else sets[0]
            counter[0] += 1
            loss, _, _ = L.calc(st_['traj'], times_d, st_['batch'])
            loss.backward()
            st_['traj'].grad = None
            if producer is not None:
                producer.wait()      # the previous step's all-reduce must be done before "the optimizer" (and before the gradients are zeroed)
                producer.step()      # network forward + backward; buckets are exchanged as they fill, overlapping the next step's loss
            elif reducer is not None:
                reducer.wait()
                reducer.start()      # (gloo debugging backend: the flat buffer as it is)
            return loss

        # set-up, not part of the W warm-up steps: bring the caching allocator to its steady state (the
        # per-step workspaces are first hipMalloc'ed over several steps) and the device out of its idle clocks
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        # EXACTLY `steps` steps per timed block, each block bracketed by barrier + synchronize and reduced with
        # MAX over ranks; three consecutive blocks, the MEDIAN one is reported (a fresh box showed one-off host
        # stalls of ~15 ms that land in one block); all block times go into the JSON line
        blocks, own_blocks = [], []
        for _ in range(3):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            counter[0] = 0           # (every block: the same `steps` steps, batch 0 first)
            t0 = time.perf_counter()
            for _ in range(steps):
                last = step()
            if producer is not None:
                producer.wait()
            elif reducer is not None:
                reducer.wait()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            own = time.perf_counter() - t0
            own_blocks.append(own)
            blocks.append(dp.max_over_ranks(own, comm_dev))
        dt = sorted(blocks)[1]
        total_valid = dp.sum_over_ranks(valid_local, comm_dev)
        # the same number of steps on ONE resident batch (what rounds 1-4 timed), beside the rotating figure
        single = None
        if nbat > 1 and reducer is None and producer is None:
            sb = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    last1 = step(rotate=False)
                torch.cuda.synchronize()
                sb.append(time.perf_counter() - t0)
            s_dt = dp.max_over_ranks(sorted(sb)[1], comm_dev)
            single = {'ms_per_step': round(1e3 * s_dt / steps, 4), 'value': round(dp.sum_over_ranks(sets[0]['valid'], comm_dev) * steps / s_dt / 1e6, 3),
                      'unit': 'Mevents/s', 'note': 'every timed step on the same resident batch (inputs may sit in the Infinity Cache)'}
            del last1
        # (the last timed step ran on batch (steps - 1) % nbat: the cross-checks below compare against that one)
        lastset = sets[(steps - 1) % nbat]
        trajd, batch, batch_time, evd = lastset['traj'], lastset['batch'], lastset['batch_time'], lastset['batch_time']['events']
        # every rank's own time for the median block and its own event count (the headline uses the MAX over ranks)
        per_rank = None
        if world > 1:
            mine = torch.tensor([sorted(own_blocks)[1], valid_local], dtype=torch.float64, device=comm_dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = {'ms_per_step': [round(1e3 * float(a[0]) / steps, 4) for a in allr],
                        'value': [round(float(a[1]) * steps / float(a[0]) / 1e6, 3) for a in allr], 'unit': 'Mevents/s'}
            per_rank['balance_min_over_max'] = round(min(per_rank['value']) / max(per_rank['value']), 4)
        # the exchange alone (no loss step beside it): achieved all-reduce rate against the xGMI bounds
        comm_alone = None
        if reducer is not None and world > 1:
            for _ in range(2):
                reducer.start(); reducer.wait()
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                reducer.start(); reducer.wait()
            torch.cuda.synchronize()
            tc = dp.max_over_ranks((time.perf_counter() - t0) / 5, comm_dev)
            nbytes = dp.UNET_GRAD_NUMEL * 4
            link = 153.0                                          # GB/s per xGMI link and direction (7 links per GPU)
            ring = 2.0 * (world - 1) / world * nbytes / 1e9       # GB every rank sends (and receives) in a ring all-reduce
            comm_alone = {'ms': round(1e3 * tc, 4), 'algbw_GBps': round(nbytes / tc / 1e9, 1), 'busbw_GBps': round(ring / tc, 1),
                          'bound_one_link_ms': round(1e3 * ring / link, 4), 'bound_seven_links_ms': round(1e3 * ring / (7 * link), 4),
                          'note': 'busbw = 2(N-1)/N x bytes / time; a ring is bound by ONE link per neighbour (153 GB/s per direction), '
                                  'direct reduce-scatter + all-gather by all seven'}

        # instrumented pass: HIP events around every C-ABI call, on the stream they launch on
        stages = {}
        kernels = {}
        if instrument:
            def plain_step(i):          # (the timed loop's step without the reducer, rotating over the batches the same way)
                st_ = sets[i % nbat]
                loss, _, _ = L.calc(st_['traj'], times_d, st_['batch'])
                loss.backward()
                st_['traj'].grad = None
            ops.STAGE_TIMER = ops.StageTimer()
            for i in range(steps):
                plain_step(i)
            stages = ops.STAGE_TIMER.summary()
            ops.STAGE_TIMER = None
            # ... and HIP events around every KERNEL (recorded by the library on its launch stream), over the same steps issued
            # exactly as the timed loop issues them (mpc_focus_fwd / mpc_focus_bwd)
            with ops.KernelTimer() as kt:
                for i in range(steps):
                    plain_step(i)
            kernels = kt.summary()
        # the same step captured once into a HIP graph and replayed (static shapes; N = 1 only): what the host-side
        # launch overhead of the eager path costs -- reported beside the eager number, never as `value`
        graph_ms = None
        if world == 1 and instrument and not args.no_hip_graph:
            try:
                if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
                    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    # a fresh leaf: the gradient-accumulation node of `trajd` belongs to the default stream, which
                    # must not take part in a capture
                    tg = trajd.detach().clone().requires_grad_(True)
                    for _ in range(2):
                        lg, _, _ = L.calc(tg, times_d, batch)
                        lg.backward()
                        tg.grad = None
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    lg, _, _ = L.calc(tg, times_d, batch)
                    lg.backward()
                for _ in range(3):
                    gr.replay()
                gts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        gr.replay()
                    torch.cuda.synchronize()
                    gts.append(time.perf_counter() - t0)
                graph_ms = 1e3 * sorted(gts)[1] / steps
                del gr, lg, tg
            except Exception as e:      # informational
                graph_ms = repr(e)[:120]
        # the same steps through FocusLoss(static_shapes=True): the library's own capture-once-and-replay (two HIP graphs per
        # shape), no capture code on the caller's side
        static_ms = None
        if world == 1 and instrument and not args.no_hip_graph:
            try:
                Lst = LossFactory.get_loss_calculator('FOCUS', dict(loss_config(wl), static_shapes=True))

                def sstep():
                    ls_, _, _ = Lst.calc(trajd, times_d, batch)
                    ls_.backward()
                    trajd.grad = None
                    return ls_
                for _ in range(5):
                    sstep()
                sts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        ls_ = sstep()
                    torch.cuda.synchronize()
                    sts.append(time.perf_counter() - t0)
                static_ms = {'ms_per_step': round(1e3 * sorted(sts)[1] / steps, 4), 'loss_equal': bool(ls_.item() == last.item())}
                del Lst
            except Exception as e:      # informational
                static_ms = {'error': repr(e)[:160]}
        # the other event layout beside the headline: the reference loader's time-ordered tensor when the headline runs on
        # bucket-ordered events (and the other way round), plus what the ordering costs ingest
        ordered = None
        if instrument and world == 1:
            other = batch_time if args.events_layout == 'bucket' else L.order_events(batch_time)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                L.order_events(batch_time)
            torch.cuda.synchronize()
            order_ms = 1e3 * (time.perf_counter() - t0) / 10

            def ostep():
                lo_, _, _ = L.calc(trajd, times_d, other)
                lo_.backward()
                trajd.grad = None
                return lo_
            for _ in range(3):
                ostep()
            ots = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    lo_ = ostep()
                torch.cuda.synchronize()
                ots.append(time.perf_counter() - t0)
            oms = 1e3 * sorted(ots)[1] / steps
            ordered = {'layout': 'time' if args.events_layout == 'bucket' else 'bucket', 'ms_per_step': round(oms, 4),
                       'value': round(valid_local / (oms * 1e-3) / 1e6, 3), 'unit': 'Mevents/s',
                       'order_events_standalone_ms': round(order_ms, 4), 'loss_equal': bool(lo_.item() == last.item())}
        # one cross-check of the number the timed steps computed, outside the timed region: the same step through the
        # in-library global-atomic event kernels (a second implementation of warp + vote + backward)
        check = None
        if instrument:
            La = LossFactory.get_loss_calculator('FOCUS', dict(loss_config(wl), debug_atomic_path=True))
            ta = trajd.detach().clone().requires_grad_(True)
            la, _, _ = La.calc(ta, times_d, batch_time)       # (the debugging path takes the plain tensor; the layouts agree bit for bit)
            la.backward()
            tb = trajd.detach().clone().requires_grad_(True)
            lb, _, _ = L.calc(tb, times_d, batch)
            lb.backward()
            check = {'loss_rel_diff_vs_atomic_path': abs(la.item() - lb.item()) / abs(lb.item()),
                     'grad_rel_l2_vs_atomic_path': float((ta.grad - tb.grad).norm() / tb.grad.norm())}
        return dict(wl=wl, dt=dt, blocks=blocks, steps=steps, total_valid=total_valid, stages=stages, kernels=kernels, graph_ms=graph_ms,
                    loss=float(last.item()), n=traj.shape[2], check=check, ordered=ordered, static=static_ms, per_rank=per_rank,
                    comm_alone=comm_alone, single=single, nbat=nbat)

    r = run_workload(args.workload, args.steps, args.warmup, args.grad_allreduce)
    r_comm = None
    if world > 1 and not args.grad_allreduce:
        # the data-parallel training step's one real exchange: the same steps with the averaged
        # network gradient (UNet-sized, 124 MB fp32) all-reduced over RCCL on a side stream, overlapping the next loss
        r_comm = run_workload(args.workload, args.steps, args.warmup, True, instrument=False)
    wl = r['wl']
    ms_per_step = 1e3 * r['dt'] / r['steps']
    value = r['total_valid'] * r['steps'] / r['dt'] / 1e6

    def event_path_roofline(per_step, wl_):
        """the event path alone (A6-A11 with the LUT given) against the HBM roofline."""
        names = ('mpc_event_splat_fwd', 'mpc_contrast_fwd', 'mpc_event_splat_bwd', 'mpc_finalize')
        us = sum(per_step[k]['us_per_step'] for k in names if k in per_step)
        b = algorithmic_bytes(wl_['B'], wl_['M'], wl_['nb'])
        return {'us_per_step': round(us, 1), 'achieved': round(b / (us * 1e-6) / 1e9, 1) if us > 0 else 0.0,
                'frac': round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else 0.0}

    def roofline_of(res, wname=None):
        wl_ = res['wl']
        sb = stage_bytes(wl_['B'], wl_['M'], wl_['nb'], res['n'])
        per_step = {}
        for name, st in res['stages'].items():
            calls = st['launches'] / res['steps']
            per_step[name] = {'us_per_step': st['avg_us'] * calls, 'calls_per_step': calls,
                              'algorithmic_MB': sb.get(name, 0) / 1e6}
        dom = max(per_step, key=lambda k: per_step[k]['us_per_step'])
        d = per_step[dom]
        ach = d['algorithmic_MB'] * 1e6 / (d['us_per_step'] * 1e-6) / 1e9 if d['us_per_step'] > 0 else 0.0
        gpu_us = sum(v['us_per_step'] for v in per_step.values())
        path_b = algorithmic_bytes(wl_['B'], wl_['M'], wl_['nb'])
        pk, pfile = profile_kernels(wname) if wname else ({}, None)
        in_stage = [k for k, _, _, _ in pk.get(dom, [])]
        # live per-kernel durations (HIP events recorded by the library around every launch of these steps)
        live = {k: {'us_per_launch': round(v['avg_us'], 2), 'launches_per_step': round(v['launches'] / res['steps'], 2),
                    'stage': stage_of_kernel(k)} for k, v in sorted(res.get('kernels', {}).items(), key=lambda kv: -kv[1]['total_us'])}
        dom_k = next(iter(live), None)
        if dom_k is not None and live[dom_k]['stage'] in per_step:
            # the dominant KERNEL of the step and its own duration: algorithmic bytes of its stage per launch / that duration
            dom = live[dom_k]['stage']
            d = dict(per_step[dom])
            d['us_per_step'] = live[dom_k]['us_per_launch'] * live[dom_k]['launches_per_step']
            ach = d['algorithmic_MB'] * 1e6 / (d['us_per_step'] * 1e-6) / 1e9 if d['us_per_step'] > 0 else 0.0
            in_stage = [k for k in live if live[k]['stage'] == dom]
        t_up, t_lo = pmc_traffic(wname, dom, {k: v['launches_per_step'] for k, v in live.items()}) if wname else (None, None)
        ev_path = event_path_roofline(per_step, wl_)
        if live:
            # the per-kernel times supersede the stage brackets for the sums (a stage bracket also holds the gaps between its
            # kernels and, for the staged entry points, launches the fused calls do not make)
            gpu_us = sum(v['us_per_launch'] * v['launches_per_step'] for v in live.values())
            ev_us = sum(v['us_per_launch'] * v['launches_per_step'] for v in live.values()
                        if v['stage'] in ('mpc_event_splat_fwd', 'mpc_contrast_fwd', 'mpc_event_splat_bwd', 'mpc_finalize'))
            if ev_us > 0:
                ev_path = {'us_per_step': round(ev_us, 1), 'achieved': round(path_b / (ev_us * 1e-6) / 1e9, 1),
                           'frac': round(path_b / (ev_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), 'from': 'per-kernel HIP events'}
        knn_info = knn_ceiling(wname, per_step, wl_['B'], wl_['nb']) if wname else {}
        # what bounds the dominant kernel: the KNN kernels issue vector instructions (exact K-nearest selection) and leave HBM idle --
        # their ceiling is the SIMDs' issue rate (one wave64 instruction per 2 cycles, MI355X_MICROARCH.md), the HBM figures below
        # stay as measured; every other kernel of the path is HBM-bound by nature
        is_knn = dom.startswith('mpc_knn')
        valu = None
        kprof = (knn_info.get('from_committed_profile') or {}).get('kernels', {}).get(dom_k) if is_knn else None
        if kprof:
            valu = {'achieved': round(2.0 / kprof['simd_cycles_per_valu_instr'], 3), 'unit': 'fraction of 1 wave64 instruction per 2 SIMD cycles',
                    'simd_cycles_per_valu_instr': kprof['simd_cycles_per_valu_instr'], 'from': 'committed SQ counters of this library (knn.from_committed_profile)'}
            # (round 6) ... and against the floor of ITS OWN instruction mix (tools/valu_floor.py): the 2-cycle figure is an ideal no mix reaches
            fl = (knn_info.get('valu_floor') or {}).get(dom_k.split('<')[0]) if dom_k else None
            if fl:
                live_us = live[dom_k]['us_per_launch'] if dom_k in live else None
                valu.update({'floor_us': fl['floor_us'], 'two_cycle_ideal_us': fl['two_cycle_ideal_us'],
                             'achieved_over_floor': round(fl['floor_us'] / live_us, 3) if live_us else fl['achieved_over_floor'],
                             'mean_issue_cycles_of_the_mix': fl['mean_issue_cycles'],
                             'floor_note': 'floor_us: the kernel\'s own dynamic instruction mix at the measured issue cost of every class '
                                           '(knn.valu_floor, tools/valu_floor.py); achieved_over_floor = floor_us / live kernel_us'})
        return {
            'bound': 'valu_issue' if is_knn else 'hbm', 'valu_issue': valu,
            'kernel': dom_k if dom_k is not None else (in_stage[0] if in_stage else dom), 'stage': dom,
            'note': 'kernel_us = the average launch of `kernel`, HIP events recorded by the library around that launch on its stream, '
                    'over the timed steps (kernels_us has every kernel); achieved = algorithmic bytes of its stage / kernel_us'
                    + ('; this stage is VALU-issue bound (exact K-nearest selection: `knn` below has its queries/s and issue-slot '
                       'share), the HBM fraction is reported as measured' if dom.startswith('mpc_knn') else ''),
            'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': t_up, 'traffic_lower_bound': t_lo,
            'traffic_note': 'PMC FETCH_SIZE x 2 + WRITE_SIZE per step of this stage (reads counted at half on gfx950); the raw counters are the lower bound',
            'algorithmic_bytes': int(d['algorithmic_MB'] * 1e6),
            'kernel_us': round(d['us_per_step'], 1), 'kernels_in_stage': in_stage, 'profile_summary': pfile,
            'kernels_us': live,
            'event_path': ev_path,
            'knn': knn_info,
            'path': {'algorithmic_MB': round(path_b / 1e6, 2), 'gpu_us_per_step': round(gpu_us, 1),
                     'achieved': round(path_b / (gpu_us * 1e-6) / 1e9, 1) if gpu_us > 0 else 0.0,
                     'frac': round(path_b / (gpu_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if gpu_us > 0 else 0.0},
            'stages_us_per_step': {k: round(v['us_per_step'], 1) for k, v in sorted(per_step.items())},
        }

    out = {
        'metric': 'Mevents/s through CMax loss fwd+bwd, DSEC 480x640',
        'value': round(value, 3), 'unit': 'Mevents/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'blocks_ms_per_step': [round(1e3 * x / r['steps'], 4) for x in r['blocks']], 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f"{args.workload}: {wl['desc']}", 'batch_per_gpu': wl['B'], 'events_layout': args.events_layout,
                   'resident_batches_rotated': r.get('nbat', 1),
                   'global_batch': wl['B'] * world, 'events_per_sample': wl['M'], 'num_bins': wl['nb'],
                   'num_knn': KNN, 'image': [H, W], 'parallelism': f'dp{world}',
                   'grad_allreduce_MB': round(dp.UNET_GRAD_NUMEL * 4 / 1e6, 2) if (world > 1 and args.grad_allreduce) else 0},
        'rccl_ranks': rccl_ranks, 'per_rank': r.get('per_rank'),
        'loss': r['loss'], 'loss_check': r.get('check'),
        'single_batch': r.get('single'),
        'other_event_layout': r.get('ordered'),
        'static_shapes': r.get('static'),
        'roofline': roofline_of(r, args.workload),
    }
    if isinstance(r.get('graph_ms'), float):
        out['hip_graph'] = {'ms_per_step': round(r['graph_ms'], 4), 'value': round(r['total_valid'] / (r['graph_ms'] * 1e-3) / 1e6, 3),
                            'unit': 'Mevents/s', 'note': 'calc + backward captured once (torch.cuda.CUDAGraph) and replayed'}
    elif r.get('graph_ms') is not None:
        out['hip_graph'] = {'error': r['graph_ms']}
    if r_comm is not None:
        out['dp_with_grad_allreduce'] = {
            'value': round(r_comm['total_valid'] * r_comm['steps'] / r_comm['dt'] / 1e6, 3), 'unit': 'Mevents/s',
            'ms_per_step': round(1e3 * r_comm['dt'] / r_comm['steps'], 4),
            'grad_allreduce_MB': round(dp.UNET_GRAD_NUMEL * 4 / 1e6, 2), 'allreduce_alone': r_comm.get('comm_alone'),
            'per_rank': r_comm.get('per_rank'),
            'note': 'same steps + a stand-in network of the reference UNet\'s 31 044 610 parameters (dp.StandInNetwork: dense layers, forward + backward on '
                    '512 rows) whose gradients fill the flat buffer as views; every bucket of the 124 MB is all-reduced (RCCL, '
                    'side stream) the moment its last gradient is written, overlapping the rest of that backward and the next step\'s loss'}
        r_net = run_workload(args.workload, args.steps, args.warmup, False, instrument=False, net_only=True)
        out['dp_with_grad_allreduce']['same_steps_without_the_exchange_ms'] = round(1e3 * r_net['dt'] / r_net['steps'], 4)
    if rank == 0 and world == 1:
        also = {}
        # the data-parallel leg on ONE GPU: a process group of one rank, so that the bucketed all-reduce goes through the RCCL call path
        # (stream ordering, bucket events) beside a real gradient producer; what it measures here is the contention of the network's
        # backward and the exchange's kernels with the loss kernels -- the xGMI traffic itself needs the 8-GPU node
        if not args.no_nondefault:
            try:
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', str(_free_port()))
                dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
                r1 = run_workload(args.workload, max(5, args.steps // 2), 2, True, instrument=False, force_group=True)
                r0 = run_workload(args.workload, max(5, args.steps // 2), 2, False, instrument=False, net_only=True)
                dist.destroy_process_group()
                also['dp_producer_1gpu'] = {
                    'loss_only_ms_per_step': round(ms_per_step, 4),
                    'loss_plus_network_ms_per_step': round(1e3 * r0['dt'] / r0['steps'], 4),
                    'loss_plus_network_plus_bucketed_allreduce_ms_per_step': round(1e3 * r1['dt'] / r1['steps'], 4),
                    'grad_allreduce_MB': round(dp.UNET_GRAD_NUMEL * 4 / 1e6, 2),
                    'note': 'RCCL group of ONE rank: stand-in network of 31 044 610 parameters (dp.StandInNetwork), gradients as views into the flat '
                            'buffer, 4 buckets all-reduced on a side stream as they fill; the network step itself dominates these figures -- compare '
                            'the last two'}
            except Exception as e:      # informational
                also['dp_producer_1gpu'] = {'error': repr(e)[:300]}
        for name in [a for a in args.also.split(',') if a and a != args.workload]:
            # each extra workload runs in a fresh child process: the caching allocator of this process
            # (sized by the main workload) otherwise frees/reallocates inside the child's timed steps
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--workload', name, '--steps', str(args.steps),
                   '--warmup', str(args.warmup), '--no-cpu-baseline', '--no-realistic', '--no-nondefault', '--also', '', '--events-layout', args.events_layout,
                   '--batches', str(args.batches)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                dj = json.loads(r.stdout.strip().splitlines()[-1])
                also[name] = {'value': dj['value'], 'ms_per_step': dj['ms_per_step'],
                              'single_batch_ms_per_step': (dj.get('single_batch') or {}).get('ms_per_step'),
                              'hip_graph_ms_per_step': dj.get('hip_graph', {}).get('ms_per_step'),
                              'static_shapes_ms_per_step': (dj.get('static_shapes') or {}).get('ms_per_step'),
                              'time_ordered_ms_per_step': (dj.get('other_event_layout') or {}).get('ms_per_step'),
                              'path_frac': dj['roofline']['path']['frac'],
                              'stages_us_per_step': dj['roofline']['stages_us_per_step']}
            except Exception as e:      # informational field: never fail the main line
                also[name] = {'error': repr(e)[:200]}
        # the same workload on the inputs training produces instead of white-noise coefficients (smooth flow fields, zero flow,
        # ragged batches): step time relative to the headline's inputs, share of the KNN queries handed to the fallback
        if not args.no_realistic:
            try:
                import subprocess
                import tempfile
                with tempfile.NamedTemporaryFile(suffix='.json') as tf:
                    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'realistic_probe.py'), '--workload', args.workload,
                           '--steps', str(max(5, args.steps // 2)), '--layout', args.events_layout, '--json', tf.name]
                    subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True)
                    rows = json.load(open(tf.name))
                also['realistic_inputs'] = {
                    'note': 'C3-shaped steps; vs_white = step time / step time on the white-noise coefficients of the headline; '
                            'knn_fail_frac = queries the strip kernel handed to the one-wavefront-per-query search of k_knn_tail',
                    'variants': {r['variant']: {'ms_per_step': r['ms_per_step'], 'vs_white': r['vs_first'],
                                                'knn_fail_frac': round(r['knn_fail_frac'], 5),
                                                'k_knn_tail_us': r['kernels_us_per_step'].get('k_knn_tail'),
                                                'k_knn_strip_us': r['kernels_us_per_step'].get('k_knn_strip'),
                                                'k_knn_bwd_tile_us': r['kernels_us_per_step'].get('k_knn_bwd_tile'),
                                                'k_knn_bwd_far_us': r['kernels_us_per_step'].get('k_knn_bwd_far')} for r in rows},
                    'worst_vs_white': max(r['vs_first'] for r in rows)}
                # the input the reference's own network produces (src/modules/trajectory_net.py:142-161: a smooth mixture, not white noise),
                # beside the headline
                un = next((r for r in rows if r['variant'] == 'unet'), None)
                if un is not None:
                    out['value_unet'] = {'value': round(un['valid_events'] / (un['ms_per_step'] * 1e-3) / 1e6, 1), 'unit': 'Mevents/s',
                                         'ms_per_step': un['ms_per_step'], 'vs_white': un['vs_first'],
                                         'note': 'the same C3-shaped step on a UNet-like smooth flow field (tools/realistic_probe.py, family unet)'}
            except Exception as e:
                also['realistic_inputs'] = {'error': repr(e)[:300]}
        # configurations outside the shipped yaml files: the headline's workload with one loss setting changed
        try:
            if args.no_nondefault:
                raise RuntimeError('skipped (--no-nondefault)')
            from motionpriorcmax_amd import LossFactory as _LF
            nd = {}
            for tag, over, trefs in (('iwd', {'interpolation_scheme': 'iwd'}, (0.41,)), ('dist_l1', {'dist_norm': 'l1'}, (0.41,)),
                                     ('num_tref_2', {'num_tref': 2, 'scale_iwe_by_dt': False, 'polarity_aware_batching': False}, (0.41, 0.77))):
                evn, npn, trn, tmn = synth_inputs(wl, seed=1, trefs=trefs)
                Ln = _LF.get_loss_calculator('FOCUS', dict(loss_config(wl), **over))
                evd_, tmd_ = evn.to(dev), tmn.to(dev)
                trd_ = trn.to(dev).requires_grad_(True)
                bn = {'events': evd_, 'num_pos_events': npn}

                def _st():
                    l_, _, _ = Ln.calc(trd_, tmd_, bn)
                    l_.backward()
                    trd_.grad = None
                for _ in range(6):
                    _st()
                blocks_ = []
                for _ in range(3):                       # (median of three blocks of ten steps, as the headline)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        _st()
                    torch.cuda.synchronize()
                    blocks_.append(1e2 * (time.perf_counter() - t0))
                nd[tag] = {'ms_per_step': round(sorted(blocks_)[1], 4)}
                del Ln, trd_, evd_
            nd['note'] = ('same workload and inputs as the headline with one loss setting changed; iwd: strip forward, tile-gather backward with '
                          'distance weights (round 6; the point gather k_knn_bwd_points before: 1.14 ms); dist_l1: strip forward with 128 slots per query (round 6; the tile kernel k_knn_query before: 1.94 ms); '
                          'num_tref = 2: the two reference times of a sample as two samples of the num_tref == 1 kernels (round 6, FocusLoss._calc_trefs_as_samples; the general kernels before: 3.73 ms)')
            also['non_default_configs'] = nd
        except Exception as e:
            also['non_default_configs'] = {'error': repr(e)[:200]}
        # UNPINNED extension: the per-event continuous-time basis warp (no LUT, no KNN) on the headline's events and coefficients
        try:
            if args.no_nondefault:
                raise RuntimeError('skipped (--no-nondefault)')
            from motionpriorcmax_amd import LossFactory as _LF2
            kb = wl['k'] if wl['k'] <= 5 else 3
            evp, npp, _, tmp_ = synth_inputs(wl, seed=1)
            gpe = torch.Generator().manual_seed(8)
            cgrid = (torch.randn(wl['B'], 1, 2 * kb, H, W, generator=gpe) * (3.0 if kb == 1 else 1.0)).to(dev).requires_grad_(True)
            Lp = _LF2.get_loss_calculator('FOCUS', loss_config(wl))
            bp = Lp.order_events({'events': evp.to(dev), 'num_pos_events': npp})      # (bucket-ordered rows: the backward without global atomics)

            def _pe():
                l_, _, _ = Lp.calc_per_event_basis(cgrid, 0.41, bp, kb)
                l_.backward()
                cgrid.grad = None
            for _ in range(4):
                _pe()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                _pe()
            torch.cuda.synchronize()
            tpe = (time.perf_counter() - t0) / 10
            nvalid = float(evp[..., 5].sum())
            with ops.KernelTimer() as ktp:
                for _ in range(5):
                    _pe()
            kus = sum(v['total_us'] for v in ktp.summary().values()) / 5
            also['per_event_basis'] = {'ms_per_step': round(1e3 * tpe, 4), 'value': round(nvalid / tpe / 1e6, 1), 'unit': 'Mevents/s',
                                       'library_kernels_us_per_step': round(kus, 1),
                                       'note': 'UNPINNED extension (no reference code): every event warped with the motion basis at its own '
                                               'timestamp and the coefficients of its tile, no flow LUT and no KNN (mpc_pe_warp, the vote / blur / '
                                               'objective kernels, mpc_pe_grad_ordered on bucket-ordered rows); the rest of the step is torch: the '
                                               'tile coefficients sliced out of the dense [B,1,2k,H,W] grid and its backward, the smoothness field'}
            del Lp, cgrid, bp
        except Exception as e:
            also['per_event_basis'] = {'error': repr(e)[:200]}
        # the two UNPINNED extensions BASELINE.json's configs name, at the size their config names ("pyramid variant
        # reported separately", "cubic B-spline (unpinned extension)"); the reference has neither
        try:
            if args.no_nondefault:
                raise RuntimeError('skipped (--no-nondefault)')
            from motionpriorcmax_amd import LossFactory as _LF3, utils as _U3
            from motionpriorcmax_amd.utils.synth import synth_events as _se3, bin_mid_times as _bm3

            def _time10(fn):
                """median of three blocks of ten steps (one block of ten caught an allocator refill on one box: 4.3 ms against 0.73)"""
                for _ in range(4):
                    fn()
                blocks_ = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        fn()
                    torch.cuda.synchronize()
                    blocks_.append((time.perf_counter() - t0) / 10)
                return sorted(blocks_)[1]
            # (a) configs[2]: 3-level IWE pyramid on the headline's workload (stage calls, no fused path)
            wl3 = WORKLOADS['C3']
            ev3, np3, tr3, tm3 = synth_inputs(wl3, seed=1)
            Lp3 = _LF3.get_loss_calculator('FOCUS', dict(loss_config(wl3), pyramid_levels=3))
            e3, t3, m3 = ev3.to(dev), tr3.to(dev).requires_grad_(True), tm3.to(dev)
            b3 = {'events': e3, 'num_pos_events': np3}

            def _p3():
                l_, _, _ = Lp3.calc(t3, m3, b3)
                l_.backward()
                t3.grad = None
            tp3 = _time10(_p3)
            also['pyramid3_C3'] = {'ms_per_step': round(1e3 * tp3, 4), 'value': round(float(ev3[..., 5].sum()) / tp3 / 1e6, 1), 'unit': 'Mevents/s',
                                   'note': 'UNPINNED extension: FocusLoss(pyramid_levels=3) on the C3 workload (2x2 averages of the raw IWE, the '
                                           'reference objective on every level, focus = sum over levels); tests/test_gpu_xrows_fullsize.py'}
            del Lp3, e3, t3, b3
            # (b) configs[3]: cubic B-spline trajectories (10 free control points) on the C4 workload, basis evaluated on the device
            wl4 = WORKLOADS['C4']
            ev4, np4 = _se3(1, wl4['M'], (H, W), wl4['nb'], seed=1, pad_frac=0.02, time_sorted=True)
            tm4 = torch.cat((torch.tensor([0.41]), _bm3(wl4['nb']))).to(dev)
            g4 = torch.Generator().manual_seed(9)
            pr4 = (torch.randn(1, 20, H // PATCH, W // PATCH, generator=g4) * 2.0).to(dev).requires_grad_(True)
            L4 = _LF3.get_loss_calculator('FOCUS', loss_config(wl4))
            b4 = {'events': ev4.to(dev), 'num_pos_events': np4}

            def _b4():
                tj, _ = _U3.trajectories_from_bspline(pr4, tm4, PATCH, (H, W))
                l_, _, _ = L4.calc(tj, tm4, b4)
                l_.backward()
                pr4.grad = None
            tb4 = _time10(_b4)
            also['bspline_C4'] = {'ms_per_step': round(1e3 * tb4, 4), 'value': round(float(ev4[..., 5].sum()) / tb4 / 1e6, 1), 'unit': 'Mevents/s',
                                  'note': 'UNPINNED extension: clamped uniform cubic B-spline flow curves (utils.trajectories_from_bspline, autograd to '
                                          'the control points) through FocusLoss.calc on the C4 workload; the step includes the basis evaluation'}
            del L4, b4, pr4
        except Exception as e:
            also['pyramid3_C3'] = also.get('pyramid3_C3', {'error': repr(e)[:200]})
            also['bspline_C4'] = also.get('bspline_C4', {'error': repr(e)[:200]})
        # next row 8f-2: voxel-grid builder on the same window shape (network input; not part of `value`)
        try:
            from motionpriorcmax_amd.utils import voxel_grids
            from oracle import voxel_oracle as VO
            Bv, Nv, vshape = wl['B'], wl['M'], (wl['nb'], H, W)
            xs = [torch.stack(VO.synth_raw_events(Nv, vshape, seed=900 + b), -1) for b in range(Bv)]
            evv = torch.stack(xs).to(dev)
            cntv = torch.full((Bv,), Nv, dtype=torch.int32, device=dev)
            for _ in range(5):
                voxel_grids(evv, cntv, vshape, 'mean_std')
            torch.cuda.synchronize()
            tvs = []
            for _ in range(11):                       # median: the allocator may reshuffle once after the main workload
                t0 = time.perf_counter()
                voxel_grids(evv, cntv, vshape, 'mean_std')
                torch.cuda.synchronize()
                tvs.append(time.perf_counter() - t0)
            tv = sorted(tvs)[len(tvs) // 2]
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            x, y, t, p = (xs[0][:, k] for k in range(4))
            t0 = time.perf_counter()
            VO.voxel_grid(x, y, t, p, vshape, 'mean_std')
            tc = time.perf_counter() - t0
            also['voxel_grid'] = {'ms_per_batch': round(1e3 * tv, 4), 'value': round(Bv * Nv / tv / 1e6, 1),
                                  'unit': 'Mevents/s', 'batch': Bv, 'events_per_sample': Nv,
                                  'algorithmic_MB': round((16 * Bv * Nv + 4 * Bv * wl['nb'] * H * W) / 1e6, 1),
                                  'cpu_oracle_Mevents_per_s': round(Nv / tc / 1e6, 3)}
        except Exception as e:
            also['voxel_grid'] = {'error': repr(e)[:200]}
        # next row 8f-1: event ingest (raw window -> padded [B,M,6] tensor) on the same window shape
        try:
            from motionpriorcmax_amd.utils import ingest_events
            from oracle import ingest_oracle as IO
            Bi, Ni = wl['B'], wl['M']
            raws = [IO.synth_raw(Ni, H, W, seed=950 + b) for b in range(Bi)]
            tx = [torch.from_numpy(__import__('numpy').stack([r[k] for r in raws])).to(dev) for k in range(4)]
            cnti = torch.full((Bi,), Ni, dtype=torch.int32)
            for _ in range(3):
                ingest_events(tx[0], tx[1], tx[2], tx[3], cnti, (H, W), wl['nb'])
            tis = []
            for _ in range(11):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ingest_events(tx[0], tx[1], tx[2], tx[3], cnti, (H, W), wl['nb'])
                torch.cuda.synchronize(); tis.append(time.perf_counter() - t0)
            ti = sorted(tis)[len(tis) // 2]
            t0 = time.perf_counter()
            IO.collate([IO.sample_events(*r, H, W, wl['nb']) for r in raws[:2]])
            tci = (time.perf_counter() - t0) / 2
            also['event_ingest'] = {'ms_per_batch': round(1e3 * ti, 4), 'value': round(Bi * Ni / ti / 1e6, 1), 'unit': 'Mevents/s',
                                    'batch': Bi, 'events_per_sample': Ni, 'algorithmic_MB': round((20 + 24) * Bi * Ni / 1e6, 1),
                                    'cpu_oracle_Mevents_per_s': round(Ni / tci / 1e6, 2), 'note': 'includes the 2-integer host read that sizes the output'}
        except Exception as e:
            also['event_ingest'] = {'error': repr(e)[:200]}
        # next row 8f-3: dense flow from tile trajectories + flow metrics at the DSEC validation shape
        try:
            from motionpriorcmax_amd.utils import dense_flow_from_traj, calculate_flow_error, get_optical_flow_tile_mask
            from oracle import flow_oracle as FO
            Bf = wl['B']
            pixf = torch.nonzero(get_optical_flow_tile_mask((H, W), 4)).to(dev)
            tff = torch.randn(Bf, pixf.shape[0], 2, device=dev)
            gtf, prf, emf, _ = FO.synth_flow_case(Bf, H, W, seed=77, with_scale=False)
            gtf, prf, emf = gtf.to(dev), prf.to(dev), emf.to(dev)

            def med(fn, reps=11):
                for _ in range(3):
                    fn()
                ts_ = []
                for _ in range(reps):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
                return sorted(ts_)[len(ts_) // 2]
            td = med(lambda: dense_flow_from_traj(tff, pixf, 4, (H, W)))
            te = med(lambda: calculate_flow_error(gtf, prf, emf))
            t0 = time.perf_counter()
            FO.calculate_flow_error(gtf[:1].cpu(), prf[:1].cpu(), emf[:1].cpu())
            tce = time.perf_counter() - t0
            also['dense_flow'] = {'ms_per_batch': round(1e3 * td, 4), 'batch': Bf,
                                  'algorithmic_MB': round(Bf * 2 * 4 * (H * W + 2 * (H // 4) * (W // 4)) / 1e6, 2)}
            also['flow_error'] = {'ms_per_batch': round(1e3 * te, 4), 'batch': Bf,
                                  'algorithmic_MB': round(Bf * H * W * 17 / 1e6, 2),
                                  'cpu_oracle_ms_per_sample': round(1e3 * tce, 2)}
        except Exception as e:
            also['dense_flow'] = {'error': repr(e)[:200]}
        out['also'] = also
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
