"""Randomised differential test of the next-row operators (voxel grid, event ingest, dense flow, flow metrics)
against their CPU oracles over random small shapes and ragged batches.  Diagnostics; run on a GPU box, ideally
with PYTORCH_NO_CUDA_MEMORY_CACHING=1 so that an out-of-bounds access faults instead of landing in cached memory:

    python tools/fuzz_aux.py [n_cases] [seed]"""
import os
import sys
import random

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motionpriorcmax_amd.utils import (voxel_grids, ingest_events, dense_flow_from_traj, calculate_flow_error,
                                       get_optical_flow_tile_mask)
from oracle import voxel_oracle as V, ingest_oracle as I, flow_oracle as F

DEV = 'cuda:0'


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0

    def report(kind, tag, msg):
        nonlocal bad
        bad += 1
        print('MISMATCH', kind, tag, msg, flush=True)

    for case in range(n_cases):
        H, W = rng.randrange(8, 150), rng.randrange(8, 200)
        seed = rng.randrange(1 << 30)
        # ---- voxel grid ------------------------------------------------------------------------------
        C = rng.choice([1, 2, 5, 15])
        ns = [rng.choice([2, 3, 100, 5000, 30000]) for _ in range(rng.choice([1, 2, 3]))]
        norm = rng.choice(['mean_std', 'max', None])
        tag = f'case {case} voxel {C}x{H}x{W} ns={ns} norm={norm}'
        if os.environ.get('FUZZ_VERBOSE'):
            print(tag, flush=True)
        N = max(ns)
        ev = torch.zeros(len(ns), N, 4)
        raws = []
        for b, n in enumerate(ns):
            x, y, t, p = V.synth_raw_events(n, (C, H, W), seed=seed + b)
            ev[b, :n] = torch.stack((x, y, t, p), -1)
            raws.append((x, y, t, p))
        out = voxel_grids(ev.to(DEV), torch.tensor(ns, dtype=torch.int32), (C, H, W), norm).cpu()
        for b in range(len(ns)):
            ref = V.voxel_grid(*raws[b], (C, H, W), norm)
            if not torch.isfinite(ref).all():
                continue
            scale = max(1.0, float(ref.abs().max()))
            nbad = int(((out[b] - ref).abs() > 2e-5 * scale).sum())
            if nbad > 2:      # an entry that cancels to exactly 0 in one arithmetic only flips its "non-zero" status
                report('voxel', tag, f'sample {b}: {nbad} entries differ, max {float((out[b] - ref).abs().max())}')
        # ---- ingest -----------------------------------------------------------------------------------
        nb = rng.choice([1, 3, 15])
        ns = [rng.choice([0, 1, 7, 300, 20000]) for _ in range(rng.choice([1, 2, 4]))]
        tag = f'case {case} ingest {H}x{W} nb={nb} ns={ns}'
        if os.environ.get('FUZZ_VERBOSE'):
            print(tag, flush=True)
        N = max(max(ns), 1)
        raws = [I.synth_raw(n, H, W, seed=seed + 10 + b) if n else None for b, n in enumerate(ns)]
        pad = lambda k, dt: np.stack([np.concatenate((r[k], np.zeros(N - len(r[k]), dt))) if r else np.zeros(N, dt) for r in raws])
        x, y, t, p = pad(0, 'float32'), pad(1, 'float32'), pad(2, 'int64'), pad(3, 'float32')
        o = ingest_events(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV), torch.from_numpy(t).to(DEV),
                          torch.from_numpy(p).to(DEV), torch.tensor(ns, dtype=torch.int32), (H, W), nb, want_voxel_input=True)
        empty = (np.zeros((0, 5), np.float32), np.zeros((0, 5), np.float32))
        ref, num_pos = I.collate([I.sample_events(*r, H, W, nb) if r else empty for r in raws])
        if o['num_pos_events'] != num_pos or not np.array_equal(o['events'].cpu().numpy(), ref, equal_nan=True):
            report('ingest', tag, f"num_pos {o['num_pos_events']} vs {num_pos}")
        # ---- dense flow + metrics -------------------------------------------------------------------
        ps = rng.choice([1, 2, 3, 4, 8])
        Hf, Wf = max(H, 2 * ps), max(W, 2 * ps)
        B = rng.choice([1, 2, 5])
        tag = f'case {case} flow {Hf}x{Wf} patch={ps} B={B}'
        if os.environ.get('FUZZ_VERBOSE'):
            print(tag, flush=True)
        g = torch.Generator().manual_seed(seed)
        mask = get_optical_flow_tile_mask((Hf, Wf), ps)
        pix = torch.nonzero(mask)
        keep = pix[:, 0] // ps < Hf // ps
        keep &= pix[:, 1] // ps < Wf // ps
        pix = pix[keep]
        tf = torch.randn(B, pix.shape[0], 2, generator=g) * 4
        dense, patch = dense_flow_from_traj(tf.to(DEV), pix.to(DEV), ps, (Hf, Wf))
        rd, rp = F.dense_flow_from_traj(tf.numpy(), pix.numpy(), ps, (Hf, Wf))
        if not np.array_equal(patch.cpu().numpy(), rp) or np.abs(dense.cpu().numpy() - rd).max() > 2e-5 * max(1.0, np.abs(rd).max()):
            report('dense_flow', tag, f'max diff {np.abs(dense.cpu().numpy() - rd).max()}')
        gt, pr, em, ts = F.synth_flow_case(B, Hf, Wf, seed=seed, with_mask=rng.random() < 0.7, with_scale=rng.random() < 0.5)
        ref = F.calculate_flow_error(gt, pr, em, ts)
        err = calculate_flow_error(gt.to(DEV), pr.to(DEV), None if em is None else em.to(DEV), None if ts is None else ts.to(DEV))
        for k in ('EPE', '1PE', '2PE', '3PE', 'AE'):
            if abs(float(err[k]) - float(ref[k])) > 3e-5 * max(abs(float(ref[k])), 1e-6):
                report('flow_error', tag, f'{k}: {float(err[k])} vs {float(ref[k])}')
    print(f'{n_cases} cases, {bad} bad')
    # (the bounds-checked debug build, tools/bounds_run.sh: out-of-range indices its accessors recorded; -1 = product build)
    from motionpriorcmax_amd import _lib as _C
    _n = _C.lib().mpc_bounds_check()
    print('mpc_bounds_check:', _n, _C.lib().mpc_last_error_string().decode() if _n > 0 else '')


if __name__ == '__main__':
    main()
