"""The ctypes stub printed in INTEGRATION.md section 2 is executed as written and must reproduce FocusLoss.calc
(loss and gradient) -- the document is the binding a maintainer would copy, so it is tested."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def test_integration_md_ctypes_stub_runs_and_matches():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 2.'):text.index('## 3.')]
    code = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    cwd = os.getcwd()
    os.chdir(ROOT)                                       # the stub loads the library by its repository-relative path
    try:
        ns = {}
        exec(compile(code, 'INTEGRATION.md#2', 'exec'), ns)
    finally:
        os.chdir(cwd)
    from motionpriorcmax_amd import LossFactory, _lib as C
    g = load_golden('g1_allflags')
    cfg = g['cfg']
    dev = 'cuda:0'
    traj = torch.from_numpy(g['trajectories']).to(dev)
    events = torch.from_numpy(g['events']).to(dev)
    times = torch.from_numpy(g['times']).to(dev)
    B, M = events.shape[:2]
    H, W = cfg['image_shape']
    sp = cfg['lut_superpixel_size']
    shape = ns['MpcShape'](B=B, M=M, Mp=int(g['num_pos']), nb=cfg['num_bins'], T=1, H=H, W=W, sp=sp,
                           hq=(H + sp - 1) // sp, wq=(W + sp - 1) // sp, n=traj.shape[2], K=cfg['num_knn'],
                           flags=C.F_SCALE_BY_DT | C.F_MASK_BORDER | C.F_POLARITY_SPLIT)
    t_ref = times[:1].contiguous()
    scal, blur, saved = ns['focus_forward'](shape, traj.contiguous(), events.contiguous(), t_ref, cfg['smooth_weight'])
    grad_out = torch.ones(1, device=dev)
    gtraj = ns['focus_backward'](shape, traj.contiguous(), events.contiguous(), t_ref, saved, scal, grad_out)
    torch.cuda.synchronize()
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    t = traj.clone().requires_grad_(True)
    loss, _, misc = L.calc(t, times, {'events': events, 'num_pos_events': int(g['num_pos'])})
    loss.backward()
    assert abs(float(scal[0]) - float(loss.detach())) <= 1e-6 * abs(float(loss.detach()))
    assert torch.allclose(gtraj, t.grad, rtol=1e-5, atol=1e-7)
    assert torch.equal(blur.reshape(misc['iwes'].shape), misc['iwes'])
