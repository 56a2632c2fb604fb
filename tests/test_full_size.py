"""
GPU checks at BASELINE.json's FULL sizes (cfg 2, 4, 5), where the oracle is too slow for the whole batch:
size-independent properties of the domain + the oracle on a random subset of samples.
  * unit norm of every final state, outputs inside [lo, hi]
  * adjoint gradient == parameter-shift rule evaluated with the HIP forward pass itself (exact for these gates)
  * linearity in the upstream weight, additivity of grad_w over a batch split
  * oracle (C restatement) on 6 random samples of the batch, 1e-10
"""
import numpy as np
import pytest
import torch

from oracle import hea_oracle as O
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
TOL = 1e-10

CASES = {
    'cfg2': (5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 1024),      # Advection QuanONet Q5 Net40-2-20-2
    'cfg4': (8, O.block_configs_heaqnn(8, (20, 2)), 2048),               # RDiffusion HEAQNN Q8 depth 20 x 2
    'cfg5': (12, O.block_configs_quanonet(12, (40, 2, 20, 2)), 1024),    # Advection QuanONet Q12, 8192 / 8 GPUs
}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', list(CASES))
def test_full_size_properties(dev, name):
    from quanonet_amd import _lib
    n, cfgs, B = CASES[name]
    rng = np.random.default_rng(hash(name) % 1000)
    E, blk = O.circuit_sizes(n, cfgs)
    lo, hi = -5.0, 5.0
    off, co = O.ham_params(n, lo, hi)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    x = rng.uniform(-np.pi, np.pi, (B, E)); w = rng.uniform(-np.pi, np.pi, (blk, 3, n)); g = rng.normal(size=B)
    sh = _lib.CircuitShape(n, cfgs)
    xd, wd, gd = t(x), t(w), t(g)

    out, st = _lib.hea_forward(sh, xd, wd, off, co, return_state=True)
    norms = (st ** 2).sum(dim=(1, 2))
    assert float((norms - 1).abs().max()) < 1e-11
    assert float(out.min()) >= lo - 1e-9 and float(out.max()) <= hi + 1e-9

    gx, gw, out2 = _lib.hea_backward(sh, xd, wd, gd, off, co, state=None, want_out=True)
    assert float((out2 - out).abs().max()) < TOL

    # oracle on a random subset of the batch
    idx = np.sort(rng.choice(B, 6, replace=False))
    ro, rgx, _ = C.hea_backward(n, cfgs, x[idx], w, g[idx], off, co)
    np.testing.assert_allclose(out.cpu().numpy()[idx], ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(gx.cpu().numpy()[idx], rgx, rtol=0, atol=TOL)

    # parameter-shift with the HIP forward itself: d/dw sum_b g_b f_b = sum_b g_b (f_b(+pi/2) - f_b(-pi/2)) / 2
    for (s, k, q) in [(0, 0, 0), (blk - 1, 2, n - 1), (blk // 2, 1, n // 2)]:
        wp, wm = w.copy(), w.copy()
        wp[s, k, q] += np.pi / 2; wm[s, k, q] -= np.pi / 2
        fp = _lib.hea_forward(sh, xd, t(wp), off, co); fm = _lib.hea_forward(sh, xd, t(wm), off, co)
        ps = float((gd * 0.5 * (fp - fm)).sum())
        assert abs(ps - float(gw[s, k, q])) < 1e-9 * max(1.0, abs(ps)), (s, k, q)
    for col in (0, E // 2, E - 1):
        xp, xm = x.copy(), x.copy()
        xp[:, col] += np.pi / 2; xm[:, col] -= np.pi / 2
        fp = _lib.hea_forward(sh, t(xp), wd, off, co); fm = _lib.hea_forward(sh, t(xm), wd, off, co)
        ps = gd * 0.5 * (fp - fm)
        assert float((ps - gx[:, col]).abs().max()) < TOL

    # linear in the upstream weight; grad_w additive over a batch split (fixed summation order -> tiny tolerance)
    gx2, gw2 = _lib.hea_backward(sh, xd, wd, 2.0 * gd, off, co, state=st)
    assert float((gx2 - 2 * gx).abs().max()) < TOL and float((gw2 - 2 * gw).abs().max()) < 1e-9
    h = B // 2
    _, gwa = _lib.hea_backward(sh, xd[:h].contiguous(), wd, gd[:h].contiguous(), off, co)
    _, gwb = _lib.hea_backward(sh, xd[h:].contiguous(), wd, gd[h:].contiguous(), off, co)
    assert float((gwa + gwb - gw).abs().max()) < 1e-9
