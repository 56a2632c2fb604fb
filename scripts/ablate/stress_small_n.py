import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import hea_oracle as O, c_oracle as C
from quanonet_amd import _lib
dev = torch.device('cuda:0')
def t(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
rng = np.random.default_rng(12345)
worst = 0.0
for trial in range(160):
    n = int(rng.integers(2, 6))
    nb = int(rng.integers(1, 12))
    if trial % 4 == 3:      # block-unrolled shapes (what the reference builds): one full RX chunk + the same 1 or 2 sub-layers
        ld = int(rng.integers(1, 3))
        cfgs = [(n, ld)] * nb
    else:
        cfgs = [(int(rng.choice([0, 1, n - 1, n, n + 1, 2 * n + 1])), int(rng.integers(0, 4))) for _ in range(nb)]
    E, blk = O.circuit_sizes(n, cfgs)
    if E == 0 and blk == 0: continue
    B = int(rng.choice([1, 3, 17, 64, 129, 300, 777, 1100]))
    x = rng.uniform(-3, 3, (B, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=B)
    off, co = O.ham_params(n, -1.0, 3.0)
    ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
    sh = _lib.CircuitShape(n, cfgs)
    for v in ('auto', 'ztri', 'ztri2', 'zpacked', 'tri', 'pair', 'packed'):
        _lib.set_backward_variant(v)
        gx, gw, out = _lib.hea_backward(sh, t(x), t(w), t(g), off, co, want_out=True)
        e = max(np.abs(out.cpu().numpy() - ro).max(), np.abs(gx.cpu().numpy() - rgx).max() if E else 0.0,
                np.abs(gw.cpu().numpy() - rgw).max() if blk else 0.0)
        worst = max(worst, e)
        assert e < 1e-10, (trial, v, n, cfgs, B, e)
_lib.set_backward_variant('auto')
_lib.check_status(dev)
print('stress ok, worst error', worst)
