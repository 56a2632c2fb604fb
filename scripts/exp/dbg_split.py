import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
n, cfgs, B = 5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 1024
rng = np.random.default_rng(1)
E, blk = O.circuit_sizes(n, cfgs)
off, co = O.ham_params(n, -5.0, 5.0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
x = rng.uniform(-np.pi, np.pi, (B, E)); w = rng.uniform(-np.pi, np.pi, (blk, 3, n)); g = rng.normal(size=B)
sh = _lib.CircuitShape(n, cfgs)
xd, wd, gd = t(x), t(w), t(g)
out = _lib.hea_forward(sh, xd, wd, off, co)
for rep in range(3):
    gx, gw, out2 = _lib.hea_backward(sh, xd, wd, gd, off, co, state=None, want_out=True)
    d = (out2 - out).abs().cpu().numpy()
    bad = np.nonzero(~(d < 1e-10))[0]
    print('rep', rep, 'bad samples', len(bad), bad[:40], d[bad][:8])
