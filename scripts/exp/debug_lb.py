import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hea_oracle as O, c_oracle as C
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
_lib.set_backward_variant('packed')
n = 5
rng = np.random.default_rng(0)
for cfgs in ([(5, 0)], [(0, 1)], [(5, 1)], [(5, 2), (5, 1)]):
    E, blk = O.circuit_sizes(n, cfgs)
    for B in (1, 4, 9):
        x = rng.uniform(-3, 3, (B, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=B)
        off, co = O.ham_params(n)
        ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True)
        _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
        sh = _lib.CircuitShape(n, cfgs)
        out, st = _lib.hea_forward(sh, t(x), t(w), off, co, return_state=True)
        gx, gw = _lib.hea_backward(sh, t(x), t(w), t(g), off, co)
        e = lambda a, b: float(np.abs(a.cpu().numpy() - b).max()) if b.size else 0.0
        print(cfgs, 'B', B, 'out %.1e state %.1e gx %.1e gw %.1e' % (e(out, ro), e(st, rst), e(gx, rgx), e(gw, rgw)), flush=True)
