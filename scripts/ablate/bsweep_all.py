"""Q5 cfg-2 circuit: backward (own forward sweep, incl. prep / reduce) of every variant over a batch sweep; timing only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
n = 5; cfgs = O.block_configs_quanonet(5, (40, 2, 20, 2)); E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
def med(fn, reps=15):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3
for B in (512, 1024, 2048, 3072, 4096, 8192, 16384):
    x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
    for _ in range(40): _lib.hea_backward(sh, x, w, g, off, co)      # back to full clocks after the host-side set-up of this batch
    res = []
    for v in ('auto', 'packed', 'zpacked', 'tri', 'ztri', 'ztri2'):
        _lib.set_backward_variant(v)
        res.append(f'{v} {med(lambda: _lib.hea_backward(sh, x, w, g, off, co)):.1f}')
    _lib.set_backward_variant('auto'); fa = med(lambda: _lib.hea_forward(sh, x, w, off, co))
    _lib.set_backward_variant('packed'); fo = med(lambda: _lib.hea_forward(sh, x, w, off, co))
    _lib.set_backward_variant('ztri'); fz = med(lambda: _lib.hea_forward(sh, x, w, off, co))
    print(f'B={B}: bwd us: ' + '  '.join(res) + f' | fwd us: auto {fa:.1f} first-gen {fo:.1f} zyz {fz:.1f}', flush=True)
_lib.set_backward_variant('auto')
_lib.check_status(dev)
