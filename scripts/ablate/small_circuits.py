"""Small circuits at the paper's batch size (100): backward call (own forward sweep, incl. prep / reduce) of every n <= 5
variant -- where the pipeline's start-up and drain are most of the kernel.  Timing only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
def med(fn, reps=30):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, net in ((2, (5, 1, 5, 1)), (2, (20, 2, 10, 2)), (3, (20, 2, 10, 2)), (5, (20, 2, 10, 2)), (5, (40, 2, 20, 2))):
    cfgs = O.block_configs_quanonet(n, net); E, blk = O.circuit_sizes(n, cfgs)
    w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
    for B in (32, 100, 256):
        x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
        res = []
        for v in ('auto', 'packed', 'zpacked', 'pair', 'tri', 'ztri'):
            _lib.set_backward_variant(v)
            res.append(f'{v} {med(lambda: _lib.hea_backward(sh, x, w, g, off, co)):.1f}')
        _lib.set_backward_variant('auto'); fa = med(lambda: _lib.hea_forward(sh, x, w, off, co))
        _lib.set_backward_variant('packed'); fo = med(lambda: _lib.hea_forward(sh, x, w, off, co))
        print(f'n={n} net={net} blk={blk} B={B}: bwd call us (back-to-back): ' + '  '.join(res) + f' | fwd: auto {fa:.1f} first-gen {fo:.1f}', flush=True)
_lib.set_backward_variant('auto')
_lib.check_status(dev)
