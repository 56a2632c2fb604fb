"""Fixed cost of the n = 5 training call: backward call (own forward sweep, incl. prep / reduce) against the number of
blocks, extrapolated to zero blocks; and the same with a given final state (no forward phase)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
def med(fn, reps=40):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n = 5
for B in (100, 1024):
    pts = []
    for nbk in (2, 4, 8, 16, 32, 60):
        td = nbk // 3 or 1; bd = nbk - td
        cfgs = O.block_configs_quanonet(n, (bd, 2, td, 2)); E, blk = O.circuit_sizes(n, cfgs)
        w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
        x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
        tb = med(lambda: _lib.hea_backward(sh, x, w, g, off, co))
        tf = med(lambda: _lib.hea_forward(sh, x, w, off, co))
        pts.append((nbk, tb, tf))
        print(f'B={B} blocks={nbk}: backward call {tb:.1f} us  forward call {tf:.1f} us', flush=True)
    (b0, y0, f0), (b1, y1, f1) = pts[-2], pts[-1]
    sb = (y1 - y0) / (b1 - b0); sf = (f1 - f0) / (b1 - b0)
    print(f'B={B}: backward {sb:.3f} us per block, intercept {y1 - sb * b1:.1f} us; forward {sf:.3f} us per block, intercept {f1 - sf * b1:.1f} us')
