"""Batch sweep of the Q5 backward variants (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
def t(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
n = 5; cfgs = O.block_configs_quanonet(5, (40, 2, 20, 2)); E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
for B in (256, 512, 768, 1024, 1280, 1536, 2048):
    x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
    res = []
    for v in ('packed', 'pair', 'tri'):
        _lib.set_backward_variant(v)
        for _ in range(5): _lib.hea_backward(sh, x, w, g, off, co)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); _lib.hea_backward(sh, x, w, g, off, co); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        res.append(f'{v} {ts[len(ts)//2]*1e3:.1f}')
    print(f'B={B}: ' + '  '.join(res), flush=True)
