"""n = 6..9 timing at two batch sizes (HIP events around the raw C-ABI calls)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
def t(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
for n in ([int(a) for a in sys.argv[1:]] or [6, 7, 8, 9]):
    cfgs = O.block_configs_quanonet(n, (8, 2, 4, 2)); E, blk = O.circuit_sizes(n, cfgs)
    rng = np.random.default_rng(0)
    w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
    for B in (1024, 2048, 4096):
        x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
        for _ in range(3): _lib.hea_backward(sh, x, w, g, off, co)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for a, b in ev:
            a.record(); _lib.hea_backward(sh, x, w, g, off, co); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        print(f'n={n} B={B}: fwd+bwd {ts[len(ts)//2]*1e3:.1f} us', flush=True)
