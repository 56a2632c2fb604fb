"""One forward + one backward launch at a given batch (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
cfgs = O.block_configs_quanonet(n, (40, 2, 20, 2))
E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
x, w, g = t(rng.uniform(-3, 3, (B, E))), t(rng.uniform(-3, 3, (blk, 3, n))), t(rng.normal(size=B))
sh = _lib.CircuitShape(n, cfgs)
off, co = O.ham_params(n)
for _ in range(3):
    out, st = _lib.hea_forward(sh, x, w, off, co, return_state=True)
    _lib.hea_backward(sh, x, w, g, off, co, state=st)
torch.cuda.synchronize()
