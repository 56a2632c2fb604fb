"""One backward call of the cfg-2 circuit with the wait-profiling build of the pipeline kernel (make ... -DQHEA_PROFILE_WAITS,
QHEA_LIB=...): workgroup 100 prints, per wave, its time in hand-off waits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
n = 5; cfgs = O.block_configs_quanonet(5, (40, 2, 20, 2)); E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = t(rng.uniform(-3, 3, (blk, 3, n))); sh = _lib.CircuitShape(n, cfgs); off, co = O.ham_params(n)
x = t(rng.uniform(-3, 3, (B, E))); g = t(rng.normal(size=B))
for i in range(3):
    _lib.hea_backward(sh, x, w, g, off, co)
    torch.cuda.synchronize()
    print('--- call', i, flush=True)
