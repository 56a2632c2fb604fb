import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
dev = torch.device('cuda:0')
def t(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
def run(n, cfgs, B, seed=0, zero_w=False, zero_x=False):
    rng = np.random.default_rng(seed)
    E, blk = O.circuit_sizes(n, cfgs)
    x = rng.uniform(-3, 3, (B, E)) * (0 if zero_x else 1)
    w = rng.uniform(-3, 3, (blk, 3, n)) * (0 if zero_w else 1)
    off, co = O.ham_params(n)
    sh = _lib.CircuitShape(n, cfgs)
    out, st = _lib.hea_forward(sh, t(x), t(w), off, co, return_state=True)
    torch.cuda.synchronize()
    ref = O.hea_state(n, cfgs, x, w)
    st = st.cpu().numpy(); stc = st[..., 0] + 1j * st[..., 1]
    ro = O.hea_forward(n, cfgs, x, w, off, co)
    print(f"n={n} cfgs={cfgs} B={B} zw={zero_w} zx={zero_x}: state err {np.abs(stc-ref).max():.2e} out err {np.abs(out.cpu().numpy()-ro).max():.2e}")
    if np.abs(stc-ref).max() > 1e-9 and B <= 2 and n <= 3:
        print(' got', np.round(stc, 4)); print(' ref', np.round(ref, 4))
for n in (2, 3, 5):
    run(n, [], 1)
    run(n, [(n, 0)], 1)
    run(n, [(0, 1)], 1, zero_w=True)
    run(n, [(0, 1)], 1)
    run(n, [(n, 1)], 2)
    run(n, [(n, 2)] * 2, 5)
