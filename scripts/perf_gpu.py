"""Dev loop: numerics vs the C oracle + event timing of the raw C-ABI calls (cfg 2 by default)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hea_oracle as O, c_oracle as C
from quanonet_amd import _lib
dev = torch.device('cuda:0')
def t(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
def case(n, cfgs, B, check=True, reps=50):
    rng = np.random.default_rng(0)
    E, blk = O.circuit_sizes(n, cfgs)
    x = rng.uniform(-3, 3, (B, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=B)
    off, co = O.ham_params(n)
    sh = _lib.CircuitShape(n, cfgs)
    xd, wd, gd = t(x), t(w), t(g)
    out, st = _lib.hea_forward(sh, xd, wd, off, co, return_state=True)
    gx, gw = _lib.hea_backward(sh, xd, wd, gd, off, co, state=st)
    gx2, gw2 = _lib.hea_backward(sh, xd, wd, gd, off, co, state=None)
    torch.cuda.synchronize()
    msg = ''
    if check:
        nb = min(B, 64)
        ro, rgx, rgw_ = C.hea_backward(n, cfgs, x[:nb], w, g[:nb], off, co)
        e1 = np.abs(out.cpu().numpy()[:nb] - ro).max(); e2 = np.abs(gx.cpu().numpy()[:nb] - rgx).max()
        _, _, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
        e3 = np.abs(gw.cpu().numpy() - rgw).max(); e4 = (gw - gw2).abs().max().item(); e5 = (gx - gx2).abs().max().item()
        msg = f'err out {e1:.1e} gx {e2:.1e} gw {e3:.1e} recompute-vs-state {e4:.1e}/{e5:.1e}'
    def timeit(fn):
        for _ in range(5): fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts)//2] * 1e3
    tf = timeit(lambda: _lib.hea_forward(sh, xd, wd, off, co))
    tb = timeit(lambda: _lib.hea_backward(sh, xd, wd, gd, off, co, state=st))
    tb2 = timeit(lambda: _lib.hea_backward(sh, xd, wd, gd, off, co, state=None))
    print(f'n={n} B={B} E={E} blk={blk}: fwd {tf:.1f} us  bwd(state) {tb:.1f} us  bwd(recompute) {tb2:.1f} us | {msg}', flush=True)
if __name__ == '__main__':
    which = sys.argv[1:] or ['cfg2']
    if 'cfg2' in which:
        case(5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 1024)
        case(5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 4096, check=False)
        case(5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 16384, check=False)
    if 'cfg2nocheck' in which:
        case(5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 1024, check=False)
    if 'cfg1' in which:
        case(2, O.block_configs_quanonet(2, (5, 1, 5, 1)), 32)
        case(2, O.block_configs_quanonet(2, (5, 1, 5, 1)), 4096, check=False)
    if 'cfg4' in which:
        case(8, O.block_configs_heaqnn(8, (20, 2)), 2048)
    if 'cfg5' in which:
        case(12, O.block_configs_quanonet(12, (40, 2, 20, 2)), 1024, check=False, reps=5)
    if 'bign' in which:
        for n in (8, 9, 10, 11, 12):
            case(n, O.block_configs_quanonet(n, (4, 2, 2, 2)), 1024, check=False, reps=5)
    if 'ldsbig' in which:
        for n in (10, 11, 12):
            case(n, O.block_configs_quanonet(n, (4, 2, 2, 2)), 1024, check=False, reps=5)
