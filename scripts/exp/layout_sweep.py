"""Layout experiment: Q5 packed kernels with 5 / 4 / 3 lane bits per sample (1 / 2 / 4 amplitudes
per lane, 2 / 4 / 8 samples per wave).  Each library runs in its own child process (QHEA_LIB); numerics against the
C oracle first, then HIP-event medians of qhea_forward and qhea_backward (own forward sweep, packed variant).
Build the two experiment libraries with `make -C quanonet_amd/csrc explb LB=4` and `... LB=3`."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import hea_oracle as O, c_oracle as C
from quanonet_amd import _lib
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
n = 5; cfgs = O.block_configs_quanonet(5, (40, 2, 20, 2)); E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
w = rng.uniform(-3, 3, (blk, 3, n)); off, co = O.ham_params(n)
sh = _lib.CircuitShape(n, cfgs)
_lib.set_backward_variant('packed')
x = rng.uniform(-3, 3, (37, E)); g = rng.normal(size=37)
ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
gx, gw, out = _lib.hea_backward(sh, t(x), t(w), t(g), off, co, want_out=True)
err = max(float(np.abs(out.cpu().numpy() - ro).max()), float(np.abs(gx.cpu().numpy() - rgx).max()), float(np.abs(gw.cpu().numpy() - rgw).max()))
res = {'max_err_vs_oracle': err}
wd = t(w)
for B in (1024, 4096, 16384):
    xd = t(rng.uniform(-3, 3, (B, E))); gd = t(rng.normal(size=B))
    def timeit(fn, reps=30):
        for _ in range(5): fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return round(1e3 * ts[len(ts) // 2], 1)
    res[f'B{B}_fwd_us'] = timeit(lambda: _lib.hea_forward(sh, xd, wd, off, co))
    res[f'B{B}_bwd_us'] = timeit(lambda: _lib.hea_backward(sh, xd, wd, gd, off, co))
print('RESULT ' + json.dumps(res))
'''
libs = [('LB=5 (shipped: 1 amplitude/lane, 2 samples/wave)', os.path.join(ROOT, 'quanonet_amd', 'libquanonet_hea.so')),
        ('LB=4 (2 amplitudes/lane, 4 samples/wave)', os.path.join(ROOT, 'scripts', 'exp', 'libquanonet_hea_lb4.so')),
        ('LB=3 (4 amplitudes/lane, 8 samples/wave)', os.path.join(ROOT, 'scripts', 'exp', 'libquanonet_hea_lb3.so'))]
for name, lib in libs:
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}], env=dict(os.environ, QHEA_LIB=lib), capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
    print(name, line[0][7:] if line else ('FAILED: ' + r.stderr[-600:]), flush=True)
