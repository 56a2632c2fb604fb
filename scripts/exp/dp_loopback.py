"""On-device cost of the data-parallel exchange MECHANISM, without a second rank's kernels competing for the GPU (the only
multi-rank runs possible on a one-GPU box share the device, and there the waiting blocks of one rank take compute units from the
other's circuit kernel -- DESIGN.md section 8).  One process, world = 2, with a library built with -DQHEA_DP_LOOPBACK (hea_dp.hpp:
a publisher stores its tagged words under EVERY rank's slot of both buffers, so that "rank 1"'s contribution -- a copy of rank 0's --
is found by the poll): every block of rank 0's reduce kernel does all of a real exchange's work (the stores into both buffers,
the poll of the peer's slots in its own, the rank-order sum) with no second rank on the GPU.  The step time against the
single-device step is what the mechanism costs on the device; what a real peer adds on top is the xGMI hop and the skew between
the ranks.  (The sums are then twice the local gradients: timing only.)
Usage: make -C quanonet_amd/csrc QUBITS="2 5" SUBSET="-D'QHEA_SUBSET(X)=X(2) X(5)' -D'QHEA_ZSUBSET(X)=X(2) X(5)' -DQHEA_DP_LOOPBACK" \
            OBJDIR=../../build/obj_loop TARGET=../../scripts/exp/libq_loop.so all
       QHEA_LIB=scripts/exp/libq_loop.so python scripts/exp/dp_loopback.py"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer

dev = torch.device('cuda', 0)


def timeit(fn, reps=25, inner=5, per=8):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / (inner * per))
    return 1e6 * float(np.median(ts))


out = {}
for batch in (100, 512, 1024):
    torch.manual_seed(0)
    tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
    n_values = tr.numel + 2
    own = _lib.dp_alloc(n_values, 2, dev)
    peer = _lib.dp_alloc(n_values, 2, dev)
    rng = np.random.default_rng(0); nb = 8
    br = torch.tensor(rng.normal(size=(nb * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 2)), device=dev)
    y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
    rows = torch.zeros(nb, n_values, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
    opt = tr.optimizer; g = opt.param_groups[0]
    seq = [0]

    def single():
        tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)

    def fused():
        _lib.model_dp_train_steps(tr.desc, bounds, [batch] * nb, br, tk, y.reshape(-1), tr.pflat, rows, opt.exp_avg, opt.exp_avg_sq,
                                  opt.t + 1, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], 0, 2, [own, peer],
                                  n_values, seq[0] + 1, timeout_ms=2000.0)
        opt.t += nb; seq[0] += nb

    def separate():
        for i in range(nb):
            a, b = bounds[i], bounds[i + 1]
            _lib.model_loss_grad(tr.desc, br[a:b], tk[a:b], y[a:b].reshape(-1), tr.pflat, 1.0 / batch, rows[i])
            opt.t += 1; seq[0] += 1
            _lib.dp_allreduce_adam(0, 2, [own, peer], seq[0], rows[i], rows[i], tr.pflat, opt.exp_avg, opt.exp_avg_sq, opt.t, g['lr'],
                                   timeout_ms=2000.0)
    t1 = timeit(single)
    tf = timeit(fused)
    ts = timeit(separate)
    _lib.dp_status(own, dev)
    assert torch.isfinite(rows).all()
    out[batch] = {'single_device_step_us': round(t1, 2), 'exchange_inside_reduce_us': round(tf, 2), 'separate_exchange_kernel_us': round(ts, 2),
                  'mechanism_cost_inside_reduce_us': round(tf - t1, 2), 'mechanism_cost_separate_kernel_us': round(ts - t1, 2)}
    print(batch, out[batch], flush=True)
    _lib.dp_free(own, dev); _lib.dp_free(peer, dev)
print(json.dumps(out))
