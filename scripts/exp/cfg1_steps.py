"""cfg 1 (Antideriv QuanONet Q2 Net5-1-5-1, batch 32) and the reference's batch of 100: step and forward time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda', 0)
for batch in (32, 100, 1024):
    torch.manual_seed(0)
    tr = DataParallelTrainer(QuanONetPT(2, 10, 1, (5, 1, 5, 1), scale_coeff=0.001, if_trainable_freq=True).to(dev), lr=1e-4)
    rng = np.random.default_rng(0); nb = 8
    br = torch.tensor(rng.normal(size=(nb * batch, 10)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 1)), device=dev)
    y = torch.tensor(rng.normal(size=(nb * batch, 1)), device=dev); o = torch.empty(nb * batch, dtype=torch.float64, device=dev)
    rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
    def timeit(fn):
        for _ in range(3): fn()
        ts = []
        for _ in range(15):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
        return 1e6 * float(np.median(ts))
    print(batch, 'step %.2f us' % timeit(lambda: tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)),
          'forward %.2f us' % timeit(lambda: _lib.model_forward_chunks(tr.desc, br, tk, tr.pflat, batch, out=o)), flush=True)
