"""Training-step time of cfg 2's model (Q5 Net40-2-20-2, trainable frequency) by batch and backward variant, issued as
PTSolver's epoch loop issues it (runs of 8 steps per host call).  Usage: python scripts/exp/small_batch_steps.py [variants...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer

variants = sys.argv[1:] or ['ztri', 'zquad', 'auto']
batches = [int(v) for v in os.environ.get('BATCHES', '100,256,512,768,1024,1280,1536').split(',')]
dev = torch.device('cuda', 0)
out = {}
for batch in batches:
    rng = np.random.default_rng(0)
    nb = 8
    br = torch.tensor(rng.normal(size=(nb * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 2)), device=dev)
    y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
    bounds = [i * batch for i in range(nb + 1)]
    for var in variants:
        _lib.set_backward_variant(var)
        torch.manual_seed(0)
        tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
        rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev)
        for _ in range(3):
            tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)
        ts = []
        for _ in range(15):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
        tr.check_status()
        out[f'{batch}/{var}'] = round(1e6 * float(np.median(ts)), 2)
        print(batch, var, out[f'{batch}/{var}'], 'us per step', flush=True)
_lib.set_backward_variant('auto')
print(json.dumps(out))
