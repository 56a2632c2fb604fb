"""Forward-only evaluation time of cfg 2's model by chunk size, issued as PTSolver.predict issues it (qhea_model_forward_chunks
over 8 chunks).  Usage: BATCHES=256,512,1024 python scripts/exp/fwd_steps.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer
batches = [int(v) for v in os.environ.get('BATCHES', '128,256,512,768,1024,2048,4096').split(',')]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
out = {}
for batch in batches:
    rng = np.random.default_rng(0)
    nb = 8
    br = torch.tensor(rng.normal(size=(nb * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 2)), device=dev)
    o = torch.empty(nb * batch, dtype=torch.float64, device=dev)
    for _ in range(3):
        _lib.model_forward_chunks(tr.desc, br, tk, tr.pflat, batch, out=o)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            _lib.model_forward_chunks(tr.desc, br, tk, tr.pflat, batch, out=o)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
    t = float(np.median(ts))
    out[batch] = {'us_per_chunk': round(1e6 * t, 2), 'M_evals_per_s': round(batch / t / 1e6, 2)}
    print(batch, out[batch], flush=True)
print(json.dumps(out))
