"""Soak: tens of thousands of training steps of cfg 2's model at several batches with the status word and the results checked
(a hand-off overrun would show as QHEA_EPIPELINE / NaN).  Usage: python scripts/exp/train_soak.py [seconds per batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda', 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
out = {}
for batch in (100, 512, 1024, 1280, 2048):
    torch.manual_seed(0)
    tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
    rng = np.random.default_rng(0); nb = 8
    br = torch.tensor(rng.normal(size=(nb * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 2)), device=dev)
    y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
    rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)
        steps += 400
        torch.cuda.synchronize()
        tr.check_status()
        assert torch.isfinite(rows).all() and torch.isfinite(tr.pflat).all()
    out[batch] = {'steps': steps, 'us_per_step': round(1e6 * (time.perf_counter() - t0) / steps, 2), 'sse_last': float(rows[-1, -2])}
    print(batch, out[batch], flush=True)
print(json.dumps(out))
