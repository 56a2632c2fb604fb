"""cfg 4 (HEAQNN Q8, B = 2048): per-window step times in issue order, to see whether the slow values are a warm-up effect or a
second mode.  Usage: python scripts/exp/q8_windows.py [batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd.models import HEAQNNPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda', 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
tr = DataParallelTrainer(HEAQNNPT(8, 102, (20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
rng = np.random.default_rng(0); nb = 4
x = torch.tensor(rng.normal(size=(nb * batch, 102)), device=dev)
y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
ts = []
for w in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_steps([x], y, bounds, [batch] * nb, rows)
    torch.cuda.synchronize(); ts.append(round(1e6 * (time.perf_counter() - t0) / nb, 1))
print(json.dumps({'batch': batch, 'us_per_step_by_window': ts}))
