import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/quanonet_amd') else os.getcwd())
import numpy as np, torch
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda', 0)
batch = 1024; nb = 4
torch.manual_seed(0)
tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4)
rng = np.random.default_rng(0)
br = torch.tensor(rng.normal(size=(nb * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(nb * batch, 2)), device=dev)
y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
for _ in range(200):
    tr.train_steps([br, tk], y, bounds, [batch] * nb, rows)
torch.cuda.synchronize()
