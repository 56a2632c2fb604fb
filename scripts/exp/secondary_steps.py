"""Training steps of ONE of bench.py's secondary configurations, for a kernel trace (rocprofv3 --kernel-trace --stats -- python3
scripts/exp/secondary_steps.py <cfg1|cfg3|cfg4|cfg5>): the kernels behind `secondary` in the bench line and their durations
-> profiles/r03_secondary_kernel_stats.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd.models import QuanONetPT, HEAQNNPT
from quanonet_amd.solver import DataParallelTrainer

CASES = {'cfg1': ('quanonet', 2, (5, 1, 5, 1), 10, 1, 32, 2000), 'cfg3': ('quanonet', 5, (40, 2, 20, 2), 100, 2, 512, 1000),
         'cfg4': ('heaqnn', 8, (20, 2), 102, 0, 2048, 400), 'cfg5': ('quanonet', 12, (40, 2, 20, 2), 100, 2, 1024, 24)}
kind, n, net, b_in, t_in, batch, steps = CASES[sys.argv[1]]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = (QuanONetPT(n, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=True) if kind == 'quanonet'
         else HEAQNNPT(n, b_in, net, scale_coeff=0.1, if_trainable_freq=True)).to(dev)
tr = DataParallelTrainer(model, lr=1e-4)
nbat = 4
rng = np.random.default_rng(7)
ins = [torch.tensor(rng.normal(size=(nbat * batch, b_in)), device=dev)]
if kind == 'quanonet':
    ins.append(torch.tensor(rng.uniform(size=(nbat * batch, t_in)), device=dev))
y = torch.tensor(rng.normal(scale=0.5, size=(nbat * batch, 1)), device=dev)
rows = torch.zeros(nbat, tr.numel + 2, dtype=torch.float64, device=dev)
bounds = [j * batch for j in range(nbat + 1)]
for _ in range(steps // nbat):
    tr.train_steps(ins, y, bounds, [batch] * nbat, rows)
torch.cuda.synchronize()
tr.check_status()
print(sys.argv[1], 'ok', steps, 'steps')
