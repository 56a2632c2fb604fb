"""Throughput of the WHOLE PTSolver.train loop (epoch permutation, batch gathering, step, loss bookkeeping) at the
headline shape -- cfg 2, batch 1024, synthetic rows resident on the device -- beside bench.py's bare training step.
Usage: python scripts/solver_loop_rate.py [rows] [epochs] [batch] [qubits] [net_size e.g. 5,1,5,1] [b_in] [t_in]"""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quanonet_amd.solver import PTSolver, set_random_seed

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100 * 1024
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 5
net = [int(v) for v in sys.argv[5].split(',')] if len(sys.argv) > 5 else [40, 2, 20, 2]
b_in = int(sys.argv[6]) if len(sys.argv) > 6 else 100
t_in = int(sys.argv[7]) if len(sys.argv) > 7 else 2
rng = np.random.default_rng(0)
data = {'train_branch_input': rng.normal(size=(rows, b_in)), 'train_trunk_input': rng.uniform(size=(rows, t_in)),
        'train_output': rng.normal(scale=0.5, size=(rows, 1)),
        'test_branch_input': rng.normal(size=(4096, b_in)), 'test_trunk_input': rng.uniform(size=(4096, t_in)),
        'test_output': rng.normal(scale=0.5, size=(4096, 1))}
cfg = {'model_type': 'QuanONet', 'operator': 'Advection', 'num_qubits': nq, 'net_size': net,
       'scale_coeff': 0.1, 'if_trainable_freq': 'true', 'learning_rate': 1e-4, 'batch_size': batch,
       'num_epochs': 1, 'if_save': False, 'prefix': tempfile.mkdtemp()}
set_random_seed(0)
s = PTSolver(cfg, data, device=torch.device('cuda', 0), log=lambda *a, **k: None)
s.train()                                                    # warm-up epoch
torch.cuda.synchronize()
s.config['num_epochs'] = epochs
t0 = time.perf_counter()
s.train()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = epochs * int(np.ceil(rows / batch))
print(json.dumps({'qubits': nq, 'net_size': net, 'batch': batch, 'rows': rows, 'epochs': epochs, 'steps': steps, 'ms_per_step': 1e3 * dt / steps,
                  'train_samples_per_s': epochs * rows / dt}))
