"""cProfile of one PTSolver epoch at the headline shape: where the HOST spends its time per step."""
import cProfile, pstats, os, sys, tempfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd.solver import PTSolver, set_random_seed
rows = 100 * 1024
rng = np.random.default_rng(0)
data = {'train_branch_input': rng.normal(size=(rows, 100)), 'train_trunk_input': rng.uniform(size=(rows, 2)),
        'train_output': rng.normal(scale=0.5, size=(rows, 1)),
        'test_branch_input': rng.normal(size=(64, 100)), 'test_trunk_input': rng.uniform(size=(64, 2)),
        'test_output': rng.normal(scale=0.5, size=(64, 1))}
cfg = {'model_type': 'QuanONet', 'operator': 'Advection', 'num_qubits': 5, 'net_size': [40, 2, 20, 2],
       'scale_coeff': 0.1, 'if_trainable_freq': 'true', 'learning_rate': 1e-4, 'batch_size': 1024,
       'num_epochs': 1, 'if_save': False, 'prefix': tempfile.mkdtemp()}
set_random_seed(0)
s = PTSolver(cfg, data, device=torch.device('cuda', 0), log=lambda *a, **k: None)
s.train(); torch.cuda.synchronize()
s.config['num_epochs'] = 3
pr = cProfile.Profile(); pr.enable(); s.train(); torch.cuda.synchronize(); pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(22); print(out.getvalue()[:6000])
