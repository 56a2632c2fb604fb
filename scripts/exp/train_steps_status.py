"""30 training steps of the cfg-2 model with the status word checked after every step: python train_steps_status.py [variant] [batch].
With QHEA_LIB pointing at a -DQHEA_DEBUG_ABORT build an overrun prints which wait it was."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from quanonet_amd.models import QuanONetPT
from quanonet_amd import _lib
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
_lib.set_backward_variant(sys.argv[1] if len(sys.argv) > 1 else 'auto')
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
model = QuanONetPT(bench.N_QUBITS, bench.B_IN, bench.T_IN, bench.NET, scale_coeff=0.1, if_trainable_freq=True).to(dev)
trainer = DataParallelTrainer(model, lr=1e-4, world_size=1, dist=None)
branch, trunk, y = bench.synth(0, 2 * batch)
branch = torch.tensor(branch, device=dev); trunk = torch.tensor(trunk, device=dev); y = torch.tensor(y, device=dev)
for i in range(30):
    s = (i % 2) * batch
    trainer.train_step(branch[s:s + batch], trunk[s:s + batch], y[s:s + batch], global_batch=batch)
    torch.cuda.synchronize()
    try:
        _lib.check_status(dev)
    except Exception as e:
        print('step', i, 'FAILED:', str(e)[:80], flush=True); break
else:
    print('30 steps ok', flush=True)
