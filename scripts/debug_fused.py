import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hea_oracle as O
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
torch.manual_seed(3)
n, net, b_in, t_in, B = 5, (3, 2, 2, 1), 7, 2, 4
model = QuanONetPT(n, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=True).to(dev)
rng = np.random.default_rng(5)
br = rng.normal(size=(B, b_in)); tr = rng.uniform(size=(B, t_in)); y = rng.normal(size=B)
tr_ = DataParallelTrainer(model, fused=True)
d = tr_.desc
print('desc', d.model, d.n_qubits, list(d.net), d.branch_in, d.trunk_in, d.trainable_freq, d.scale_coeff, d.ham_offset, d.ham_coeff, 'P', _lib.model_param_count(d), tr_.numel)
pred = _lib.model_forward(d, t(br), t(tr), tr_.pflat).cpu().numpy()
print('fused', pred)
# inspect cs table in workspace
ws = _lib._workspaces[('cuda', 0)]
sh = model.quantum_layer._shape
E = sh.E
offU = 0; szU = ((sh.blk + 2) * n * 64 + 255) // 256 * 256
cs = ws[szU:szU + B * E * 16].view(torch.float64).view(B, E, 2).cpu().numpy()
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
xt = O.tiled_elementwise(tr, sd['trunk_freq.weights'], sd['trunk_freq.bias']); xb = O.tiled_elementwise(br, sd['branch_freq.weights'], sd['branch_freq.bias'])
x = np.concatenate([xt, xb], 1)
print('cs err', np.abs(cs[..., 0] - np.cos(x / 2)).max(), np.abs(cs[..., 1] - np.sin(x / 2)).max())

gt = ws[:szU].view(torch.float64).view(-1, 8).cpu().numpy()
w = sd['quantum_layer.ansatz_weights']
def RY(t): c,s=np.cos(t/2),np.sin(t/2); return np.array([[c,-s],[s,c]],complex)
def RZ(t): return np.diag([np.exp(-0.5j*t),np.exp(0.5j*t)])
U = RY(w[0,2,0])@RZ(w[0,1,0])@RY(w[0,0,0])
print('gate0 table', gt[n], 'expected', U[0,0].real, U[0,0].imag, U[0,1].real, U[0,1].imag)
with torch.no_grad():
    p2 = model(t(br), t(tr))[:, 0].cpu().numpy()
print('module', p2)
pf = tr_.pflat.cpu().numpy()
print('pflat[50:53]', pf[50:53], 'w[0,0,:3]', w[0,0,:3])
pred = _lib.model_forward(d, t(br), t(tr), tr_.pflat).cpu().numpy()
torch.cuda.synchronize()
gt = ws[:szU].view(torch.float64).view(-1, 8).cpu().numpy()
cs = ws[szU:szU + B * E * 16].view(torch.float64).view(B, E, 2).cpu().numpy()
xs = 2 * np.arctan2(cs[..., 1], cs[..., 0])
print('x got  ', np.round(xs[0, :12], 4)); print('x want ', np.round(x[0, :12], 4))
print('trunk row0', tr[0], 'branch row0', br[0, :5])
for g in range(3):
    print('gate', g, np.round(gt[n + g, :4], 4))
for q in range(3):
    U = RY(w[0,2,q])@RZ(w[0,1,q])@RY(w[0,0,q]); print('exp ', q, np.round([U[0,0].real, U[0,0].imag, U[0,1].real, U[0,1].imag], 4))
