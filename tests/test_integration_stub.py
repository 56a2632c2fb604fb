"""
The reference-side binding of INTEGRATION.md (integration/quantum_circuits_hip.py) as a tested artefact: the stub is
imported on its own (it uses neither quanonet_amd nor the reference's core package), wrapped the way the reference's
QuanONetPT / HEAQNNPT wrap a quantum layer (core/models_pt.py:103-213: float32 modules, tile * w + b, trunk first,
+ bias), trained for three steps of the reference's loop body (solvers/solver_pt.py:230-236) and compared with the
oracle doing the same three steps in fp64 -- tolerance 2e-5 (float32 pre/post-processing and parameters).
"""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import hea_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_stub():
    os.environ.setdefault('QHEA_LIB', os.path.join(ROOT, 'quanonet_amd', 'libquanonet_hea.so'))
    spec = importlib.util.spec_from_file_location('quantum_circuits_hip', os.path.join(ROOT, 'integration', 'quantum_circuits_hip.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stub_is_self_contained_and_binds_only_the_header():
    src = open(os.path.join(ROOT, 'integration', 'quantum_circuits_hip.py')).read()
    assert 'import quanonet_amd' not in src and 'from quanonet_amd' not in src and 'from core' not in src
    hdr = open(os.path.join(ROOT, 'include', 'quanonet_hea.h')).read()
    import re
    for sym in set(re.findall(r'\b(qhea_[a-z_]+)\b', src)):
        assert re.search(r'\b%s\s*\(' % sym, hdr), f"{sym} is not declared in include/quanonet_hea.h"
    stub = _load_stub()                     # loads the shared library; no GPU needed for construction
    torch.manual_seed(42)
    layer = stub.build_quanonet_hip(2, 8, 1, (2, 1, 2, 1))
    torch.manual_seed(42)
    ref = torch.empty(4, 3, 2).uniform_(-np.pi, np.pi)
    assert torch.equal(layer.ansatz_weights.detach(), ref) and layer.ansatz_weights.dtype == torch.float32
    assert [k for k, _ in layer.named_parameters()] == ['ansatz_weights'] and layer.block_configs == [(2, 1)] * 4
    h = stub.build_heaqnn_hip(3, 6, (2, 2, 0, 0), ham_diag=np.arange(8.0))
    assert 'ham_diag' in dict(h.named_buffers()) and h.use_full_ham and h.ansatz_weights.shape == (4, 3, 3)
    with pytest.raises(RuntimeError):
        layer(torch.zeros(3, 8))            # CPU tensor: no CPU path


class _Tiled(nn.Module):                     # core/models_pt.py:14-41 restated (float32)
    def __init__(self, n_in, n_out, scale):
        super().__init__()
        self.n_in, self.n_out = n_in, n_out
        self.weights = nn.Parameter(torch.full((n_out,), scale))
        self.bias = nn.Parameter(torch.zeros(n_out))

    def forward(self, x):
        idx = torch.arange(self.n_out, device=x.device) % self.n_in
        return x[:, idx] * self.weights + self.bias


class _QuanONetLike(nn.Module):              # core/models_pt.py:103-166 with the stub as quantum layer
    def __init__(self, stub, n, b_in, t_in, net, scale):
        super().__init__()
        self.branch_freq = _Tiled(b_in, net[0] * n, scale)
        self.trunk_freq = _Tiled(t_in, net[2] * n, scale)
        self.quantum_layer = stub.build_quanonet_hip(n, b_in, t_in, net)
        self.bias = nn.Parameter(torch.zeros(1))

    def forward(self, branch, trunk):
        return self.quantum_layer(torch.cat([self.trunk_freq(trunk), self.branch_freq(branch)], dim=1)) + self.bias


class _HEAQNNLike(nn.Module):                # core/models_pt.py:169-213
    def __init__(self, stub, n, n_in, net, scale):
        super().__init__()
        self.freq = _Tiled(n_in, net[0] * n, scale)
        self.quantum_layer = stub.build_heaqnn_hip(n, n_in, net)

    def forward(self, x):
        return self.quantum_layer(self.freq(x))


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['quanonet', 'heaqnn'])
def test_three_training_steps_through_the_stub_match_the_oracle(kind):
    stub = _load_stub()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(21)
    torch.manual_seed(3)
    n, B, lr = 4, 23, 5e-2
    if kind == 'quanonet':
        net = (2, 2, 2, 1)
        model = _QuanONetLike(stub, n, 6, 2, net, 0.1).to(dev)
        ins = (rng.normal(size=(B, 6)).astype(np.float32), rng.uniform(size=(B, 2)).astype(np.float32))
    else:
        net = (3, 2)
        model = _HEAQNNLike(stub, n, 7, net, 0.1).to(dev)
        ins = (rng.normal(size=(B, 7)).astype(np.float32),)
    y = rng.normal(scale=0.5, size=(B, 1)).astype(np.float32)
    # plain SGD: Adam's g/sqrt(v) turns a float32-sized error on a near-zero gradient into a full-size step, which
    # would test the optimizer's conditioning, not the binding
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    loss_fn = nn.MSELoss()
    # the oracle's copy of the same training run, fp64, the same optimizer on CPU
    names = [k for k, _ in model.named_parameters()]
    ref = {k: torch.tensor(v.detach().cpu().numpy().astype(np.float64), requires_grad=True) for k, v in model.named_parameters()}
    ropt = torch.optim.SGD([ref[k] for k in names], lr=lr)
    tin = [torch.tensor(a, device=dev) for a in ins]
    ty = torch.tensor(y, device=dev)
    for step in range(3):
        opt.zero_grad()
        pred = model(*tin)
        loss = loss_fn(pred, ty)
        loss.backward()
        cur = {k: v.detach().numpy() for k, v in ref.items()}
        if kind == 'quanonet':
            rl, rg, ro = O.quanonet_loss_and_grads(cur, ins[0], ins[1], y[:, 0], n, net)
        else:
            rl, rg, ro = O.heaqnn_loss_and_grads(cur, ins[0], y[:, 0], n, net)
        np.testing.assert_allclose(pred[:, 0].detach().cpu().numpy(), ro, rtol=0, atol=2e-5, err_msg=f'step {step} forward')
        assert abs(loss.item() - rl) < 2e-5
        for k, prm in model.named_parameters():
            np.testing.assert_allclose(prm.grad.cpu().numpy().reshape(-1), rg[k].reshape(-1), rtol=0, atol=2e-5,
                                       err_msg=f'step {step} grad {k}')
        opt.step()
        ropt.zero_grad()
        for k in names:
            ref[k].grad = torch.tensor(np.asarray(rg[k], np.float64).reshape(ref[k].shape))
        ropt.step()
    for k, prm in model.named_parameters():
        np.testing.assert_allclose(prm.detach().cpu().numpy().reshape(-1), ref[k].detach().numpy().reshape(-1), rtol=0,
                                   atol=2e-5, err_msg=f'parameters after 3 steps: {k}')
