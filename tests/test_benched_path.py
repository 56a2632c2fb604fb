"""
The EXACT path bench.py times, at its own size, against the oracle.

bench.py's timed region is ``DataParallelTrainer.train_steps`` -> ``qhea_model_train_steps`` on
``QuanONetPT(5, 100, 2, (40, 2, 20, 2))`` with trainable frequency, float64: the pipelined backward kernel with in-kernel
encoding angles from branch / trunk (60 blocks, b_in = 100 -> 200 tiled columns), the reduce kernel with the fused Adam
update that also writes the next step's layer records.  Here three consecutive steps of that call are compared with an
independent CPU loop: oracle gradients (``oracle.hea_oracle.quanonet_loss_and_grads`` on the C engine) +
``torch.optim.Adam`` -- every step's ``[grads | sse | sum y^2]`` row and the parameters after the last step at 1e-9
(reference: core/models_pt.py:103-166, solvers/solver_pt.py:226-241).

Also here: the trainable-frequency model at n = 10 and n = 12 (cfg 5's model) through ``qhea_model_loss_grad`` against
the oracle -- the workgroup-resident kernels at model level.
"""
import numpy as np
import pytest
import torch

from oracle import hea_oracle as O
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
TOL = 1e-9

N_QUBITS, NET, B_IN, T_IN = 5, (40, 2, 20, 2), 100, 2          # bench.py's workload


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _oracle_adam_loop(model_cpu, names, branch, trunk, y, bounds, gbs, n, net, lr):
    """Independent CPU loop: oracle loss/gradients + torch.optim.Adam.  Returns (rows [steps, P+2], final flat params)."""
    params = [p for _, p in model_cpu.named_parameters()]
    opt = torch.optim.Adam(params, lr=lr)
    rows = []
    for i in range(len(gbs)):
        lo, hi = bounds[i], bounds[i + 1]
        sd = {k: v.detach().numpy() for k, v in model_cpu.state_dict().items()}
        loss, grads, _ = O.quanonet_loss_and_grads(sd, branch[lo:hi], trunk[lo:hi], y[lo:hi], n, net,
                                                   batch_total=gbs[i], engine=C)
        flat = np.concatenate([np.asarray(grads[k], np.float64).reshape(-1) for k in names])
        rows.append(np.concatenate([flat, [loss * gbs[i], float((y[lo:hi] ** 2).sum())]]))
        opt.zero_grad()
        for k, p in zip(names, params):
            p.grad = torch.from_numpy(np.asarray(grads[k], np.float64).reshape(p.shape).copy())
        opt.step()
    return np.stack(rows), np.concatenate([p.detach().numpy().reshape(-1) for p in params])


@pytest.mark.parametrize('batch,variant', [(1024, 'auto'), (512, 'auto'), (1000, 'auto'), (1024, 'ztri'), (512, 'ztri'),
                                           (100, 'auto'), (1024, 'zquad'), (333, 'zquad'), (1280, 'auto')])
def test_benched_train_steps_match_oracle_and_torch_adam(dev, batch, variant):
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    steps, lr = 3, 1e-3                           # a larger lr than the bench's 1e-4: the steps must differ visibly
    rng = np.random.default_rng(77 + batch)
    n_rows = steps * batch
    branch = rng.normal(size=(n_rows, B_IN)); trunk = rng.uniform(size=(n_rows, T_IN))
    y = rng.normal(scale=0.5, size=n_rows)
    bounds = [i * batch for i in range(steps + 1)]
    gbs = [batch] * steps

    torch.manual_seed(0)
    model = QuanONetPT(N_QUBITS, B_IN, T_IN, NET, scale_coeff=0.1, if_trainable_freq=True)
    with torch.no_grad():                           # non-trivial frequency biases (the init is zeros)
        model.branch_freq.bias.copy_(torch.from_numpy(rng.normal(scale=0.3, size=model.branch_freq.bias.shape)))
        model.trunk_freq.bias.copy_(torch.from_numpy(rng.normal(scale=0.3, size=model.trunk_freq.bias.shape)))
        model.bias.fill_(0.2)
    import copy
    cpu_model = copy.deepcopy(model).double()
    names = [k for k, _ in cpu_model.named_parameters()]
    want_rows, want_params = _oracle_adam_loop(cpu_model, names, branch, trunk, y, bounds, gbs, N_QUBITS, NET, lr)

    _lib.set_backward_variant(variant)
    try:
        tr = DataParallelTrainer(model.double().to(dev), lr=lr)
        assert tr.accepts_out and tr.numel == 2401
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        rows = torch.zeros(steps, tr.numel + 2, dtype=torch.float64, device=dev)
        tr.train_steps([t(branch), t(trunk)], t(y).reshape(-1, 1), bounds, gbs, rows)
        torch.cuda.synchronize()
        tr.check_status()
    finally:
        _lib.set_backward_variant('auto')
    got = rows.cpu().numpy()
    for i in range(steps):
        np.testing.assert_allclose(got[i], want_rows[i], rtol=0, atol=TOL, err_msg=f'step {i}')
    np.testing.assert_allclose(tr.pflat.cpu().numpy(), want_params, rtol=0, atol=TOL)
    # the steps did move the parameters (the comparison is not vacuous)
    assert np.abs(want_rows[0][:-2] - want_rows[2][:-2]).max() > 1e-6


@pytest.mark.parametrize('n,B', [(10, 12), (12, 8)])
def test_trainable_frequency_model_at_n10_n12_matches_oracle(dev, n, B):
    """cfg 5's model (QuanONetPT(12,100,2,(40,2,20,2)), trainable frequency) through qhea_model_loss_grad /
    qhea_model_forward: the workgroup-resident kernels with in-kernel encoding, against the oracle (C engine)."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    rng = np.random.default_rng(n)
    torch.manual_seed(n)
    model = QuanONetPT(n, B_IN, T_IN, NET, scale_coeff=0.1, if_trainable_freq=True).double()
    with torch.no_grad():
        model.branch_freq.bias.copy_(torch.from_numpy(rng.normal(scale=0.3, size=model.branch_freq.bias.shape)))
        model.trunk_freq.bias.copy_(torch.from_numpy(rng.normal(scale=0.3, size=model.trunk_freq.bias.shape)))
        model.bias.fill_(-0.1)
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    branch = rng.normal(size=(B, B_IN)); trunk = rng.uniform(size=(B, T_IN)); y = rng.normal(scale=0.5, size=B)
    loss, grads, out = O.quanonet_loss_and_grads(sd, branch, trunk, y, n, NET, batch_total=3 * B, engine=C)
    want = np.concatenate([np.asarray(grads[k], np.float64).reshape(-1) for k in names])

    tr = DataParallelTrainer(model.to(dev), lr=1e-3, fused=True)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    flat = tr.loss_and_grad(t(branch), t(trunk), t(y), global_batch=3 * B).clone()
    pred = _lib.model_forward(tr.desc, t(branch), t(trunk), tr.pflat).cpu().numpy()
    tr.check_status()
    np.testing.assert_allclose(pred, out, rtol=0, atol=1e-10)
    np.testing.assert_allclose(flat[:-2].cpu().numpy(), want, rtol=0, atol=1e-10)
    assert abs(flat[-2].item() - loss * 3 * B) < 1e-9
    assert abs(flat[-1].item() - float((y ** 2).sum())) < 1e-9
