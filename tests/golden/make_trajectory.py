#!/usr/bin/env python3
"""
Generates tests/golden/ptsolver_trajectory.npz: the (indices, loss-per-step, parameters) trace of an INDEPENDENT
restatement of the reference's training loop (solvers/solver_pt.py:191-277) that SURVEY.md 8(a) row A0 asks for.

Nothing of the product is used: the model is the oracle (oracle/hea_oracle.py, C engine for the circuit), the
gradients are the oracle's closed-form sums, the optimizer is ``torch.optim.Adam`` on CPU float64 tensors with
torch's defaults (what the reference builds, solver_pt.py:149-163), the batch order is ``np.random.permutation``
from the seeded global generator (solver_pt.py:220; seeding as utils/common.py:154-170 does it).  The sample count
is not a multiple of the batch size, so every epoch ends in a short batch whose MSE is a mean over ITS rows
(nn.MSELoss on the batch, solver_pt.py:232-236), and the learning rate is large enough that the best epoch (lowest
mean of the batch losses, solver_pt.py:243-257) is not the last one.

    python tests/golden/make_trajectory.py        # rewrites the fixture (deterministic)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hea_oracle as O      # noqa: E402
from oracle import c_oracle as C        # noqa: E402

CASES = {
    # name: (model_type, n, net, b_in, t_in, N, batch, epochs, lr, scale, trainable_freq, seed)
    'quanonet_tf': ('QuanONet', 3, (3, 1, 2, 2), 6, 2, 230, 100, 6, 0.15, 0.1, True, 3),
    'heaqnn_tf': ('HEAQNN', 4, (3, 2), 7, 0, 130, 50, 5, 0.1, 0.2, True, 5),
    'quanonet_ff': ('QuanONet', 2, (2, 1, 2, 1), 4, 1, 90, 40, 4, 0.2, 0.3, False, 7),
}


def param_order(model_type, trainable):
    """nn.Module.parameters() order of the reference classes (core/models_pt.py:124-151, 188-203)."""
    if model_type == 'QuanONet':
        freq = ['branch_freq.weights', 'branch_freq.bias', 'trunk_freq.weights', 'trunk_freq.bias'] if trainable else []
        return ['bias'] + freq + ['quantum_layer.ansatz_weights']
    return (['freq.weights', 'freq.bias'] if trainable else []) + ['quantum_layer.ansatz_weights']


def init_params(model_type, n, net, scale, trainable, seed):
    """Initial parameters as the reference constructs them after torch.manual_seed(seed): the circuit weights are the
    first draw of the CPU generator, U(-pi,pi) in float32 (core/quantum_circuits_tq.py:50-53); frequency weights =
    scale, biases = 0 (core/models_pt.py:35-36), model bias 0 (:151)."""
    torch.manual_seed(seed)
    blk = (net[0] * net[1] + net[2] * net[3]) if model_type == 'QuanONet' else net[0] * net[1]
    w = torch.empty(blk, 3, n).uniform_(-np.pi, np.pi).numpy().astype(np.float64)
    p = {'quantum_layer.ansatz_weights': w}
    if model_type == 'QuanONet':
        p['bias'] = np.zeros(1)
        if trainable:
            for nm, d in (('branch_freq', net[0] * n), ('trunk_freq', net[2] * n)):
                p[nm + '.weights'] = np.full(d, scale)
                p[nm + '.bias'] = np.zeros(d)
    elif trainable:
        p['freq.weights'] = np.full(net[0] * n, scale)
        p['freq.bias'] = np.zeros(net[0] * n)
    return p


def make_data(model_type, b_in, t_in, N, seed):
    rng = np.random.default_rng(1000 + seed)
    if model_type == 'QuanONet':
        br = rng.normal(size=(N, b_in)); tr = rng.uniform(size=(N, t_in))
        y = (np.sin(2.0 * tr[:, 0]) * br[:, 0] * 0.5 + 0.1 * br[:, 1])[:, None]
        return (br, tr), y
    x = rng.normal(size=(N, b_in))
    y = (np.tanh(x[:, 0]) * 0.7 - 0.2 * x[:, 1])[:, None]
    return (x,), y


def reference_loop(name):
    model_type, n, net, b_in, t_in, N, bs, epochs, lr, scale, trainable, seed = CASES[name]
    inputs, y = make_data(model_type, b_in, t_in, N, seed)
    p = init_params(model_type, n, net, scale, trainable, seed)
    order = param_order(model_type, trainable)
    tens = {k: torch.tensor(p[k], dtype=torch.float64, requires_grad=True) for k in order}
    opt = torch.optim.Adam([tens[k] for k in order], lr=lr)
    np.random.seed(seed)
    nb = max(1, int(np.ceil(N / bs)))
    all_idx, step_loss, epoch_loss, epoch_rel = [], [], [], []
    best, best_epoch, best_params = float('inf'), -1, None
    for epoch in range(epochs):
        indices = np.random.permutation(N)
        all_idx.append(indices)
        tot, sse, ysq = 0.0, 0.0, 0.0
        for i in range(nb):
            idx = indices[i * bs:(i + 1) * bs]
            cur = {k: tens[k].detach().numpy() for k in order}
            sc = None if trainable else scale
            if model_type == 'QuanONet':
                loss, grads, _ = O.quanonet_loss_and_grads(cur, inputs[0][idx], inputs[1][idx], y[idx, 0], n, net,
                                                           scale_coeff=sc, engine=C)
            else:
                loss, grads, _ = O.heaqnn_loss_and_grads(cur, inputs[0][idx], y[idx, 0], n, net, scale_coeff=sc, engine=C)
            opt.zero_grad()
            for k in order:
                tens[k].grad = torch.tensor(np.asarray(grads[k], np.float64).reshape(tens[k].shape))
            opt.step()
            step_loss.append(loss)
            tot += loss
            sse += loss * len(idx)
            ysq += float(np.sum(y[idx] ** 2))
        avg = tot / nb
        epoch_loss.append(avg)
        epoch_rel.append(np.sqrt(sse) / (np.sqrt(ysq) + 1e-8))
        if avg < best:
            best, best_epoch = avg, epoch
            best_params = {k: tens[k].detach().numpy().copy() for k in order}
    out = {'indices': np.stack(all_idx), 'step_loss': np.array(step_loss), 'epoch_loss': np.array(epoch_loss),
           'epoch_rel': np.array(epoch_rel), 'best_epoch': np.array(best_epoch), 'y': y}
    for i, a in enumerate(inputs):
        out[f'input{i}'] = a
    for k in order:
        out['final.' + k] = tens[k].detach().numpy()
        out['best.' + k] = best_params[k]
        out['init.' + k] = p[k]
    return out


def main():
    blob = {}
    for name in CASES:
        for k, v in reference_loop(name).items():
            blob[f'{name}/{k}'] = v
        print(name, 'epoch losses', blob[f'{name}/epoch_loss'], 'best epoch', int(blob[f'{name}/best_epoch']))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ptsolver_trajectory.npz'), **blob)


if __name__ == '__main__':
    main()
