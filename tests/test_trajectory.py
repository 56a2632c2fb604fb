"""
Training-loop parity: PTSolver.train against the committed trace of an independent loop
(tests/golden/make_trajectory.py: oracle gradients + torch.optim.Adam on CPU, np.random.permutation order, a short
last batch every epoch, best-by-mean-train-loss rule; reference solvers/solver_pt.py:191-277).

* CPU (`-m "not gpu"`): the product's PTSolver / DataParallelTrainer / model classes with the quantum layer swapped
  for the oracle double -- host logic only (batch order, last-batch normalisation, best-checkpoint rule).
* GPU (`-m gpu`): the real HIP path (fused three-launch step with in-kernel Adam) against the same trace.
"""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

NAMES = ['quanonet_tf', 'heaqnn_tf', 'quanonet_ff']


def _build_cpu_model(cfg, data):
    from quanonet_amd.models import QuanONetPT, HEAQNNPT
    tf = cfg['if_trainable_freq'] == 'true'
    if cfg['model_type'] == 'QuanONet':
        m = QuanONetPT(cfg['num_qubits'], data['train_branch_input'].shape[1], data['train_trunk_input'].shape[1],
                       tuple(cfg['net_size']), scale_coeff=cfg['scale_coeff'], if_trainable_freq=tf)
    else:
        m = HEAQNNPT(cfg['num_qubits'], data['train_input'].shape[1], tuple(cfg['net_size']),
                     scale_coeff=cfg['scale_coeff'], if_trainable_freq=tf)
    return H.swap_in_oracle_layer(m)


def _assert_matches_trace(solver, hist, a, order, tol):
    np.testing.assert_array_equal(np.stack(hist['indices']), a['indices'])          # same epoch orders
    np.testing.assert_allclose(hist['loss_steps'], a['step_loss'], rtol=0, atol=tol)   # every batch, short ones included
    np.testing.assert_allclose(hist['loss_train'], a['epoch_loss'], rtol=0, atol=tol)
    sd = {k: v.detach().cpu().numpy() for k, v in solver.model.state_dict().items()}
    for k in order:
        np.testing.assert_allclose(sd[k].reshape(-1), a['final.' + k].reshape(-1), rtol=0, atol=tol, err_msg='final ' + k)
    best = np.load(os.path.join(solver.out_dir, 'best_model.npz'))
    assert int(np.argmin(hist['loss_train'])) == int(a['best_epoch'])
    for k in order:
        np.testing.assert_allclose(best[k].reshape(-1), a['best.' + k].reshape(-1), rtol=0, atol=tol, err_msg='best ' + k)
    fin = np.load(os.path.join(solver.out_dir, 'final.npz'))
    for k in order:
        np.testing.assert_allclose(fin[k].reshape(-1), a['final.' + k].reshape(-1), rtol=0, atol=tol)


@pytest.mark.parametrize('name', NAMES)
def test_ptsolver_host_loop_reproduces_the_reference_trace_cpu(name, tmp_path):
    from quanonet_amd.solver import PTSolver, set_random_seed
    cfg, data, a, order, seed = H.trajectory_solver_inputs(name, str(tmp_path))
    set_random_seed(seed)
    model = _build_cpu_model(cfg, data)
    for k in order:                                                                  # identical initial weights
        np.testing.assert_array_equal(dict(model.state_dict())[k].numpy().reshape(-1), a['init.' + k].reshape(-1))
    s = PTSolver(cfg, data, device=torch.device('cpu'), log=lambda *x, **k: None, model=model)
    hist = s.train()
    _assert_matches_trace(s, hist, a, order, 1e-11)
    # evaluate() reloads the BEST weights (solver_pt.py:282-286), not the final ones
    s.evaluate(hist)
    sd = {k: v.detach().numpy() for k, v in s.model.state_dict().items()}
    for k in order:
        np.testing.assert_allclose(sd[k].reshape(-1), a['best.' + k].reshape(-1), rtol=0, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_ptsolver_on_hip_reproduces_the_reference_trace(name, tmp_path):
    from quanonet_amd.solver import PTSolver, set_random_seed, FlatAdam
    assert torch.cuda.is_available()
    cfg, data, a, order, seed = H.trajectory_solver_inputs(name, str(tmp_path))
    set_random_seed(seed)
    s = PTSolver(cfg, data, device=torch.device('cuda:0'), log=lambda *x, **k: None)
    assert s.trainer.desc is not None and isinstance(s.trainer.optimizer, FlatAdam)   # the fused three-launch step
    sd0 = {k: v.detach().cpu().numpy() for k, v in s.model.state_dict().items()}
    for k in order:
        np.testing.assert_array_equal(sd0[k].reshape(-1), a['init.' + k].reshape(-1))
    hist = s.train()
    _assert_matches_trace(s, hist, a, order, 1e-9)
    m = s.evaluate(hist)
    assert np.isfinite(m['rel_l2'])
    sd = {k: v.detach().cpu().numpy() for k, v in s.model.state_dict().items()}
    for k in order:
        np.testing.assert_allclose(sd[k].reshape(-1), a['best.' + k].reshape(-1), rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_ptsolver_autograd_module_path_reproduces_the_trace(tmp_path):
    """Same loop through torch autograd on the HIP module (HEACircuitHIP's autograd.Function) and torch's own Adam."""
    from quanonet_amd.solver import PTSolver, set_random_seed
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', str(tmp_path), optimizer_kwargs={'amsgrad': False})
    set_random_seed(seed)
    s = PTSolver(cfg, data, device=torch.device('cuda:0'), log=lambda *x, **k: None)
    s.trainer.desc = None                                           # force the module path
    assert not type(s.trainer.optimizer).__name__ == 'FlatAdam'     # optimizer_kwargs beyond betas/eps/wd -> torch Adam
    hist = s.train()
    _assert_matches_trace(s, hist, a, order, 1e-9)


# ---------------------------------------------------------------------------------------------------------------
# the whole PTSolver loop over several ranks (gloo, CPU): every rank slices the SAME global batch although only
# rank 0's NumPy generator is seeded; the run equals the single-process trace; evaluate() uses rank 0's best weights
# on every rank; a rank with an empty test shard does not break the metrics
# ---------------------------------------------------------------------------------------------------------------
def _solver_worker(rank, world, port, q, tmp, n_test):
    import socket  # noqa: F401
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from quanonet_amd.solver import PTSolver, set_random_seed, shard_slice
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', tmp)
    for k in list(data):
        if k.startswith('test_'):
            data[k] = data[k][:n_test]
    # only rank 0 is seeded like the reference's launcher; the others get unrelated generator states and weights
    set_random_seed(seed if rank == 0 else 12345 + rank)
    model = _build_cpu_model(cfg, data)
    s = PTSolver(cfg, data, device=torch.device('cpu'), dist=dist, rank=rank, world_size=world,
                 log=lambda *x, **k: None, model=model)
    # record which rows every rank takes per step
    taken = []
    orig = s.trainer.train_step

    def spy(*batch, global_batch=None, **kw):
        taken.append((batch[-1].reshape(-1).clone().numpy(), global_batch))
        return orig(*batch, global_batch=global_batch, **kw)
    s.trainer.train_step = spy
    hist = s.train()
    metrics = s.evaluate(hist)
    sd = {k: v.detach().numpy().copy() for k, v in s.model.state_dict().items()}
    q.put((rank, hist['loss_train'], hist['loss_steps'], [i.tolist() for i in hist['indices']],
           [(t.tolist(), g) for t, g in taken], metrics, sd))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_test', [(2, 40), (3, 2)])
def test_multi_rank_ptsolver_equals_the_single_process_trace(world, n_test, tmp_path):
    import socket
    import torch.multiprocessing as mp
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', str(tmp_path))
    sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_solver_worker, args=(r, world, port, q, str(tmp_path), n_test)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    y = a['y'][:, 0]
    bs = cfg['batch_size']
    for rank, loss_train, loss_steps, indices, taken, metrics, sd in res:
        np.testing.assert_array_equal(np.array(indices), a['indices'])               # rank 0's order on every rank
        np.testing.assert_allclose(loss_steps, a['step_loss'], rtol=0, atol=1e-11)
        np.testing.assert_allclose(loss_train, a['epoch_loss'], rtol=0, atol=1e-11)
        for k in order:                                                              # evaluate(): best weights everywhere
            np.testing.assert_allclose(sd[k].reshape(-1), a['best.' + k].reshape(-1), rtol=0, atol=1e-11, err_msg=k)
        assert metrics == res[0][5]
    # the union of the ranks' shards of every step is exactly the single-process batch, in order
    nb = int(np.ceil(len(y) / bs))
    for step in range(len(res[0][4])):
        ep, i = divmod(step, nb)
        want = y[a['indices'][ep][i * bs:(i + 1) * bs]]
        got = np.concatenate([np.asarray(r[4][step][0]) for r in res])
        np.testing.assert_array_equal(got, want)
        assert all(r[4][step][1] == len(want) for r in res)
    # metrics of the sharded evaluation == oracle forward of the best weights on the whole test set
    from oracle import hea_oracle as O
    best = {k: a['best.' + k] for k in order}
    ref = O.quanonet_forward(best, data['test_branch_input'][:n_test], data['test_trunk_input'][:n_test],
                             cfg['num_qubits'], tuple(cfg['net_size']))
    d = ref - data['test_output'][:n_test, 0]
    assert abs(res[0][5]['MSE'] - np.mean(d ** 2)) < 1e-12
    assert abs(res[0][5]['Max_Error'] - np.max(np.abs(d))) < 1e-12
