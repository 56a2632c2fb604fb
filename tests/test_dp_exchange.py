"""
Data-parallel exchange over peer-mapped buffers (include/quanonet_hea.h: qhea_dp_*, csrc/hea_dp.hip), exercised with
several processes that share the ONE GPU of the test box (hipIpc maps a buffer of another process on the same device
just as it maps one on a peer device; what a one-GPU box cannot show is the xGMI hop itself).

* raw exchange: sums in rank order, both slot parities, slot reuse over many rounds, in-place operation, Adam fused;
* the whole PTSolver loop over 2 and 3 ranks through the exchange equals the committed single-process trace
  (tests/golden/ptsolver_trajectory.npz, the independent CPU loop of make_trajectory.py);
* a rank that never hears from a peer reports QHEA_EEXCHANGE, leaves NaN gradients and does not update.
"""
import os
import socket

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return dist


def _raw_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from quanonet_amd import _lib
    from quanonet_amd.solver import PeerExchange, FlatAdam
    dev = torch.device('cuda', 0)
    n = 2403
    px = PeerExchange.create(dist, rank, world, n, dev)
    assert px is not None, "peer exchange not available"
    rng = np.random.default_rng(100 + rank)
    worst = 0.0
    for rnd in range(40):                                          # many rounds: parities alternate, slots are reused
        local = torch.tensor(rng.normal(size=n), device=dev)
        all_local = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(all_local, local)
        want = torch.zeros_like(local)
        for t in all_local:                                        # rank order, as the kernel adds
            want = want + t
        px.seq += 1
        _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, local, local)         # in place
        worst = max(worst, float((local - want).abs().max()))
    px.check_status()
    # fused Adam == qhea_adam_step on the summed gradient
    p0 = torch.tensor(np.random.default_rng(7).normal(size=n - 2), device=dev)
    g_local = torch.tensor(np.random.default_rng(200 + rank).normal(size=n), device=dev)
    gs = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gs, g_local)
    gsum = torch.zeros_like(g_local)
    for t in gs:
        gsum = gsum + t
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    _lib.adam_step(pa, gsum, ma, va, 1, 1e-2)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    px.seq += 1
    _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, g_local, g_local, pb, mb, vb, 1, 1e-2)
    torch.cuda.synchronize()
    adam_err = float(max((pa - pb).abs().max(), (ma - mb).abs().max(), (va - vb).abs().max()))
    q.put((rank, worst, adam_err, pb.cpu().numpy()))
    dist.barrier()
    px.close()
    dist.destroy_process_group()


def _spawn(target, world, extra=()):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('world', [2, 4])
def test_exchange_sums_in_rank_order_and_fuses_adam(world):
    res = _spawn(_raw_worker, world)
    for rank, worst, adam_err, p in res:
        assert worst == 0.0                      # the same additions in the same order as the host-side rank-order sum
        assert adam_err == 0.0
        np.testing.assert_array_equal(p, res[0][3])              # replicas bitwise identical


def _solver_worker(rank, world, port, q, tmp, exchange):
    os.environ['QHEA_DP_EXCHANGE'] = exchange
    dist = _init(rank, world, port)
    from quanonet_amd.solver import PTSolver, set_random_seed
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', tmp)
    set_random_seed(seed if rank == 0 else 999 + rank)
    s = PTSolver(cfg, data, device=torch.device('cuda', 0), dist=dist, rank=rank, world_size=world,
                 log=lambda *x, **k: None)
    used_peer = s.trainer.peer is not None
    hist = s.train()
    s.trainer.check_status()
    sd = {k: v.detach().cpu().numpy().copy() for k, v in s.model.state_dict().items()}
    q.put((rank, used_peer, hist['loss_train'], hist['loss_steps'], sd))
    dist.barrier()
    if s.trainer.peer is not None:
        s.trainer.peer.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,exchange', [(2, 'peer'), (3, 'peer'), (2, 'rccl')])
def test_multi_rank_ptsolver_through_the_peer_exchange_equals_the_trace(world, exchange, tmp_path):
    """exchange = 'rccl': the fallback (all_reduce of the step's row + qhea_adam_step) through the same loop."""
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', str(tmp_path))
    res = _spawn(_solver_worker, world, extra=(str(tmp_path), exchange))
    for rank, used_peer, loss_train, loss_steps, sd in res:
        assert used_peer == (exchange == 'peer'), "the data-parallel step did not take the requested exchange"
        np.testing.assert_allclose(loss_steps, a['step_loss'], rtol=0, atol=1e-9)
        np.testing.assert_allclose(loss_train, a['epoch_loss'], rtol=0, atol=1e-9)
        for k in order:
            np.testing.assert_allclose(sd[k].reshape(-1), a['final.' + k].reshape(-1), rtol=0, atol=1e-9, err_msg=k)
            np.testing.assert_array_equal(sd[k], res[0][4][k])                   # replicas bitwise identical


def _timeout_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from quanonet_amd import _lib
    from quanonet_amd.solver import PeerExchange
    dev = torch.device('cuda', 0)
    n = 64
    px = PeerExchange.create(dist, rank, world, n, dev)
    assert px is not None
    out = None
    if rank == 0:                                                  # rank 1 never publishes this round
        local = torch.ones(n, dtype=torch.float64, device=dev)
        p = torch.ones(n - 2, dtype=torch.float64, device=dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        px.seq += 1
        _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, local, local, p, m, v, 1, 1e-2, timeout_ms=50.0)
        try:
            px.check_status()
            raised = None
        except _lib.QheaError as e:
            raised = str(e)
        out = (bool(torch.isnan(local).all()), bool((p == 1.0).all()), raised)
    q.put((rank, out))
    dist.barrier()
    px.close()
    dist.destroy_process_group()


def test_missing_contribution_is_reported_not_computed_through():
    res = _spawn(_timeout_worker, 2)
    nan_out, untouched, raised = res[0][1]
    assert nan_out and untouched
    assert raised is not None and '(-7)' in raised
