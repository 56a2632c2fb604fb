"""
Data-parallel exchange over peer-mapped buffers (include/quanonet_hea.h: qhea_dp_*, csrc/hea_dp.hip), exercised with
several processes that share the ONE GPU of the test box (hipIpc maps a buffer of another process on the same device
just as it maps one on a peer device; what a one-GPU box cannot show is the xGMI hop itself).

* raw exchange: sums in rank order, both slot parities, slot reuse over many rounds, in-place operation, Adam fused;
* the whole PTSolver loop over 2 and 3 ranks through the exchange equals the committed single-process trace
  (tests/golden/ptsolver_trajectory.npz, the independent CPU loop of make_trajectory.py);
* a rank that never hears from a peer reports QHEA_EEXCHANGE, leaves NaN gradients and does not update -- and so does the
  late rank and every later exchange (a timeout is fatal on every rank);
* the data-parallel step with the exchange inside the reduce kernel (qhea_model_dp_train_steps) equals the separate
  exchange bitwise.
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
    px, reason = PeerExchange.create(dist, rank, world, n, dev)
    assert px is not None, reason
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
    """Rank 1 stays away from exchange 1 until rank 0 has given up on it (50 ms), then issues it: the late rank must fail
    too and leave its parameters alone (a timeout is fatal on every rank), and so must every later exchange."""
    import time
    dist = _init(rank, world, port)
    from quanonet_amd import _lib
    from quanonet_amd.solver import PeerExchange
    dev = torch.device('cuda', 0)
    n = 64
    px, reason = PeerExchange.create(dist, rank, world, n, dev)
    assert px is not None, reason

    def one(timeout_ms):
        local = torch.ones(n, dtype=torch.float64, device=dev)
        p = torch.ones(n - 2, dtype=torch.float64, device=dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        px.seq += 1
        _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, local, local, p, m, v, 1, 1e-2, timeout_ms=timeout_ms)
        try:
            px.check_status()
            raised = None
        except _lib.QheaError as e:
            raised = str(e)
        return bool(torch.isnan(local).all()), bool((p == 1.0).all()), raised
    if rank == 0:
        first = one(50.0)                                          # rank 1 has not published: overrun
        dist.barrier()
    else:
        dist.barrier()                                             # ... rank 0 has given up by now
        t0 = time.perf_counter()
        first = one(5000.0)                                        # the late rank: fails at once, does not wait out its bound
        assert time.perf_counter() - t0 < 2.0
    dist.barrier()
    second = one(5000.0)                                           # sticky: both ranks fail the next exchange at once
    q.put((rank, first, second))
    dist.barrier()
    px.close()
    dist.destroy_process_group()


def test_missing_contribution_is_fatal_on_every_rank():
    res = _spawn(_timeout_worker, 2)
    for rank, first, second in res:
        for nan_out, untouched, raised in (first, second):
            assert nan_out and untouched, (rank, first, second)
            assert raised is not None and '(-7)' in raised


# ------------------------------------------------------------------------------------------------------------------
# the data-parallel step with the exchange INSIDE the reduce kernel (qhea_model_dp_train_steps): bitwise the separate path
# ------------------------------------------------------------------------------------------------------------------
def _fused_worker(rank, world, port, q, nq, net, kind, rows_per_step):
    dist = _init(rank, world, port)
    from quanonet_amd.models import QuanONetPT, HEAQNNPT
    from quanonet_amd.solver import DataParallelTrainer, shard_slice
    dev = torch.device('cuda', 0)
    steps = len(rows_per_step)
    rng = np.random.default_rng(5)                                  # the same global data on every rank
    n_rows = sum(rows_per_step)
    b_in = 7
    branch = rng.normal(size=(n_rows, b_in)); trunk = rng.uniform(size=(n_rows, 2)); y = rng.normal(size=(n_rows, 1))
    # this rank's contiguous shard of every global batch, concatenated (what PTSolver._stage_epoch gathers)
    sel, bounds, off = [], [0], 0
    for gb in rows_per_step:
        lo, hi = shard_slice(gb, rank, world)
        sel.extend(range(off + lo, off + hi)); bounds.append(bounds[-1] + hi - lo); off += gb
    t = lambda a: torch.tensor(np.ascontiguousarray(a[sel]), dtype=torch.float64, device=dev)
    ins = [t(branch), t(trunk)] if kind == 'quanonet' else [t(np.concatenate([branch, trunk], axis=1))]
    yd = t(y)

    def make(fused):
        torch.manual_seed(11)
        if kind == 'quanonet':
            m = QuanONetPT(nq, b_in, 2, net, scale_coeff=0.1, if_trainable_freq=True).double().to(dev)
        else:
            m = HEAQNNPT(nq, b_in + 2, net, scale_coeff=0.1, if_trainable_freq=True).double().to(dev)
        tr = DataParallelTrainer(m, lr=1e-2, world_size=world, dist=dist)
        assert tr.peer is not None and tr.peer_fused, tr.dp_exchange_reason
        tr.peer_fused = fused
        return tr
    a, b = make(False), make(True)
    rows_a = torch.zeros(steps, a.numel + 2, dtype=torch.float64, device=dev)
    rows_b = torch.zeros_like(rows_a)
    for i in range(steps):                                          # separate path: loss_grad launches + one exchange kernel
        lo, hi = bounds[i], bounds[i + 1]
        a.train_step(*[x[lo:hi] for x in ins], yd[lo:hi], global_batch=rows_per_step[i], out=rows_a[i])
    b.train_steps(ins, yd, bounds, rows_per_step, rows_b)          # exchange inside the reduce kernels, one host call
    torch.cuda.synchronize()
    a.check_status(); b.check_status()
    same_rows = bool(torch.equal(rows_a, rows_b))
    same_p = bool(torch.equal(a.pflat, b.pflat)) and bool(torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq))
    # one more single step through train_step on both (b: qhea_model_dp_train_steps with one step)
    lo, hi = bounds[0], bounds[1]
    fa = a.train_step(*[x[lo:hi] for x in ins], yd[lo:hi], global_batch=rows_per_step[0]).clone()
    fb = b.train_step(*[x[lo:hi] for x in ins], yd[lo:hi], global_batch=rows_per_step[0]).clone()
    torch.cuda.synchronize()
    same_single = bool(torch.equal(fa, fb)) and bool(torch.equal(a.pflat, b.pflat))
    q.put((rank, same_rows, same_p, same_single, b.peer_fused, b.pflat.cpu().numpy(), rows_b.cpu().numpy()))
    dist.barrier()
    a.peer.close(); b.peer.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,nq,net,kind,rows_per_step', [
    (2, 5, (3, 2, 2, 2), 'quanonet', [64, 64, 64, 37]),      # block-unrolled shape: steps 2, 3 take their records from the previous reduce
    (3, 5, (3, 2, 2, 2), 'quanonet', [64, 64, 64, 37]),      # uneven shards (64 = 22 + 21 + 21)
    (2, 4, (3, 2, 2, 1), 'quanonet', [50, 50, 31]),          # mixed sub-layer counts: generic walk, a prep launch per step
    (2, 2, (4, 2, 3, 2), 'quanonet', [32, 32, 32]),          # n = 2
    (2, 8, (2, 2), 'heaqnn', [40, 40, 24]),                  # first-generation kernels (n = 8), no bias
    (2, 10, (2, 1, 1, 1), 'quanonet', [12, 12]),             # workgroup-resident kernels (n = 10)
    (2, 5, (2, 2, 2, 2), 'quanonet', [4608, 4608, 1000]),    # one-wave ZYZ kernel (2304 rows per rank)
])
def test_exchange_inside_the_reduce_kernel_equals_the_separate_exchange_bitwise(world, nq, net, kind, rows_per_step):
    res = _spawn(_fused_worker, world, extra=(nq, net, kind, rows_per_step))
    for rank, same_rows, same_p, same_single, still_fused, p, rows in res:
        assert still_fused, "the fused data-parallel step fell back to the separate exchange"
        assert same_rows and same_p and same_single, (rank, same_rows, same_p, same_single)
        np.testing.assert_array_equal(p, res[0][5])                  # replicas bitwise identical
        np.testing.assert_array_equal(rows, res[0][6])               # every rank holds the same global rows
        assert np.isfinite(rows).all() and (rows[:, -2] > 0).all()


def _calibrate_worker(rank, world, port, q, fault_rank):
    dist = _init(rank, world, port)
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(9)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    branch, trunk, y = t(rng.normal(size=(48, 7))), t(rng.uniform(size=(48, 2))), t(rng.normal(size=(48, 1)))
    torch.manual_seed(3)
    tr = DataParallelTrainer(QuanONetPT(5, 7, 2, (3, 2, 2, 2), scale_coeff=0.1, if_trainable_freq=True).double().to(dev),
                             lr=1e-2, world_size=world, dist=dist)
    assert tr.peer is not None, tr.dp_exchange_reason
    tr.train_step(branch, trunk, y)                              # a state with non-zero Adam moments
    torch.cuda.synchronize()
    before = (tr.pflat.clone(), tr.optimizer.exp_avg.clone(), tr.optimizer.exp_avg_sq.clone(), tr.optimizer.t)
    if rank == fault_rank:                                        # this rank's replica came out one rounding apart
        agree = tr._agree_bitwise
        tr._agree_bitwise = lambda a, b: agree(a, b + 1e-13)
    got = tr.calibrate_exchange(branch, trunk, y, steps=3)
    torch.cuda.synchronize()
    restored = (torch.equal(before[0], tr.pflat) and torch.equal(before[1], tr.optimizer.exp_avg)
                and torch.equal(before[2], tr.optimizer.exp_avg_sq) and before[3] == tr.optimizer.t)
    tr.train_step(branch, trunk, y)                              # the run goes on through whatever was kept
    torch.cuda.synchronize()
    tr.check_status()
    q.put((rank, got is not None, tr.peer is not None, restored, tr.dp_exchange_reason, tr.pflat.cpu().numpy()))
    dist.barrier()
    if tr.peer is not None:
        tr.peer.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('fault_rank', [-1, 1])
def test_calibration_checks_both_forms_bitwise_and_falls_back_to_the_collective(fault_rank):
    """calibrate_exchange runs the same steps through both forms of the peer exchange: they must end on bitwise identical
    parameters on every rank (reason string says so, state restored); a rank whose replica differs by one bit's worth
    sends EVERY rank to the all-reduce."""
    res = _spawn(_calibrate_worker, 2, extra=(fault_rank,))
    for rank, chose, has_peer, restored, reason, p in res:
        assert restored, rank
        if fault_rank < 0:
            assert has_peer and 'bitwise identical' in reason, reason
            # (with two ranks on ONE device the fused trial may also time out under its short bound: then no choice was made)
            assert chose or 'did not complete' in reason
        else:
            assert not chose and not has_peer and 'did NOT end on bitwise identical' in reason, reason
        np.testing.assert_array_equal(p, res[0][5])


def _fused_timeout_worker(rank, world, port, q):
    """The fused step with a missing peer: every block of rank 0's reduce kernel gives up, nothing is updated, the late rank
    fails as well."""
    dist = _init(rank, world, port)
    from quanonet_amd import _lib
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    dev = torch.device('cuda', 0)
    torch.manual_seed(3)
    tr = DataParallelTrainer(QuanONetPT(5, 7, 2, (3, 2, 2, 2), scale_coeff=0.1, if_trainable_freq=True).double().to(dev),
                             lr=1e-2, world_size=world, dist=dist)
    assert tr.peer_fused, tr.dp_exchange_reason
    rng = np.random.default_rng(rank)
    br = torch.tensor(rng.normal(size=(20, 7)), device=dev); tk = torch.tensor(rng.uniform(size=(20, 2)), device=dev)
    y = torch.tensor(rng.normal(size=(20, 1)), device=dev)
    before = tr.pflat.clone()
    tr.peer.timeout_ms = 50.0 if rank == 0 else 5000.0
    if rank == 1:
        dist.barrier()
    flat = tr.train_step(br, tk, y, global_batch=40).clone()
    torch.cuda.synchronize()
    if rank == 0:
        dist.barrier()
    try:
        tr.check_status()                                           # collective: agrees on the outcome across the ranks
        raised = None
    except _lib.QheaError as e:
        raised = str(e)
    q.put((rank, bool(torch.isnan(flat).all()), bool(torch.equal(before, tr.pflat)), raised))
    dist.barrier()
    tr.peer.close()
    dist.destroy_process_group()


def test_fused_step_with_a_missing_peer_fails_on_every_rank():
    res = _spawn(_fused_timeout_worker, 2)
    for rank, all_nan, untouched, raised in res:
        assert all_nan and untouched, (rank, all_nan, untouched)
        assert raised is not None and ('(-7)' in raised or 'peer rank' in raised)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE cfg 3 / cfg 5 in their data-parallel form, as far as one GPU allows: the stated per-GPU shard (512 rows of the
# Q5 Net40-2-20-2 model; a few rows of the Q12 one) on every rank, the GLOBAL batch in the loss weight, ranks sharing the
# device -- against the oracle + torch.optim.Adam on the whole global batch
# ------------------------------------------------------------------------------------------------------------------
def _shard_worker(rank, world, port, q, nq, per_rank, steps):
    dist = _init(rank, world, port)
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    dev = torch.device('cuda', 0)
    net, b_in, t_in = (40, 2, 20, 2), 100, 2
    gb = per_rank * world
    rng = np.random.default_rng(3)                                  # the same global data on every rank
    branch = rng.normal(size=(steps * gb, b_in)); trunk = rng.uniform(size=(steps * gb, t_in))
    y = rng.normal(scale=0.5, size=(steps * gb, 1))
    torch.manual_seed(0)
    model = QuanONetPT(nq, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=True).double().to(dev)
    tr = DataParallelTrainer(model, lr=1e-3, world_size=world, dist=dist)
    assert tr.peer is not None, tr.dp_exchange_reason
    tr.peer_fused = False                                           # ranks share the GPU here: the separate exchange kernel
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    rows = []
    for i in range(steps):
        lo = i * gb + rank * per_rank
        flat = tr.train_step(t(branch[lo:lo + per_rank]), t(trunk[lo:lo + per_rank]), t(y[lo:lo + per_rank]), global_batch=gb)
        rows.append(flat.clone())
    torch.cuda.synchronize()
    tr.check_status()
    q.put((rank, torch.stack(rows).cpu().numpy(), tr.pflat.cpu().numpy()))
    dist.barrier()
    tr.peer.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,nq,per_rank,steps', [(4, 5, 512, 2), (2, 12, 6, 2)])
def test_data_parallel_shards_at_the_baseline_configs_match_the_oracle(world, nq, per_rank, steps):
    """cfg 3 (Darcy-shaped Q5 Net40-2-20-2, 512 rows per GPU; 4 of its 8 ranks fit the one-GPU box's process limit) and
    cfg 5's model (Q12, a few rows per rank): every step's GLOBAL [grads | sse | sum y^2] row and the final parameters
    against oracle gradients + torch.optim.Adam over the whole global batch, 1e-9."""
    from oracle import hea_oracle as O
    from oracle import c_oracle as C
    res = _spawn(_shard_worker, world, extra=(nq, per_rank, steps))
    net, b_in, t_in = (40, 2, 20, 2), 100, 2
    gb = per_rank * world
    rng = np.random.default_rng(3)
    branch = rng.normal(size=(steps * gb, b_in)); trunk = rng.uniform(size=(steps * gb, t_in))
    y = rng.normal(scale=0.5, size=(steps * gb, 1))[:, 0]
    # initial parameters exactly as the workers' QuanONetPT constructor draws them under the same seed (the ansatz angles are
    # torch's first draw after manual_seed, in float32: core/quantum_circuits_tq.py:50-53; frequency weights = scale, biases 0)
    import torch.nn as nn
    torch.manual_seed(0)
    w32 = torch.empty(120, 3, nq); nn.init.uniform_(w32, -np.pi, np.pi)
    bd, bl, td, tl = net
    params = {'bias': torch.zeros(1, dtype=torch.float64),
              'branch_freq.weights': torch.full((bd * nq,), 0.1, dtype=torch.float64), 'branch_freq.bias': torch.zeros(bd * nq, dtype=torch.float64),
              'trunk_freq.weights': torch.full((td * nq,), 0.1, dtype=torch.float64), 'trunk_freq.bias': torch.zeros(td * nq, dtype=torch.float64),
              'quantum_layer.ansatz_weights': w32.double()}
    names = list(params)                                             # nn.Module.parameters() order of QuanONetPT
    tp = [nn.Parameter(v.clone()) for v in params.values()]
    opt = torch.optim.Adam(tp, lr=1e-3)
    want_rows = []
    for i in range(steps):
        lo, hi = i * gb, (i + 1) * gb
        sd = {k: p.detach().numpy() for k, p in zip(names, tp)}
        loss, grads, _ = O.quanonet_loss_and_grads(sd, branch[lo:hi], trunk[lo:hi], y[lo:hi], nq, net, engine=C)
        flat = np.concatenate([np.asarray(grads[k], np.float64).reshape(-1) for k in names])
        want_rows.append(np.concatenate([flat, [loss * gb, float((y[lo:hi] ** 2).sum())]]))
        opt.zero_grad()
        for k, p in zip(names, tp):
            p.grad = torch.from_numpy(np.asarray(grads[k], np.float64).reshape(p.shape).copy())
        opt.step()
    want_p = np.concatenate([p.detach().numpy().reshape(-1) for p in tp])
    for rank, rows, p in res:
        for i in range(steps):
            np.testing.assert_allclose(rows[i], want_rows[i], rtol=0, atol=1e-9, err_msg=f'rank {rank} step {i}')
        np.testing.assert_allclose(p, want_p, rtol=0, atol=1e-9)
        np.testing.assert_array_equal(p, res[0][2])                  # replicas bitwise identical
