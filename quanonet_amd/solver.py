"""
Training loop for QuanONetPT / HEAQNNPT on the HIP quantum layer: host-side mirror of the
reference's ``solvers/solver_pt.py`` (PTSolver) plus the data-parallel step the reference lacks.

* ``DataParallelTrainer``: one training step = forward -> MSE -> in-kernel adjoint backward ->
  ONE sum over the ranks of a flat fp64 buffer [all gradients | sse | sum y^2] -> Adam (``PeerExchange``: both in
  one kernel over peer-mapped buffers; ``dist.all_reduce`` + an Adam launch where that is not available).
  Each rank's loss is ``sum((pred-y)^2) / global_batch`` so that a plain SUM reproduces the
  gradient of ``MSELoss(mean)`` over the global batch (solver_pt.py:232-236), also for uneven
  shards.  The two logging scalars ride in the same buffer, removing the reference's two
  ``.item()`` host syncs per batch (solver_pt.py:238-241).
* ``PTSolver``: config-dict driven epoch/batch loop with the reference's semantics
  (solver_pt.py:191-329): ``np.random.permutation`` batch order, best-by-train-loss checkpoint
  as ``.pt`` + ``.npz`` with the reference's state_dict keys, final checkpoint, evaluate() with
  rel-L2 / MSE / MAE / max-error.  Training data stay resident on the device.
"""
import json
import os
import random

import numpy as np
import torch
import torch.nn as nn


def set_random_seed(seed):
    """
    What the reference's launcher does before it builds the solver (utils/common.py:154-180, main.py): seed
    ``random``, NumPy's global generator (the per-epoch ``np.random.permutation``, solver_pt.py:220) and torch (the
    U(-pi,pi) draw of the circuit weights, core/quantum_circuits_tq.py:50-53).  PTSolver itself never seeds, exactly
    like the reference's; in a multi-rank run only rank 0's NumPy state matters (it draws, the others receive).
    """
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def shard_slice(n_items, rank, world):
    """Contiguous, near-even shard [lo, hi) of a batch of n_items for `rank` of `world`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def regression_metrics(y_pred, y_true, dist=None, world=1):
    """
    MSE / MAE / Max_Error (utils/metrics.py:6-29) and relative L2 (solvers/solver_pt.py:316-321) of a test set that is
    sharded over `world` ranks: every rank passes ITS slice of predictions and targets (torch tensors, any device);
    four sums and one maximum are all-reduced, so every rank returns the metrics of the whole set.
    """
    d = (y_pred.reshape(-1) - y_true.reshape(-1)).to(torch.float64)
    yt = y_true.reshape(-1).to(torch.float64)
    sums = torch.stack([(d * d).sum(), d.abs().sum(), (yt * yt).sum(),
                        torch.tensor(float(d.numel()), dtype=torch.float64, device=d.device)])
    mx = d.abs().max() if d.numel() else torch.zeros((), dtype=torch.float64, device=d.device)
    mx = mx.reshape(1)
    if dist is not None and world > 1:
        dist.all_reduce(sums)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sse, sae, sy2, cnt = sums.tolist()
    return {'MSE': sse / cnt, 'MAE': sae / cnt, 'Max_Error': float(mx.item()),
            'rel_l2': float(np.sqrt(sse) / (np.sqrt(sy2) + 1e-8))}


class FlatAdam(torch.optim.Optimizer):
    """
    torch.optim.Adam arithmetic (amsgrad=False) on the trainer's flat fp64 vector through ``qhea_adam_step``:
    one launch per update instead of torch's two multi-tensor kernels.  A regular ``Optimizer`` (param_groups,
    lr schedulers, state_dict) so that PTSolver's scheduler handling is unchanged.
    """

    def __init__(self, params, pflat, gflat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(list(params), dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.pflat, self.gflat = pflat, gflat
        self.exp_avg = torch.zeros_like(pflat)
        self.exp_avg_sq = torch.zeros_like(pflat)
        self.t = 0

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        g = self.param_groups[0]
        self.t += 1
        _lib.adam_step(self.pflat, self.gflat, self.exp_avg, self.exp_avg_sq, self.t, g['lr'], g['betas'][0],
                       g['betas'][1], g['eps'], g['weight_decay'])

    def state_dict(self):
        sd = super().state_dict()
        sd['flat_state'] = {'step': self.t, 'exp_avg': self.exp_avg.clone(), 'exp_avg_sq': self.exp_avg_sq.clone()}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        fs = sd.pop('flat_state', None)
        super().load_state_dict(sd)
        if fs is not None:
            self.t = int(fs['step'])
            self.exp_avg.copy_(fs['exp_avg'])
            self.exp_avg_sq.copy_(fs['exp_avg_sq'])


class PeerExchange:
    """
    The data-parallel step's gradient sum + Adam update as ONE kernel per rank over peer-mapped exchange buffers
    (include/quanonet_hea.h: qhea_dp_*, csrc/hea_dp.hip) instead of ``dist.all_reduce`` + a separate Adam launch.

    Set-up uses the process group only for hand-shakes: the 64-byte hipIpc handles travel by ``all_gather_object``,
    every rank maps the others' buffers, and a three-round self-check (both slot parities, slot reuse) is compared with
    the analytic sum.  ``create`` returns None -- on EVERY rank, the outcome is agreed by an all-reduce -- when any
    rank could not map a buffer or saw a wrong sum; the trainer then keeps the RCCL all-reduce.  ``QHEA_DP_EXCHANGE=rccl``
    switches the peer path off.
    """

    def __init__(self, dist, rank, world, n_values, device, timeout_ms=5000.0):
        self.dist, self.rank, self.world, self.n, self.device = dist, rank, world, int(n_values), device
        self.timeout_ms = float(timeout_ms)
        self.seq = 0
        self.own = None
        self.bufs = [None] * world

    @classmethod
    def create(cls, dist, rank, world, n_values, device, log=None):
        """Returns (exchange or None, reason): the reason says in words which path the data-parallel step takes and why
        (bench.py prints it as config.dp_exchange_reason; rank 0 logs it)."""
        from . import _lib
        say = log if (log is not None and rank == 0) else (lambda *a, **k: None)

        def out(px, reason):
            say("data-parallel exchange: " + reason)
            return px, reason
        if world < 2:
            return None, "single rank: no exchange"
        if world > _lib.DP_MAX_RANKS or device.type != 'cuda':
            return out(None, f"all_reduce: world size {world} on {device.type} is outside the peer exchange "
                             f"(HIP devices, at most {_lib.DP_MAX_RANKS} ranks)")
        if os.environ.get('QHEA_DP_EXCHANGE', 'peer').lower() in ('rccl', 'nccl', 'off', '0'):
            return out(None, "all_reduce: QHEA_DP_EXCHANGE=" + os.environ['QHEA_DP_EXCHANGE'])
        env = os.environ.get(_lib.IPC_ENV)
        env_note = ""
        if env != '0':
            env_note = f" ({_lib.IPC_ENV}={env!r}: hipIpc handles need the dmabuf mode, value 0)"
        elif _lib.IPC_ENV_SET_LATE:
            env_note = f" ({_lib.IPC_ENV}=0 was set after this process had initialised the GPU: export it before launch)"
        self = cls(dist, rank, world, n_values, device)
        ok, why = 1, ""
        handle = None
        try:
            self.own = _lib.dp_alloc(self.n, world, device)
            handle = _lib.dp_export(self.own, device)
        except _lib.QheaError as e:
            ok, why = 0, f"rank {rank}: {e}"
        handles = [None] * world
        dist.all_gather_object(handles, handle)
        if ok and all(h is not None for h in handles):
            self.bufs[rank] = self.own
            try:
                for r in range(world):
                    if r != rank:
                        self.bufs[r] = _lib.dp_import(handles[r], device)
            except _lib.QheaError as e:
                ok, why = 0, f"rank {rank}: {e}"
        elif ok:
            ok, why = 0, "a peer could not allocate or export its buffer"
        whys = [None] * world
        dist.all_gather_object(whys, why)                      # everybody learns the first failure's text
        if not self._agree(ok):
            self.close()
            first = next((w for w in whys if w), "unknown")
            return out(None, f"all_reduce: peer buffers could not be mapped on every rank [{first}]{env_note}")
        if not self._agree(self._self_check()):
            self.close()
            return out(None, f"all_reduce: the peer exchange's self-check (three rounds against the analytic sum) failed{env_note}")
        return out(self, f"peer-mapped buffers over hipIpc, {world} ranks, {self.n} values per rank; self-check passed")

    def _agree(self, ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def _self_check(self):
        from . import _lib
        good = True
        base = torch.arange(1, self.n + 1, dtype=torch.float64, device=self.device) * 1e-3
        for rnd in range(3):
            local = base * (self.rank + 1) + rnd
            out = torch.empty_like(local)
            self.seq += 1
            _lib.dp_allreduce_adam(self.rank, self.world, self.bufs, self.seq, local, out, timeout_ms=2000.0)
            try:
                _lib.dp_status(self.own, self.device)
            except _lib.QheaError:
                good = False
            want = base * (self.world * (self.world + 1) / 2) + rnd * self.world
            good = good and bool(torch.allclose(out, want, rtol=1e-13, atol=0.0))
        return good

    def allreduce_adam(self, flat, pflat, opt):
        """flat <- sum over ranks of flat (rank order); FlatAdam `opt`'s update of pflat with it; one launch."""
        from . import _lib
        g = opt.param_groups[0]
        opt.t += 1
        self.seq += 1
        _lib.dp_allreduce_adam(self.rank, self.world, self.bufs, self.seq, flat, flat, pflat, opt.exp_avg,
                               opt.exp_avg_sq, opt.t, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                               g['weight_decay'], timeout_ms=self.timeout_ms)

    def check_status(self):
        from . import _lib
        _lib.dp_status(self.own, self.device)

    def close(self):
        """Collective: every rank calls it (also the ranks whose own allocation failed in create)."""
        from . import _lib
        for r, b in enumerate(self.bufs):
            if b is not None and r != self.rank:
                try:
                    _lib.dp_close(b, self.device)
                except _lib.QheaError:
                    pass
        self.bufs = [None] * self.world
        self.dist.barrier()                            # every importer has closed its mapping of the buffers freed next
        if self.own is not None:
            _lib.dp_free(self.own, self.device)
            self.own = None


class DataParallelTrainer:
    """
    fused='auto': when the model is one of this package's QuanONetPT / HEAQNNPT (fp64, on a HIP
    device) the step runs through the model-level C ABI (qhea_model_loss_grad: three launches);
    otherwise through torch autograd on the module (HEACircuitHIP's autograd.Function).  Both paths
    fill the same flat buffer, so the collective and the optimizer step are shared.
    """

    def __init__(self, model, lr=1e-4, world_size=1, dist=None, optimizer='adam', optimizer_kwargs=None,
                 fused='auto', peer_exchange=True, log=None):
        self.model = model
        self.world = int(world_size)
        self.dist = dist
        self.params = [p for p in model.parameters() if p.requires_grad]
        p0 = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        for p in self.params:
            if p.dtype != torch.float64:
                raise ValueError("DataParallelTrainer expects float64 parameters (fp64 training path)")
        # flat parameter vector (reference state_dict order) and flat gradient buffer (+2 logging scalars);
        # every parameter / .grad becomes a view into them
        self.pflat = torch.empty(self.numel, dtype=torch.float64, device=p0.device)
        self.flat = torch.zeros(self.numel + 2, dtype=torch.float64, device=p0.device)
        off = 0
        for p in self.params:
            n = p.numel()
            self.pflat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.pflat[off:off + n].view(p.shape)
            p.grad = self.flat[off:off + n].view(p.shape)
            off += n
        self._last = self.flat
        self.desc = None
        if fused and hasattr(model, 'fused_desc') and p0.is_cuda and len(self.params) == len(list(model.parameters())):
            from . import _lib
            desc = model.fused_desc()
            if _lib.model_param_count(desc) == self.numel:
                self.desc = desc
        if fused is True and self.desc is None:
            raise RuntimeError("fused training path requested but the model does not support it")
        kw = dict(optimizer_kwargs or {})
        if optimizer.lower() == 'lbfgs':
            # as the reference (solvers/solver_pt.py:154-155)
            raise NotImplementedError("LBFGS requires a closure; not supported by the current training loop.")
        opt_map = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW, 'sgd': torch.optim.SGD,
                   'rmsprop': torch.optim.RMSprop}
        cls = opt_map.get(optimizer.lower(), torch.optim.Adam)
        if cls is torch.optim.Adam and p0.is_cuda and set(kw) <= {'betas', 'eps', 'weight_decay'}:
            self.optimizer = FlatAdam(self.params, self.pflat, self.flat, lr=lr, **kw)     # one launch per update
        else:
            if cls in (torch.optim.Adam, torch.optim.AdamW) and p0.is_cuda and 'fused' not in kw:
                kw['fused'] = True
            self.optimizer = cls(self.params, lr=lr, **kw)
        self.peer = None
        self.peer_fused = False              # the exchange runs inside the reduce kernel (qhea_model_dp_train_steps)
        self.fused_ok = True                 # ... and has not been found to stall on this placement of ranks (calibrate_exchange)
        self.dp_exchange_reason = "single rank: no exchange"
        if self.world > 1:
            self.broadcast_parameters()
            if not isinstance(self.optimizer, FlatAdam):
                self.dp_exchange_reason = "all_reduce: the peer exchange fuses the flat Adam update; this optimizer is torch's"
            elif not peer_exchange:
                self.dp_exchange_reason = "all_reduce: requested (peer_exchange=False / config dp_exchange)"
            else:
                self.peer, self.dp_exchange_reason = PeerExchange.create(dist, dist.get_rank(), self.world, self.numel + 2,
                                                                         p0.device, log=log)
                self.peer_fused = self.peer is not None and self.desc is not None

    def broadcast_parameters(self):
        self.dist.broadcast(self.pflat, src=0)

    def _forward(self, inputs):
        if isinstance(inputs, (tuple, list)):
            return self.model(*inputs)
        return self.model(inputs)

    def _ham_diag(self):
        q = getattr(self.model, 'quantum_layer', None)
        return q.ham_diag if (q is not None and getattr(q, 'use_full_ham', False)) else None

    @property
    def accepts_out(self):
        """Whether train_step can leave [gradients | sse | sum y^2] in a caller's buffer instead of self.flat (the fused
        model-level path with the flat Adam: nothing reads the parameters' .grad views there)."""
        return self.desc is not None and isinstance(self.optimizer, FlatAdam)

    def loss_and_grad(self, *batch, global_batch=None, out=None):
        """Fill self.flat (or `out`, see accepts_out) with this shard's [gradients | sse | sum y^2] (no collective, no
        update)."""
        *inputs, y = batch
        gb = float(global_batch if global_batch is not None else y.shape[0] * self.world)
        if self.desc is not None:
            from . import _lib
            branch = inputs[0]
            trunk = inputs[1] if len(inputs) > 1 else None
            _lib.model_loss_grad(self.desc, branch, trunk, y.reshape(-1), self.pflat, 1.0 / gb,
                                 self.flat if out is None else out, ham_diag=self._ham_diag())
        else:
            self.flat.zero_()
            pred = self._forward(inputs)
            resid = pred - y.reshape(pred.shape)
            sse = (resid * resid).sum()
            (sse / gb).backward()
            with torch.no_grad():
                self.flat[self.numel] = sse
                self.flat[self.numel + 1] = (y * y).sum()
        return self.flat

    def train_step(self, *batch, global_batch=None, out=None):
        """batch = (branch, trunk, y) or (x, y): this rank's shard.  Returns the flat buffer (device): self.flat, or
        `out` -- a caller's [numel + 2] fp64 buffer, honoured where accepts_out says so (PTSolver hands in one row per
        step of an epoch, so that the two logging scalars of every step stay on the device without a copy)."""
        if out is not None and not self.accepts_out:
            out = None
        flat = self._last = self.flat if out is None else out
        if self.world == 1 and self.desc is not None and isinstance(self.optimizer, FlatAdam):
            # single device: loss, gradients and the Adam update in three launches (qhea_model_train_step)
            from . import _lib
            *inputs, y = batch
            gb = float(global_batch if global_batch is not None else y.shape[0])
            opt, g = self.optimizer, self.optimizer.param_groups[0]
            opt.t += 1
            _lib.model_train_step(self.desc, inputs[0], inputs[1] if len(inputs) > 1 else None, y.reshape(-1),
                                  self.pflat, 1.0 / gb, flat, opt.exp_avg, opt.exp_avg_sq, opt.t, g['lr'],
                                  g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'],
                                  ham_diag=self._ham_diag())
            return flat
        gb_all = float(global_batch if global_batch is not None else batch[-1].shape[0] * self.world)
        if self.peer_fused and gb_all >= self.world and batch[-1].shape[0] > 0:
            # the sum over the ranks inside this step's reduce kernel (qhea_model_dp_train_steps, one step).  Every rank
            # must take the same path: the choice rests on the GLOBAL batch (contiguous near-even shards, shard_slice,
            # are all non-empty when it is at least the world size)
            from . import _lib
            *inputs, y = batch
            gb = gb_all
            try:
                self._dp_steps(inputs, y, [0, y.shape[0]], [gb], flat.view(1, -1))
                return flat
            except _lib.Unsupported:
                self.peer_fused = False                 # reduce grid not resident at once: separate exchange kernel from now on
        self.loss_and_grad(*batch, global_batch=global_batch, out=out)
        if self.peer is not None:
            # sum over the ranks' peer-mapped buffers + Adam in ONE launch (csrc/hea_dp.hip)
            self.peer.allreduce_adam(flat, self.pflat, self.optimizer)
            return flat
        if self.world > 1:
            self.dist.all_reduce(flat)                 # SUM; one latency-bound message (19 KB at Q5)
        if out is not None:                            # FlatAdam reads its gradient from self.flat
            self.flat.copy_(out)
        self.optimizer.step()
        return flat

    def _dp_steps(self, inputs, y, bounds, global_batches, rows):
        from . import _lib
        opt, g, px = self.optimizer, self.optimizer.param_groups[0], self.peer
        n_steps = len(bounds) - 1
        _lib.model_dp_train_steps(self.desc, bounds, global_batches, inputs[0], inputs[1] if len(inputs) > 1 else None,
                                  y.reshape(-1), self.pflat, rows, opt.exp_avg, opt.exp_avg_sq, opt.t + 1, g['lr'],
                                  g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], px.rank, px.world, px.bufs,
                                  px.n, px.seq + 1, timeout_ms=px.timeout_ms, ham_diag=self._ham_diag())
        opt.t += n_steps
        px.seq += n_steps

    def calibrate_exchange(self, *batch, global_batch=None, steps=30):
        """
        Several ranks, peer buffers mapped: time this very step through both forms of the peer exchange -- inside the
        reduce kernel (two launches per step) and as a separate one-workgroup kernel (four) -- and keep the faster one.
        Which one wins depends on where the ranks run: the reduce kernel's waiting blocks hold compute units, which costs
        nothing when every rank has a GPU of its own and a lot when ranks share one (a rehearsal).  Collective; parameters,
        Adam state and step count are restored afterwards, so a run is the same with or without the calibration.  Returns
        (ms per step fused, ms per step separate) or None when there is nothing to choose.
        """
        if not self.peer_fused or self.world < 2:
            return None
        import time
        gb = float(global_batch if global_batch is not None else batch[-1].shape[0] * self.world)
        if gb < self.world or batch[-1].shape[0] == 0:
            return None
        from . import _lib
        opt = self.optimizer
        keep = (self.pflat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t)
        dev = self.pflat.device

        def restore():
            self.pflat.copy_(keep[0]); opt.exp_avg.copy_(keep[1]); opt.exp_avg_sq.copy_(keep[2]); opt.t = keep[3]

        def timed(fused):
            self.peer_fused = fused
            for _ in range(2):                                   # the first pass warms up
                self.dist.barrier(); torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    self.train_step(*batch, global_batch=global_batch)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            bad = 0.0
            try:
                self.peer.check_status()
            except _lib.QheaError:
                bad = 1.0
            t = torch.tensor([dt, bad], dtype=torch.float64, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return 1e3 * float(t[0].item()) / steps, bool(t[1].item() > 0.5)
        # The trial inside the reduce kernel runs with a short bound: where ranks SHARE a GPU its waiting blocks can keep a
        # peer's kernels off the compute units for good (seen with three ranks on one device).  A failed trial has poisoned
        # the buffers (a timeout is fatal by design): they are re-created and the separate kernel is kept.
        long_timeout, self.peer.timeout_ms = self.peer.timeout_ms, 250.0
        t_fused, failed = timed(True)
        self.peer.timeout_ms = long_timeout
        if failed:
            restore()
            torch.cuda.synchronize(dev)
            self.peer.close()
            self.peer, why = PeerExchange.create(self.dist, self.dist.get_rank(), self.world, self.numel + 2, dev)
            self.peer_fused = False
            self.fused_ok = False
            self.dp_exchange_reason = (why + "; calibrated on this step: the exchange inside the reduce kernel did not complete "
                                       "within 250 ms (ranks sharing a GPU?) -> separate kernel, buffers re-created")
            return None
        p_fused = self.pflat.clone()
        restore()
        t_sep, failed = timed(False)
        p_sep = self.pflat.clone()
        restore()
        if failed:
            raise _lib.QheaError("data-parallel exchange failed during calibration")
        # Both trials ran the same steps from the same state, and both add the ranks' numbers in rank order: the parameters
        # they end on must be BITWISE the same, on every rank.  Checked here because this is the first place the exchange
        # meets the run's real placement of ranks (separate GPUs, the link between them): if the two forms disagree or
        # the replicas differ, neither is trusted and every rank keeps the collective library's all-reduce.
        if not self._agree_bitwise(p_fused, p_sep):
            torch.cuda.synchronize(dev)
            self.peer.close()
            self.peer = None
            self.peer_fused = False
            self.fused_ok = False
            self.dp_exchange_reason += ("; calibrated on this step: the two forms of the peer exchange did NOT end on bitwise "
                                        "identical parameters on every rank -> all_reduce (collective library) + Adam launch")
            return None
        # (the same numbers on every rank: one decision.  Within 3 % the form inside the reduce kernel is kept: it lets an epoch's
        # inner loop be ONE host call, which this per-step timing does not credit it with)
        self.peer_fused = t_fused <= 1.03 * t_sep
        self.dp_exchange_reason += (f"; calibrated on this step: {t_fused:.4f} ms inside the reduce kernel, {t_sep:.4f} ms "
                                    f"as a separate kernel, parameters bitwise identical through both and on every rank -> "
                                    + ("inside the reduce kernel" if self.peer_fused else "separate kernel"))
        return t_fused, t_sep

    def _agree_bitwise(self, a, b):
        """Collective: True on every rank iff a == b bitwise on every rank AND every rank holds rank 0's a and b."""
        ref = torch.stack([a, b])
        self.dist.broadcast(ref, src=0)
        wrong = torch.tensor([0.0 if (torch.equal(a, b) and torch.equal(ref[0], a) and torch.equal(ref[1], b)) else 1.0],
                             dtype=torch.float64, device=a.device)
        self.dist.all_reduce(wrong, op=self.dist.ReduceOp.MAX)
        return wrong.item() < 0.5

    @property
    def epoch_call(self):
        """Whether train_steps can issue a run of steps from one host call: the fused single-device path, or the
        data-parallel path with the exchange inside the reduce kernel."""
        return self.accepts_out and (self.world == 1 or self.peer_fused)

    def train_steps(self, inputs, y, bounds, global_batches, rows):
        """
        The steps of a whole epoch from ONE host call -- step i on rows bounds[i]:bounds[i+1] of the contiguous `inputs` /
        `y` (this rank's shards), its [gradients | sse | sum y^2] (summed over the ranks) left in rows[i].  One device:
        qhea_model_train_steps; several: qhea_model_dp_train_steps, the sum over the ranks inside each step's reduce kernel
        (raises _lib.Unsupported, nothing launched, if a shard is empty).  Bitwise the result of calling
        train_step(..., out=rows[i]) in a loop, without the interpreter between launches.
        """
        assert self.epoch_call
        from . import _lib
        n_steps = len(bounds) - 1
        if self.world > 1:
            self._dp_steps(inputs, y, bounds, global_batches, rows)
            self._last = rows[n_steps - 1]
            return rows
        opt, g = self.optimizer, self.optimizer.param_groups[0]
        _lib.model_train_steps(self.desc, bounds, global_batches, inputs[0], inputs[1] if len(inputs) > 1 else None,
                               y.reshape(-1), self.pflat, rows, opt.exp_avg, opt.exp_avg_sq, opt.t + 1, g['lr'],
                               g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], ham_diag=self._ham_diag())
        opt.t += n_steps
        self._last = rows[n_steps - 1]
        return rows

    def loss_scalars(self):
        """(sse, sum y^2) of the last global batch -- forces a device sync; call per epoch, not per step."""
        v = self._last[self.numel:].tolist()        # the buffer the last step left its results in (self.flat or `out`)
        return v[0], v[1]

    def check_status(self):
        """Raise if a backward kernel reported a hand-off overrun since the last check (qhea_check_status; that call's
        gradients were NaN-poisoned and its Adam update skipped).  Synchronises: call where the host waits anyway."""
        if not self.pflat.is_cuda:
            return
        from . import _lib
        err = None
        try:
            _lib.check_status(self.pflat.device)
            if self.peer is not None:
                self.peer.check_status()
        except _lib.QheaError as e:
            err = e
        if self.world > 1:
            # the ranks agree on the outcome BEFORE anybody acts on it (rank 0 saving a checkpoint): a failure
            # on any rank raises on every rank
            t = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=self.pflat.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            if err is None and t.item() > 0.5:
                err = _lib.QheaError("a peer rank reported a failed step (pipeline overrun or exchange timeout): "
                                     "the replicas are no longer in step")
        if err is not None:
            raise err


class PTSolver:
    """
    Mirror of the reference PTSolver (solvers/solver_pt.py:21-329).  ``config`` uses the reference's
    keys (model_type, operator, num_qubits, net_size, scale_coeff, if_trainable_freq, ham_bound,
    ham_diag, learning_rate, batch_size, num_epochs, optimizer, lr_scheduler, if_save, prefix, seed).
    ``data_dict`` has the DataManager output keys (data_utils/data_manager.py:74-106):
    train_branch_input/train_trunk_input/train_output/test_* (QuanONet) or train_input/... (HEAQNN).
    """

    def __init__(self, config, data_dict, device=None, dist=None, rank=0, world_size=1, log=print, model=None):
        self.config = config
        self.data_dict = data_dict
        self.model_type = config['model_type']
        self.dist, self.rank, self.world = dist, rank, world_size
        self.log = log if rank == 0 else (lambda *a, **k: None)
        self.device = device if device is not None else torch.device('cuda')
        # `model` is a seam for the host-logic tests (a module whose quantum layer is the oracle test double runs
        # this very loop on the CPU); the solver never builds anything but the HIP modules itself
        if self.device.type != 'cuda' and model is None:
            raise RuntimeError("PTSolver runs on a HIP device only (no CPU fallback)")
        self.out_dir = os.path.join(config.get('prefix') or 'outputs', config.get('operator', 'Op'),
                                    config.get('run_id', 'run'))
        self.model = (model if model is not None else self._create_model()).to(self.device)
        self.trainer = DataParallelTrainer(self.model, lr=config['learning_rate'], world_size=world_size,
                                           dist=dist, optimizer=config.get('optimizer', 'adam'),
                                           optimizer_kwargs=config.get('optimizer_kwargs', {}),
                                           peer_exchange=str(config.get('dp_exchange', 'peer')).lower() == 'peer',
                                           log=self.log)
        self.lr_scheduler = self._build_scheduler()
        self.best_loss = float('inf')
        self.best_model_path = None
        self._setup_data()

    # ---- model / data -------------------------------------------------------------------------
    def _create_model(self):
        from .models import QuanONetPT, HEAQNNPT
        c = self.config
        ham_bound = tuple(c.get('ham_bound', [-5, 5]))
        net_size = tuple(c.get('net_size', [20, 2, 10, 2]))
        if_tf = str(c.get('if_trainable_freq', 'true')).lower() == 'true'
        scale = float(c.get('scale_coeff', 0.01))
        n = int(c['num_qubits'])
        # --ham_diag strictly overrides --ham_pauli (utils/common.py:84)
        ham_pauli = 'Z' if c.get('ham_diag') is not None else (c.get('ham_pauli') or 'Z')
        if self.model_type == 'QuanONet':
            return QuanONetPT(n, self.data_dict['train_branch_input'].shape[1],
                              self.data_dict['train_trunk_input'].shape[1], net_size, scale_coeff=scale,
                              if_trainable_freq=if_tf, ham_bound=ham_bound, ham_diag=c.get('ham_diag'),
                              ham_pauli=ham_pauli)
        if self.model_type == 'HEAQNN':
            return HEAQNNPT(n, self.data_dict['train_input'].shape[1], net_size, scale_coeff=scale,
                            if_trainable_freq=if_tf, ham_bound=ham_bound, ham_diag=c.get('ham_diag'),
                            ham_pauli=ham_pauli)
        raise ValueError(f"PTSolver does not support model_type='{self.model_type}'")

    def _dev(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.float64).to(self.device)

    def _setup_data(self):
        d = self.data_dict
        if self.model_type == 'HEAQNN':
            self.train_input = (self._dev(d['train_input']),)
            self.test_input = (self._dev(d['test_input']),)
        else:
            self.train_input = (self._dev(d['train_branch_input']), self._dev(d['train_trunk_input']))
            self.test_input = (self._dev(d['test_branch_input']), self._dev(d['test_trunk_input']))
        self.train_output = self._dev(d['train_output']).reshape(len(d['train_output']), -1)
        self.test_output = np.asarray(d['test_output'], dtype=np.float64).reshape(len(d['test_output']), -1)

    def _build_scheduler(self):
        name = str(self.config.get('lr_scheduler', 'none')).lower()
        kw = self.config.get('lr_scheduler_kwargs', {})
        opt = self.trainer.optimizer
        if name == 'cosine':
            return torch.optim.lr_scheduler.CosineAnnealingLR(
                opt, T_max=kw.get('T_max', self.config.get('num_epochs', 1000)), eta_min=kw.get('eta_min', 0.0))
        if name == 'step':
            return torch.optim.lr_scheduler.StepLR(opt, step_size=kw.get('step_size', 100), gamma=kw.get('gamma', 0.5))
        if name == 'exponential':
            return torch.optim.lr_scheduler.ExponentialLR(opt, gamma=kw.get('gamma', 0.99))
        return None

    # ---- training (solver_pt.py:191-277) ------------------------------------------------------------
    def _save(self, path, flat=None):
        """state_dict as .pt + .npz.  `flat`: a (host) snapshot of the trainer's flat parameter vector to save INSTEAD of the
        live parameters -- the epoch loop has already queued the next epoch when it decides about the checkpoint (train())."""
        sd = self.model.state_dict()
        if flat is not None:
            sd = dict(sd)
            own = {id(p): i for i, p in enumerate(self.trainer.params)}
            offs = np.cumsum([0] + [p.numel() for p in self.trainer.params])
            for k, p in self.model.named_parameters():
                i = own.get(id(p))
                if i is not None and k in sd:
                    sd[k] = flat[offs[i]:offs[i + 1]].view(p.shape).to(sd[k].dtype)
        torch.save(sd, path)
        np.savez(path.replace('.pt', '.npz'), **{k: v.detach().cpu().numpy() for k, v in sd.items()})

    def _stage_epoch(self, n, bs, nb):
        """
        One epoch's batch order and this rank's rows of it, gathered ONCE (three gathers per epoch instead of three per
        step): step i of the reference takes idx[i*bs:(i+1)*bs] (solver_pt.py:226-228), this rank its contiguous shard
        of that.  Returns (order, row bounds per step, gathered inputs, gathered targets).
        """
        idx_dev = self._epoch_permutation(n)
        bounds, pieces = [0], []
        for i in range(nb):
            gb_i = min(bs, n - i * bs)
            lo, hi = shard_slice(gb_i, self.rank, self.world)
            pieces.append((i * bs + lo, i * bs + hi))
            bounds.append(bounds[-1] + hi - lo)
        if self.world == 1:
            sel_all = idx_dev
        else:
            sel_all = torch.cat([idx_dev[a:b] for a, b in pieces]) if bounds[-1] else idx_dev[:0]
        return idx_dev, bounds, [t[sel_all] for t in self.train_input], self.train_output[sel_all]

    def _epoch_permutation(self, n):
        """
        ``np.random.permutation(n)`` from NumPy's global generator, as the reference draws it (solver_pt.py:220), so a
        launcher that seeds like the reference's (``set_random_seed``) reproduces its batch order.  With several
        ranks ONLY rank 0 draws and the order is broadcast: every rank then slices the same global batch whatever
        its own generator state is (separately launched ranks are not seeded alike by anything).
        """
        if self.world > 1:
            idx = torch.empty(n, dtype=torch.int64, device=self.device)
            if self.rank == 0:
                idx.copy_(torch.as_tensor(np.random.permutation(n)))
            self.dist.broadcast(idx, src=0)
            return idx
        return torch.as_tensor(np.random.permutation(n), device=self.device)

    def is_completed(self):
        """Whether this run's directory already holds a metric.json -- the reference's "experiment already completed" test
        (utils/logger.py:182-185), on which its train() ends the process (solvers/solver_pt.py:192-194)."""
        return os.path.exists(os.path.join(self.out_dir, 'metric.json'))

    def train(self):
        if self.config.get('skip_completed', False) and self.is_completed():
            # the reference calls sys.exit(0) here; a library returns instead (the launcher decides what to do with None)
            self.log("Experiment already completed (metric.json exists): training skipped.")
            return None
        n = self.train_output.shape[0]
        bs = min(int(self.config.get('batch_size', 100)), n)
        epochs = int(self.config['num_epochs'])
        nb = max(1, int(np.ceil(n / bs)))
        history = {'loss_train': [], 'loss_test': []}
        trace = bool(self.config.get('trace_steps', False))     # parity tests: per-step MSE and the epoch orders
        if trace:
            history['loss_steps'], history['indices'] = [], []
        os.makedirs(self.out_dir, exist_ok=True)
        self.best_model_path = os.path.join(self.out_dir, 'best_model.pt')
        staged = self._stage_epoch(n, bs, nb) if epochs > 0 else None
        if staged is not None and self.world > 1 and self.config.get('dp_calibrate', True):
            # both forms of the peer exchange timed on the first batch, the faster one kept (state restored afterwards)
            _, b0, in0, out0 = staged
            r = self.trainer.calibrate_exchange(*[t[b0[0]:b0[1]] for t in in0], out0[b0[0]:b0[1]], global_batch=min(bs, n))
            if r is not None:
                self.log("data-parallel exchange: " + self.trainer.dp_exchange_reason)
        want_save = self.config.get('if_save', True) and self.rank == 0
        nm = self.trainer.numel

        def issue(staged):
            """queue one epoch's steps; returns the device rows of its [sse | sum y^2] and its order"""
            self.model.train()
            idx_dev, bounds, ep_inputs, ep_output = staged
            # (sse, sum y^2) of every global batch stay on the device until the epoch ends.  On the fused path every step
            # leaves its [gradients | sse | sum y^2] in a row of its own (no copy kernel per step; nb x 19 KB at Q5)
            rows = (torch.zeros(nb, nm + 2, dtype=torch.float64, device=self.device) if self.trainer.accepts_out
                    else None)
            tails = rows[:, nm:] if rows is not None else torch.zeros(nb, 2, dtype=torch.float64, device=self.device)
            # the whole epoch's inner loop from one host call (no interpreter between the launches; config['epoch_call'] =
            # False keeps one host call per step).  Several ranks: every rank must take the same path, so the choice rests on
            # what all of them know -- no global batch smaller than the world size (an empty shard somewhere)
            one_call = (rows is not None and self.trainer.epoch_call and self.config.get('epoch_call', True) and
                        min(bs, n - (nb - 1) * bs) >= self.world)
            if one_call:
                self.trainer.train_steps(ep_inputs, ep_output, bounds, [min(bs, n - i * bs) for i in range(nb)], rows)
            else:
                for i in range(nb):
                    a, b = bounds[i], bounds[i + 1]
                    gb = min(bs, n - i * bs)
                    flat = self.trainer.train_step(*[t[a:b] for t in ep_inputs], ep_output[a:b], global_batch=gb,
                                                   out=None if rows is None else rows[i])
                    if rows is None:
                        tails[i].copy_(flat[nm:])
            return tails, idx_dev

        cur = issue(staged) if epochs > 0 else None
        for epoch in range(epochs):
            tails, idx_dev = cur
            # the next epoch's order and rows are drawn and gathered BEFORE the host waits for this epoch: the draw (a
            # millisecond of host time at 10^5 rows) overlaps the steps still queued on the device
            staged = self._stage_epoch(n, bs, nb) if epoch + 1 < epochs else None
            tl = tails.tolist()                                 # one host sync per epoch
            self.trainer.check_status()                         # a kernel-side pipeline failure ends the run here
            # What the bookkeeping below needs from the END of this epoch is on the host now -- the losses, and (19 KB) the
            # parameters for the best-by-train-loss checkpoint -- so the NEXT epoch is queued first and the host does its
            # sums, its checkpoint and its log line while the device already works (the schedulers offered are functions
            # of the epoch count alone).
            snap = self.trainer.pflat.detach().to('cpu', copy=True) if want_save else None
            idx_host = idx_dev.cpu().numpy() if trace else None
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            cur = issue(staged) if staged is not None else None
            s = [0.0, 0.0, 0.0]                                 # sum of batch MSE, sse, sum y^2 -- added in step order
            step_mse = []
            for i in range(nb):
                gb = min(bs, n - i * bs)
                step_mse.append(tl[i][0] / gb)
                s[0] += tl[i][0] / gb
                s[1] += tl[i][0]
                s[2] += tl[i][1]
            if trace:
                history['loss_steps'].extend(step_mse)
                history['indices'].append(idx_host)
            avg_loss = s[0] / nb
            avg_rel = np.sqrt(s[1]) / (np.sqrt(s[2]) + 1e-8)
            history['loss_train'].append(avg_loss)
            if avg_loss < self.best_loss:
                self.best_loss = avg_loss
                if want_save:
                    self._save(self.best_model_path, flat=snap)
            if epoch % 10 == 0:
                self.log(f"Epoch {epoch} | MSE: {avg_loss:.6e} | Rel_L2: {avg_rel:.4%}")
        if self.config.get('if_save', True) and self.rank == 0:
            self._save(os.path.join(self.out_dir, 'final.pt'))                 # logger.py:174-177 + solver_pt.py:266-272: final.pt / final.npz
        return history

    # ---- evaluation (solver_pt.py:279-329, utils/metrics.py:6-29) -----------------------------------------
    def predict(self, inputs, batch_size=None):
        bs = int(batch_size or self.config.get('batch_size', 100))
        n = inputs[0].shape[0]
        outs = []
        self.model.eval()
        tr = self.trainer
        with torch.no_grad():
            if tr.desc is not None and n > 0:
                # all chunks from one host call: the layer records are prepared once for the whole set (the parameters
                # do not change during an evaluation)
                from . import _lib
                ins = [t.contiguous() for t in inputs]
                o = _lib.model_forward_chunks(tr.desc, ins[0], ins[1] if len(ins) > 1 else None, tr.pflat, bs,
                                              ham_diag=tr._ham_diag())
                return o.unsqueeze(-1)
            for s in range(0, n, bs):
                chunk = [t[s:s + bs] for t in inputs]
                outs.append(self.model(*chunk))
        if not outs:                                            # an empty shard (fewer test rows than ranks)
            return torch.empty((0, 1), dtype=torch.float64, device=inputs[0].device)
        return torch.cat(outs, dim=0)

    def evaluate(self, history=None):
        # rank 0 alone wrote best_model.pt (train()) and alone reads it back; the other ranks receive the weights by
        # broadcast (parameters are views into the trainer's flat vector), so no rank can read a half-written file
        # or, on node-local storage, silently evaluate different weights
        if self.rank == 0 and self.best_model_path and os.path.exists(self.best_model_path):
            sd = torch.load(self.best_model_path, map_location=self.device, weights_only=True)
            self.model.load_state_dict(sd)
        if self.world > 1:
            self.trainer.broadcast_parameters()
        # every rank evaluates its contiguous slice of the test set (10-100x a training batch: SURVEY.md 8(f)-2) in
        # chunks of 16384 rows (Q5: 87 M evaluations/s against 27 M at 1024 rows per call, profiles/r02_batch_sweep.txt);
        # the metrics are reduced from four sums and one maximum
        n_test = self.test_input[0].shape[0]
        lo, hi = shard_slice(n_test, self.rank, self.world)
        y_pred = self.predict([t[lo:hi] for t in self.test_input], batch_size=self.config.get('eval_batch_size', 16384))
        y_true = torch.as_tensor(np.asarray(self.test_output)[lo:hi], dtype=torch.float64).to(y_pred.device)
        metrics = regression_metrics(y_pred, y_true, self.dist, self.world)
        if self.rank == 0:
            os.makedirs(self.out_dir, exist_ok=True)
            with open(os.path.join(self.out_dir, 'metric.json'), 'w') as f:
                json.dump({'metrics': metrics, 'history': history}, f, default=lambda o: o.tolist())
        self.log(f"Test Relative L2 Error: {metrics['rel_l2']:.6f}")
        return metrics
