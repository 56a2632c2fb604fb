#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X-native QuanONet training path.

Workload (BASELINE.json configs[1]): Advection-shaped QuanONet, Q=5, Net40-2-20-2, b_in=100, t_in=2,
batch 1024 PER GPU (weak scaling), fp64, synthetic data.  One "step" is one
full training step of the hot path: frequency layers -> HIP circuit forward -> MSE -> in-kernel adjoint
backward -> (RCCL all-reduce of the flat gradient when N>1) -> Adam.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run,
one rank per GPU.  Rank 0 prints ONE JSON line.  `value` = train samples/s over all ranks; the same line
also carries forward-only circuit-evals/s, the roofline object of the dominant kernel and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_QUBITS, NET, B_IN, T_IN, BATCH = 5, (40, 2, 20, 2), 100, 2, 1024

def synth(rank, n_samples):
    rng = np.random.default_rng(1000 + rank)
    branch = rng.normal(size=(n_samples, B_IN))
    trunk = rng.uniform(size=(n_samples, T_IN))
    y = rng.normal(scale=0.5, size=(n_samples, 1))
    return branch, trunk, y


def cpu_baseline(seconds_target=12.0):
    """Oracle (C + OpenMP over the batch) timed on this host: training = forward + adjoint backward."""
    from oracle import c_oracle as C, hea_oracle as O
    cfgs = O.block_configs_quanonet(N_QUBITS, NET)
    E, blk = O.circuit_sizes(N_QUBITS, cfgs)
    rng = np.random.default_rng(0)
    w = rng.uniform(-np.pi, np.pi, (blk, 3, N_QUBITS))
    off, co = O.ham_params(N_QUBITS)
    cores = C.threads()
    nb = 256 * max(1, cores // 4)
    x = rng.uniform(-np.pi, np.pi, (nb, E))
    g = rng.normal(size=nb)
    C.hea_backward(N_QUBITS, cfgs, x[:64], w, g[:64], off, co)            # warm
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds_target:
        C.hea_backward(N_QUBITS, cfgs, x, w, g, off, co)
        done += nb
    dt = time.perf_counter() - t0
    C.hea_forward(N_QUBITS, cfgs, x, w, off, co)                            # warm
    t1 = time.perf_counter()
    fdone = 0
    while time.perf_counter() - t1 < seconds_target / 4:
        C.hea_forward(N_QUBITS, cfgs, x, w, off, co)
        fdone += nb
    dtf = time.perf_counter() - t1
    return {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{done} train samples (forward + adjoint backward of the Q5 Net40-2-20-2 circuit, "
                      f"oracle/hea_oracle.c, OpenMP over the batch), {dt:.1f} s; forward-only: {fdone} evals, {dtf:.1f} s",
            "forward_evals_per_s": fdone / dtf}


def spawn_ranks(args):
    """`python bench.py --gpus N` invoked plainly (no launcher): start the N ranks as CHILDREN of this process --
    which has not touched the GPU -- through torch.distributed.run, relay rank 0's JSON line, exit with their code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            print(ln)
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='collective backend; "gloo" + --same-device rehearses N>1 on one GPU')
    ap.add_argument('--same-device', action='store_true', help='rehearsal only: every rank uses cuda:0')
    ap.add_argument('--batch', type=int, default=BATCH, help='samples per GPU and step (the headline is 1024)')
    ap.add_argument('--backward-variant', default='auto',
                    help='measurement only: force an n <= 5 kernel variant (quanonet_amd._lib.BWD_VARIANTS); the headline is "auto"')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                     # never returns

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    from quanonet_amd.models import QuanONetPT
    from quanonet_amd import _lib
    from quanonet_amd.solver import DataParallelTrainer
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import roofline_from_profiles as RF

    batch = args.batch
    _lib.set_backward_variant(args.backward_variant)          # before any workspace is sized
    torch.manual_seed(0)
    model = QuanONetPT(N_QUBITS, B_IN, T_IN, NET, scale_coeff=0.1, if_trainable_freq=True).to(dev)
    trainer = DataParallelTrainer(model, lr=1e-4, world_size=world, dist=dist)

    n_batches = 8                                             # device-resident synthetic set, cycled
    branch, trunk, y = synth(rank, n_batches * batch)
    branch = torch.tensor(branch, device=dev); trunk = torch.tensor(trunk, device=dev); y = torch.tensor(y, device=dev)

    def step(i):
        s = (i % n_batches) * batch
        return trainer.train_step(branch[s:s + batch], trunk[s:s + batch], y[s:s + batch],
                                  global_batch=batch * world)

    # One device: the steps are issued as the product's epoch loop issues them (PTSolver.train -> qhea_model_train_steps):
    # runs of consecutive steps from one host call, inside which a step's reduce kernel writes the next step's layer
    # records instead of a prep launch.  Same batches in the same order as the per-step loop (step i: batch i mod 8).
    epoch_call = world == 1 and trainer.accepts_out and not os.environ.get('QHEA_BENCH_PER_STEP')
    rows = torch.zeros(n_batches, trainer.numel + 2, dtype=torch.float64, device=dev) if epoch_call else None
    bounds = [j * batch for j in range(n_batches + 1)]

    def run_steps(first, k):
        done = 0
        while done < k:
            start = (first + done) % n_batches
            m = min(n_batches - start, k - done)
            trainer.train_steps([branch, trunk], y, bounds[start:start + m + 1], [batch * world] * m, rows[start:start + m])
            done += m

    if epoch_call:
        step.many = run_steps

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_window(fn, k, first):
        """EXACTLY k calls bracketed by barrier + synchronize on both sides; max over ranks."""
        fence()
        t0 = time.perf_counter()
        many = getattr(fn, 'many', None)                 # a function that issues k steps itself (first, k)
        if many is not None:
            many(first, k)
        else:
            for i in range(k):
                fn(first + i)
        fence()
        return rank_max(time.perf_counter() - t0)

    def measure(fn, k, budget_s=0.4):
        """Median over repeated windows of exactly k steps each: one 20-step window is 3 ms, short enough for the
        clock ramp after the idle fence to move it by several percent (BASELINE.md section 3 asks for the median)."""
        w0 = timed_window(fn, k, 0)
        n_win = int(min(50, max(5, np.ceil(budget_s / max(w0, 1e-6)))))
        wins = [w0] + [timed_window(fn, k, (j + 1) * k) for j in range(n_win - 1)]
        return float(np.median(wins)), wins

    if epoch_call:
        run_steps(0, args.warmup)
    else:
        for i in range(args.warmup):
            step(i)
    elapsed, windows = measure(step, args.steps)
    samples_per_s = batch * world * args.steps / elapsed
    trainer.check_status()

    # forward-only circuit evaluations/s (evaluation path): the resident set in chunks of one batch, issued as
    # PTSolver.predict issues an evaluation (qhea_model_forward_chunks: the layer records are prepared once per call, the
    # parameters being fixed); a "step" is one batch's forward launch
    fwd_out = torch.empty(n_batches * batch, dtype=torch.float64, device=dev)

    def fwd_only(_i=0):
        return _lib.model_forward(trainer.desc, branch[:batch], trunk[:batch], trainer.pflat)

    def fwd_steps(first, k):
        done = 0
        while done < k:
            m = min(n_batches, k - done)
            _lib.model_forward_chunks(trainer.desc, branch[:m * batch], trunk[:m * batch], trainer.pflat, batch,
                                      out=fwd_out[:m * batch])
            done += m
    for i in range(5):
        fwd_only()
    fwd_steps.many = fwd_steps
    fwd_elapsed, _ = measure(fwd_steps if not os.environ.get('QHEA_BENCH_PER_STEP') else fwd_only, args.steps, budget_s=0.2)
    evals_per_s = batch * world * args.steps / fwd_elapsed

    # dominant kernel: the fused circuit kernel (forward sweep + MSE residual + adjoint reverse sweep) launched by
    # qhea_model_loss_grad.  Timed ALONE with HIP events recorded by the library immediately around that launch on
    # the launch stream (qhea_profile_next_circuit_kernel); profiles/ holds the rocprofv3 summary of this command.
    roof = None
    if rank == 0:
        yb = y[:batch].reshape(-1).contiguous()
        reps = max(20, min(args.steps, 200))

        def kernel_ms(fn):
            for _ in range(5):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record(); b.record()                      # materialise the hipEvent handles
            torch.cuda.synchronize()
            for a, b in ev:
                _lib.profile_next_circuit_kernel(a, b)
                fn()
            torch.cuda.synchronize()
            return float(np.median([a.elapsed_time(b) for a, b in ev]))

        lg_ms = kernel_ms(lambda: trainer.loss_and_grad(branch[:batch], trunk[:batch], yb, global_batch=batch * world))
        fwd_ms = kernel_ms(fwd_only)
        pmc, pmc_src = RF.load_pmc()
        roof = RF.build_roofline(lg_ms, fwd_ms, pmc if batch == BATCH else None,
                                 f"HIP events around the kernel's launch on its stream, median of {reps} launches, this run",
                                 batch=batch)
        if roof.get('traffic') is not None:
            roof['traffic_source'] = pmc_src

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        line = {
            "metric": "train samples/sec (circuit-evals/sec alongside), QuanONet Q=5 Net40-2-20-2 Advection",
            "value": samples_per_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Advection QuanONet Q=5 Net40-2-20-2 b_in=100 t_in=2, batch {batch} per GPU, "
                                   "fp64, Adam lr=1e-4, trainable frequency",
                       "global_batch": batch * world, "parallelism": f"dp{world}",
                       "dp_exchange": (None if world == 1 else
                                       "peer-mapped buffers, sum + Adam in one kernel (csrc/hea_dp.hip)"
                                       if trainer.peer is not None else "all_reduce (" + args.backend + ") + Adam launch")},
            "timing": {"windows": len(windows), "steps_per_window": args.steps,
                       "ms_per_step_median": 1e3 * elapsed / args.steps,
                       "ms_per_step_first_window": 1e3 * windows[0] / args.steps,
                       "ms_per_step_min": 1e3 * min(windows) / args.steps,
                       "ms_per_step_max": 1e3 * max(windows) / args.steps,
                       "issue": ("runs of up to 8 consecutive steps per host call (qhea_model_train_steps, the epoch loop of "
                                 "PTSolver.train: a step's reduce kernel writes the next step's layer records)"
                                 if epoch_call else "one host call per step (prep, circuit, reduce launches each)"),
                       "note": "value = median over windows of EXACTLY `steps` training steps each, every window "
                               "bracketed by barrier + device synchronize, max over ranks per window"},
            "circuit_evals_per_s": evals_per_s,
            "circuit_evals_issue": ("one host call per forward launch (record preparation every call)"
                                    if os.environ.get('QHEA_BENCH_PER_STEP') else
                                    "the resident set in chunks of one batch per host call (qhea_model_forward_chunks, what "
                                    "PTSolver.predict runs: one record preparation per call, one forward launch per batch)"),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if trainer.peer is not None:
        trainer.peer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
