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
import quanonet_amd._lib          # noqa: E402,F401  (first: it sets the IPC mode default before anything initialises the GPU)

N_QUBITS, NET, B_IN, T_IN, BATCH = 5, (40, 2, 20, 2), 100, 2, 1024

def synth(rank, n_samples):
    rng = np.random.default_rng(1000 + rank)
    branch = rng.normal(size=(n_samples, B_IN))
    trunk = rng.uniform(size=(n_samples, T_IN))
    y = rng.normal(scale=0.5, size=(n_samples, 1))
    return branch, trunk, y


def cpu_baseline(seconds_target=12.0):
    """Oracle (C + OpenMP over the batch) timed on this host: training = forward + adjoint backward.  Beside the all-core
    figure (`value`): the same port on ONE thread, and the reference's own SHAPE of computation -- one batched contraction
    per gate, autograd keeping every intermediate state (core/quantum_circuits_tq.py:65-127 on TorchQuantum) -- restated in
    plain PyTorch (scripts/tq_shaped_baseline.py) on the same host cores."""
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
    # one thread
    C.set_threads(1)
    try:
        n1 = 512
        t2 = time.perf_counter()
        d1 = 0
        while time.perf_counter() - t2 < seconds_target / 4:
            C.hea_backward(N_QUBITS, cfgs, x[:n1], w, g[:n1], off, co)
            d1 += n1
        dt1 = time.perf_counter() - t2
    finally:
        C.set_threads(cores)
    out = {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{done} train samples (forward + adjoint backward of the Q5 Net40-2-20-2 circuit, "
                     f"oracle/hea_oracle.c, OpenMP over the batch), {dt:.1f} s; forward-only: {fdone} evals, {dtf:.1f} s",
           "forward_evals_per_s": fdone / dtf,
           "one_thread": {"value": d1 / dt1, "unit": "samples/s", "cores": 1, "kind": "port",
                          "sample": f"{d1} train samples, {dt1:.1f} s"}}
    try:
        out["reference_shaped"] = tq_shaped_cpu()
    except Exception as e:                                   # a baseline leg must not take the benchmark down
        out["reference_shaped"] = {"error": repr(e)}
    return out


def tq_shaped_cpu(batch=128):
    """ONE training step (and one forward pass) of the gate-by-gate PyTorch restatement of the reference's TorchQuantum
    layer on the host CPU, complex128, all cores, on a batch of 128 (a step at the headline batch takes ~40 s here)."""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import tq_shaped_baseline as TQ
    torch.manual_seed(0)
    model = TQ.TQShapedQuanONet(torch.complex128)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    rng = np.random.default_rng(1000)
    branch = torch.tensor(rng.normal(size=(batch, B_IN))); trunk = torch.tensor(rng.uniform(size=(batch, T_IN)))
    y = torch.tensor(rng.normal(scale=0.5, size=(batch, 1)))
    with torch.no_grad():
        t0 = time.perf_counter()
        model(branch, trunk)
        tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    opt.zero_grad()
    torch.nn.functional.mse_loss(model(branch, trunk), y).backward()
    opt.step()
    tt = time.perf_counter() - t0
    return {"value": batch / tt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "what": "TorchQuantum-shaped gate-by-gate PyTorch (scripts/tq_shaped_baseline.py), complex128, host CPU",
            "sample": f"1 training step on {batch} samples, {tt:.1f} s; 1 forward pass, {tf:.1f} s",
            "forward_evals_per_s": batch / tf}


FP64_VECTOR_PEAK_TFLOPS = 78.6


def secondary_configs(dev, measure_many):
    """The other BASELINE.json configurations (and the reference's own training batch) on this one GPU, each timed the way
    the headline is: runs of consecutive training steps from one host call where the product issues them so, the median
    over windows of k steps.  `frac_fp64_vector_step` prices the WHOLE step (all its launches) against the fp64 vector peak
    with SURVEY.md 8(d)'s algorithmic count 22 * 2^n * R per training sample -- the headline's `roofline.frac` prices the
    circuit kernel alone."""
    from quanonet_amd.models import QuanONetPT, HEAQNNPT
    from quanonet_amd import _lib
    from quanonet_amd.solver import DataParallelTrainer, PTSolver, set_random_seed
    out = {}
    cases = [
        ('cfg1_antideriv_q2_b32', 'Antideriv QuanONet Q=2 Net5-1-5-1 b_in=10 t_in=1, batch 32', 'quanonet', 2, (5, 1, 5, 1), 10, 1, 32, 100),
        ('cfg3_shard_darcy_q5_b512', 'Darcy QuanONet Q=5 Net40-2-20-2, batch 4096 over 8 GPUs: one GPU\'s shard of 512', 'quanonet', 5, (40, 2, 20, 2), 100, 2, 512, 40),
        ('cfg4_rdiffusion_heaqnn_q8_b2048', 'RDiffusion HEAQNN Q=8 depth 20 x 2, input 102, batch 2048', 'heaqnn', 8, (20, 2), 102, 0, 2048, 20),
        ('cfg5_shard_advection_q12_b1024', 'Advection QuanONet Q=12 Net40-2-20-2, batch 8192 over 8 GPUs: one GPU\'s shard of 1024', 'quanonet', 12, (40, 2, 20, 2), 100, 2, 1024, 3),
    ]
    for key, what, kind, n, net, b_in, t_in, batch, k in cases:
        torch.manual_seed(0)
        if kind == 'quanonet':
            model = QuanONetPT(n, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=True).to(dev)
            bd, bl, td, tl = net
            E, blk = (bd + td) * n, bd * bl + td * tl
        else:
            model = HEAQNNPT(n, b_in, net, scale_coeff=0.1, if_trainable_freq=True).to(dev)
            E, blk = net[0] * n, net[0] * net[1]
        R = E + 3 * n * blk
        tr = DataParallelTrainer(model, lr=1e-4)
        nbat = 4
        rng = np.random.default_rng(7)
        ins = [torch.tensor(rng.normal(size=(nbat * batch, b_in)), device=dev)]
        if kind == 'quanonet':
            ins.append(torch.tensor(rng.uniform(size=(nbat * batch, t_in)), device=dev))
        y = torch.tensor(rng.normal(scale=0.5, size=(nbat * batch, 1)), device=dev)
        rows = torch.zeros(nbat, tr.numel + 2, dtype=torch.float64, device=dev)
        bounds = [j * batch for j in range(nbat + 1)]

        def run_steps(first, kk, tr=tr, ins=ins, y=y, rows=rows, bounds=bounds, nbat=nbat, batch=batch):
            done = 0
            while done < kk:
                start = (first + done) % nbat
                m = min(nbat - start, kk - done)
                tr.train_steps(ins, y, bounds[start:start + m + 1], [batch] * m, rows[start:start + m])
                done += m
        fwd_out = torch.empty(nbat * batch, dtype=torch.float64, device=dev)

        def fwd_steps(first, kk, tr=tr, ins=ins, fwd_out=fwd_out, nbat=nbat, batch=batch):
            done = 0
            while done < kk:
                m = min(nbat, kk - done)
                _lib.model_forward_chunks(tr.desc, ins[0][:m * batch], ins[1][:m * batch] if len(ins) > 1 else None,
                                          tr.pflat, batch, out=fwd_out[:m * batch])
                done += m
        run_steps(0, max(2, k // 4))
        el, wins = measure_many(run_steps, k, 0.35)
        fwd_steps(0, 2)
        elf, _ = measure_many(fwd_steps, k, 0.15)
        tr.check_status()
        ms = 1e3 * el / k
        flops = 22.0 * (1 << n) * R * batch
        out[key] = {"workload": what, "batch": batch, "n_qubits": n, "rotation_gates_R": R,
                    "ms_per_step": ms, "train_samples_per_s": batch * k / el,
                    "circuit_evals_per_s": batch * k / elf, "windows": len(wins), "steps_per_window": k,
                    "algorithmic_flops_per_step": flops,
                    "frac_fp64_vector_step": flops / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        del tr, model, ins, y, rows, fwd_out
    # the whole PTSolver.train loop (permutation, one gather per epoch, steps, bookkeeping) over 100 000 resident rows:
    # at the headline batch and at the reference's own training batch of 100 (scripts/reproduce_benchmarks1.sh:15-21)
    import tempfile
    rng = np.random.default_rng(0)
    nrows = 100000
    data = {'train_branch_input': rng.normal(size=(nrows, B_IN)), 'train_trunk_input': rng.uniform(size=(nrows, T_IN)),
            'train_output': rng.normal(scale=0.5, size=(nrows, 1)),
            'test_branch_input': rng.normal(size=(256, B_IN)), 'test_trunk_input': rng.uniform(size=(256, T_IN)),
            'test_output': rng.normal(scale=0.5, size=(256, 1))}
    for bs, epochs in ((1024, 6), (100, 3)):
        cfg = {'model_type': 'QuanONet', 'operator': 'Advection', 'num_qubits': N_QUBITS, 'net_size': list(NET),
               'scale_coeff': 0.1, 'if_trainable_freq': 'true', 'learning_rate': 1e-4, 'batch_size': bs,
               'num_epochs': 1, 'if_save': False, 'prefix': tempfile.mkdtemp()}
        set_random_seed(0)
        sv = PTSolver(cfg, data, device=dev, log=lambda *a, **k: None)
        sv.config['num_epochs'] = 4 if bs >= 512 else 1          # warm-up: >= 30 ms of load (the clock ramps that long after idle)
        sv.train()
        torch.cuda.synchronize()
        sv.config['num_epochs'] = epochs
        t0 = time.perf_counter()
        sv.train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = epochs * int(np.ceil(nrows / bs))
        out[f'ptsolver_loop_q5_batch{bs}'] = {
            "workload": f"PTSolver.train whole loop, Advection QuanONet Q=5 Net40-2-20-2, {nrows} resident rows, batch {bs}",
            "batch": bs, "epochs": epochs, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "train_samples_per_s": epochs * nrows / dt,
            "frac_fp64_vector_step": 22.0 * 32 * 2100 * (nrows * epochs) / dt / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        del sv
    # the reference's own training configurations through the same loop (batch 100, Adam 1e-4, scripts/reproduce_benchmarks1.sh:15-21):
    # the README's default model -- TF-QuanONet Antideriv Q5 Net20-2-10-2, 1000 epochs x 10 000 rows = 10^5 steps, "~80 min on a
    # server-class CPU" with MindQuantum (README.md:167-178; the only published wall time) -- and the shipped Q2 Net5-1-5-1 model
    for key, what, nq, net, b_in, anchor in (
            ('ptsolver_loop_readme_default_q5_net20-2-10-2_batch100', 'README default: TF-QuanONet Antideriv Q=5 Net20-2-10-2, num_points_0 100', 5, [20, 2, 10, 2], 100,
             {"reference_wall_time": "~80 min (server-class CPU, MindQuantum), README.md:167-178", "reference_steps": 100000}),
            ('ptsolver_loop_antideriv_q2_net5-1-5-1_batch100', 'shipped checkpoint\'s model: QuanONet Antideriv Q=2 Net5-1-5-1, num_points_0 10', 2, [5, 1, 5, 1], 10, None)):
        nrows = 10000
        data = {'train_branch_input': rng.normal(size=(nrows, b_in)), 'train_trunk_input': rng.uniform(size=(nrows, 1)),
                'train_output': rng.normal(scale=0.5, size=(nrows, 1)),
                'test_branch_input': rng.normal(size=(64, b_in)), 'test_trunk_input': rng.uniform(size=(64, 1)),
                'test_output': rng.normal(scale=0.5, size=(64, 1))}
        cfg = {'model_type': 'QuanONet', 'operator': 'Antideriv', 'num_qubits': nq, 'net_size': net, 'scale_coeff': 0.01,
               'if_trainable_freq': 'true', 'learning_rate': 1e-4, 'batch_size': 100, 'num_epochs': 8, 'if_save': False,
               'prefix': tempfile.mkdtemp()}
        set_random_seed(0)
        sv = PTSolver(cfg, data, device=dev, log=lambda *a, **k: None)
        sv.train()                                               # warm-up: 800 steps
        torch.cuda.synchronize()
        sv.config['num_epochs'] = 20
        t0 = time.perf_counter()
        sv.train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = 20 * (nrows // 100)
        out[key] = {"workload": f"PTSolver.train whole loop, {what}, {nrows} resident rows, batch 100", "batch": 100, "steps": steps,
                    "ms_per_step": 1e3 * dt / steps, "train_samples_per_s": 20 * nrows / dt,
                    "projected_seconds_for_the_reference_run": 1e5 * dt / steps}
        if anchor:
            out[key].update(anchor)
        del sv
    return out



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
    ap.add_argument('--no-secondary', action='store_true', help='skip the other BASELINE configs (N = 1 only)')
    ap.add_argument('--budget-scale', type=float, default=1.0, help='scales the seconds of repeated windows per measurement')
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

    n_batches = 32                                            # device-resident synthetic set, cycled (PTSolver's epoch at this batch: 97 steps per host call)
    branch, trunk, y = synth(rank, n_batches * batch)
    branch = torch.tensor(branch, device=dev); trunk = torch.tensor(trunk, device=dev); y = torch.tensor(y, device=dev)

    def step(i):
        s = (i % n_batches) * batch
        return trainer.train_step(branch[s:s + batch], trunk[s:s + batch], y[s:s + batch],
                                  global_batch=batch * world)

    # One device: the steps are issued as the product's epoch loop issues them (PTSolver.train -> qhea_model_train_steps):
    # runs of consecutive steps from one host call, inside which a step's reduce kernel writes the next step's layer
    # records instead of a prep launch.  Same batches in the same order as the per-step loop (step i: batch i mod 32).
    epoch_call = trainer.epoch_call and not os.environ.get('QHEA_BENCH_PER_STEP')
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

    def measure(fn, k, budget_s=2.0):
        """Median over repeated windows of exactly k steps each: one 20-step window is 3 ms, short enough for the
        clock ramp after the idle fence to move it by several percent (BASELINE.md section 3 asks for the median).  The
        budgets add up to > 3 s of GPU work per run, so that a sampler outside the process sees the device busy."""
        w0 = timed_window(fn, k, 0)
        n_win = int(min(1000, max(5, np.ceil(args.budget_scale * budget_s / max(w0, 1e-6)))))
        wins = [w0] + [timed_window(fn, k, (j + 1) * k) for j in range(n_win - 1)]
        return float(np.median(wins)), wins

    if world > 1 and not os.environ.get('QHEA_BENCH_NO_CALIBRATION'):
        # as PTSolver.train does before its first epoch: both forms of the peer exchange timed on one batch, the faster kept
        trainer.calibrate_exchange(branch[:batch], trunk[:batch], y[:batch], global_batch=batch * world)
        epoch_call = trainer.epoch_call and not os.environ.get('QHEA_BENCH_PER_STEP')
        if rank == 0:
            print("data-parallel exchange: " + trainer.dp_exchange_reason, file=sys.stderr, flush=True)
        if not epoch_call and hasattr(step, 'many'):
            del step.many
    if epoch_call:
        run_steps(0, args.warmup)
    else:
        for i in range(args.warmup):
            step(i)
    elapsed, windows = measure(step, args.steps)
    samples_per_s = batch * world * args.steps / elapsed
    trainer.check_status()

    # N > 1: which devices the ranks ran on, and the same step through the OTHER exchanges in the same run (every rank
    # switches together), so that the line explains itself: the exchange inside the reduce kernel (the headline when the
    # peer buffers could be mapped), the separate one-workgroup exchange kernel, and the collective library's all-reduce
    devices, alternatives = None, None
    if world > 1:
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": pr.name,
                "pci_bus_id": getattr(pr, 'pci_bus_id', None), "pci_device_id": getattr(pr, 'pci_device_id', None),
                "pci_domain_id": getattr(pr, 'pci_domain_id', None), "uuid": str(getattr(pr, 'uuid', '')),
                "compute_units": pr.multi_processor_count, "hostname": os.uname().nodename, "pid": os.getpid()}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        per_step = lambda i: step(i)                                   # no .many: one host call per step
        alternatives = {"headline (the product's calibrated choice): " +
                        ("exchange inside the reduce kernel, runs of steps per host call" if epoch_call else
                         "one host call per step"): 1e3 * elapsed / args.steps}
        saved = (trainer.peer, trainer.peer_fused)
        if trainer.peer is not None:
            if trainer.desc is not None and trainer.fused_ok:
                trainer.peer_fused = True
                el, _ = measure(per_step, args.steps, budget_s=0.5)
                alternatives["exchange inside the reduce kernel, one host call per step"] = 1e3 * el / args.steps
                if trainer.peer_fused and not saved[1]:
                    run_steps.many = run_steps
                    el, _ = measure(run_steps, args.steps, budget_s=0.5)
                    alternatives["exchange inside the reduce kernel, runs of steps per host call"] = 1e3 * el / args.steps
            trainer.peer_fused = False
            el, _ = measure(per_step, args.steps, budget_s=0.5)
            alternatives["separate one-workgroup exchange kernel (prep, circuit, reduce, exchange + Adam)"] = 1e3 * el / args.steps
        trainer.peer, trainer.peer_fused = None, False
        el, _ = measure(per_step, args.steps, budget_s=0.5)
        alternatives[f"all_reduce ({args.backend}) + Adam launch"] = 1e3 * el / args.steps
        trainer.peer, trainer.peer_fused = saved
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
    fwd_elapsed, _ = measure(fwd_steps if not os.environ.get('QHEA_BENCH_PER_STEP') else fwd_only, args.steps, budget_s=1.0)
    evals_per_s = batch * world * args.steps / fwd_elapsed

    # the same rows in the chunks PTSolver.evaluate uses for a test set (16384 rows per forward launch):
    # what an evaluation of a trained model runs at, beside the per-training-batch figure above
    eval_chunk = 16384
    n_rows = n_batches * batch

    def fwd_eval(_i=0):
        _lib.model_forward_chunks(trainer.desc, branch, trunk, trainer.pflat, eval_chunk, out=fwd_out)
    for i in range(3):
        fwd_eval()
    ev_elapsed, _ = measure(fwd_eval, max(4, args.steps // 8), budget_s=0.5)
    evals_per_s_eval = n_rows * world * max(4, args.steps // 8) / ev_elapsed

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

    # the clock this device's shader engines run at inside a kernel (diagnostic: devices of one model differ by several
    # percent, and the latency-bound kernels with them -- the box-to-box spread of these numbers)
    clock = None
    if rank == 0:
        med, lo, hi = _lib.clock_probe(dev)
        clock = {"in_kernel_clock_mhz_median": med, "min": lo, "max": hi,
                 "how": "1024 one-wave workgroups, dependent fp64 FMA chain, s_memtime / s_memrealtime x 100 MHz (qhea_clock_probe)"}

    secondary = None
    if world == 1 and not args.no_secondary and batch == BATCH and args.backward_variant == 'auto':
        def measure_many(fn, k, budget_s):
            fn.many = fn
            return measure(fn, k, budget_s=budget_s)
        try:                                            # (never at the headline's expense: the line is printed whatever happens here)
            secondary = secondary_configs(dev, measure_many)
        except Exception as e:                          # noqa: BLE001
            secondary = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline()
            except Exception as e:                      # noqa: BLE001
                cpu = {"error": f"{type(e).__name__}: {e}"}
        line = {
            "metric": "train samples/sec (circuit-evals/sec alongside), QuanONet Q=5 Net40-2-20-2 Advection",
            "value": samples_per_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Advection QuanONet Q=5 Net40-2-20-2 b_in=100 t_in=2, batch {batch} per GPU, "
                                   "fp64, Adam lr=1e-4, trainable frequency",
                       "global_batch": batch * world, "parallelism": f"dp{world}",
                       "dp_exchange": (None if world == 1 else
                                       "peer-mapped buffers, sum over the ranks + Adam inside the reduce kernel "
                                       "(qhea_model_dp_train_steps: two launches per step)" if trainer.peer_fused else
                                       "peer-mapped buffers, sum + Adam in one exchange kernel (csrc/hea_dp.hip)"
                                       if trainer.peer is not None else "all_reduce (" + args.backend + ") + Adam launch"),
                       "dp_exchange_reason": trainer.dp_exchange_reason,
                       "world_size": world, "backend": None if world == 1 else args.backend,
                       "same_device_rehearsal": bool(args.same_device), "devices": devices,
                       "ms_per_step_by_exchange": alternatives},
            "timing": {"windows": len(windows), "steps_per_window": args.steps,
                       "ms_per_step_median": 1e3 * elapsed / args.steps,
                       "ms_per_step_first_window": 1e3 * windows[0] / args.steps,
                       "ms_per_step_min": 1e3 * min(windows) / args.steps,
                       "ms_per_step_max": 1e3 * max(windows) / args.steps,
                       "issue": ("runs of up to 32 consecutive steps per host call (qhea_model_train_steps / "
                                 "qhea_model_dp_train_steps, the epoch loop of PTSolver.train: a step's reduce kernel writes "
                                 "the next step's layer records)"
                                 if epoch_call else "one host call per step (prep, circuit, reduce launches each)"),
                       "note": "value = median over windows of EXACTLY `steps` training steps each, every window "
                               "bracketed by barrier + device synchronize, max over ranks per window"},
            "circuit_evals_per_s": evals_per_s,
            "circuit_evals_issue": ("one host call per forward launch (record preparation every call)"
                                    if os.environ.get('QHEA_BENCH_PER_STEP') else
                                    "the resident set in chunks of one batch per host call (qhea_model_forward_chunks, what "
                                    "PTSolver.predict runs: one record preparation per call, one forward launch per batch)"),
            "circuit_evals_per_s_evaluation_chunks": {"value": evals_per_s_eval, "rows_per_launch": eval_chunk,
                                                      "what": "the same rows in PTSolver.evaluate's chunks (a test set, not a training batch)"},
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary, "device_clock": clock,
        }
        print(json.dumps(line))
    if trainer.peer is not None:
        trainer.peer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
