"""Debug: N same-device ranks, bench-like loop, status checked after every window."""
import os, sys, socket, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch


def worker(rank, world, port, fused, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    torch.manual_seed(0)
    tr = DataParallelTrainer(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev), lr=1e-4,
                             world_size=world, dist=dist)
    tr.peer_fused = tr.peer_fused and fused
    batch = int(os.environ.get('DBG_BATCH', '1024'))
    rng = np.random.default_rng(1000 + rank)
    br = torch.tensor(rng.normal(size=(8 * batch, 100)), device=dev); tk = torch.tensor(rng.uniform(size=(8 * batch, 2)), device=dev)
    y = torch.tensor(rng.normal(scale=0.5, size=(8 * batch, 1)), device=dev)
    log = []
    for win in range(30):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            s = ((win * 5 + i) % 8) * batch
            tr.train_step(br[s:s + batch], tk[s:s + batch], y[s:s + batch], global_batch=batch * world)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            _lib.check_status(dev); pipe = 'ok'
        except Exception as e:
            pipe = 'PIPE'
        try:
            tr.peer.check_status(); ex = 'ok'
        except Exception as e:
            ex = 'EXCH'
        log.append((win, round(dt * 1e3, 2), pipe, ex))
        if ex != 'ok' or pipe != 'ok':
            break
    q.put((rank, log[-3:], len(log)))
    dist.barrier()
    tr.peer.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    world = int(sys.argv[1]); fused = int(sys.argv[2])
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=worker, args=(r, world, port, fused, q)) for r in range(world)]
    for p in procs: p.start()
    for _ in procs: print(q.get(timeout=300), flush=True)
    for p in procs: p.join(timeout=60)
