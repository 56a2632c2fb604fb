"""Cost of the data-parallel exchange kernel (csrc/hea_dp.hip: sum over the ranks' peer-mapped buffers + Adam in one
launch) beside what it replaces (all_reduce + qhea_adam_step), on the ONE GPU of the box:
  * world 1, in process: the kernel's floor (publish to its own buffer, flag, collect, Adam) for 2403 / 5763 doubles;
  * world 2 and 4, processes sharing the GPU (gloo for the hand-shakes): exchange kernel with real peer traffic through
    hipIpc-mapped buffers on the same device -- no xGMI hop, which a one-GPU box cannot show.
Usage: python scripts/dp_exchange_cost.py            (spawns its own ranks)"""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * ts[len(ts) // 2]


def floor_world1():
    from quanonet_amd import _lib
    dev = torch.device('cuda', 0)
    res = {}
    for n in (2403, 5763):
        buf = _lib.dp_alloc(n, 1, dev)
        g = torch.randn(n, dtype=torch.float64, device=dev)
        p = torch.randn(n - 2, dtype=torch.float64, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
        seq = [0]
        def ex():
            seq[0] += 1
            _lib.dp_allreduce_adam(0, 1, [buf], seq[0], g, g, p, m, v, seq[0], 1e-4)
        t_ex = timed(ex)
        t_adam = timed(lambda: _lib.adam_step(p, g, m, v, 1, 1e-4))
        _lib.dp_status(buf, dev)
        _lib.dp_free(buf, dev)
        res[f'{n}_doubles'] = {'exchange_plus_adam_us': t_ex, 'adam_launch_alone_us': t_adam}
    return res


def worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from quanonet_amd import _lib
    from quanonet_amd.solver import PeerExchange
    dev = torch.device('cuda', 0)
    res = {}
    for n in (2403, 5763):
        px, _reason = PeerExchange.create(dist, rank, world, n, dev)
        g = torch.randn(n, dtype=torch.float64, device=dev)
        p = torch.randn(n - 2, dtype=torch.float64, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
        def ex():
            px.seq += 1
            _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, g, g, p, m, v, px.seq, 1e-4)
        dist.barrier()
        t = timed(ex)
        px.check_status()
        res[f'{n}_doubles'] = {'exchange_plus_adam_us': t}
        dist.barrier()
        px.close()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    out = {}
    for world in (2, 4):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=300) for _ in procs])
        for p in procs:
            p.join(timeout=60)
        out[f'world{world}_same_device'] = res[0][1]
    out['world1_floor'] = floor_world1()          # after the children: the parent touches the GPU only now
    print(json.dumps(out))
