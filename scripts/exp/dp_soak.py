"""Soak test of the data-parallel exchange (csrc/hea_dp.hip) with several processes sharing the GPU: many thousands of
back-to-back exchanges without a host sync, ranks skewed against each other by random device-side sleeps, every result
checked on the device against the closed-form sum (rank r contributes (r + 1) * (round + i * 1e-3) at element i).
Usage: python scripts/exp/dp_soak.py [world] [rounds]"""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def worker(rank, world, port, rounds, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from quanonet_amd import _lib
    from quanonet_amd.solver import PeerExchange
    dev = torch.device('cuda', 0)
    n = 2403
    px, _reason = PeerExchange.create(dist, rank, world, n, dev)
    assert px is not None
    ramp = torch.arange(n, dtype=torch.float64, device=dev) * 1e-3
    worst = torch.zeros((), dtype=torch.float64, device=dev)
    buf = torch.empty(n, dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(1000 + rank)
    tri = world * (world + 1) / 2
    for rnd in range(1, rounds + 1):
        if torch.rand((), generator=gen).item() < 0.05:                  # skew: this rank falls behind now and then
            torch.cuda._sleep(int(torch.randint(10_000, 400_000, (), generator=gen).item()))
        torch.add(ramp, float(rnd), out=buf); buf.mul_(rank + 1)
        px.seq += 1
        _lib.dp_allreduce_adam(rank, world, px.bufs, px.seq, buf, buf)
        # expected: sum over r of (r+1)*(rnd + ramp) in rank order; products (r+1)*x are exact multiples, the sum is
        # formed here in the same order as the kernel forms it
        want = torch.zeros_like(buf)
        base = ramp + float(rnd)
        for r in range(world):
            want += base * (r + 1)
        worst = torch.maximum(worst, (buf - want).abs().max())
        if rnd % 5000 == 0 and rank == 0:
            print(f'round {rnd} worst {worst.item():.1e}', flush=True)
    px.check_status()
    q.put((rank, float(worst.item())))
    dist.barrier(); px.close(); dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=worker, args=(r, world, port, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    print(json.dumps({'world': world, 'rounds': rounds, 'worst_abs_error_per_rank': [w for _, w in res],
                      'exit_codes': [p.exitcode for p in procs]}))
