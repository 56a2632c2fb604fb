"""Fixed cost of ONE torch.distributed all_reduce (SUM) of the flat gradient buffer (2403 doubles = 19 KB at Q5) as the
data-parallel step issues it (solver.py: DataParallelTrainer.train_step).  On the one-GPU box only world_size 1 can run
over RCCL: that is the launch + enqueue floor of the collective, not the xGMI latency, which is stated as an
estimate in DESIGN.md section 8.  Launch: python -m torch.distributed.run --standalone --nproc-per-node 1 scripts/allreduce_cost.py"""
import os, sys, time, json
import torch
import torch.distributed as dist
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
res = {}
for n in (2403, 5763):
    buf = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(20):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); dist.all_reduce(buf); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / len(ev)
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    res[f'{n}_doubles'] = {'device_us_median': 1e3 * ts[len(ts) // 2], 'host_us_per_call': 1e6 * wall}
if dist.get_rank() == 0:
    print(json.dumps({'world': dist.get_world_size(), 'backend': 'nccl (RCCL)', 'all_reduce': res}))
dist.destroy_process_group()
