"""Box-to-box spread ("Q8 288 <-> 323 us, no cause given"): the in-kernel shader clock of THIS device
(qhea_clock_probe) beside the step times of cfg 4 (HEAQNN Q8, B = 2048) and cfg 2 (Q5, B = 1024).  Run on several boxes;
one JSON line per run is appended to profiles/r03_box_spread.jsonl by the caller."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quanonet_amd import _lib
from quanonet_amd.models import QuanONetPT, HEAQNNPT
from quanonet_amd.solver import DataParallelTrainer
dev = torch.device('cuda', 0)


def step_us(model, ins_shapes, batch):
    tr = DataParallelTrainer(model.to(dev), lr=1e-4)
    rng = np.random.default_rng(0); nb = 4
    ins = [torch.tensor(rng.normal(size=(nb * batch, w)), device=dev) for w in ins_shapes]
    y = torch.tensor(rng.normal(scale=0.5, size=(nb * batch, 1)), device=dev)
    rows = torch.zeros(nb, tr.numel + 2, dtype=torch.float64, device=dev); bounds = [i * batch for i in range(nb + 1)]
    for _ in range(5):
        tr.train_steps(ins, y, bounds, [batch] * nb, rows)
    ts = []
    for _ in range(25):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            tr.train_steps(ins, y, bounds, [batch] * nb, rows)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / (5 * nb))
    tr.check_status()
    return 1e6 * float(np.median(ts)), 1e6 * float(np.min(ts)), 1e6 * float(np.max(ts))


torch.manual_seed(0)
q8 = step_us(HEAQNNPT(8, 102, (20, 2), scale_coeff=0.1, if_trainable_freq=True), [102], 2048)
q5 = step_us(QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True), [100, 2], 1024)
clk = _lib.clock_probe(dev)                                    # after the load, as a kernel would see it
pr = torch.cuda.get_device_properties(dev)
print(json.dumps({'uuid': str(getattr(pr, 'uuid', '')), 'pci_bus_id': getattr(pr, 'pci_bus_id', None),
                  'clock_mhz_median_min_max': [round(v, 1) for v in clk],
                  'q8_b2048_step_us_median_min_max': [round(v, 2) for v in q8],
                  'q5_b1024_step_us_median_min_max': [round(v, 2) for v in q5]}))
