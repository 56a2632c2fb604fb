"""Randomised parity sweep of the model-level training step against the oracle (C engine): random qubit counts, block lists,
batch sizes, frequency modes and model kinds; two consecutive steps each (the second runs on records written by the first
step's reduce kernel where the shape allows), gradients / sse at 1e-9, parameters after Adam against torch.optim.Adam.
Usage: python scripts/exp/random_parity.py [cases] [seed]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hea_oracle as O
from oracle import c_oracle as C
from quanonet_amd.models import QuanONetPT, HEAQNNPT
from quanonet_amd.solver import DataParallelTrainer

def run(cases=60, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    dev = torch.device('cuda', 0)
    worst = 0.0
    for case in range(cases):
        n = int(rng.choice([2, 3, 4, 5, 5, 5, 6, 8, 10]))
        kind = 'heaqnn' if rng.random() < 0.3 else 'quanonet'
        ld = int(rng.integers(1, 3))
        same_ld = rng.random() < 0.7
        depth = lambda: int(rng.integers(1, 5 if n <= 8 else 3))
        net = (depth(), ld) if kind == 'heaqnn' else (depth(), ld, depth(), ld if same_ld else 3 - ld)
        B = int(rng.choice([1, 3, 17, 32, 53, 64, 100, 129, 257, 513 if n <= 6 else 40]))
        tf = bool(rng.random() < 0.7)
        b_in, t_in = int(rng.integers(2, 9)), int(rng.integers(1, 3))
        torch.manual_seed(case)
        if kind == 'heaqnn':
            model = HEAQNNPT(n, b_in + t_in, net, scale_coeff=0.3, if_trainable_freq=tf)
        else:
            model = QuanONetPT(n, b_in, t_in, net, scale_coeff=0.3, if_trainable_freq=tf)
        model = model.double()
        with torch.no_grad():
            for k, p in model.named_parameters():
                if 'freq' in k and 'bias' in k:
                    p.copy_(torch.from_numpy(rng.normal(scale=0.3, size=p.shape)))
        cpu = copy.deepcopy(model)
        names = [k for k, _ in cpu.named_parameters()]
        params = [p for _, p in cpu.named_parameters()]
        opt = torch.optim.Adam(params, lr=1e-2)
        tr = DataParallelTrainer(model.to(dev), lr=1e-2)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        for step in range(2):
            br = rng.normal(size=(B, b_in)); tk = rng.uniform(size=(B, t_in)); y = rng.normal(scale=0.5, size=B)
            sd = {k: v.detach().numpy() for k, v in cpu.state_dict().items()}
            if kind == 'heaqnn':
                loss, grads, _ = O.heaqnn_loss_and_grads(sd, np.concatenate([br, tk], 1), y, n, net, scale_coeff=None if tf else 0.3, engine=C)
                batch = (t(np.concatenate([br, tk], 1)), t(y))
            else:
                loss, grads, _ = O.quanonet_loss_and_grads(sd, br, tk, y, n, net, scale_coeff=None if tf else 0.3, engine=C)
                batch = (t(br), t(tk), t(y))
            flat = tr.train_step(*batch).clone()
            torch.cuda.synchronize(); tr.check_status()
            want = np.concatenate([np.asarray(grads[k], np.float64).reshape(-1) for k in names])
            err = float(np.abs(flat[:-2].cpu().numpy() - want).max()); worst = max(worst, err)
            assert err < 1e-9, (case, n, kind, net, B, tf, step, err)
            assert abs(flat[-2].item() - loss * B) < 1e-9 * max(1.0, loss * B)
            opt.zero_grad()
            for k, p in zip(names, params):
                p.grad = torch.from_numpy(np.asarray(grads[k], np.float64).reshape(p.shape).copy())
            opt.step()
            perr = float(np.abs(tr.pflat.cpu().numpy() - np.concatenate([p.detach().numpy().reshape(-1) for p in params])).max())
            assert perr < 1e-9, (case, 'params', perr)
        if verbose: print(case, n, kind, net, B, tf, 'ok', f'{err:.1e}', flush=True)
    if verbose: print('all', cases, 'cases ok; worst gradient error', worst)
    return worst



if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
