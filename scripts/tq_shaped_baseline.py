#!/usr/bin/env python3
"""
A TorchQuantum-SHAPED baseline, restated in plain PyTorch (SURVEY.md section 8(d), item ii).

What the reference's PyTorch path does per training step (core/quantum_circuits_tq.py:65-127 driving
TorchQuantum 0.1.8, which is not installed here): keep the batch of statevectors as a dense tensor, apply every
gate as its own batched contraction on the target wire, and let autograd store every intermediate state for the
backward pass.  This file is that structure and nothing more -- one framework kernel (or several) per gate, state
streamed through HBM every time -- so that the same MI355X (eager PyTorch-ROCm) and the same host CPU can be timed
running the reference's *shape* of computation next to this repository's fused kernels.  It is measurement
scaffolding: nothing in quanonet_amd/ imports it.

    python scripts/tq_shaped_baseline.py [--device cuda|cpu] [--dtype c128|c64] [--steps K] [--check]

Prints one JSON line: train samples/s and forward evals/s for cfg 2 (Q5, Net40-2-20-2, B = 1024).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, NET, B_IN, T_IN, BATCH = 5, (40, 2, 20, 2), 100, 2, 1024


def block_configs(n, net):
    bd, bl, td, tl = net
    return [(n, tl)] * td + [(n, bl)] * bd                 # trunk blocks first (quantum_circuits_tq.py:130-138)


class GateByGateHEA(torch.nn.Module):
    """Dense state (B, 2, ..., 2) with wire i on axis n - i (so that bit i of the flat index is wire i)."""

    def __init__(self, n, cfgs, cdtype):
        super().__init__()
        self.n, self.cfgs, self.cdtype = n, cfgs, cdtype
        self.rdtype = torch.float64 if cdtype == torch.complex128 else torch.float32
        blk = sum(c[1] for c in cfgs)
        w = torch.empty(blk, 3, n).uniform_(-math.pi, math.pi)
        self.ansatz_weights = torch.nn.Parameter(w.to(self.rdtype))

    def _axis(self, wire):
        return self.n - wire                                  # axis 0 is the batch

    def _gate(self, state, mat, wire):
        """mat: (2,2) shared or (B,2,2) per sample; contraction on the wire's axis (what tq's bmm does)."""
        ax = self._axis(wire)
        st = state.movedim(ax, -1)                            # (..., 2)
        if mat.dim() == 2:
            st = torch.matmul(st, mat.transpose(0, 1))
        else:
            shp = st.shape
            st = torch.bmm(st.reshape(shp[0], -1, 2), mat.transpose(1, 2)).reshape(shp)
        return st.movedim(-1, ax)

    def _rx(self, theta):
        c, s = torch.cos(theta / 2), torch.sin(theta / 2)
        z = torch.zeros_like(c)
        return torch.stack([torch.stack([torch.complex(c, z), torch.complex(z, -s)], -1),
                            torch.stack([torch.complex(z, -s), torch.complex(c, z)], -1)], -2)

    def _ry(self, theta):
        c, s = torch.cos(theta / 2), torch.sin(theta / 2)
        z = torch.zeros_like(c)
        return torch.stack([torch.stack([torch.complex(c, z), torch.complex(-s, z)], -1),
                            torch.stack([torch.complex(s, z), torch.complex(c, z)], -1)], -2)

    def _rz(self, theta):
        c, s = torch.cos(theta / 2), torch.sin(theta / 2)
        z = torch.zeros_like(c)
        return torch.stack([torch.stack([torch.complex(c, -s), torch.complex(z, z)], -1),
                            torch.stack([torch.complex(z, z), torch.complex(c, s)], -1)], -2)

    def _cnot(self, state, control, target):
        ac, at = self._axis(control), self._axis(target)
        on = state.select(ac, 1).flip(at - 1 if at > ac else at)   # target axis index shifts after select
        return torch.stack([state.select(ac, 0), on], dim=ac)

    def forward(self, x, offset, coeff):
        Bn, n = x.shape[0], self.n
        state = torch.zeros((Bn,) + (2,) * n, dtype=self.cdtype, device=x.device)
        state[(slice(None),) + (0,) * n] = 1.0
        col = blk = 0
        for n_enc, ld in self.cfgs:
            for j in range(n_enc):
                state = self._gate(state, self._rx(x[:, col]), j % n)
                col += 1
            for _ in range(ld):
                for i in range(n):
                    w = self.ansatz_weights[blk, :, i]
                    state = self._gate(state, self._ry(w[0]), i)
                    state = self._gate(state, self._rz(w[1]), i)
                    state = self._gate(state, self._ry(w[2]), i)
                for i in range(n):
                    state = self._cnot(state, (i + 1) % n, i)
                blk += 1
        probs = (state.real ** 2 + state.imag ** 2).reshape(Bn, -1)
        k = torch.arange(1 << n, device=x.device)
        zsum = sum(1.0 - 2.0 * ((k >> i) & 1).to(probs.dtype) for i in range(n))
        return offset + coeff * (probs * zsum[None, :]).sum(dim=1, keepdim=True)


class TQShapedQuanONet(torch.nn.Module):
    def __init__(self, cdtype):
        super().__init__()
        rd = torch.float64 if cdtype == torch.complex128 else torch.float32
        bd, bl, td, tl = NET
        self.q = GateByGateHEA(N, block_configs(N, NET), cdtype)
        self.bw = torch.nn.Parameter(torch.full((bd * N,), 0.1, dtype=rd)); self.bb = torch.nn.Parameter(torch.zeros(bd * N, dtype=rd))
        self.tw = torch.nn.Parameter(torch.full((td * N,), 0.1, dtype=rd)); self.tb = torch.nn.Parameter(torch.zeros(td * N, dtype=rd))
        self.bias = torch.nn.Parameter(torch.zeros(1, dtype=rd))
        self.register_buffer('bidx', torch.arange(bd * N) % B_IN)
        self.register_buffer('tidx', torch.arange(td * N) % T_IN)

    def forward(self, branch, trunk):
        x = torch.cat([trunk[:, self.tidx] * self.tw + self.tb, branch[:, self.bidx] * self.bw + self.bb], dim=1)
        return self.q(x, 0.0, 5.0 / N) + self.bias


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    ap.add_argument('--dtype', default='c128', choices=['c128', 'c64'])
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--check', action='store_true', help='compare a small circuit with the oracle first')
    args = ap.parse_args()
    dev = torch.device(args.device)
    cdtype = torch.complex128 if args.dtype == 'c128' else torch.complex64
    rd = torch.float64 if args.dtype == 'c128' else torch.float32

    if args.check:
        from oracle import hea_oracle as O
        cfgs = [(3, 2), (4, 1)]
        m = GateByGateHEA(3, cfgs, torch.complex128).to(dev)
        rng = np.random.default_rng(0)
        x = rng.uniform(-3, 3, (4, 7))
        out = m(torch.tensor(x, device=dev), 0.3, 0.7)[:, 0].detach().cpu().numpy()
        ref = O.hea_forward(3, cfgs, x, m.ansatz_weights.detach().cpu().numpy(), 0.3, 0.7)
        assert np.abs(out - ref).max() < 1e-12, np.abs(out - ref).max()

    torch.manual_seed(0)
    model = TQShapedQuanONet(cdtype).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    rng = np.random.default_rng(1000)
    branch = torch.tensor(rng.normal(size=(BATCH, B_IN)), dtype=rd, device=dev)
    trunk = torch.tensor(rng.uniform(size=(BATCH, T_IN)), dtype=rd, device=dev)
    y = torch.tensor(rng.normal(scale=0.5, size=(BATCH, 1)), dtype=rd, device=dev)

    def sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize()

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(branch, trunk), y)
        loss.backward()
        opt.step()

    step(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    t_train = (time.perf_counter() - t0) / args.steps
    with torch.no_grad():
        model(branch, trunk); sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(branch, trunk)
        sync()
        t_fwd = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"baseline": "TorchQuantum-shaped gate-by-gate PyTorch (this script)", "device": str(dev),
                      "dtype": args.dtype, "threads": torch.get_num_threads() if dev.type == 'cpu' else None,
                      "batch": BATCH, "train_ms_per_step": 1e3 * t_train, "train_samples_per_s": BATCH / t_train,
                      "forward_ms": 1e3 * t_fwd, "forward_evals_per_s": BATCH / t_fwd}))


if __name__ == '__main__':
    main()
