#!/usr/bin/env python3
"""
Generates the committed golden fixtures under tests/golden/.  Runs ONLY in the
build container (it reads /root/reference); the GPU box never runs it.

What it writes (all data, no reference source text):
  * antideriv_q2.npz            - the reference's shipped Antideriv Q2 weight file
                                  (pretrained_weights/Antideriv/.../best_model.npz), verbatim arrays
  * {advection,rdiffusion,darcy}_q5.npz - the three shipped MindSpore .ckpt files decoded
                                  with quanonet_amd.checkpoint.read_mindspore_ckpt (MindSpore key names)
  * pde_truths.npz              - numerical PDE solutions for the notebook's OOD inputs, produced by
                                  importing the reference's own numpy/scipy solvers
                                  (data_utils/data_generation.py:224-352); deterministic, no RNG
  * known_answers.json          - K1..K8 expected figures and their sources
  * hea_vectors.npz             - oracle-generated random vectors (x, w, g -> out, grad_x, grad_w)
                                  at cfg-1/2/4-shaped sizes, fixed seeds (regression pins for the oracle
                                  itself and inputs for the GPU parity tests)
"""
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from quanonet_amd.checkpoint import read_mindspore_ckpt          # noqa: E402
from oracle import hea_oracle as O                               # noqa: E402


def main():
    # 1. Antideriv npz: copy arrays
    src = os.path.join(REF, 'pretrained_weights/Antideriv/'
                       'Antideriv_QuanONet_Net5-1-5-1_Q2_TF_S0.001_1000x100_Seed0/best_model.npz')
    d = np.load(src, allow_pickle=False)
    np.savez(os.path.join(HERE, 'antideriv_q2.npz'), **{k: d[k] for k in d.files})

    # 2. decode the three Q5 checkpoints
    for op, sub in [('advection', 'Advection/Advection_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0'),
                    ('rdiffusion', 'RDiffusion/RDiffusion_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0'),
                    ('darcy', 'Darcy/Darcy_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x25_Seed0')]:
        st = read_mindspore_ckpt(os.path.join(REF, 'pretrained_weights', sub, 'best_model.ckpt'))
        print(op, {k: v.shape for k, v in st.items()})
        np.savez(os.path.join(HERE, f'{op}_q5.npz'), **st)

    # 3. PDE truths via the reference's own solvers (notebook cell 7 recipe)
    sys.path.insert(0, REF)
    from data_utils.data_generation import (solve_advection_pde, solve_rdiffusion_pde,
                                            solve_darcy_pde)
    x_cal = np.linspace(0, 1, 100).astype(np.float32)
    truths = {}
    for name, solver, ncal, kw in [
            ('advection', solve_advection_pde, 100, dict(c=1.0)),
            ('rdiffusion', solve_rdiffusion_pde, 100, dict(D=0.01, k=0.01)),
            ('darcy', solve_darcy_pde, 25, dict(K=0.1, f=-1.0))]:
        for tag, fn in [('sin2pi', lambda x: np.sin(2 * np.pi * x)),
                        ('sin4pi', lambda x: np.sin(4 * np.pi * x))]:
            raw, _ = solver(ncal, 0.2, u0_cal=fn(x_cal), **kw)
            truths[f'{name}_{tag}'] = np.asarray(raw.T, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'pde_truths.npz'), **truths)

    # 4. known answers
    ka = {
        'K1': {'source': 'ibm_inference.py:180-183', 'weights': 'antideriv_q2.npz',
               'truth': 'sin(pi x)/pi', 'rel_l2_max': 0.05, 'survey_rel_l2': 0.0269},
        'K2': {'source': 'ibm_inference.py:185-187', 'weights': 'antideriv_q2.npz',
               'truth': 'x^2/2', 'rel_l2_max': 0.12, 'survey_rel_l2': 0.089},
        'K3': {'source': 'visualization.ipynb cell 7 output 1', 'weights': 'advection_q5.npz',
               'truth': 'advection_sin2pi', 'mse': '3.0e-03', 'mae': '4.5e-02'},
        'K4': {'source': 'visualization.ipynb cell 7 output 2', 'weights': 'advection_q5.npz',
               'truth': 'advection_sin4pi', 'mse': '1.2e-02', 'mae': '8.8e-02'},
        'K5': {'source': 'visualization.ipynb cell 7 output 3', 'weights': 'rdiffusion_q5.npz',
               'truth': 'rdiffusion_sin2pi', 'mse': '1.0e-04', 'mae': '8.1e-03'},
        'K6': {'source': 'visualization.ipynb cell 7 output 4', 'weights': 'rdiffusion_q5.npz',
               'truth': 'rdiffusion_sin4pi', 'mse': '7.0e-04', 'mae': '2.1e-02'},
        'K7': {'source': 'visualization.ipynb cell 7 output 5', 'weights': 'darcy_q5.npz',
               'truth': 'darcy_sin2pi', 'mse': '7.6e-04', 'mae': '2.1e-02'},
        'K8': {'source': 'visualization.ipynb cell 7 output 6', 'weights': 'darcy_q5.npz',
               'truth': 'darcy_sin4pi', 'mse': '9.2e-03', 'mae': '7.7e-02'},
    }
    with open(os.path.join(HERE, 'known_answers.json'), 'w') as f:
        json.dump(ka, f, indent=1)

    # 5. oracle-generated random vectors
    vec = {}
    cases = [
        ('cfg1_q2', 2, O.block_configs_quanonet(2, (5, 1, 5, 1)), 32, 11),
        ('cfg2_q5', 5, O.block_configs_quanonet(5, (40, 2, 20, 2)), 16, 12),
        ('cfg4_q8', 8, O.block_configs_heaqnn(8, (20, 2)), 6, 13),
        ('q3_small', 3, O.block_configs_quanonet(3, (2, 1, 1, 3)), 7, 14),
        ('q6_small', 6, O.block_configs_heaqnn(6, (3, 2)), 5, 15),
    ]
    for name, n, cfgs, B, seed in cases:
        rng = np.random.default_rng(seed)
        E, blk = O.circuit_sizes(n, cfgs)
        x = rng.uniform(-np.pi, np.pi, size=(B, E))
        w = rng.uniform(-np.pi, np.pi, size=(blk, 3, n))
        g = rng.normal(size=B)
        off, co = O.ham_params(n, -5.0, 5.0)
        out, gx, gw = O.hea_backward(n, cfgs, x, w, g, off, co)
        vec[f'{name}.n'] = np.array(n)
        vec[f'{name}.cfgs'] = np.array(cfgs, dtype=np.int64)
        vec[f'{name}.x'] = x
        vec[f'{name}.w'] = w
        vec[f'{name}.g'] = g
        vec[f'{name}.out'] = out
        vec[f'{name}.grad_x'] = gx
        vec[f'{name}.grad_w'] = gw
    np.savez_compressed(os.path.join(HERE, 'hea_vectors.npz'), **vec)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
