"""
Randomised parity of the model-level training step (scripts/exp/random_parity.py): 40 random (qubits, block list, batch,
frequency mode, model kind) cases, two consecutive steps each, gradients and the Adam-updated parameters against the oracle
(C engine) + torch.optim.Adam at 1e-9.  The seeds 1-4 x 80-120 cases were run when the round's kernel changes went in.
"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_two_steps_each():
    spec = importlib.util.spec_from_file_location('random_parity', os.path.join(ROOT, 'scripts', 'exp', 'random_parity.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst = mod.run(cases=40, seed=0, verbose=False)
    assert worst < 1e-9
