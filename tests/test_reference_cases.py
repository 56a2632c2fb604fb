"""
GPU parity tests that restate the REFERENCE'S OWN test cases and paper configurations on the HIP path:

* the four quantum cases of compare_backends.py (:140-212 QuanONet TQ/PL/Qiskit, :219-281 HEAQNN TQ/PL/Qiskit,
  :288-376 QuanONet MindSpore vs TorchQuantum on the pretrained Antideriv npz, :383-449 HEAQNN MS vs TQ) with the
  reference's seeds, shapes, input stream (one module-level ``np.random.default_rng(0)`` consumed in the order
  ``__main__`` runs the cases, :644-671) and tolerances (1e-4 forward / 1e-4 and 5e-4 gradients on float32 modules).
  The reference compares simulator back-ends with each other; none of them is installed here, so the other side of
  each comparison is the fp64 oracle, and the same case is repeated on the fp64 module at 1e-10;
* HEAQNNPT at BASELINE.json's cfg-4 model dimensions (Q8, depth 20 x 2, 102 inputs tiled to 160 angles);
* the fixed-frequency paper configuration (FF 40-2-40-2, scripts/reproduce_benchmarks1.sh:53-68);
* K2 (ibm_inference.py:185-187) on the GPU.
Every model-level path (fused C-ABI call and torch-autograd module) is compared with the ORACLE, never only with
another HIP path.
"""
import numpy as np
import pytest
import torch

from oracle import hea_oracle as O
from oracle import c_oracle as C
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-10
ATOL_PT, ATOL_GRAD_PT, ATOL_MSPT, ATOL_GRAD_MS = 1e-4, 1e-4, 1e-4, 5e-4      # compare_backends.py:26-30


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device('cuda:0')


def _t(a, dev, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)


def _oracle(model, ins, y, n, net, scale, batch_total=None):
    sd = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    ins64 = [np.asarray(a, np.float64) for a in ins]
    sc = None if model.if_trainable_freq else scale
    if len(ins) == 2:
        return O.quanonet_loss_and_grads(sd, ins64[0], ins64[1], y, n, net, scale_coeff=sc, batch_total=batch_total,
                                         engine=C)
    return O.heaqnn_loss_and_grads(sd, ins64[0], y, n, net, scale_coeff=sc, batch_total=batch_total, engine=C)


def _check_model_paths(model, ins, y, n, net, scale, dev, batch_total=None):
    """Fused C-ABI path and autograd module path of an fp64 model, each against the oracle at 1e-10."""
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    B = len(y)
    rl, rg, ro = _oracle(model, ins, y, n, net, scale, batch_total)
    ref = np.concatenate([rg[k].reshape(-1) for k, _ in model.named_parameters()])
    tin = [_t(a, dev) for a in ins]
    fused = DataParallelTrainer(model, lr=1e-3, fused=True)
    flat = fused.loss_and_grad(*tin, _t(y, dev), global_batch=batch_total or B).clone()
    pred = _lib.model_forward(fused.desc, tin[0], tin[1] if len(tin) > 1 else None, fused.pflat).cpu().numpy()
    np.testing.assert_allclose(pred, ro, rtol=0, atol=TOL, err_msg="fused forward")
    np.testing.assert_allclose(flat[:-2].cpu().numpy(), ref, rtol=0, atol=TOL, err_msg="fused gradients")
    assert abs(flat[-2].item() - rl * (batch_total or B)) < 1e-9
    auto = DataParallelTrainer(model, lr=1e-3, fused=False)
    flat2 = auto.loss_and_grad(*tin, _t(y, dev).unsqueeze(-1), global_batch=batch_total or B)
    np.testing.assert_allclose(flat2[:-2].cpu().numpy(), ref, rtol=0, atol=TOL, err_msg="autograd gradients")
    with torch.no_grad():
        p2 = model(*tin)[:, 0].cpu().numpy()
    np.testing.assert_allclose(p2, ro, rtol=0, atol=TOL, err_msg="module forward")
    return rl, rg, ro


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json cfg 4 and the FF paper configuration
# ------------------------------------------------------------------------------------------------------------
def test_heaqnn_cfg4_model_dimensions(dev):
    """HEAQNNPT(8, 102, (20, 2)): 102 inputs tiled x2 and sliced to 160 angles, 40 sub-layers, 1280 parameters."""
    from quanonet_amd.models import HEAQNNPT
    torch.manual_seed(7)
    n, net, B = 8, (20, 2), 21
    model = HEAQNNPT(n, 102, net, scale_coeff=0.1, if_trainable_freq=True).to(dev)
    assert sum(p.numel() for p in model.parameters()) == 1280
    rng = np.random.default_rng(70)
    with torch.no_grad():
        model.freq.bias.copy_(_t(rng.normal(scale=0.3, size=160), dev))
    x = rng.normal(size=(B, 102)); y = rng.normal(scale=0.5, size=B)
    _check_model_paths(model, (x,), y, n, net, 0.1, dev, batch_total=2 * B)


@pytest.mark.parametrize('tf', [True, False])
def test_heaqnn_small_both_frequency_modes(dev, tf):
    from quanonet_amd.models import HEAQNNPT
    torch.manual_seed(8)
    n, net, B = 5, (3, 2), 33
    model = HEAQNNPT(n, 9, net, scale_coeff=0.2, if_trainable_freq=tf).to(dev)
    rng = np.random.default_rng(80)
    x = rng.normal(size=(B, 9)); y = rng.normal(size=B)
    _check_model_paths(model, (x,), y, n, net, 0.2, dev)


def test_fixed_frequency_quanonet_40_2_40_2(dev):
    """FF QuanONet at the paper's PDE shape: Q5, Net40-2-40-2, 100 sensors, 2 coordinates, scale 0.1; only the bias
    and the 2400 circuit angles are trainable."""
    from quanonet_amd.models import QuanONetPT
    torch.manual_seed(9)
    n, net, B = 5, (40, 2, 40, 2), 19
    model = QuanONetPT(n, 100, 2, net, scale_coeff=0.1, if_trainable_freq=False).to(dev)
    assert [k for k, _ in model.named_parameters()] == ['bias', 'quantum_layer.ansatz_weights']
    rng = np.random.default_rng(90)
    with torch.no_grad():
        model.bias.fill_(0.05)
    br = rng.normal(size=(B, 100)); tr = rng.uniform(size=(B, 2)); y = rng.normal(scale=0.5, size=B)
    _check_model_paths(model, (br, tr), y, n, net, 0.1, dev)


@pytest.mark.parametrize('n,net', [(2, (3, 1, 2, 2)), (6, (2, 1, 2, 1)), (10, (1, 1, 1, 1))])
def test_fixed_frequency_quanonet_other_kernels(dev, n, net):
    from quanonet_amd.models import QuanONetPT
    torch.manual_seed(10 + n)
    B = 11
    model = QuanONetPT(n, 7, 2, net, scale_coeff=0.3, if_trainable_freq=False).to(dev)
    rng = np.random.default_rng(100 + n)
    br = rng.normal(size=(B, 7)); tr = rng.uniform(size=(B, 2)); y = rng.normal(size=B)
    _check_model_paths(model, (br, tr), y, n, net, 0.3, dev, batch_total=3 * B)


# ------------------------------------------------------------------------------------------------------------
# compare_backends.py, the four quantum cases
# ------------------------------------------------------------------------------------------------------------
def _compare_backends_stream():
    """The inputs each case draws from the module-level RNG (compare_backends.py:51) when __main__ runs the cases
    in order (:644-671).  data/Antideriv/... is not shipped, so case 3 takes its random branch (:327-329)."""
    rng = np.random.default_rng(0)
    f32 = np.float32
    s = {}
    s['quanonet_pt'] = dict(branch=rng.random((6, 8)).astype(f32), trunk=rng.random((6, 1)).astype(f32),
                            tgt=rng.random((6, 1)).astype(f32))
    s['heaqnn_pt'] = dict(x=rng.random((6, 6)).astype(f32), tgt=rng.random((6, 1)).astype(f32))
    s['quanonet_ms'] = dict(branch=rng.random((16, 10)).astype(f32), trunk=rng.random((16, 1)).astype(f32),
                            tgt=rng.random((16, 1)).astype(f32))
    s['heaqnn_ms'] = dict(x=rng.random((8, 6)).astype(f32), tgt=rng.random((8, 1)).astype(f32))
    return s


def _float32_case(model32, ins, tgt, dev):
    """The reference's procedure on a float32 module: forward under no_grad, then mean((out - tgt)^2).backward()."""
    tin = [torch.tensor(a, device=dev) for a in ins]
    model32.eval()
    with torch.no_grad():
        out = model32(*tin).cpu().numpy()
    model32.zero_grad()
    ((model32(*tin) - torch.tensor(tgt, device=dev)) ** 2).mean().backward()
    return out, {k: p.grad.detach().cpu().numpy() for k, p in model32.named_parameters()}


def test_compare_backends_quanonet_pt_case(dev):
    """compare_backends.py:140-212: Q2, net (2,1,2,1), b_in 8, t_in 1, batch 6, TF, scale 0.1, torch.manual_seed(42)."""
    from quanonet_amd.models import QuanONetPT
    d = _compare_backends_stream()['quanonet_pt']
    cfg = dict(num_qubits=2, branch_input_size=8, trunk_input_size=1, net_size=(2, 1, 2, 1), scale_coeff=0.1,
               if_trainable_freq=True, ham_bound=(-5.0, 5.0))
    torch.manual_seed(42)
    m32 = QuanONetPT(**cfg, dtype=torch.float32).to(dev)
    torch.manual_seed(42)
    w_ref = torch.empty(4, 3, 2).uniform_(-np.pi, np.pi)           # core/quantum_circuits_tq.py:50-53 under seed 42
    assert torch.equal(m32.quantum_layer.ansatz_weights.detach().cpu(), w_ref)
    out, grads = _float32_case(m32, (d['branch'], d['trunk']), d['tgt'], dev)
    rl, rg, ro = _oracle(m32, (d['branch'], d['trunk']), d['tgt'][:, 0], 2, (2, 1, 2, 1), 0.1)
    np.testing.assert_allclose(out[:, 0], ro, rtol=0, atol=ATOL_PT)
    np.testing.assert_allclose(grads['quantum_layer.ansatz_weights'], rg['quantum_layer.ansatz_weights'], rtol=0,
                               atol=ATOL_GRAD_PT)
    torch.manual_seed(42)
    m64 = QuanONetPT(**cfg).to(dev)
    _check_model_paths(m64, (d['branch'], d['trunk']), d['tgt'][:, 0].astype(np.float64), 2, (2, 1, 2, 1), 0.1, dev)


def test_compare_backends_heaqnn_pt_case(dev):
    """compare_backends.py:219-281: Q2, net (2,1,0,0), 6 inputs, batch 6, TF, scale 0.1, torch.manual_seed(42)."""
    from quanonet_amd.models import HEAQNNPT
    d = _compare_backends_stream()['heaqnn_pt']
    cfg = dict(num_qubits=2, input_size=6, net_size=(2, 1, 0, 0), scale_coeff=0.1, if_trainable_freq=True,
               ham_bound=(-5.0, 5.0))
    torch.manual_seed(42)
    m32 = HEAQNNPT(**cfg, dtype=torch.float32).to(dev)
    out, grads = _float32_case(m32, (d['x'],), d['tgt'], dev)
    rl, rg, ro = _oracle(m32, (d['x'],), d['tgt'][:, 0], 2, (2, 1), 0.1)
    np.testing.assert_allclose(out[:, 0], ro, rtol=0, atol=ATOL_PT)
    np.testing.assert_allclose(grads['quantum_layer.ansatz_weights'], rg['quantum_layer.ansatz_weights'], rtol=0,
                               atol=ATOL_GRAD_PT)
    torch.manual_seed(42)
    m64 = HEAQNNPT(**cfg).to(dev)
    _check_model_paths(m64, (d['x'],), d['tgt'][:, 0].astype(np.float64), 2, (2, 1), 0.1, dev)


def test_compare_backends_quanonet_pretrained_antideriv_case(dev):
    """compare_backends.py:288-376: the shipped Antideriv Q2 npz through the MS->PT key map, 16 rows; gradients of the
    circuit weights AND of branch_freq.weights / trunk_freq.weights (:366-376)."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.checkpoint import ms_to_pt_state
    d = _compare_backends_stream()['quanonet_ms']
    st = dict(np.load(H.GOLDEN + '/antideriv_q2.npz'))
    cfg = dict(num_qubits=2, branch_input_size=10, trunk_input_size=1, net_size=(5, 1, 5, 1), scale_coeff=0.001,
               if_trainable_freq=True, ham_bound=(-5.0, 5.0))
    m32 = QuanONetPT(**cfg, dtype=torch.float32)
    m32.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in ms_to_pt_state(st, 2, (5, 1, 5, 1)).items()})
    m32 = m32.to(dev)
    out, grads = _float32_case(m32, (d['branch'], d['trunk']), d['tgt'], dev)
    rl, rg, ro = _oracle(m32, (d['branch'], d['trunk']), d['tgt'][:, 0], 2, (5, 1, 5, 1), 0.001)
    np.testing.assert_allclose(out[:, 0], ro, rtol=0, atol=ATOL_MSPT)
    for k in ('quantum_layer.ansatz_weights', 'branch_freq.weights', 'trunk_freq.weights'):
        np.testing.assert_allclose(grads[k], rg[k], rtol=0, atol=ATOL_GRAD_MS, err_msg=k)
    m64 = QuanONetPT(**cfg)
    m64.load_state_dict({k: torch.tensor(v) for k, v in ms_to_pt_state(st, 2, (5, 1, 5, 1)).items()})
    m64 = m64.to(dev)
    _check_model_paths(m64, (d['branch'], d['trunk']), d['tgt'][:, 0].astype(np.float64), 2, (5, 1, 5, 1), 0.001, dev)


def test_compare_backends_heaqnn_ms_case(dev):
    """compare_backends.py:383-449: Q2, 6 inputs, depth 2 x 1, batch 8, gradients of the circuit and of freq.weights.
    The reference copies a random MindSpore initialisation (circuit U(-pi,pi), frequency bias U(-pi,pi):
    core/layers.py:24-27) into the PT module; MindSpore's generator is not available, so the same distributions are
    drawn from numpy here (the case's shapes, inputs and tolerances are the reference's)."""
    from quanonet_amd.models import HEAQNNPT
    d = _compare_backends_stream()['heaqnn_ms']
    rng = np.random.default_rng(383)
    sd = {'freq.weights': np.full(4, 0.1), 'freq.bias': rng.uniform(-np.pi, np.pi, 4),
          'quantum_layer.ansatz_weights': rng.uniform(-np.pi, np.pi, (2, 3, 2))}
    cfg = dict(num_qubits=2, input_size=6, net_size=(2, 1, 0, 0), scale_coeff=0.1, if_trainable_freq=True,
               ham_bound=(-5.0, 5.0))
    m32 = HEAQNNPT(**cfg, dtype=torch.float32)
    m32.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in sd.items()})
    m32 = m32.to(dev)
    out, grads = _float32_case(m32, (d['x'],), d['tgt'], dev)
    rl, rg, ro = _oracle(m32, (d['x'],), d['tgt'][:, 0], 2, (2, 1), 0.1)
    np.testing.assert_allclose(out[:, 0], ro, rtol=0, atol=ATOL_MSPT)
    for k in ('quantum_layer.ansatz_weights', 'freq.weights'):
        np.testing.assert_allclose(grads[k], rg[k], rtol=0, atol=ATOL_GRAD_MS, err_msg=k)
    m64 = HEAQNNPT(**cfg)
    m64.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    m64 = m64.to(dev)
    _check_model_paths(m64, (d['x'],), d['tgt'][:, 0].astype(np.float64), 2, (2, 1), 0.1, dev)


# ------------------------------------------------------------------------------------------------------------
# K1 / K2 on the GPU (ibm_inference.py:176-187)
# ------------------------------------------------------------------------------------------------------------
def test_known_answers_k1_k2_on_gpu(dev):
    from quanonet_amd.models import QuanONetPT
    p = H.load_pt_params('antideriv_q2.npz', 2, (5, 1, 5, 1))
    model = QuanONetPT(2, 10, 1, (5, 1, 5, 1), scale_coeff=0.001, if_trainable_freq=True)
    model.load_state_dict({k: torch.tensor(v) for k, v in p.items()})
    model = model.to(dev).eval()
    trunk = np.linspace(0, 1, 100)[:, None]
    ka = H.known_answers()
    for key, bv, truth in [('K1', np.cos(np.pi * np.linspace(0, 1, 10)), np.sin(np.pi * trunk[:, 0]) / np.pi),
                           ('K2', np.linspace(0, 1, 10), 0.5 * trunk[:, 0] ** 2)]:
        branch = np.tile(bv, (100, 1))
        with torch.no_grad():
            out = model(_t(branch, dev), _t(trunk, dev))[:, 0].cpu().numpy()
        rel = np.linalg.norm(out - truth) / np.linalg.norm(truth)
        assert rel < ka[key]['rel_l2_max'], key
        assert abs(rel - ka[key]['survey_rel_l2']) < 2e-3, key
        np.testing.assert_allclose(out, O.quanonet_forward(p, branch, trunk, 2, (5, 1, 5, 1)), rtol=0, atol=TOL)


def test_readme_demo_k9_on_gpu(dev):
    """README.md:137-155 on the HIP path: 100 000 forward evaluations of the shipped Q2 checkpoint (PTSolver.predict's
    batched qhea_model_forward) on the demo's test set; figures as in tests/test_oracle_golden.py::test_k9..."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import regression_metrics
    ka = H.known_answers()['K9']
    p = H.load_pt_params('antideriv_q2.npz', 2, (5, 1, 5, 1))
    model = QuanONetPT(2, 10, 1, (5, 1, 5, 1), scale_coeff=0.001, if_trainable_freq=True)
    model.load_state_dict({k: torch.tensor(v) for k, v in p.items()})
    model = model.to(dev).eval()
    d = np.load(H.GOLDEN + '/antideriv_demo.npz')
    branch = np.repeat(d['u0'].astype(np.float64), 100, axis=0)
    trunk = np.tile(d['x'].astype(np.float64), 1000)[:, None]
    y = d['u'].astype(np.float64).reshape(-1, 1)
    with torch.no_grad():
        outs = [model(_t(branch[s:s + 20000], dev), _t(trunk[s:s + 20000], dev)) for s in range(0, 100000, 20000)]
    pred = torch.cat(outs, dim=0)
    m = regression_metrics(pred, _t(y, dev))
    assert abs(m['rel_l2'] - ka['seed0']['rel_l2']) < 5e-4 and abs(m['MSE'] - ka['seed0']['mse']) < 5e-6
    assert abs(m['MAE'] - ka['seed0']['mae']) < 5e-5
    assert abs(m['rel_l2'] - ka['readme']['rel_l2']) < 0.1 * ka['readme']['rel_l2']
