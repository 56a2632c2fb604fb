"""
GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden fixtures.  Tolerance: 1e-10 absolute on expectation values and gradients (north_star).
"""
import numpy as np
import pytest
import torch

from oracle import hea_oracle as O
from oracle import c_oracle as C
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device('cuda:0')


def _t(a, dev):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)


def _run(n, cfgs, x, w, g, dev, off, co, diag=None, use_state=True, pauli='Z'):
    from quanonet_amd import _lib
    sh = _lib.CircuitShape(n, cfgs)
    xd, wd, gd = _t(x, dev), _t(w, dev), _t(g, dev)
    dd = None if diag is None else _t(diag, dev)
    out, st = _lib.hea_forward(sh, xd, wd, off, co, dd, return_state=True, ham_pauli=pauli)
    gx, gw, out2 = _lib.hea_backward(sh, xd, wd, gd, off, co, dd, state=st if use_state else None, want_out=True,
                                     ham_pauli=pauli)
    torch.cuda.synchronize()
    return (out.cpu().numpy(), st.cpu().numpy(), gx.cpu().numpy(), gw.cpu().numpy(), out2.cpu().numpy())


@pytest.fixture(params=['packed', 'pair', 'tri', 'ztri', 'zpacked', 'zquad'])
def backward_variant(request, dev):
    """The backward kernels for n <= 5 (one wave per sample group, the psi-wave / lambda-wave pipeline, the
    psi / lambda / sigma three-wave pipeline, and that pipeline on the ZYZ form of the gates -- 'ztri', which with
    its forward kernel is what the default 'auto' runs), forced through qhea_set_backward_variant; afterwards the status word
    must be clean (no hand-off overrun) and the choice goes back to automatic."""
    from quanonet_amd import _lib
    _lib.set_backward_variant(request.param)
    yield request.param
    _lib.set_backward_variant('auto')
    _lib.check_status(dev)


def test_golden_vectors(dev, backward_variant):
    for nm, v in H.golden_vectors().items():
        off, co = O.ham_params(v['n'])
        for use_state in (True, False):
            out, st, gx, gw, out2 = _run(v['n'], v['cfgs'], v['x'], v['w'], v['g'], dev, off, co,
                                         use_state=use_state)
            np.testing.assert_allclose(out, v['out'], rtol=0, atol=TOL, err_msg=nm)
            np.testing.assert_allclose(out2, v['out'], rtol=0, atol=TOL, err_msg=nm)
            np.testing.assert_allclose(gx, v['grad_x'], rtol=0, atol=TOL, err_msg=nm)
            np.testing.assert_allclose(gw, v['grad_w'], rtol=0, atol=TOL, err_msg=nm)


@pytest.mark.parametrize('n', list(range(2, 13)))
def test_every_qubit_count_against_oracle(dev, n, backward_variant):
    if n > 5 and backward_variant != 'packed':
        pytest.skip("the pipelined kernels exist for n <= 5 only")
    rng = np.random.default_rng(100 + n)
    cfgs = [(n, 2), (n, 1), (n, 2)]
    E, blk = O.circuit_sizes(n, cfgs)
    spw = max(1, 64 >> n)
    B = {True: 3 * spw * 4 + 5, False: 7}[n < 9]         # ragged: not a multiple of samples/wave x waves/WG
    x = rng.uniform(-np.pi, np.pi, (B, E))
    w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
    g = rng.normal(size=B)
    off, co = O.ham_params(n, -3.0, 7.0)
    ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True)
    _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
    for use_state in (True, False):
        out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=use_state)
        np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(st, rst, rtol=0, atol=TOL)
        np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
        np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL)


@pytest.mark.parametrize('n', [10, 11, 12])
def test_lds_resident_kernels_against_oracle(dev, n):
    """hea_lds.hip (the kernels for n >= 10): ragged encodings (more and fewer than n per block), a block without
    sub-layers and one without encodings included."""
    rng = np.random.default_rng(300 + n)
    cfgs = [(n, 2), (n - 3, 1), (n + 3, 2), (n, 0), (0, 1)]
    E, blk = O.circuit_sizes(n, cfgs)
    B = 5
    x = rng.uniform(-np.pi, np.pi, (B, E))
    w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
    g = rng.normal(size=B)
    off, co = O.ham_params(n, -3.0, 7.0)
    ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True)
    _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
    for use_state in (True, False):
        out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=use_state)
        np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(st, rst, rtol=0, atol=TOL)
        np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
        np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL)


@pytest.mark.parametrize('n', list(range(2, 13)))
def test_random_block_shapes(dev, n, backward_variant):
    """Seeded random block lists: encodings per block from {0, 1, n-1, n, n+1, 2n+1} (none, ragged, exactly the
    wires, more than one RX layer), 0..3 sub-layers, ragged batches.  Exercises the RX fold of the n >= 6 kernels
    (folded / not folded / folded + extra RX layers / blocks without sub-layers) against the gate-by-gate oracle."""
    if n > 5 and backward_variant != 'packed':
        pytest.skip("the pipelined kernels exist for n <= 5 only")
    rng = np.random.default_rng(9000 + n)
    for trial in range(4):
        nb = int(rng.integers(1, 6))
        cfgs = [(int(rng.choice([0, 1, n - 1, n, n + 1, 2 * n + 1])), int(rng.integers(0, 4))) for _ in range(nb)]
        E, blk = O.circuit_sizes(n, cfgs)
        if E == 0 and blk == 0:
            cfgs.append((n, 1))
            E, blk = O.circuit_sizes(n, cfgs)
        B = int(rng.integers(1, 10))
        x = rng.uniform(-np.pi, np.pi, (B, E))
        w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
        g = rng.normal(size=B)
        off, co = O.ham_params(n, -1.0, 3.0)
        ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True)
        _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
        for use_state in (True, False):
            out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=use_state)
            msg = f"n={n} cfgs={cfgs} B={B} use_state={use_state}"
            np.testing.assert_allclose(out, ro, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(st, rst, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL, err_msg=msg)


@pytest.mark.parametrize('ld', [1, 2])
@pytest.mark.parametrize('n', [2, 3, 4, 5])
def test_uniform_block_shapes(dev, n, ld, backward_variant):
    """Circuits whose blocks all have one RX chunk (0 < enc <= n) and the same number of sub-layers (what the reference
    builds: ld = 2 in every script, 1 in the shipped Q2 checkpoint) take the block-unrolled walk of hea_zyz.hpp under
    'ztri' / 'auto'; runs of different enc, a single block, ragged batches, with and without the saved final state."""
    rng = np.random.default_rng(4000 + 10 * n + ld)
    for cfgs, B in [([(n, ld)] * 3 + [(n - 1, ld)] * 2 + [(1, ld)], 2 * max(1, 64 >> n) + 1), ([(n, ld)], 3),
                    ([(1, ld)] * 2 + [(n, ld)] * 5, 70)]:
        E, blk = O.circuit_sizes(n, cfgs)
        x = rng.uniform(-np.pi, np.pi, (B, E))
        w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
        g = rng.normal(size=B)
        off, co = O.ham_params(n, -2.0, 5.0)
        ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True)
        _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
        for use_state in (True, False):
            out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=use_state)
            msg = f"n={n} cfgs={cfgs} B={B} use_state={use_state}"
            np.testing.assert_allclose(out, ro, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(st, rst, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL, err_msg=msg)
            np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL, err_msg=msg)


@pytest.mark.parametrize('ld', [1, 2])
def test_two_pipelines_per_workgroup(dev, ld):
    """'ztri2' at n = 5 batches of more sample groups than CUs: the pipelined backward kernel runs two groups per
    workgroup and adds their gradient sums in LDS (hea_zyz.hpp, PIPES = 2); an odd group count leaves the last
    workgroup's second pipeline without samples.  The forward phase sweeps in the split layout."""
    from quanonet_amd import _lib
    n = 5
    rng = np.random.default_rng(5150 + ld)
    _lib.set_backward_variant('ztri2')
    try:
        for cfgs, B in [([(n, ld)] * 4, 2 * 259), ([(n, ld)] * 3, 2 * 258 + 1)]:
            E, blk = O.circuit_sizes(n, cfgs)
            x = rng.uniform(-np.pi, np.pi, (B, E))
            w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
            g = rng.normal(size=B)
            off, co = O.ham_params(n, -2.0, 5.0)
            ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
            out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=False)
            np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
            np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL)
            np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
            np.testing.assert_allclose(gw, rgw, rtol=0, atol=1e-9)      # sums over 500+ samples
    finally:
        _lib.set_backward_variant('auto')
        _lib.check_status(dev)


@pytest.mark.parametrize('B', [513, 640, 1100, 1537, 2100])
def test_auto_dispatch_across_batch_regimes(dev, B):
    """n = 5, block-unrolled shape, default (AUTO) dispatch at the batch sizes where it changes kernels or occupancy:
    one pipeline workgroup per CU (<= 512), two per CU (513 ... 1024), a third that LDS does not admit (... 1536), the
    one-wave ZYZ kernel beyond; odd batches leave a half-filled last sample group.  Forward through the split-layout
    kernel up to one sample per SIMD, the all-lane kernels beyond."""
    n, cfgs = 5, [(5, 2)] * 3
    rng = np.random.default_rng(9000 + B)
    E, blk = O.circuit_sizes(n, cfgs)
    x = rng.uniform(-np.pi, np.pi, (B, E))
    w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
    g = rng.normal(size=B)
    off, co = O.ham_params(n, -2.0, 5.0)
    ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co)
    out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=False)
    np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
    np.testing.assert_allclose(gw, rgw, rtol=0, atol=1e-9)          # sums over up to 2100 samples
    from quanonet_amd import _lib
    _lib.check_status(dev)


def test_ham_diag_readout(dev):
    n, cfgs = 4, [(4, 1), (4, 2)]
    rng = np.random.default_rng(7)
    E, blk = O.circuit_sizes(n, cfgs)
    x = rng.uniform(-3, 3, (9, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=9)
    d = rng.normal(size=16)
    ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, 0.0, 0.0, ham_diag=d)
    out, st, gx, gw, _ = _run(n, cfgs, x, w, g, dev, 0.0, 0.0, diag=d)
    np.testing.assert_allclose(out, ro, atol=TOL)
    np.testing.assert_allclose(gx, rgx, atol=TOL)
    np.testing.assert_allclose(gw, rgw, atol=TOL)


@pytest.mark.parametrize('pauli', ['X', 'Y'])
@pytest.mark.parametrize('n', [2, 4, 5, 6, 8, 10, 12])
def test_pauli_xy_readout(dev, n, pauli, backward_variant):
    """H = offset + coeff * sum_i P_i for P = X, Y (generate_simple_hamiltonian's `pauli`,
    core/quantum_circuits_ms.py:28-39).  The oracle applies the Paulis one by one; the kernels rotate the
    measurement basis.  state_out stays psi_N (before that rotation)."""
    if n > 5 and backward_variant != 'packed':
        pytest.skip("the pipelined kernels exist for n <= 5 only")
    rng = np.random.default_rng(500 + n)
    cfgs = [(n, 2), (n + 1, 1)]
    E, blk = O.circuit_sizes(n, cfgs)
    B = 7 if n >= 9 else 3 * max(1, 64 >> n) + 2
    x = rng.uniform(-np.pi, np.pi, (B, E))
    w = rng.uniform(-np.pi, np.pi, (blk, 3, n))
    g = rng.normal(size=B)
    off, co = O.ham_params(n, -2.0, 6.0)
    ro, rst = C.hea_forward(n, cfgs, x, w, off, co, return_state=True, ham_pauli=pauli)
    _, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, off, co, ham_pauli=pauli)
    for use_state in (True, False):
        out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, off, co, use_state=use_state, pauli=pauli)
        np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(st, rst, rtol=0, atol=TOL)
        np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
        np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL)


@pytest.mark.parametrize('pauli', ['X', 'Y'])
def test_pauli_xy_lds_kernels_and_model_path(dev, pauli):
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    # fused model path + autograd module path, n = 5
    torch.manual_seed(11)
    n, net, b_in, t_in, B = 5, (3, 2, 2, 1), 6, 2, 37
    model = QuanONetPT(n, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=True, ham_pauli=pauli).to(dev)
    rng = np.random.default_rng(17)
    br = rng.normal(size=(B, b_in)); tr = rng.uniform(size=(B, t_in)); y = rng.normal(size=B)
    with torch.no_grad():
        model.bias.fill_(-0.2)
    fused = DataParallelTrainer(model, lr=1e-3, fused=True)
    flat = fused.loss_and_grad(_t(br, dev), _t(tr, dev), _t(y, dev)).clone()
    pred = _lib.model_forward(fused.desc, _t(br, dev), _t(tr, dev), fused.pflat).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rl, rg, ro = O.quanonet_loss_and_grads(sd, br, tr, y, n, net, ham_pauli=pauli)
    ref = np.concatenate([rg[k].reshape(-1) for k, _ in model.named_parameters()])
    np.testing.assert_allclose(pred, ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(flat[:-2].cpu().numpy(), ref, rtol=0, atol=TOL)
    auto = DataParallelTrainer(model, lr=1e-3, fused=False)
    flat2 = auto.loss_and_grad(_t(br, dev), _t(tr, dev), _t(y, dev).unsqueeze(-1))
    np.testing.assert_allclose(flat.cpu().numpy(), flat2.cpu().numpy(), rtol=0, atol=TOL)
    # workgroup-resident kernels (n >= 10)
    n, cfgs = 10, [(10, 2), (13, 1)]
    E, blk = O.circuit_sizes(n, cfgs)
    x = rng.uniform(-3, 3, (4, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=4)
    ro, rgx, rgw = C.hea_backward(n, cfgs, x, w, g, 0.4, 1.3, ham_pauli=pauli)
    for use_state in (True, False):
        out, st, gx, gw, out2 = _run(n, cfgs, x, w, g, dev, 0.4, 1.3, use_state=use_state, pauli=pauli)
        np.testing.assert_allclose(out, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(out2, ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(gx, rgx, rtol=0, atol=TOL)
        np.testing.assert_allclose(gw, rgw, rtol=0, atol=TOL)
    # X/Y together with a diagonal Hamiltonian is an argument error
    sh = _lib.CircuitShape(n, cfgs)
    with pytest.raises(_lib.QheaError):
        _lib.hea_forward(sh, _t(x, dev), _t(w, dev), 0.0, 1.0, _t(np.ones(1 << n), dev), ham_pauli=pauli)


def test_edge_shapes(dev, backward_variant):
    from quanonet_amd import _lib
    # batch 1, a block with no ansatz sub-layer, a block with no encoding, more encodings than wires
    for n, cfgs, B in [(5, [(5, 2)], 1), (3, [(3, 0), (3, 1)], 4), (4, [(0, 2), (4, 1)], 5), (2, [(5, 1), (3, 2)], 6)]:
        rng = np.random.default_rng(B)
        E, blk = O.circuit_sizes(n, cfgs)
        x = rng.uniform(-3, 3, (B, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=B)
        off, co = O.ham_params(n)
        ro, rgx, rgw = O.hea_backward(n, cfgs, x, w, g, off, co)
        out, st, gx, gw, _ = _run(n, cfgs, x, w, g, dev, off, co)
        np.testing.assert_allclose(out, ro, atol=TOL)
        np.testing.assert_allclose(gx, rgx, atol=TOL)
        np.testing.assert_allclose(gw, rgw, atol=TOL)
    # empty batch: returns without touching memory, grad_w zeroed
    sh = _lib.CircuitShape(3, [(3, 1)])
    gx, gw = _lib.hea_backward(sh, torch.empty(0, 3, dtype=torch.float64, device=dev),
                               torch.zeros(1, 3, 3, dtype=torch.float64, device=dev),
                               torch.empty(0, dtype=torch.float64, device=dev), 0.0, 1.0)
    assert gx.shape == (0, 3) and float(gw.abs().sum()) == 0.0


def test_cpu_tensor_is_rejected():
    from quanonet_amd import _lib
    sh = _lib.CircuitShape(2, [(2, 1)])
    with pytest.raises(_lib.QheaError):
        _lib.hea_forward(sh, torch.zeros(1, 2, dtype=torch.float64), torch.zeros(1, 3, 2, dtype=torch.float64), 0.0, 1.0)


@pytest.mark.parametrize('key,op,tag,npts', H.PDE_CASES)
def test_known_answers_k3_k8_on_gpu(dev, key, op, tag, npts):
    from quanonet_amd.models import QuanONetPT
    ka = H.known_answers()[key]
    p = H.load_pt_params(f'{op}_q5.npz', 5, (40, 2, 20, 2))
    model = QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True)
    model.load_state_dict({k: torch.tensor(v) for k, v in p.items()})
    model = model.to(dev).eval()
    branch, trunk = H.notebook_inputs(npts, H.U0[tag])
    with torch.no_grad():
        out = model(torch.tensor(branch, device=dev, dtype=torch.float64),
                    torch.tensor(trunk, device=dev, dtype=torch.float64))[:, 0].cpu().numpy()
    truth = np.load(H.GOLDEN + '/pde_truths.npz')[f'{op}_{tag}']
    diff = truth - out.reshape(npts, npts)
    assert H.fmt1e(np.mean(diff ** 2)) == ka['mse']
    assert H.fmt1e(np.mean(np.abs(diff))) == ka['mae']


def test_model_grads_match_oracle(dev):
    from quanonet_amd.models import QuanONetPT, HEAQNNPT
    torch.manual_seed(1)
    n, net = 5, (3, 2, 2, 1)
    model = QuanONetPT(n, 10, 2, net, scale_coeff=0.1, if_trainable_freq=True).to(dev)
    rng = np.random.default_rng(2)
    B = 50
    br = rng.normal(size=(B, 10)); tr = rng.uniform(size=(B, 2)); y = rng.normal(size=B)
    out = model(_t(br, dev), _t(tr, dev))
    loss = torch.nn.functional.mse_loss(out, _t(y, dev).unsqueeze(-1))
    loss.backward()
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rl, rg, ro = O.quanonet_loss_and_grads(params, br, tr, y, n, net)
    assert abs(loss.item() - rl) < TOL
    for k, prm in model.named_parameters():
        np.testing.assert_allclose(prm.grad.cpu().numpy().reshape(-1), rg[k].reshape(-1), rtol=0, atol=TOL, err_msg=k)
    # HEAQNN forward only (no reference checkpoint exists: parity unpinned beyond the oracle)
    hm = HEAQNNPT(4, 7, (3, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev)
    xin = rng.normal(size=(9, 7))
    o = hm(_t(xin, dev))[:, 0].detach().cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in hm.state_dict().items()}
    xe = O.tiled_elementwise(xin, sd['freq.weights'], sd['freq.bias'])
    ref = O.hea_forward(4, O.block_configs_heaqnn(4, (3, 2)), xe, sd['quantum_layer.ansatz_weights'], *O.ham_params(4))
    np.testing.assert_allclose(o, ref, atol=TOL)


@pytest.mark.parametrize('n,net,tf', [(5, (3, 2, 2, 1), True), (2, (5, 1, 5, 1), True), (5, (2, 2, 2, 2), False),
                                      (8, (2, 1, 1, 2), True)])
def test_fused_model_path_matches_oracle_and_autograd(dev, n, net, tf, backward_variant):
    """qhea_model_loss_grad / qhea_model_forward == oracle == the autograd module path."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    torch.manual_seed(3)
    b_in, t_in, B = 7, 2, 45
    model = QuanONetPT(n, b_in, t_in, net, scale_coeff=0.1, if_trainable_freq=tf).to(dev)
    rng = np.random.default_rng(n)
    br = rng.normal(size=(B, b_in)); tr = rng.uniform(size=(B, t_in)); y = rng.normal(size=B)
    if tf:
        with torch.no_grad():
            model.branch_freq.bias.copy_(_t(rng.normal(size=model.branch_freq.bias.shape), dev))
            model.trunk_freq.bias.copy_(_t(rng.normal(size=model.trunk_freq.bias.shape), dev))
            model.bias.fill_(0.3)
    fused = DataParallelTrainer(model, lr=1e-3, fused=True)
    flat = fused.loss_and_grad(_t(br, dev), _t(tr, dev), _t(y, dev), global_batch=2 * B).clone()
    pred = _lib.model_forward(fused.desc, _t(br, dev), _t(tr, dev), fused.pflat).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    # trainable-frequency and fixed-frequency (_ScaleRepeat, core/models_pt.py:44-68) forms both against the oracle
    rl, rg, ro = O.quanonet_loss_and_grads(sd, br, tr, y, n, net, batch_total=2 * B, scale_coeff=None if tf else 0.1)
    ref = np.concatenate([rg[k].reshape(-1) for k, _ in model.named_parameters()])
    np.testing.assert_allclose(pred, ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(flat[:-2].cpu().numpy(), ref, rtol=0, atol=TOL)
    assert abs(flat[-2].item() - rl * 2 * B) < 1e-9
    assert abs(flat[-1].item() - float((y ** 2).sum())) < 1e-9
    # autograd path on the same module
    auto = DataParallelTrainer(model, lr=1e-3, fused=False)
    flat2 = auto.loss_and_grad(_t(br, dev), _t(tr, dev), _t(y, dev).unsqueeze(-1), global_batch=2 * B)
    np.testing.assert_allclose(flat.cpu().numpy(), flat2.cpu().numpy(), rtol=0, atol=TOL)
    with torch.no_grad():
        p2 = model(_t(br, dev), _t(tr, dev))[:, 0].cpu().numpy()
    np.testing.assert_allclose(pred, p2, rtol=0, atol=TOL)


def test_fused_heaqnn_path(dev):
    from quanonet_amd.models import HEAQNNPT
    from quanonet_amd.solver import DataParallelTrainer
    torch.manual_seed(4)
    model = HEAQNNPT(5, 9, (3, 2), scale_coeff=0.1, if_trainable_freq=True).to(dev)
    rng = np.random.default_rng(9)
    x = rng.normal(size=(33, 9)); y = rng.normal(size=33)
    fused = DataParallelTrainer(model, fused=True)
    f1 = fused.loss_and_grad(_t(x, dev), _t(y, dev)).clone()
    auto = DataParallelTrainer(model, fused=False)
    f2 = auto.loss_and_grad(_t(x, dev), _t(y, dev).unsqueeze(-1))
    np.testing.assert_allclose(f1.cpu().numpy(), f2.cpu().numpy(), rtol=0, atol=TOL)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rl, rg, ro = O.heaqnn_loss_and_grads(sd, x, y, 5, (3, 2))
    ref = np.concatenate([rg[k].reshape(-1) for k, _ in model.named_parameters()])
    np.testing.assert_allclose(f1[:-2].cpu().numpy(), ref, rtol=0, atol=TOL)
    np.testing.assert_allclose(f2[:-2].cpu().numpy(), ref, rtol=0, atol=TOL)
    assert abs(f1[-2].item() - rl * 33) < 1e-9


def test_training_reduces_loss_and_steps_are_deterministic(dev):
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    rng = np.random.default_rng(0)
    br = _t(rng.normal(size=(256, 6)), dev); tr = _t(rng.uniform(size=(256, 1)), dev)
    y = torch.sin(3 * tr[:, 0]) * br[:, 0]
    losses = []
    for rep in range(2):
        torch.manual_seed(0)
        model = QuanONetPT(2, 6, 1, (3, 1, 3, 1), scale_coeff=0.1, if_trainable_freq=True).to(dev)
        t = DataParallelTrainer(model, lr=5e-2)
        hist = []
        for it in range(60):
            t.train_step(br, tr, y)
            if it % 10 == 0:
                hist.append(t.loss_scalars()[0])
        losses.append(hist)
    assert losses[0] == losses[1]                  # bitwise reproducible
    assert losses[0][-1] < 0.7 * losses[0][0]


def test_ptsolver_end_to_end(dev, tmp_path):
    """PTSolver (mirror of solvers/solver_pt.py): train, best/final checkpoints with the reference's keys, evaluate."""
    from quanonet_amd.solver import PTSolver
    rng = np.random.default_rng(0)
    ntr, nte, b_in, t_in = 600, 200, 8, 1
    def make(nrows):
        br = rng.normal(size=(nrows, b_in)); tr = rng.uniform(size=(nrows, t_in))
        y = (np.sin(2.0 * tr[:, 0]) * br[:, 0] * 0.5)[:, None]
        return br, tr, y
    br, tr, y = make(ntr); bre, tre, ye = make(nte)
    data = {'train_branch_input': br, 'train_trunk_input': tr, 'train_output': y,
            'test_branch_input': bre, 'test_trunk_input': tre, 'test_output': ye}
    cfg = {'model_type': 'QuanONet', 'operator': 'Synthetic', 'num_qubits': 3, 'net_size': [3, 1, 2, 1],
           'scale_coeff': 0.1, 'if_trainable_freq': 'true', 'learning_rate': 2e-2, 'batch_size': 100,
           'num_epochs': 12, 'prefix': str(tmp_path), 'run_id': 'run0'}
    np.random.seed(0); torch.manual_seed(0)
    s = PTSolver(cfg, data, device=dev, log=lambda *a, **k: None)
    hist = s.train()
    assert len(hist['loss_train']) == 12 and hist['loss_train'][-1] < 0.8 * hist['loss_train'][0]
    m = s.evaluate(hist)
    assert set(m) == {'MSE', 'MAE', 'Max_Error', 'rel_l2'} and np.isfinite(m['rel_l2'])
    out_dir = s.out_dir
    import os, json
    for f in ('best_model.pt', 'best_model.npz', 'final.pt', 'final.npz', 'metric.json'):
        assert os.path.exists(os.path.join(out_dir, f)), f
    z = np.load(os.path.join(out_dir, 'best_model.npz'))
    assert sorted(z.files) == sorted(['bias', 'branch_freq.weights', 'branch_freq.bias', 'trunk_freq.weights',
                                      'trunk_freq.bias', 'quantum_layer.ansatz_weights'])
    # the evaluation path (qhea_model_forward) equals the oracle on the saved best weights
    params = {k: z[k] for k in z.files}
    ref = O.quanonet_forward(params, bre, tre, 3, (3, 1, 2, 1))
    pred = s.predict(s.test_input)[:, 0].cpu().numpy()
    np.testing.assert_allclose(pred, ref, rtol=0, atol=TOL)
    assert abs(json.load(open(os.path.join(out_dir, 'metric.json')))['metrics']['MSE'] - m['MSE']) < 1e-15
    # reproducible run-to-run (same seeds -> same history, bitwise)
    np.random.seed(0); torch.manual_seed(0)
    cfg2 = dict(cfg, run_id='run1', if_save=False)
    h2 = PTSolver(cfg2, data, device=dev, log=lambda *a, **k: None).train()
    assert h2['loss_train'] == hist['loss_train']


def test_reference_checkpoint_loads_into_model(dev):
    """Weights shipped by the reference (decoded .ckpt fixture) -> QuanONetPT via checkpoint.ms_to_pt_state."""
    from quanonet_amd.checkpoint import ms_to_pt_state
    from quanonet_amd.models import QuanONetPT
    st = dict(np.load(H.GOLDEN + '/antideriv_q2.npz'))
    sd = ms_to_pt_state(st, 2, (5, 1, 5, 1))
    model = QuanONetPT(2, 10, 1, (5, 1, 5, 1), scale_coeff=0.001, if_trainable_freq=True)
    model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    trunk = np.linspace(0, 1, 100)[:, None]
    branch = np.tile(np.cos(np.pi * np.linspace(0, 1, 10)), (100, 1))
    with torch.no_grad():
        out = model(_t(branch, dev), _t(trunk, dev))[:, 0].cpu().numpy()
    truth = np.sin(np.pi * trunk[:, 0]) / np.pi                     # K1 (ibm_inference.py:180-183)
    assert np.linalg.norm(out - truth) / np.linalg.norm(truth) < 0.05


def test_flat_adam_equals_torch_adam(dev):
    from quanonet_amd.solver import FlatAdam
    rng = np.random.default_rng(1)
    n = 2401
    p0 = _t(rng.normal(size=n), dev)
    pa, pb = p0.clone(), p0.clone().requires_grad_(True)
    ga = torch.zeros(n + 2, dtype=torch.float64, device=dev)
    flat = FlatAdam([torch.nn.Parameter(pa)], pa, ga, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ref = torch.optim.Adam([pb], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for it in range(25):
        g = _t(rng.normal(size=n) * (1.0 + it), dev)
        ga[:n] = g
        pb.grad = g.clone()
        flat.step(); ref.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(pa.cpu().numpy(), pb.detach().cpu().numpy(), rtol=1e-13, atol=1e-15)
    sd = flat.state_dict()
    assert sd['flat_state']['step'] == 25


@pytest.mark.parametrize('kind', ['quanonet', 'quanonet_fixed_freq', 'heaqnn'])
def test_fused_train_step_is_bitwise_the_two_call_path(dev, kind):
    """qhea_model_train_step (Adam applied inside the reduce kernel) == qhea_model_loss_grad + qhea_adam_step."""
    from quanonet_amd.models import QuanONetPT, HEAQNNPT
    from quanonet_amd.solver import DataParallelTrainer
    rng = np.random.default_rng(77)
    B = 53

    def make():
        torch.manual_seed(5)
        if kind == 'heaqnn':
            return HEAQNNPT(4, 6, (3, 2), scale_coeff=0.2, if_trainable_freq=True).to(dev)
        return QuanONetPT(5, 7, 2, (3, 2, 2, 1), scale_coeff=0.1, if_trainable_freq=(kind == 'quanonet')).to(dev)

    if kind == 'heaqnn':
        data = [(_t(rng.normal(size=(B, 6)), dev), _t(rng.normal(size=B), dev)) for _ in range(6)]
    else:
        data = [(_t(rng.normal(size=(B, 7)), dev), _t(rng.uniform(size=(B, 2)), dev), _t(rng.normal(size=B), dev))
                for _ in range(6)]
    fused = DataParallelTrainer(make(), lr=3e-3, optimizer_kwargs={'weight_decay': 0.01})
    split = DataParallelTrainer(make(), lr=3e-3, optimizer_kwargs={'weight_decay': 0.01})
    assert torch.equal(fused.pflat, split.pflat)
    for batch in data:
        f1 = fused.train_step(*batch).clone()                       # world == 1 -> qhea_model_train_step
        split.loss_and_grad(*batch)
        f2 = split.flat.clone()
        split.optimizer.step()
        assert torch.equal(f1, f2)
        assert torch.equal(fused.pflat, split.pflat)
    assert torch.equal(fused.optimizer.exp_avg, split.optimizer.exp_avg)
    assert torch.equal(fused.optimizer.exp_avg_sq, split.optimizer.exp_avg_sq)
    assert fused.optimizer.t == split.optimizer.t == len(data)


def test_float32_module_like_the_reference(dev):
    """The reference keeps float32 parameters/inputs (solver_pt.py:132); the op computes in fp64 and casts back."""
    from quanonet_amd.models import QuanONetPT
    torch.manual_seed(5)
    n, net = 3, (2, 1, 2, 2)
    model = QuanONetPT(n, 6, 2, net, scale_coeff=0.1, if_trainable_freq=True, dtype=torch.float32).to(dev)
    rng = np.random.default_rng(8)
    br = rng.normal(size=(20, 6)).astype(np.float32); tr = rng.uniform(size=(20, 2)).astype(np.float32)
    y = rng.normal(size=(20, 1)).astype(np.float32)
    out = model(torch.tensor(br, device=dev), torch.tensor(tr, device=dev))
    assert out.dtype == torch.float32 and out.shape == (20, 1)
    loss = torch.nn.functional.mse_loss(out, torch.tensor(y, device=dev))
    loss.backward()
    params = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    rl, rg, ro = O.quanonet_loss_and_grads(params, br.astype(np.float64), tr.astype(np.float64), y[:, 0].astype(np.float64), n, net)
    np.testing.assert_allclose(out[:, 0].detach().cpu().numpy(), ro, rtol=0, atol=2e-5)     # float32 pre/post-processing
    for k, p in model.named_parameters():
        assert p.grad.dtype == torch.float32
        np.testing.assert_allclose(p.grad.cpu().numpy().reshape(-1), rg[k].reshape(-1), rtol=0, atol=2e-5, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize('nq,net,n_rows,bs', [
    (4, (3, 2, 2, 1), 230, 64),          # mixed sub-layer counts: the generic walk, a prep launch per step
    (5, (3, 2, 2, 2), 230, 64),          # block-unrolled shape, LD = 2: steps 2, 3 take their records from the previous reduce
    (5, (3, 1, 2, 1), 200, 50),          # LD = 1
    (3, (2, 2, 3, 2), 150, 50),          # n = 3 (pad(3n) = 16 columns per sub-layer)
    (2, (4, 2, 3, 2), 96, 32),           # n = 2 (8 columns per sub-layer)
    (2, (5, 1, 5, 1), 4 * 32 + 7, 32),   # n = 2, ONE sub-layer per block (the shipped Antideriv Q2 model, cfg 1): two circuit blocks per reduce block
    (2, (3, 1, 2, 1), 4 * 32 + 7, 32),   # ... with an odd number of blocks: not eligible, a prep launch per step
    (5, (2, 2, 2, 2), 3 * 2304 + 100, 2304),   # one-wave ZYZ kernel (batch beyond 3/4 of the SIMDs)
    (5, (2, 2, 2, 2), 3 * 1024 + 100, 1024),   # two pipelines per workgroup: 256 partial rows, four per row slice
    (5, (40, 2, 20, 2), 2 * 1024 + 100, 1024), # the headline's 60-block circuit at its batch
    (5, (40, 2, 20, 2), 2 * 512 + 100, 512),   # ... and at cfg 3's shard: the quad-chain pipeline
])
def test_train_steps_from_one_host_call_equal_the_single_steps_bitwise(nq, net, n_rows, bs):
    """qhea_model_train_steps (one epoch's inner loop from one host call; on block-unrolled shapes the reduce kernel of a
    step writes the next step's layer records instead of a prep launch) against a loop of qhea_model_train_step: uneven
    last batch; parameters, Adam state and every step's [grads | sse | sum y^2] row bitwise equal; then one more single
    step on both (the records the fused path left behind are not what the next call relies on)."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    branch = torch.tensor(rng.normal(size=(n_rows, 7)), device=dev)
    trunk = torch.tensor(rng.uniform(size=(n_rows, 2)), device=dev)
    y = torch.tensor(rng.normal(size=(n_rows, 1)), device=dev)
    bounds = list(range(0, n_rows, bs)) + [n_rows]
    gbs = [bounds[i + 1] - bounds[i] for i in range(len(bounds) - 1)]
    nst = len(gbs)

    def make():
        torch.manual_seed(11)
        return DataParallelTrainer(QuanONetPT(nq, 7, 2, net, scale_coeff=0.1).double().to(dev), lr=1e-2)
    a, b = make(), make()
    assert a.accepts_out
    rows_a = torch.zeros(nst, a.numel + 2, dtype=torch.float64, device=dev)
    rows_b = torch.zeros_like(rows_a)
    for i in range(nst):
        lo, hi = bounds[i], bounds[i + 1]
        a.train_step(branch[lo:hi], trunk[lo:hi], y[lo:hi], global_batch=gbs[i], out=rows_a[i])
    b.train_steps([branch, trunk], y, bounds, gbs, rows_b)
    torch.cuda.synchronize()
    assert torch.equal(rows_a, rows_b)
    assert torch.equal(a.pflat, b.pflat)
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg) and torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)
    assert a.optimizer.t == b.optimizer.t == nst
    assert float(rows_a[:, a.numel].min()) > 0.0                # every step reported its sse
    fa = a.train_step(branch[:bs], trunk[:bs], y[:bs]).clone()
    fb = b.train_step(branch[:bs], trunk[:bs], y[:bs]).clone()
    assert torch.equal(fa, fb) and torch.equal(a.pflat, b.pflat)
    a.check_status(); b.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize('nq,net', [(5, (3, 2, 2, 2)), (4, (3, 2, 2, 1)), (8, (2, 2, 2, 2))])
def test_forward_chunks_from_one_host_call_equal_the_single_calls_bitwise(nq, net):
    """qhea_model_forward_chunks (one record preparation for all equal-sized chunks) against chunk-by-chunk
    qhea_model_forward, with a shorter last chunk; and PTSolver-style predict on top of it."""
    from quanonet_amd.models import QuanONetPT
    from quanonet_amd.solver import DataParallelTrainer
    from quanonet_amd import _lib
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(9)
    n_rows, chunk = 700, 256
    branch = torch.tensor(rng.normal(size=(n_rows, 7)), device=dev)
    trunk = torch.tensor(rng.uniform(size=(n_rows, 2)), device=dev)
    torch.manual_seed(3)
    tr = DataParallelTrainer(QuanONetPT(nq, 7, 2, net, scale_coeff=0.1).double().to(dev))
    one = torch.cat([_lib.model_forward(tr.desc, branch[s:s + chunk], trunk[s:s + chunk], tr.pflat)
                     for s in range(0, n_rows, chunk)])
    many = _lib.model_forward_chunks(tr.desc, branch, trunk, tr.pflat, chunk)
    torch.cuda.synchronize()
    assert torch.equal(one, many)
    whole = _lib.model_forward(tr.desc, branch, trunk, tr.pflat)
    assert float((whole - many).abs().max()) < 1e-12


def test_module_mirrors_the_reference_column_guard(dev):
    """core/quantum_circuits_tq.py:83: an x narrower than E skips the missing encoding gates, a wider x is not read beyond
    E; HEACircuitHIP.forward does the same (zero angles / slice), forward and both gradients against the oracle."""
    from quanonet_amd.circuit import HEACircuitHIP
    n, cfgs = 4, O.block_configs_quanonet(4, (2, 2, 1, 1))
    E, blk = O.circuit_sizes(n, cfgs)
    off, co = O.ham_params(n)
    torch.manual_seed(2)
    layer = HEACircuitHIP(n, cfgs, ham_offset=off, ham_coeff_per_qubit=co).to(dev)
    w = layer.ansatz_weights.detach().cpu().numpy()
    rng = np.random.default_rng(3)
    g = rng.normal(size=9)
    for width in (E - 3, E + 2):
        x = rng.uniform(-np.pi, np.pi, (9, width))
        xt = _t(x, dev).requires_grad_(True)
        layer.zero_grad()
        out = layer(xt)
        (out[:, 0] * _t(g, dev)).sum().backward()
        ro, rgx, rgw = O.hea_backward(n, cfgs, x, w, g, off, co)
        np.testing.assert_allclose(out[:, 0].detach().cpu().numpy(), ro, rtol=0, atol=TOL)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), rgx, rtol=0, atol=TOL)
        np.testing.assert_allclose(layer.ansatz_weights.grad.cpu().numpy(), rgw, rtol=0, atol=TOL)
