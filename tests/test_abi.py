"""
CPU-side checks of the C-ABI shared library: it loads, exports every symbol that include/*.h declares,
and its host-side argument validation behaves (no kernel is launched, no GPU needed).
"""
import ctypes
import glob
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope='module')
def lib():
    from quanonet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):            # normally built by __graft_entry__.build()
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'quanonet_amd', 'csrc'), '-j', '8'])
    return _lib.load()


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = open(h).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        for m in re.finditer(r'\b(qhea_[a-z0-9_]+)\s*\(', src):
            names.add(m.group(1))
    return sorted(names)


def test_every_declared_symbol_is_exported(lib):
    from quanonet_amd import _lib
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert getattr(lib, n) is not None, n
    assert set(_lib.EXPORTS) == set(names)


def test_version_and_strerror(lib):
    assert lib.qhea_version() >= 300
    assert lib.qhea_strerror(0) == b'ok'
    assert b'invalid' in lib.qhea_strerror(-1)
    assert b'workspace' in lib.qhea_strerror(-3)


def test_shape_validation_without_gpu(lib):
    i32 = ctypes.c_int32
    enc = (i32 * 3)(5, 5, 5)
    ld = (i32 * 3)(2, 2, 1)
    assert lib.qhea_workspace_bytes(5, 3, enc, ld, 1024) > 1024 * 15 * 16
    assert lib.qhea_workspace_bytes(1, 3, enc, ld, 1024) == 0          # n < 2 rejected
    assert lib.qhea_workspace_bytes(13, 3, enc, ld, 1024) == 0
    bad = (i32 * 3)(5, -1, 5)
    assert lib.qhea_workspace_bytes(5, 3, bad, ld, 8) == 0
    # argument errors are reported before anything touches a device
    assert lib.qhea_forward(1, 3, enc, ld, 4, None, None, 0.0, 1.0, None, 0, None, None, None, 0, None) == -1
    assert lib.qhea_forward(5, 3, enc, ld, 4, None, None, 0.0, 1.0, None, 0, None, None, None, 0, None) == -1
    assert lib.qhea_backward(5, 3, enc, ld, -1, None, None, 0.0, 1.0, None, 0, None, None, None, None, None,
                             None, 0, None) == -1
    assert lib.qhea_forward(5, 3, enc, ld, 0, None, None, 0.0, 1.0, None, 0, None, None, None, 0, None) == 0   # empty batch
    # read-out Pauli: 0/1/2 = Z/X/Y; anything else, or X/Y together with a ham_diag pointer, is rejected
    assert lib.qhea_forward(5, 3, enc, ld, 0, None, None, 0.0, 1.0, None, 2, None, None, None, 0, None) == 0
    assert lib.qhea_forward(5, 3, enc, ld, 0, None, None, 0.0, 1.0, None, 3, None, None, None, 0, None) == -1
    assert lib.qhea_forward(5, 3, enc, ld, 0, None, None, 0.0, 1.0, ctypes.c_void_p(64), 1, None, None, None, 0,
                            None) == -1
    # more than 16 distinct (enc, ld) runs is outside this build's kernel-argument budget
    many_e = (i32 * 40)(*[5] * 40)
    many_l = (i32 * 40)(*[(i % 2) + 1 for i in range(40)])
    assert lib.qhea_workspace_bytes(5, 40, many_e, many_l, 8) == 0


def test_model_descriptor(lib):
    from quanonet_amd import _lib
    d = _lib.make_model_desc(_lib.MODEL_QUANONET, 5, (40, 2, 20, 2), 100, 2, True, 0.1, 0.0, 1.0)
    assert _lib.model_param_count(d) == 2401                     # SURVEY.md section 8 table, cfg 2
    d2 = _lib.make_model_desc(_lib.MODEL_QUANONET, 2, (5, 1, 5, 1), 10, 1, True, 0.001, 0.0, 2.5)
    assert _lib.model_param_count(d2) == 101                     # cfg 1
    d3 = _lib.make_model_desc(_lib.MODEL_HEAQNN, 8, (20, 2), 102, 0, True, 0.1, 0.0, 0.625)
    assert _lib.model_param_count(d3) == 1280                    # cfg 4
    d4 = _lib.make_model_desc(_lib.MODEL_QUANONET, 12, (40, 2, 20, 2), 100, 2, True, 0.1, 0.0, 1.0)
    assert _lib.model_param_count(d4) == 5761                    # cfg 5
    ff = _lib.make_model_desc(_lib.MODEL_QUANONET, 5, (40, 2, 40, 2), 100, 2, False, 0.1, 0.0, 1.0)
    assert _lib.model_param_count(ff) == 160 * 15 + 1
    assert lib.qhea_model_workspace_bytes(ctypes.byref(d), 1024) > 0
    bad = _lib.make_model_desc(7, 5, (1, 1, 1, 1), 3, 1, True, 0.1, 0.0, 1.0)
    with pytest.raises(_lib.QheaError):
        _lib.model_param_count(bad)
    dy = _lib.make_model_desc(_lib.MODEL_QUANONET, 5, (40, 2, 20, 2), 100, 2, True, 0.1, 0.0, 1.0, ham_pauli='Y')
    assert dy.ham_pauli == 2 and _lib.model_param_count(dy) == 2401
    dy.ham_pauli = 5
    with pytest.raises(_lib.QheaError):
        _lib.model_param_count(dy)
    with pytest.raises(ValueError):
        _lib.pauli_code('W')


def test_device_count_matches_torch(lib):
    import torch
    assert lib.qhea_device_count() == (torch.cuda.device_count() if torch.cuda.is_available() else 0)


def test_module_surface_matches_reference_contract(lib):
    """parameter name/shape/init, state_dict keys, errors."""
    import numpy as np
    import torch
    from quanonet_amd.models import QuanONetPT, HEAQNNPT, _build_quantum_layer
    from quanonet_amd.circuit import HEACircuitHIP, _make_block_configs, _ham_params
    torch.manual_seed(0)
    m = QuanONetPT(5, 100, 2, (40, 2, 20, 2), scale_coeff=0.1, if_trainable_freq=True)
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [
        ('bias', (1,)), ('branch_freq.weights', (200,)), ('branch_freq.bias', (200,)),
        ('trunk_freq.weights', (100,)), ('trunk_freq.bias', (100,)),
        ('quantum_layer.ansatz_weights', (120, 3, 5))]
    # the first torch RNG draw after manual_seed is the ansatz init, float32 U(-pi, pi) (quantum_circuits_tq.py:50-53)
    torch.manual_seed(0)
    ref = torch.empty(120, 3, 5).uniform_(-np.pi, np.pi)
    assert torch.equal(sd['quantum_layer.ansatz_weights'].float(), ref)
    assert float(sd['branch_freq.weights'][0]) == 0.1 and float(sd['trunk_freq.bias'].abs().sum()) == 0.0
    h = HEAQNNPT(8, 102, (20, 2), scale_coeff=0.1, if_trainable_freq=True)
    assert [k for k in h.state_dict()] == ['freq.weights', 'freq.bias', 'quantum_layer.ansatz_weights']
    assert tuple(h.state_dict()['quantum_layer.ansatz_weights'].shape) == (40, 3, 8)
    ff = QuanONetPT(5, 100, 2, (40, 2, 40, 2), scale_coeff=0.1, if_trainable_freq=False, ham_diag=np.arange(32.0))
    assert list(ff.state_dict()) == ['bias', 'quantum_layer.ansatz_weights', 'quantum_layer.ham_diag']
    assert _make_block_configs(5, 20, 2, 40, 2) == [(5, 2)] * 60
    assert _ham_params(5, -5.0, 5.0) == (0.0, 1.0)
    with pytest.raises(ValueError):
        _build_quantum_layer('qiskit', 5, 300, (40, 2, 20, 2), (-5, 5), None, 100, 2)
    with pytest.raises(ValueError):
        HEACircuitHIP(1, [(1, 1)])
    with pytest.raises(ValueError):
        m.quantum_layer(torch.zeros(4, 7, 2, dtype=torch.float64))
    from quanonet_amd import _lib
    with pytest.raises(_lib.QheaError):     # a narrower x is legal (the reference's column guard, tq.py:83) and reaches the op
        m.quantum_layer(torch.zeros(4, 7, dtype=torch.float64))
    with pytest.raises(_lib.QheaError):                            # no CPU fallback
        m(torch.zeros(4, 100, dtype=torch.float64), torch.zeros(4, 2, dtype=torch.float64))


def test_dp_exchange_argument_validation_without_gpu(lib):
    """qhea_dp_*: sizes and argument errors are decided on the host before any HIP call."""
    vp = ctypes.c_void_p
    # header + block flags [16 ranks][512 blocks] + 2 parities x world x padded values
    assert lib.qhea_dp_buffer_bytes(2403, 8) == 256 + 2 * 8 * 2404 * 16         # header | [2 parities][world][padded n] tagged word pairs
    assert lib.qhea_dp_buffer_bytes(2403, 17) == 0                              # QHEA_DP_MAX_RANKS = 16
    assert lib.qhea_dp_buffer_bytes(0, 2) == 0
    assert lib.qhea_dp_alloc(0, 2, ctypes.byref(vp())) == -1
    assert lib.qhea_dp_export(None, None) == -1
    assert lib.qhea_dp_import(None, None) == -1
    assert lib.qhea_dp_free(None) == 0 and lib.qhea_dp_close(None) == 0
    assert lib.qhea_dp_status(None, None) == -1
    bufs = (vp * 2)(None, None)
    fake = vp(8)
    # rank outside the world, missing buffers, seq < 1, more parameters than values
    args = lambda rank, world, n, seq, npar: lib.qhea_dp_allreduce_adam(rank, world, bufs, n, seq, fake, fake, npar, None,
                                                                         None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0,
                                                                         100.0, None)
    assert args(2, 2, 16, 1, 0) == -1
    assert args(0, 2, 16, 1, 0) == -1                # buffers[r] == NULL
    assert args(0, 17, 16, 1, 0) == -1
    bufs[0] = 8; bufs[1] = 8
    assert args(0, 2, 16, 0, 0) == -1
    assert args(0, 2, 16, 1, 17) == -1
    assert b'exchange' in lib.qhea_strerror(-7)


def test_train_steps_argument_validation_without_gpu(lib):
    from quanonet_amd import _lib
    desc = _lib.ModelDesc(0, 5, (ctypes.c_int32 * 4)(40, 2, 20, 2), 100, 2, 1, 0, 0.1, 0.0, 1.0)
    i64 = ctypes.c_int64
    fake = ctypes.c_void_p(8)
    ok_rows = (i64 * 3)(0, 4, 8); bad_rows = (i64 * 3)(0, 4, 4)
    inv = (ctypes.c_double * 2)(0.25, 0.25)
    call = lambda rows, stride, first: lib.qhea_model_train_steps(ctypes.byref(desc), 2, rows, fake, fake, fake, fake, None,
                                                                  inv, fake, stride, fake, fake, first, 1e-4, 0.9, 0.999,
                                                                  1e-8, 0.0, None, 0, None)
    P = lib.qhea_model_param_count(ctypes.byref(desc))
    assert P == 2401
    assert call(bad_rows, P + 2, 1) == -1            # an empty step
    assert call(ok_rows, P + 1, 1) == -1             # rows too narrow for [grads | sse | sum y^2]
    assert call(ok_rows, P + 2, 0) == -1             # Adam update count starts at 1


def test_dp_train_steps_argument_validation_without_gpu(lib):
    """qhea_model_dp_train_steps decides everything it can on the host BEFORE the first launch: a rank that stopped half-way
    through a run of steps would leave its peers waiting."""
    from quanonet_amd import _lib
    lib.qhea_model_dp_train_steps.restype = ctypes.c_int
    desc = _lib.ModelDesc(0, 5, (ctypes.c_int32 * 4)(40, 2, 20, 2), 100, 2, 1, 0, 0.1, 0.0, 1.0)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    fake = vp(8)
    ok_rows = (i64 * 3)(0, 4, 8); empty_rows = (i64 * 3)(0, 4, 4); bad_rows = (i64 * 3)(0, 4, 2)
    inv = (ctypes.c_double * 2)(0.25, 0.25)
    bufs = (vp * 2)(8, 8)
    P = lib.qhea_model_param_count(ctypes.byref(desc))

    def call(rows=ok_rows, stride=P + 2, first=1, rank=0, world=2, buffers=bufs, dp_values=P + 2, seq=1, timeout=100.0):
        return lib.qhea_model_dp_train_steps(ctypes.byref(desc), 2, rows, fake, fake, fake, fake, None, inv, fake, i64(stride),
                                             fake, fake, i64(first), ctypes.c_double(1e-4), ctypes.c_double(0.9),
                                             ctypes.c_double(0.999), ctypes.c_double(1e-8), ctypes.c_double(0.0), rank, world,
                                             buffers, i64(dp_values), i64(seq), ctypes.c_double(timeout), None, 0, None)
    assert call(world=1) == -1 and call(world=17) == -1 and call(rank=2) == -1
    assert call(seq=0) == -1 and call(timeout=0.0) == -1 and call(first=0) == -1
    assert call(dp_values=P + 1) == -1 and call(stride=P + 1) == -1
    assert call(buffers=(vp * 2)(8, None)) == -1
    assert call(rows=bad_rows) == -1
    assert call(rows=empty_rows) == -2               # an empty shard: QHEA_EUNSUPPORTED, the caller takes the separate exchange
    assert call() == -3                              # everything valid up to the (missing) workspace: nothing was launched


def test_clock_probe_argument_validation_without_gpu(lib):
    lib.qhea_clock_probe.restype = ctypes.c_int
    lib.qhea_clock_probe.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.qhea_clock_probe(0, 10, ctypes.c_void_p(8), None) == -1
    assert lib.qhea_clock_probe(4, 0, ctypes.c_void_p(8), None) == -1
    assert lib.qhea_clock_probe(4, 10, None, None) == -1
