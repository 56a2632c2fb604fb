"""
CPU fp64 ORACLE for the QuanONet HEA circuit hot path  --  TEST INFRASTRUCTURE ONLY.

This is a numpy restatement of the reference's algorithm for the batched
hardware-efficient-ansatz (HEA) circuit forward + adjoint-gradient pass.  It is
the checker the HIP kernels are compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``quanonet_amd/`` does (the product path fails loudly when
the HIP library is missing instead of falling back to this file).

Parity pinning: forward values are pinned by the reference's own known answers
K1-K8 (analytic antiderivative checks from
``ibm_inference.py:180-187`` and the six MSE/MAE figures printed in
``visualization.ipynb`` cell 7), driven by the reference's shipped checkpoints
(``tests/golden``).  Gradients are pinned by mathematics (parameter-shift
identity + central differences) because the reference ships no gradient
vectors.  The general ``ham_diag`` read-out is "parity unpinned" (no reference
fixture exercises it; bit order chosen = MindQuantum little-endian,
``core/quantum_circuits_ms.py:41-63``).

Reference files followed (paths relative to the reference checkout):
  * circuit order / parameter layout:  core/quantum_circuits_tq.py:65-104
      (== core/quantum_circuits_ms.py:127-226, trunk blocks first)
  * read-out:                          core/quantum_circuits_tq.py:106-146
      (== core/quantum_circuits_ms.py:28-39)
  * pre/post-processing:               core/models_pt.py:14-68,153-166,205-213
  * weight mapping:                    utils/weight_transfer.py:46-98

Conventions (verified by K1-K8, a flipped CNOT gives 633 % error):
  * basis index k, bit i of k  <->  qubit (wire) i          (little-endian)
  * RX(t) = [[c,-is],[-is,c]], RY(t) = [[c,-s],[s,c]], RZ(t) = diag(e^-it/2, e^+it/2)
    with c = cos(t/2), s = sin(t/2)
  * entangler: for i in 0..n-1 in order: CNOT(control=(i+1)%n, target=i)
"""
import numpy as np


# --------------------------------------------------------------------------
# circuit shape  (core/quantum_circuits_tq.py:130-146, 149-202)
# --------------------------------------------------------------------------
def block_configs_quanonet(num_qubits, net_size):
    """[(n_encode, linear_depth)]: trunk blocks first, then branch blocks."""
    bd, bl, td, tl = net_size
    return [(num_qubits, tl)] * td + [(num_qubits, bl)] * bd


def block_configs_heaqnn(num_qubits, net_size):
    return [(num_qubits, net_size[1])] * net_size[0]


def ham_params(num_qubits, lo=-5.0, hi=5.0):
    """(offset, coeff_per_qubit) of H = offset + coeff * sum_i Z_i."""
    coff = hi - lo
    return lo + coff / 2.0, coff / 2.0 / num_qubits


def circuit_sizes(num_qubits, block_configs):
    E = sum(ne for ne, _ in block_configs)
    blk = sum(ld for _, ld in block_configs)
    return E, blk


def ham_diagonal(num_qubits, offset, coeff, ham_diag=None):
    """Real diagonal H_k.  Simple form: offset + coeff*(n - 2*popcount(k))."""
    if ham_diag is not None:
        d = np.asarray(ham_diag, dtype=np.float64).reshape(-1)
        assert d.size == 1 << num_qubits
        return d
    k = np.arange(1 << num_qubits)
    pop = np.zeros_like(k)
    for i in range(num_qubits):
        pop += (k >> i) & 1
    return offset + coeff * (num_qubits - 2.0 * pop)


# --------------------------------------------------------------------------
# gate kernels on a batched state psi[B, 2^n] (complex128)
# --------------------------------------------------------------------------
def _pairs(n, q):
    k = np.arange(1 << n)
    i0 = k[((k >> q) & 1) == 0]
    return i0, i0 | (1 << q)


def _apply_1q(psi, n, q, m00, m01, m10, m11):
    """psi <- (M on qubit q) psi; m?? are scalars or (B,1) arrays."""
    i0, i1 = _pairs(n, q)
    a0 = psi[:, i0]
    a1 = psi[:, i1]
    psi[:, i0] = m00 * a0 + m01 * a1
    psi[:, i1] = m10 * a0 + m11 * a1


def _rx(psi, n, q, theta, dagger=False):
    th = np.asarray(theta, dtype=np.float64)
    if th.ndim == 1:
        th = th[:, None]
    c = np.cos(th / 2)
    s = np.sin(th / 2)
    if dagger:
        s = -s
    _apply_1q(psi, n, q, c, -1j * s, -1j * s, c)


def _ry(psi, n, q, theta, dagger=False):
    c = np.cos(theta / 2)
    s = np.sin(theta / 2)
    if dagger:
        s = -s
    _apply_1q(psi, n, q, c, -s, s, c)


def _rz(psi, n, q, theta, dagger=False):
    if dagger:
        theta = -theta
    e0 = np.exp(-0.5j * theta)
    e1 = np.exp(+0.5j * theta)
    _apply_1q(psi, n, q, e0, 0.0, 0.0, e1)


def _cnot(psi, n, control, target):
    k = np.arange(1 << n)
    src = np.where((k >> control) & 1, k ^ (1 << target), k)
    psi[:, :] = psi[:, src]


def _im_inner_pauli(lam, psi, n, q, pauli):
    """Im <lam| sigma_q |psi> per sample, (B,)."""
    i0, i1 = _pairs(n, q)
    sp = np.empty_like(psi)
    if pauli == 'X':
        sp[:, i0] = psi[:, i1]
        sp[:, i1] = psi[:, i0]
    elif pauli == 'Y':
        sp[:, i0] = -1j * psi[:, i1]
        sp[:, i1] = 1j * psi[:, i0]
    else:
        sp[:, i0] = psi[:, i0]
        sp[:, i1] = -psi[:, i1]
    return np.imag(np.sum(np.conj(lam) * sp, axis=1))


# --------------------------------------------------------------------------
# forward / adjoint backward   (quantum_circuits_tq.py:65-127)
# --------------------------------------------------------------------------
def hea_state(num_qubits, block_configs, x, w):
    """Final statevector psi[B, 2^n] for encoding angles x[B,E], ansatz w[blk,3,n]."""
    n = num_qubits
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    B = x.shape[0]
    psi = np.zeros((B, 1 << n), dtype=np.complex128)
    psi[:, 0] = 1.0
    col = 0
    blk = 0
    for n_enc, ld in block_configs:
        for j in range(n_enc):
            if col < x.shape[1]:                          # quantum_circuits_tq.py:83: no column, no gate
                _rx(psi, n, j % n, x[:, col])
            col += 1
        for _ in range(ld):
            for i in range(n):
                _ry(psi, n, i, w[blk, 0, i])
                _rz(psi, n, i, w[blk, 1, i])
                _ry(psi, n, i, w[blk, 2, i])
            for i in range(n):
                _cnot(psi, n, (i + 1) % n, i)
            blk += 1
    return psi


def _apply_pauli_ham(psi, n, offset, coeff, pauli):
    """H psi for H = offset + coeff * sum_q P_q, P = 'X' or 'Y', Pauli by Pauli
    (generate_simple_hamiltonian's `pauli`, core/quantum_circuits_ms.py:28-39)."""
    out = offset * psi
    for q in range(n):
        i0, i1 = _pairs(n, q)
        sp = np.empty_like(psi)
        if pauli == 'X':
            sp[:, i0] = psi[:, i1]
            sp[:, i1] = psi[:, i0]
        else:
            sp[:, i0] = -1j * psi[:, i1]
            sp[:, i1] = 1j * psi[:, i0]
        out = out + coeff * sp
    return out


def _check_pauli(ham_pauli, ham_diag):
    p = str(ham_pauli).upper()
    if p not in ('X', 'Y', 'Z'):
        raise ValueError(f"ham_pauli must be X, Y or Z, got {ham_pauli!r}")
    if p != 'Z' and ham_diag is not None:
        raise ValueError("ham_diag is a Z-basis diagonal; it excludes ham_pauli X/Y (utils/common.py:84)")
    return p


def hea_forward(num_qubits, block_configs, x, w, offset=0.0, coeff=1.0, ham_diag=None, ham_pauli='Z'):
    """out[B] = <psi|H|psi>   (no bias)."""
    pauli = _check_pauli(ham_pauli, ham_diag)
    psi = hea_state(num_qubits, block_configs, x, w)
    if pauli != 'Z':
        return np.real(np.sum(np.conj(psi) * _apply_pauli_ham(psi, num_qubits, offset, coeff, pauli), axis=1))
    H = ham_diagonal(num_qubits, offset, coeff, ham_diag)
    return np.sum((psi.real ** 2 + psi.imag ** 2) * H[None, :], axis=1)


def hea_backward(num_qubits, block_configs, x, w, g, offset=0.0, coeff=1.0, ham_diag=None, ham_pauli='Z'):
    """
    Adjoint differentiation (the scheme MindQuantum's
    get_expectation_with_grad uses, core/quantum_circuits_ms.py:229-233).

    g[B] = dL/d out_b.  Returns (out[B], grad_x[B,E], grad_w[blk,3,n]) with
    grad_x[b,e] = g_b * d out_b / d x[b,e],  grad_w = sum_b g_b * d out_b / d w.
    """
    n = num_qubits
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    B, E = x.shape
    pauli = _check_pauli(ham_pauli, ham_diag)
    psi = hea_state(n, block_configs, x, w)
    if pauli != 'Z':
        hpsi = _apply_pauli_ham(psi, n, offset, coeff, pauli)
        out = np.real(np.sum(np.conj(psi) * hpsi, axis=1))
        lam = hpsi * g[:, None]
    else:
        H = ham_diagonal(n, offset, coeff, ham_diag)
        out = np.sum((psi.real ** 2 + psi.imag ** 2) * H[None, :], axis=1)
        lam = psi * H[None, :] * g[:, None]          # upstream weight folded into lambda
    grad_x = np.zeros((B, E))
    grad_w = np.zeros_like(w)

    # flatten the gate list once, then walk it in reverse
    ops = []
    col = 0
    blk = 0
    for n_enc, ld in block_configs:
        for j in range(n_enc):
            if col < E:                                   # quantum_circuits_tq.py:83 (grad_x has x's own width)
                ops.append(('rx', j % n, col))
            col += 1
        for _ in range(ld):
            for i in range(n):
                ops.append(('ry', i, (blk, 0, i)))
                ops.append(('rz', i, (blk, 1, i)))
                ops.append(('ry', i, (blk, 2, i)))
            for i in range(n):
                ops.append(('cx', (i + 1) % n, i))
            blk += 1

    for op in reversed(ops):
        kind = op[0]
        if kind == 'cx':
            _cnot(psi, n, op[1], op[2])
            _cnot(lam, n, op[1], op[2])
        elif kind == 'rx':
            q, c = op[1], op[2]
            grad_x[:, c] = _im_inner_pauli(lam, psi, n, q, 'X')
            _rx(psi, n, q, x[:, c], dagger=True)
            _rx(lam, n, q, x[:, c], dagger=True)
        elif kind == 'ry':
            q, idx = op[1], op[2]
            grad_w[idx] = np.sum(_im_inner_pauli(lam, psi, n, q, 'Y'))
            _ry(psi, n, q, w[idx], dagger=True)
            _ry(lam, n, q, w[idx], dagger=True)
        else:
            q, idx = op[1], op[2]
            grad_w[idx] = np.sum(_im_inner_pauli(lam, psi, n, q, 'Z'))
            _rz(psi, n, q, w[idx], dagger=True)
            _rz(lam, n, q, w[idx], dagger=True)
    return out, grad_x, grad_w


# --------------------------------------------------------------------------
# classical pre/post-processing   (core/models_pt.py:14-68,153-166,205-213)
# --------------------------------------------------------------------------
def tiled_elementwise(x, weights, bias):
    """y[:,k] = x[:, k mod in] * w[k] + b[k]   (models_pt.py:38-41)."""
    x = np.asarray(x, dtype=np.float64)
    out = weights.shape[0]
    idx = np.arange(out) % x.shape[1]
    return x[:, idx] * weights[None, :] + bias[None, :]


def scale_repeat(x, scale, out_features):
    """y[:,k] = scale * x[:, k mod in]   (models_pt.py:63-68)."""
    x = np.asarray(x, dtype=np.float64)
    idx = np.arange(out_features) % x.shape[1]
    return x[:, idx] * scale


def _engine(engine):
    """Circuit arithmetic behind the model-level functions: this module (numpy, gate by gate) by default, or
    ``oracle.c_oracle`` (the C restatement, same call shape) for sizes where numpy would take minutes."""
    import sys
    return engine if engine is not None else sys.modules[__name__]


def _freq_encode(x_in, params, prefix, out_features, scale_coeff):
    """
    One frequency layer (core/models_pt.py:14-68).  Trainable form when ``params`` holds
    ``<prefix>.weights`` / ``<prefix>.bias`` (_TiledElementWise, :38-41); otherwise the fixed-scale
    form y[:,k] = scale_coeff * x[:, k mod in] (_ScaleRepeat, :63-68), which has no parameters.
    Returns (encoded[B,out], tiled_input[B,out], trainable).
    """
    x_in = np.asarray(x_in, np.float64)
    idx = np.arange(out_features) % x_in.shape[1]
    tiled = x_in[:, idx]
    wkey, bkey = prefix + '.weights', prefix + '.bias'
    if wkey in params:
        w = np.asarray(params[wkey], np.float64).reshape(-1)
        b = np.asarray(params[bkey], np.float64).reshape(-1)
        assert w.shape[0] == out_features and b.shape[0] == out_features
        return tiled * w[None, :] + b[None, :], tiled, True
    if scale_coeff is None:
        raise ValueError(f"no '{wkey}' in params: fixed-frequency mode needs scale_coeff")
    return tiled * float(scale_coeff), tiled, False


def quanonet_forward(params, branch, trunk, num_qubits, net_size, ham_bound=(-5.0, 5.0),
                     ham_diag=None, ham_pauli='Z', scale_coeff=None, engine=None):
    """
    QuanONetPT.forward (models_pt.py:153-166).  params: dict with the PT state_dict keys;
    without the ``*_freq`` keys the fixed-frequency form (if_trainable_freq=False) with
    ``scale_coeff`` is evaluated.  Returns out[B] (bias included).
    """
    bd, bl, td, tl = net_size
    t_enc, _, _ = _freq_encode(trunk, params, 'trunk_freq', td * num_qubits, scale_coeff)
    b_enc, _, _ = _freq_encode(branch, params, 'branch_freq', bd * num_qubits, scale_coeff)
    x = np.concatenate([t_enc, b_enc], axis=1)            # trunk first
    off, co = ham_params(num_qubits, *ham_bound)
    cfgs = block_configs_quanonet(num_qubits, net_size)
    out = _engine(engine).hea_forward(num_qubits, cfgs, x, params['quantum_layer.ansatz_weights'], off, co, ham_diag,
                                      ham_pauli=ham_pauli)
    return out + float(np.asarray(params['bias']).reshape(-1)[0])


def quanonet_loss_and_grads(params, branch, trunk, y, num_qubits, net_size,
                            ham_bound=(-5.0, 5.0), batch_total=None, ham_pauli='Z', scale_coeff=None,
                            ham_diag=None, engine=None):
    """
    MSE(mean) loss and gradients w.r.t. every QuanONetPT parameter, restating what
    torch autograd produces for solver_pt.py:232-236.  ``batch_total`` is the global
    batch size used in the mean (defaults to len(y)); a data-parallel shard passes
    the global size so that a plain SUM over shards reproduces the full gradient.
    Fixed-frequency models (no ``*_freq`` keys in params; pass ``scale_coeff``) have only
    the ``bias`` and ``quantum_layer.ansatz_weights`` gradients.
    """
    y = np.asarray(y, np.float64).reshape(-1)
    Bt = float(batch_total if batch_total is not None else y.shape[0])
    bd, bl, td, tl = net_size
    nt = td * num_qubits
    t_enc, t_til, t_train = _freq_encode(trunk, params, 'trunk_freq', nt, scale_coeff)
    b_enc, b_til, b_train = _freq_encode(branch, params, 'branch_freq', bd * num_qubits, scale_coeff)
    x = np.concatenate([t_enc, b_enc], axis=1)
    off, co = ham_params(num_qubits, *ham_bound)
    cfgs = block_configs_quanonet(num_qubits, net_size)
    w = params['quantum_layer.ansatz_weights']
    bias = float(np.asarray(params['bias']).reshape(-1)[0])
    eng = _engine(engine)
    out = eng.hea_forward(num_qubits, cfgs, x, w, off, co, ham_diag, ham_pauli=ham_pauli) + bias
    resid = out - y
    g = 2.0 * resid / Bt
    _, gx, gw = eng.hea_backward(num_qubits, cfgs, x, w, g, off, co, ham_diag, ham_pauli=ham_pauli)
    grads = {'quantum_layer.ansatz_weights': gw, 'bias': np.array([np.sum(g)])}
    if t_train:
        grads['trunk_freq.weights'] = np.sum(gx[:, :nt] * t_til, axis=0)
        grads['trunk_freq.bias'] = np.sum(gx[:, :nt], axis=0)
    if b_train:
        grads['branch_freq.weights'] = np.sum(gx[:, nt:] * b_til, axis=0)
        grads['branch_freq.bias'] = np.sum(gx[:, nt:], axis=0)
    sse = float(np.sum(resid ** 2))
    return sse / Bt, grads, out


def heaqnn_forward(params, x_in, num_qubits, net_size, ham_bound=(-5.0, 5.0), ham_diag=None, ham_pauli='Z',
                   scale_coeff=None, engine=None):
    """HEAQNNPT.forward (models_pt.py:205-213): out = Q(F(x)), no bias.  params: PT state_dict keys."""
    enc, _, _ = _freq_encode(x_in, params, 'freq', net_size[0] * num_qubits, scale_coeff)
    off, co = ham_params(num_qubits, *ham_bound)
    cfgs = block_configs_heaqnn(num_qubits, net_size)
    return _engine(engine).hea_forward(num_qubits, cfgs, enc, params['quantum_layer.ansatz_weights'], off, co, ham_diag,
                                       ham_pauli=ham_pauli)


def heaqnn_loss_and_grads(params, x_in, y, num_qubits, net_size, ham_bound=(-5.0, 5.0), batch_total=None,
                          ham_pauli='Z', scale_coeff=None, ham_diag=None, engine=None):
    """
    MSE(mean) loss and gradients w.r.t. every HEAQNNPT parameter (freq.weights, freq.bias when trainable,
    quantum_layer.ansatz_weights); same conventions as quanonet_loss_and_grads.  The reference's own HEAQNN
    checks are compare_backends.py:219-281 and :383-449.
    """
    y = np.asarray(y, np.float64).reshape(-1)
    Bt = float(batch_total if batch_total is not None else y.shape[0])
    enc, til, train = _freq_encode(x_in, params, 'freq', net_size[0] * num_qubits, scale_coeff)
    off, co = ham_params(num_qubits, *ham_bound)
    cfgs = block_configs_heaqnn(num_qubits, net_size)
    w = params['quantum_layer.ansatz_weights']
    eng = _engine(engine)
    out = eng.hea_forward(num_qubits, cfgs, enc, w, off, co, ham_diag, ham_pauli=ham_pauli)
    resid = out - y
    g = 2.0 * resid / Bt
    _, gx, gw = eng.hea_backward(num_qubits, cfgs, enc, w, g, off, co, ham_diag, ham_pauli=ham_pauli)
    grads = {'quantum_layer.ansatz_weights': gw}
    if train:
        grads['freq.weights'] = np.sum(gx * til, axis=0)
        grads['freq.bias'] = np.sum(gx, axis=0)
    return float(np.sum(resid ** 2)) / Bt, grads, out


# --------------------------------------------------------------------------
# self-checks that need no external oracle
# --------------------------------------------------------------------------
def param_shift_grad_w(num_qubits, block_configs, x, w, g, offset, coeff, idx, ham_pauli='Z'):
    """Exact derivative of sum_b g_b out_b w.r.t. w[idx] by the parameter-shift rule."""
    wp = np.array(w, dtype=np.float64)
    wm = np.array(w, dtype=np.float64)
    wp[idx] += np.pi / 2
    wm[idx] -= np.pi / 2
    fp = hea_forward(num_qubits, block_configs, x, wp, offset, coeff, ham_pauli=ham_pauli)
    fm = hea_forward(num_qubits, block_configs, x, wm, offset, coeff, ham_pauli=ham_pauli)
    return float(np.sum(np.asarray(g) * 0.5 * (fp - fm)))


def param_shift_grad_x(num_qubits, block_configs, x, w, g, offset, coeff, col):
    xp = np.array(x, dtype=np.float64)
    xm = np.array(x, dtype=np.float64)
    xp[:, col] += np.pi / 2
    xm[:, col] -= np.pi / 2
    fp = hea_forward(num_qubits, block_configs, xp, w, offset, coeff)
    fm = hea_forward(num_qubits, block_configs, xm, w, offset, coeff)
    return np.asarray(g) * 0.5 * (fp - fm)
