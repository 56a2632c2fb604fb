"""
ctypes binding of the C ABI in include/quanonet_hea.h (libquanonet_hea.so, built in-tree by
``__graft_entry__.build()`` / ``quanonet_amd/csrc/Makefile``).

There is deliberately NO CPU fallback: if the shared library is missing, or a tensor is not on
a HIP device, the calls raise.  torch is used only for device memory and streams.
"""
import ctypes
import os

import torch

# Inter-process mapping of the data-parallel exchange buffers (qhea_dp_export / qhea_dp_import = hipIpcGetMemHandle /
# hipIpcOpenMemHandle) needs the dmabuf IPC mode on this driver stack: with the legacy mode hipIpcGetMemHandle fails with
# "invalid argument" and the trainer would fall back to the RCCL all-reduce.  The HSA runtime reads the variable when it
# is initialised, so the default is set here, at import, before this package makes its first HIP call; a process that had
# already initialised the GPU without it is told so by PeerExchange.create (its reason string).
IPC_ENV = 'HSA_ENABLE_IPC_MODE_LEGACY'
IPC_ENV_PRESET = os.environ.get(IPC_ENV)
IPC_ENV_SET_LATE = IPC_ENV_PRESET is None and torch.cuda.is_initialized()
os.environ.setdefault(IPC_ENV, '0')

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QHEA_LIB') or os.path.join(_HERE, 'libquanonet_hea.so')   # QHEA_LIB: ablation/dev builds

EXPORTS = ['qhea_version', 'qhea_strerror', 'qhea_device_count', 'qhea_workspace_bytes',
           'qhea_forward', 'qhea_backward', 'qhea_model_param_count', 'qhea_model_workspace_bytes',
           'qhea_model_forward', 'qhea_model_forward_chunks', 'qhea_model_loss_grad', 'qhea_model_train_step', 'qhea_model_train_steps',
           'qhea_profile_next_circuit_kernel',
           'qhea_adam_step', 'qhea_set_backward_variant', 'qhea_check_status',
           'qhea_dp_buffer_bytes', 'qhea_dp_alloc', 'qhea_dp_free', 'qhea_dp_export', 'qhea_dp_import', 'qhea_dp_close',
           'qhea_dp_allreduce_adam', 'qhea_dp_status', 'qhea_model_dp_train_steps', 'qhea_clock_probe']


class ModelDesc(ctypes.Structure):
    """Mirror of `qhea_model_desc` (include/quanonet_hea.h)."""
    _fields_ = [('model', ctypes.c_int32), ('n_qubits', ctypes.c_int32), ('net', ctypes.c_int32 * 4),
                ('branch_in', ctypes.c_int32), ('trunk_in', ctypes.c_int32),
                ('trainable_freq', ctypes.c_int32), ('ham_pauli', ctypes.c_int32),
                ('scale_coeff', ctypes.c_double), ('ham_offset', ctypes.c_double), ('ham_coeff', ctypes.c_double)]


MODEL_QUANONET, MODEL_HEAQNN = 0, 1
MIN_LIB_VERSION = 440           # 0.4.3: + qhea_model_dp_train_steps (exchange inside the reduce kernel)
BWD_VARIANTS = {'auto': 0, 'packed': 1, 'pair': 2, 'tri': 3, 'ztri': 4, 'zpacked': 5, 'ztri2': 6, 'zquad': 7}
PAULI = {'Z': 0, 'X': 1, 'Y': 2}


def pauli_code(p):
    """'Z'/'X'/'Y' (the reference's --ham_pauli choices, utils/common.py:81) or 0/1/2 -> QHEA_PAULI_*."""
    if isinstance(p, str):
        if p.upper() not in PAULI:
            raise ValueError(f"ham_pauli must be one of X, Y, Z (got {p!r})")
        return PAULI[p.upper()]
    if int(p) not in (0, 1, 2):
        raise ValueError(f"ham_pauli code must be 0, 1 or 2 (got {p!r})")
    return int(p)

_lib = None


class QheaError(RuntimeError):
    pass


def load():
    """Load libquanonet_hea.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QheaError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                        f"g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.qhea_version.restype = ctypes.c_int
    if lib.qhea_version() < MIN_LIB_VERSION:
        raise QheaError(f"{LIB_PATH} is version {lib.qhea_version()}, this binding needs >= {MIN_LIB_VERSION} "
                        f"(the data-parallel exchange entry points): rebuild it")
    vp, dp = ctypes.c_void_p, ctypes.c_void_p
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.qhea_version.restype = ctypes.c_int
    lib.qhea_strerror.restype = ctypes.c_char_p
    lib.qhea_strerror.argtypes = [ctypes.c_int]
    lib.qhea_device_count.restype = ctypes.c_int
    lib.qhea_workspace_bytes.restype = ctypes.c_size_t
    lib.qhea_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, i32p, i32p, ctypes.c_int64]
    lib.qhea_forward.restype = ctypes.c_int
    lib.qhea_forward.argtypes = [ctypes.c_int, ctypes.c_int, i32p, i32p, ctypes.c_int64, dp, dp,
                                 ctypes.c_double, ctypes.c_double, dp, ctypes.c_int, dp, dp, vp, ctypes.c_size_t, vp]
    lib.qhea_backward.restype = ctypes.c_int
    lib.qhea_backward.argtypes = [ctypes.c_int, ctypes.c_int, i32p, i32p, ctypes.c_int64, dp, dp,
                                  ctypes.c_double, ctypes.c_double, dp, ctypes.c_int, dp, dp, dp, dp, dp,
                                  vp, ctypes.c_size_t, vp]
    lib.qhea_profile_next_circuit_kernel.restype = ctypes.c_int
    lib.qhea_profile_next_circuit_kernel.argtypes = [vp, vp]
    lib.qhea_set_backward_variant.restype = ctypes.c_int
    lib.qhea_set_backward_variant.argtypes = [ctypes.c_int]
    lib.qhea_check_status.restype = ctypes.c_int
    lib.qhea_check_status.argtypes = [vp, ctypes.c_size_t, vp]
    lib.qhea_adam_step.restype = ctypes.c_int
    lib.qhea_adam_step.argtypes = [ctypes.c_int64, dp, dp, dp, dp, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double, vp]
    mdp = ctypes.POINTER(ModelDesc)
    i64p = ctypes.POINTER(ctypes.c_int64)
    f64p = ctypes.POINTER(ctypes.c_double)
    lib.qhea_model_forward_chunks.restype = ctypes.c_int
    lib.qhea_model_forward_chunks.argtypes = [mdp, ctypes.c_int64, i64p, dp, dp, dp, dp, dp, vp, ctypes.c_size_t, vp]
    lib.qhea_model_train_steps.restype = ctypes.c_int
    lib.qhea_model_train_steps.argtypes = [mdp, ctypes.c_int64, i64p, dp, dp, dp, dp, dp, f64p, dp, ctypes.c_int64, dp, dp,
                                           ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, vp, ctypes.c_size_t, vp]
    vpp = ctypes.POINTER(ctypes.c_void_p)
    lib.qhea_dp_buffer_bytes.restype = ctypes.c_size_t
    lib.qhea_dp_buffer_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.qhea_dp_alloc.restype = ctypes.c_int
    lib.qhea_dp_alloc.argtypes = [ctypes.c_int64, ctypes.c_int, vpp]
    lib.qhea_dp_free.restype = ctypes.c_int
    lib.qhea_dp_free.argtypes = [vp]
    lib.qhea_dp_export.restype = ctypes.c_int
    lib.qhea_dp_export.argtypes = [vp, ctypes.c_char_p]
    lib.qhea_dp_import.restype = ctypes.c_int
    lib.qhea_dp_import.argtypes = [ctypes.c_char_p, vpp]
    lib.qhea_dp_close.restype = ctypes.c_int
    lib.qhea_dp_close.argtypes = [vp]
    lib.qhea_dp_allreduce_adam.restype = ctypes.c_int
    lib.qhea_dp_allreduce_adam.argtypes = [ctypes.c_int, ctypes.c_int, vpp, ctypes.c_int64, ctypes.c_int64, dp, dp,
                                           ctypes.c_int64, dp, dp, dp, ctypes.c_int64, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, vp]
    lib.qhea_dp_status.restype = ctypes.c_int
    lib.qhea_dp_status.argtypes = [vp, vp]
    lib.qhea_model_dp_train_steps.restype = ctypes.c_int
    lib.qhea_model_dp_train_steps.argtypes = [mdp, ctypes.c_int64, i64p, dp, dp, dp, dp, dp, f64p, dp, ctypes.c_int64, dp, dp,
                                              ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, vpp,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_double, vp, ctypes.c_size_t, vp]
    lib.qhea_clock_probe.restype = ctypes.c_int
    lib.qhea_clock_probe.argtypes = [ctypes.c_int, ctypes.c_int64, vp, vp]
    lib.qhea_model_param_count.restype = ctypes.c_int64
    lib.qhea_model_param_count.argtypes = [mdp]
    lib.qhea_model_workspace_bytes.restype = ctypes.c_size_t
    lib.qhea_model_workspace_bytes.argtypes = [mdp, ctypes.c_int64]
    lib.qhea_model_forward.restype = ctypes.c_int
    lib.qhea_model_forward.argtypes = [mdp, ctypes.c_int64, dp, dp, dp, dp, dp, vp, ctypes.c_size_t, vp]
    lib.qhea_model_loss_grad.restype = ctypes.c_int
    lib.qhea_model_loss_grad.argtypes = [mdp, ctypes.c_int64, dp, dp, dp, dp, dp, ctypes.c_double, dp, dp,
                                         vp, ctypes.c_size_t, vp]
    lib.qhea_model_train_step.restype = ctypes.c_int
    lib.qhea_model_train_step.argtypes = [mdp, ctypes.c_int64, dp, dp, dp, dp, dp, ctypes.c_double, dp, dp, dp, dp,
                                          ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double, vp, ctypes.c_size_t, vp]
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise QheaError(f"{what} failed: {load().qhea_strerror(rc).decode()} ({rc})")


class CircuitShape:
    """Host-side description of the block list; owns the int32 arrays handed to the C ABI."""

    def __init__(self, num_qubits, block_configs):
        self.n = int(num_qubits)
        self.block_configs = [(int(a), int(b)) for a, b in block_configs]
        nb = len(self.block_configs)
        self._enc = (ctypes.c_int32 * max(nb, 1))(*[c[0] for c in self.block_configs])
        self._ld = (ctypes.c_int32 * max(nb, 1))(*[c[1] for c in self.block_configs])
        self.nb = nb
        self.E = sum(c[0] for c in self.block_configs)
        self.blk = sum(c[1] for c in self.block_configs)

    def workspace_bytes(self, batch):
        return int(load().qhea_workspace_bytes(self.n, self.nb, self._enc, self._ld, int(batch)))


def _dev_f64(t, name, shape=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise QheaError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float64 or not t.is_contiguous():
        raise QheaError(f"{name} must be a contiguous float64 tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise QheaError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_workspaces = {}


def _model_ws(desc, rows, device):
    """Workspace for a model-level call on `rows` rows; the size query runs with `device` current (the layout follows that
    device's SIMD count)."""
    with torch.cuda.device(device):
        nbytes = int(load().qhea_model_workspace_bytes(ctypes.byref(desc), int(rows)))
    return _workspace(device, nbytes)


def _workspace(device, nbytes):
    """Grow-only per-device scratch tensor (caller-owned from the C ABI's point of view)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            check_status(device)                    # the old buffer's status word must not be lost when growing
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def set_backward_variant(name):
    """'auto' | 'packed' | 'pair' | 'tri' | 'ztri': qhea_set_backward_variant (n <= 5 kernels; tests and sweeps)."""
    if name not in BWD_VARIANTS:
        raise ValueError(f"backward variant must be one of {sorted(BWD_VARIANTS)} (got {name!r})")
    _check(load().qhea_set_backward_variant(BWD_VARIANTS[name]), 'qhea_set_backward_variant')


def check_status(device):
    """
    qhea_check_status on this device's workspace: waits for the current stream and raises QheaError if a pipelined
    backward kernel reported a hand-off overrun since the last check (the gradients of that call were NaN-poisoned
    and its fused Adam update skipped).  No-op for a device that has not run anything yet.
    """
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ws = _workspaces.get(key)
    if ws is None:
        return
    with torch.cuda.device(device):
        rc = load().qhea_check_status(_ptr(ws), ws.numel(), _stream(device))
    _check(rc, 'qhea_check_status')


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def hea_forward(shape, x, w, ham_offset, ham_coeff, ham_diag=None, return_state=False, ham_pauli=0):
    """out[B] (and optionally the final state [B,2^n,2]) on x.device.  No bias."""
    lib = load()
    B = x.shape[0]
    _dev_f64(x, 'x', (B, shape.E))
    _dev_f64(w, 'w', (shape.blk, 3, shape.n))
    _dev_f64(ham_diag, 'ham_diag', (1 << shape.n,))
    out = torch.empty(B, dtype=torch.float64, device=x.device)
    state = torch.empty((B, 1 << shape.n, 2), dtype=torch.float64, device=x.device) if return_state else None
    with torch.cuda.device(x.device):                 # the layout follows the current device's SIMD count
        nbytes = shape.workspace_bytes(B)
    ws = _workspace(x.device, nbytes)
    with torch.cuda.device(x.device):
        rc = lib.qhea_forward(shape.n, shape.nb, shape._enc, shape._ld, B, _ptr(x), _ptr(w),
                              float(ham_offset), float(ham_coeff), _ptr(ham_diag), pauli_code(ham_pauli),
                              _ptr(out), _ptr(state),
                              _ptr(ws), ws.numel(), _stream(x.device))
    _check(rc, 'qhea_forward')
    return (out, state) if return_state else out


def hea_backward(shape, x, w, g, ham_offset, ham_coeff, ham_diag=None, state=None, want_out=False, ham_pauli=0):
    """(grad_x[B,E], grad_w[blk,3,n][, out[B]]) for upstream g[B]."""
    lib = load()
    B = x.shape[0]
    _dev_f64(x, 'x', (B, shape.E))
    _dev_f64(w, 'w', (shape.blk, 3, shape.n))
    _dev_f64(g, 'g', (B,))
    _dev_f64(ham_diag, 'ham_diag', (1 << shape.n,))
    _dev_f64(state, 'state', (B, 1 << shape.n, 2))
    grad_x = torch.empty_like(x)
    grad_w = torch.empty_like(w)
    out = torch.empty(B, dtype=torch.float64, device=x.device) if want_out else None
    with torch.cuda.device(x.device):                 # the layout follows the current device's SIMD count
        nbytes = shape.workspace_bytes(B)
    ws = _workspace(x.device, nbytes)
    with torch.cuda.device(x.device):
        rc = lib.qhea_backward(shape.n, shape.nb, shape._enc, shape._ld, B, _ptr(x), _ptr(w),
                               float(ham_offset), float(ham_coeff), _ptr(ham_diag), pauli_code(ham_pauli),
                               _ptr(g), _ptr(state),
                               _ptr(out), _ptr(grad_x), _ptr(grad_w), _ptr(ws), ws.numel(),
                               _stream(x.device))
    _check(rc, 'qhea_backward')
    return (grad_x, grad_w, out) if want_out else (grad_x, grad_w)


# ---------------------------------------------------------------------------------------------------
# model-level (fused) calls
# ---------------------------------------------------------------------------------------------------
def make_model_desc(model, n_qubits, net_size, branch_in, trunk_in, trainable_freq, scale_coeff,
                    ham_offset, ham_coeff, ham_pauli=0):
    net = list(net_size) + [0] * (4 - len(net_size))
    d = ModelDesc(int(model), int(n_qubits), (ctypes.c_int32 * 4)(*[int(v) for v in net[:4]]), int(branch_in),
                  int(trunk_in), 1 if trainable_freq else 0, pauli_code(ham_pauli), float(scale_coeff), float(ham_offset),
                  float(ham_coeff))
    return d


def model_param_count(desc):
    n = int(load().qhea_model_param_count(ctypes.byref(desc)))
    if n < 0:
        _check(n, 'qhea_model_param_count')
    return n


def model_forward(desc, branch, trunk, params, ham_diag=None, out=None):
    lib = load()
    B = branch.shape[0]
    _dev_f64(branch, 'branch', (B, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (B, desc.trunk_in))
    _dev_f64(params, 'params')
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    pred = out if out is not None else torch.empty(B, dtype=torch.float64, device=branch.device)
    ws = _model_ws(desc, B, branch.device)
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_forward(ctypes.byref(desc), B, _ptr(branch), _ptr(trunk), _ptr(params), _ptr(ham_diag),
                                    _ptr(pred), _ptr(ws), ws.numel(), _stream(branch.device))
    _check(rc, 'qhea_model_forward')
    return pred


def model_forward_chunks(desc, branch, trunk, params, chunk, ham_diag=None, out=None):
    """model_forward over all rows in chunks of `chunk` rows from ONE host call: one record preparation for all equal-sized
    chunks (qhea_model_forward_chunks); bitwise the chunk-by-chunk calls."""
    lib = load()
    N = branch.shape[0]
    _dev_f64(branch, 'branch', (N, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (N, desc.trunk_in))
    _dev_f64(params, 'params')
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    pred = out if out is not None else torch.empty(N, dtype=torch.float64, device=branch.device)
    if N == 0:
        return pred
    chunk = max(1, int(chunk))
    bounds = list(range(0, N, chunk)) + [N]
    ws = _model_ws(desc, min(chunk, N), branch.device)
    rb = (ctypes.c_int64 * len(bounds))(*bounds)
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_forward_chunks(ctypes.byref(desc), len(bounds) - 1, rb, _ptr(branch), _ptr(trunk), _ptr(params),
                                           _ptr(ham_diag), _ptr(pred), _ptr(ws), ws.numel(), _stream(branch.device))
    _check(rc, 'qhea_model_forward_chunks')
    return pred


def model_loss_grad(desc, branch, trunk, y, params, inv_batch_total, grad, ham_diag=None, pred=None):
    """Fills grad[P+2] = [d loss/d params | sse | sum y^2] for this shard; returns grad."""
    lib = load()
    B = branch.shape[0]
    _dev_f64(branch, 'branch', (B, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (B, desc.trunk_in))
    _dev_f64(y, 'y')
    if y.numel() != B:
        raise QheaError(f"y has {y.numel()} elements, expected {B}")
    _dev_f64(params, 'params')
    _dev_f64(grad, 'grad')
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    _dev_f64(pred, 'pred', (B,))
    ws = _model_ws(desc, B, branch.device)
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_loss_grad(ctypes.byref(desc), B, _ptr(branch), _ptr(trunk), _ptr(y), _ptr(params),
                                      _ptr(ham_diag), float(inv_batch_total), _ptr(grad), _ptr(pred),
                                      _ptr(ws), ws.numel(), _stream(branch.device))
    _check(rc, 'qhea_model_loss_grad')
    return grad


def model_train_step(desc, branch, trunk, y, params, inv_batch_total, grad, exp_avg, exp_avg_sq, step, lr, beta1,
                     beta2, eps, weight_decay, ham_diag=None, pred=None):
    """Single-device step: loss + gradients + Adam update of `params` in three launches; returns grad."""
    lib = load()
    B = branch.shape[0]
    _dev_f64(branch, 'branch', (B, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (B, desc.trunk_in))
    _dev_f64(y, 'y')
    if y.numel() != B:
        raise QheaError(f"y has {y.numel()} elements, expected {B}")
    for t, nm in ((params, 'params'), (grad, 'grad'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _dev_f64(t, nm)
    if not (exp_avg.numel() == exp_avg_sq.numel() == params.numel() and grad.numel() == params.numel() + 2):
        raise QheaError("model_train_step: flat vectors have inconsistent lengths")
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    _dev_f64(pred, 'pred', (B,))
    ws = _model_ws(desc, B, branch.device)
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_train_step(ctypes.byref(desc), B, _ptr(branch), _ptr(trunk), _ptr(y), _ptr(params),
                                       _ptr(ham_diag), float(inv_batch_total), _ptr(grad), _ptr(pred),
                                       _ptr(exp_avg), _ptr(exp_avg_sq), int(step), float(lr), float(beta1),
                                       float(beta2), float(eps), float(weight_decay), _ptr(ws), ws.numel(),
                                       _stream(branch.device))
    _check(rc, 'qhea_model_train_step')
    return grad


def model_train_steps(desc, bounds, global_batches, branch, trunk, y, params, rows, exp_avg, exp_avg_sq, first_step, lr,
                      beta1, beta2, eps, weight_decay, ham_diag=None):
    """
    One epoch's inner loop in one host call (qhea_model_train_steps): step i trains on rows bounds[i]:bounds[i+1] of the
    contiguous branch / trunk / y with residual weight 1 / global_batches[i] and leaves [grads | sse | sum y^2] in rows[i].
    """
    lib = load()
    n_steps = len(bounds) - 1
    if n_steps <= 0:
        return rows
    N = branch.shape[0]
    _dev_f64(branch, 'branch', (N, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (N, desc.trunk_in))
    _dev_f64(y, 'y')
    for t, nm in ((params, 'params'), (rows, 'rows'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _dev_f64(t, nm)
    P = params.numel()
    if y.numel() != N or bounds[-1] > N or len(global_batches) != n_steps:
        raise QheaError("model_train_steps: row bounds do not match the arrays")
    if rows.dim() != 2 or rows.shape[0] < n_steps or rows.shape[1] < P + 2 or exp_avg.numel() != P or exp_avg_sq.numel() != P:
        raise QheaError("model_train_steps: flat vectors have inconsistent lengths")
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    biggest = max(bounds[i + 1] - bounds[i] for i in range(n_steps))
    ws = _model_ws(desc, biggest, branch.device)
    rb = (ctypes.c_int64 * (n_steps + 1))(*[int(b) for b in bounds])
    ib = (ctypes.c_double * n_steps)(*[1.0 / float(g) for g in global_batches])
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_train_steps(ctypes.byref(desc), n_steps, rb, _ptr(branch), _ptr(trunk), _ptr(y), _ptr(params),
                                        _ptr(ham_diag), ib, _ptr(rows), int(rows.stride(0)), _ptr(exp_avg),
                                        _ptr(exp_avg_sq), int(first_step), float(lr), float(beta1), float(beta2),
                                        float(eps), float(weight_decay), _ptr(ws), ws.numel(), _stream(branch.device))
    _check(rc, 'qhea_model_train_steps')
    return rows


class Unsupported(QheaError):
    """QHEA_EUNSUPPORTED from a call that checks before it launches: the caller takes its other path."""


def model_dp_train_steps(desc, bounds, global_batches, branch, trunk, y, params, rows, exp_avg, exp_avg_sq, first_step, lr,
                         beta1, beta2, eps, weight_decay, rank, world, buffers, dp_values, first_seq, timeout_ms=5000.0,
                         ham_diag=None):
    """
    model_train_steps for `world` ranks with the sum over the ranks inside each step's reduce kernel
    (qhea_model_dp_train_steps): this rank's shard rows bounds[i]:bounds[i+1] per step, exchange numbers first_seq + i on the
    peer-mapped `buffers`; rows[i] receives the GLOBAL [grads | sse | sum y^2].  Raises Unsupported -- nothing launched --
    for an empty shard or a reduce grid that would not be resident at once.
    """
    lib = load()
    n_steps = len(bounds) - 1
    if n_steps <= 0:
        return rows
    N = branch.shape[0]
    _dev_f64(branch, 'branch', (N, desc.branch_in))
    if desc.model == MODEL_QUANONET:
        _dev_f64(trunk, 'trunk', (N, desc.trunk_in))
    _dev_f64(y, 'y')
    for t, nm in ((params, 'params'), (rows, 'rows'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _dev_f64(t, nm)
    P = params.numel()
    if y.numel() != N or bounds[-1] > N or len(global_batches) != n_steps:
        raise QheaError("model_dp_train_steps: row bounds do not match the arrays")
    if rows.dim() != 2 or rows.shape[0] < n_steps or rows.shape[1] < P + 2 or exp_avg.numel() != P or exp_avg_sq.numel() != P:
        raise QheaError("model_dp_train_steps: flat vectors have inconsistent lengths")
    _dev_f64(ham_diag, 'ham_diag', (1 << desc.n_qubits,))
    biggest = max(bounds[i + 1] - bounds[i] for i in range(n_steps))
    ws = _model_ws(desc, biggest, branch.device)
    rb = (ctypes.c_int64 * (n_steps + 1))(*[int(b) for b in bounds])
    ib = (ctypes.c_double * n_steps)(*[1.0 / float(g) for g in global_batches])
    arr = (ctypes.c_void_p * world)(*buffers)
    with torch.cuda.device(branch.device):
        rc = lib.qhea_model_dp_train_steps(ctypes.byref(desc), n_steps, rb, _ptr(branch), _ptr(trunk), _ptr(y), _ptr(params),
                                           _ptr(ham_diag), ib, _ptr(rows), int(rows.stride(0)), _ptr(exp_avg),
                                           _ptr(exp_avg_sq), int(first_step), float(lr), float(beta1), float(beta2),
                                           float(eps), float(weight_decay), int(rank), int(world), arr, int(dp_values),
                                           int(first_seq), float(timeout_ms), _ptr(ws), ws.numel(), _stream(branch.device))
    if rc == -2:
        raise Unsupported("qhea_model_dp_train_steps: empty shard or reduce grid not resident at once")
    _check(rc, 'qhea_model_dp_train_steps')
    return rows


def clock_probe(device, n_workgroups=1024, iters=200000):
    """In-kernel shader clock in MHz (median over `n_workgroups` one-wave workgroups timing a dependent fp64 FMA chain against
    the constant 100 MHz counter; qhea_clock_probe).  A diagnostic: the latency-bound kernels scale with it."""
    buf = torch.zeros(2 * n_workgroups, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _check(load().qhea_clock_probe(int(n_workgroups), int(iters), ctypes.c_void_p(buf.data_ptr()), _stream(device)),
               'qhea_clock_probe')
    t = buf.cpu().numpy().reshape(-1, 2).astype('float64')
    mhz = 100.0 * t[:, 0] / t[:, 1]
    import numpy as _np
    return float(_np.median(mhz)), float(mhz.min()), float(mhz.max())


def profile_next_circuit_kernel(start_event, stop_event):
    """Arm the measurement hook with two torch.cuda.Event(enable_timing=True) (already recorded once)."""
    rc = load().qhea_profile_next_circuit_kernel(ctypes.c_void_p(start_event.cuda_event),
                                                 ctypes.c_void_p(stop_event.cuda_event))
    _check(rc, 'qhea_profile_next_circuit_kernel')


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place Adam update of the flat fp64 parameter vector (one launch); grads may be a longer buffer."""
    n = params.numel()
    for t, nm in ((params, 'params'), (grads, 'grads'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _dev_f64(t, nm)
    if grads.numel() < n or exp_avg.numel() != n or exp_avg_sq.numel() != n:
        raise QheaError("adam_step: buffer sizes do not match the parameter vector")
    with torch.cuda.device(params.device):
        rc = load().qhea_adam_step(n, _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), int(step),
                                   float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   _stream(params.device))
    _check(rc, 'qhea_adam_step')


# ---- data-parallel exchange (include/quanonet_hea.h: qhea_dp_*) -------------------------------------------------
DP_MAX_RANKS, DP_HANDLE_BYTES = 16, 64


def dp_alloc(n_values, world, device):
    """This rank's exchange buffer (fine-grained device memory owned by the caller): its device address."""
    p = ctypes.c_void_p()
    with torch.cuda.device(device):
        _check(load().qhea_dp_alloc(int(n_values), int(world), ctypes.byref(p)), 'qhea_dp_alloc')
    return p.value


def dp_free(buf, device):
    with torch.cuda.device(device):
        _check(load().qhea_dp_free(buf), 'qhea_dp_free')


def dp_export(buf, device):
    h = ctypes.create_string_buffer(DP_HANDLE_BYTES)
    with torch.cuda.device(device):
        _check(load().qhea_dp_export(buf, h), 'qhea_dp_export')
    return h.raw


def dp_import(handle, device):
    p = ctypes.c_void_p()
    with torch.cuda.device(device):
        _check(load().qhea_dp_import(bytes(handle), ctypes.byref(p)), 'qhea_dp_import')
    return p.value


def dp_close(peer_buf, device):
    with torch.cuda.device(device):
        _check(load().qhea_dp_close(peer_buf), 'qhea_dp_close')


def dp_allreduce_adam(rank, world, buffers, seq, local, out, params=None, exp_avg=None, exp_avg_sq=None, step=1,
                      lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, timeout_ms=5000.0):
    """out = sum over ranks (in rank order) of every rank's `local`; Adam on `params` with out[:params.numel()]."""
    _dev_f64(local, 'local'); _dev_f64(out, 'out')
    n = local.numel()
    if out.numel() != n:
        raise QheaError("dp_allreduce_adam: local and out differ in size")
    npar = 0
    if params is not None:
        for t, nm in ((params, 'params'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
            _dev_f64(t, nm)
        npar = params.numel()
        if npar > n or exp_avg.numel() != npar or exp_avg_sq.numel() != npar:
            raise QheaError("dp_allreduce_adam: buffer sizes do not match the parameter vector")
    arr = (ctypes.c_void_p * world)(*buffers)
    with torch.cuda.device(local.device):
        rc = load().qhea_dp_allreduce_adam(
            int(rank), int(world), arr, n, int(seq), _ptr(local), _ptr(out), npar,
            _ptr(params) if params is not None else None, _ptr(exp_avg) if params is not None else None,
            _ptr(exp_avg_sq) if params is not None else None, int(step), float(lr), float(beta1), float(beta2),
            float(eps), float(weight_decay), float(timeout_ms), _stream(local.device))
    _check(rc, 'qhea_dp_allreduce_adam')


def dp_status(buf, device):
    """Waits for the device's current stream; raises QheaError(QHEA_EEXCHANGE) if an exchange timed out."""
    with torch.cuda.device(device):
        _check(load().qhea_dp_status(buf, _stream(device)), 'qhea_dp_status')
