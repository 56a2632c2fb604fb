"""
quantum_circuits_hip.py -- the file a maintainer of the reference drops into ``core/`` to get a ``'hip'`` quantum
back-end: a ctypes binding of ``include/quanonet_hea.h`` (``libquanonet_hea.so``) behind the reference's quantum-layer
plug-in contract (``_build_quantum_layer``, core/models_pt.py:71-100).

Self-contained on purpose: it imports nothing from this repository's ``quanonet_amd`` package and nothing from the
reference's ``core`` package (the two small shape helpers of core/quantum_circuits_tq.py:130-146 are restated below),
so it can be copied as is.  The library is looked up in ``$QHEA_LIB``, next to this file, and in the loader's path.

Surface (same as ``_TQHEACircuit``, core/quantum_circuits_tq.py:20-127):
  * ``forward(x: Tensor[B, E]) -> Tensor[B, 1]``, differentiable w.r.t. ``x`` and the parameter;
  * one ``nn.Parameter`` ``ansatz_weights`` of shape (blocks, 3, n_wires), float32, U(-pi, pi) from torch's CPU
    generator (so ``torch.manual_seed`` reproduces the reference's initial weights);
  * optional float32 buffer ``ham_diag``; attributes ``n_wires``, ``block_configs``, ``use_full_ham``;
  * builders ``build_quanonet_hip`` / ``build_heaqnn_hip`` with the signatures of ``build_quanonet_tq`` /
    ``build_heaqnn_tq`` (core/quantum_circuits_tq.py:149-202).
Parameters and activations stay float32 as in the reference's training loop (solvers/solver_pt.py:132); the op
computes in fp64 on the device and casts back.  A CPU tensor raises: the library has no CPU path.

The three edits in the reference that go with it are listed in INTEGRATION.md.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

_I32P, _VP = ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        for cand in (os.environ.get('QHEA_LIB'), os.path.join(here, 'libquanonet_hea.so'), 'libquanonet_hea.so'):
            if not cand:
                continue
            try:
                lib = ctypes.CDLL(cand)
                break
            except OSError:
                continue
        else:
            raise ImportError("libquanonet_hea.so is required but was not found (set QHEA_LIB or build it with "
                              "`make -C quanonet_amd/csrc`)")
        lib.qhea_strerror.restype = ctypes.c_char_p
        lib.qhea_strerror.argtypes = [ctypes.c_int]
        lib.qhea_workspace_bytes.restype = ctypes.c_size_t
        lib.qhea_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, _I32P, _I32P, ctypes.c_int64]
        lib.qhea_forward.restype = ctypes.c_int
        lib.qhea_forward.argtypes = [ctypes.c_int, ctypes.c_int, _I32P, _I32P, ctypes.c_int64, _VP, _VP,
                                     ctypes.c_double, ctypes.c_double, _VP, ctypes.c_int, _VP, _VP, _VP,
                                     ctypes.c_size_t, _VP]
        lib.qhea_backward.restype = ctypes.c_int
        lib.qhea_backward.argtypes = [ctypes.c_int, ctypes.c_int, _I32P, _I32P, ctypes.c_int64, _VP, _VP,
                                      ctypes.c_double, ctypes.c_double, _VP, ctypes.c_int, _VP, _VP, _VP, _VP, _VP,
                                      _VP, ctypes.c_size_t, _VP]
        _LIB = lib
    return _LIB


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {_lib().qhea_strerror(rc).decode()} ({rc})")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _HEAOp(torch.autograd.Function):
    """out[B] = <psi(x, w)| H |psi(x, w)> through qhea_forward; gradients through qhea_backward (adjoint method)."""

    @staticmethod
    def forward(ctx, x, w, layer):
        if not x.is_cuda:
            raise RuntimeError("the 'hip' quantum back-end needs its inputs on a HIP device")
        x64 = x.detach().to(torch.float64).contiguous()
        w64 = w.detach().to(torch.float64).contiguous()
        diag = layer.ham_diag.detach().to(torch.float64).contiguous() if layer.use_full_ham else None
        batch = x64.shape[0]
        out = torch.empty(batch, dtype=torch.float64, device=x.device)
        keep = x.requires_grad or w.requires_grad
        state = torch.empty((batch, 1 << layer.n_wires, 2), dtype=torch.float64, device=x.device) if keep else None
        ws = layer._workspace(batch, x.device)
        with torch.cuda.device(x.device):
            rc = _lib().qhea_forward(layer.n_wires, layer._nb, layer._enc, layer._ld, batch, _p(x64), _p(w64),
                                     layer.ham_offset, layer.ham_coeff, _p(diag), 0, _p(out), _p(state), _p(ws),
                                     ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _ok(rc, 'qhea_forward')
        if keep:
            ctx.save_for_backward(x64, w64, state)
            ctx.layer, ctx.diag, ctx.dtypes = layer, diag, (x.dtype, w.dtype)
        return out.to(x.dtype).unsqueeze(-1)

    @staticmethod
    def backward(ctx, grad_out):
        x64, w64, state = ctx.saved_tensors
        layer = ctx.layer
        g = grad_out.detach().to(torch.float64).reshape(-1).contiguous()
        gx, gw = torch.empty_like(x64), torch.empty_like(w64)
        ws = layer._workspace(x64.shape[0], x64.device)
        with torch.cuda.device(x64.device):
            rc = _lib().qhea_backward(layer.n_wires, layer._nb, layer._enc, layer._ld, x64.shape[0], _p(x64), _p(w64),
                                      layer.ham_offset, layer.ham_coeff, _p(ctx.diag), 0, _p(g), _p(state), None,
                                      _p(gx), _p(gw), _p(ws), ws.numel(),
                                      ctypes.c_void_p(torch.cuda.current_stream(x64.device).cuda_stream))
        _ok(rc, 'qhea_backward')
        return gx.to(ctx.dtypes[0]), gw.to(ctx.dtypes[1]), None


class _HIPHEACircuit(nn.Module):
    def __init__(self, n_wires, block_configs, ham_offset=0.0, ham_coeff_per_qubit=0.0, ham_diag=None):
        super().__init__()
        _lib()                                              # a missing library fails here, like a missing simulator
        self.n_wires = n_wires
        self.block_configs = [(int(e), int(d)) for e, d in block_configs]
        self._nb = len(self.block_configs)
        self._enc = (ctypes.c_int32 * max(self._nb, 1))(*[e for e, _ in self.block_configs])
        self._ld = (ctypes.c_int32 * max(self._nb, 1))(*[d for _, d in self.block_configs])
        self._n_angles = sum(e for e, _ in self.block_configs)
        self.ansatz_weights = nn.Parameter(torch.empty(sum(d for _, d in self.block_configs), 3, n_wires))
        nn.init.uniform_(self.ansatz_weights, -math.pi, math.pi)
        if ham_diag is not None:
            self.register_buffer('ham_diag', torch.tensor(ham_diag, dtype=torch.float32))
            self.use_full_ham, self.ham_offset, self.ham_coeff = True, 0.0, 0.0
        else:
            self.use_full_ham, self.ham_offset, self.ham_coeff = False, float(ham_offset), float(ham_coeff_per_qubit)
        self._ws = None

    def _workspace(self, batch, device):
        need = _lib().qhea_workspace_bytes(self.n_wires, self._nb, self._enc, self._ld, batch)
        if need == 0:
            raise RuntimeError("qhea_workspace_bytes rejected the circuit shape")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, x):
        if x.dim() != 2 or x.shape[1] != self._n_angles:
            raise ValueError(f"expected input of shape (batch, {self._n_angles}), got {tuple(x.shape)}")
        return _HEAOp.apply(x, self.ansatz_weights, self)


def _simple_hamiltonian(num_qubits, lower, upper):
    """H = offset + coeff * sum_i Z_i with spectrum [lower, upper] (core/quantum_circuits_tq.py:141-146)."""
    span = upper - lower
    return lower + 0.5 * span, 0.5 * span / num_qubits


def _layer(num_qubits, blocks, ham_bound, ham_diag):
    if ham_diag is not None:
        return _HIPHEACircuit(num_qubits, blocks, ham_diag=ham_diag)
    offset, coeff = _simple_hamiltonian(num_qubits, ham_bound[0], ham_bound[1])
    return _HIPHEACircuit(num_qubits, blocks, ham_offset=offset, ham_coeff_per_qubit=coeff)


def build_quanonet_hip(num_qubits, branch_input_size, trunk_input_size, net_size, ham_bound=(-5.0, 5.0), ham_diag=None):
    """QuanONet circuit: trunk blocks first, then branch blocks (core/quantum_circuits_tq.py:130-138, 149-176)."""
    branch_depth, branch_linear_depth, trunk_depth, trunk_linear_depth = net_size
    blocks = [(num_qubits, trunk_linear_depth)] * trunk_depth + [(num_qubits, branch_linear_depth)] * branch_depth
    return _layer(num_qubits, blocks, ham_bound, ham_diag)


def build_heaqnn_hip(num_qubits, input_size, net_size, ham_bound=(-5.0, 5.0), ham_diag=None):
    """HEAQNN circuit: net_size[0] blocks of net_size[1] sub-layers (core/quantum_circuits_tq.py:179-202)."""
    return _layer(num_qubits, [(num_qubits, net_size[1])] * net_size[0], ham_bound, ham_diag)
