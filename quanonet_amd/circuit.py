"""
HIP quantum layer: host-side mirror of the reference's TorchQuantum layer
(core/quantum_circuits_tq.py:20-202) on top of the C ABI (include/quanonet_hea.h).

Same surface as the reference module returned by ``_build_quantum_layer``
(core/models_pt.py:71-100; contract in SURVEY.md section 8b):
  * ``forward(x: Tensor[B,E]) -> Tensor[B,1]``, differentiable w.r.t. ``x`` and the parameter
  * one ``nn.Parameter`` named ``ansatz_weights`` of shape (blk,3,n), init U(-pi,pi) drawn from the
    CPU torch generator at construction (so ``torch.manual_seed(s)`` reproduces the reference's
    initial weights bit-for-bit in float32, core/quantum_circuits_tq.py:50-53)
  * optional buffer ``ham_diag`` (2^n,)
  * stateless between calls, follows ``.to(device)`` of the parent
The arithmetic runs in fp64 on the GPU whatever the parameter dtype (north-star tolerance 1e-10
against the fp64 oracle).  There is no CPU fallback: a CPU tensor raises.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _make_block_configs(num_qubits, trunk_depth, trunk_linear_depth, branch_depth, branch_linear_depth):
    """Trunk blocks first, then branch blocks (core/quantum_circuits_tq.py:130-138)."""
    return [(num_qubits, trunk_linear_depth)] * trunk_depth + [(num_qubits, branch_linear_depth)] * branch_depth


def _ham_params(num_qubits, lower_bound=-5.0, upper_bound=5.0):
    """(offset, coeff_per_qubit) of H = offset + coeff * sum_i Z_i (core/quantum_circuits_tq.py:141-146)."""
    coff = upper_bound - lower_bound
    return lower_bound + coff / 2.0, coff / 2.0 / num_qubits


class _HEAFunction(torch.autograd.Function):
    """out[B] = <psi(x,w)|H|psi(x,w)>; backward = in-kernel adjoint differentiation."""

    @staticmethod
    def forward(ctx, x, w, shape, ham_offset, ham_coeff, ham_diag, ham_pauli=0):
        x64 = x.detach().to(torch.float64).contiguous()
        w64 = w.detach().to(torch.float64).contiguous()
        d64 = None if ham_diag is None else ham_diag.detach().to(torch.float64).contiguous()
        need_grad = x.requires_grad or w.requires_grad
        if need_grad:
            out, state = _lib.hea_forward(shape, x64, w64, ham_offset, ham_coeff, d64, return_state=True,
                                          ham_pauli=ham_pauli)
            ctx.save_for_backward(x64, w64, state, d64 if d64 is not None else x64.new_empty(0))
        else:
            out = _lib.hea_forward(shape, x64, w64, ham_offset, ham_coeff, d64, ham_pauli=ham_pauli)
        ctx.shape = shape
        ctx.ham = (ham_offset, ham_coeff, d64 is not None, ham_pauli)
        ctx.dtypes = (x.dtype, w.dtype)
        return out.to(torch.promote_types(x.dtype, w.dtype))

    @staticmethod
    def backward(ctx, grad_out):
        x64, w64, state, d64 = ctx.saved_tensors
        off, co, has_diag, pauli = ctx.ham
        g = grad_out.detach().to(torch.float64).contiguous()
        gx, gw = _lib.hea_backward(ctx.shape, x64, w64, g, off, co, d64 if has_diag else None, state=state,
                                   ham_pauli=pauli)
        xd, wd = ctx.dtypes
        return gx.to(xd), gw.to(wd), None, None, None, None, None


class HEACircuitHIP(nn.Module):
    """
    Drop-in for the reference's ``_TQHEACircuit`` (core/quantum_circuits_tq.py:20-127).

    Args mirror the reference constructor; ``dtype`` selects the parameter dtype
    (float32 like the reference, or float64 for the north-star fp64 training path).
    """

    def __init__(self, n_wires, block_configs, ham_offset=0.0, ham_coeff_per_qubit=0.0, ham_diag=None,
                 dtype=torch.float64, ham_pauli='Z'):
        super().__init__()
        if n_wires < 2:
            # MindQuantum skips the entangler for n=1 (quantum_circuits_ms.py:140) while TorchQuantum
            # would issue CNOT(0,0): undefined, so reject.
            raise ValueError("HEACircuitHIP needs at least 2 qubits")
        _lib.load()                                   # fail loudly at construction if the .so is missing
        self.n_wires = int(n_wires)
        self.block_configs = [(int(a), int(b)) for a, b in block_configs]
        self._shape = _lib.CircuitShape(self.n_wires, self.block_configs)
        total_ansatz_blocks = self._shape.blk
        w32 = torch.empty(total_ansatz_blocks, 3, self.n_wires)          # float32 draw == reference draw
        nn.init.uniform_(w32, -np.pi, np.pi)
        self.ansatz_weights = nn.Parameter(w32.to(dtype))
        # read-out Pauli of the simple Hamiltonian (generate_simple_hamiltonian's `pauli`,
        # core/quantum_circuits_ms.py:28-39; the reference's PT back-ends only have Z)
        self.ham_pauli = _lib.pauli_code(ham_pauli)
        if ham_diag is not None and self.ham_pauli != 0:
            raise ValueError("ham_diag is a Z-basis diagonal: it overrides ham_pauli (utils/common.py:84); pass 'Z'")
        if ham_diag is not None:
            self.register_buffer('ham_diag', torch.as_tensor(np.asarray(ham_diag), dtype=dtype).reshape(-1))
            if self.ham_diag.numel() != 1 << self.n_wires:
                raise ValueError("ham_diag must have 2**n_wires entries")
            self.use_full_ham = True
            self.ham_offset = 0.0
            self.ham_coeff = 0.0
        else:
            self.ham_offset = float(ham_offset)
            self.ham_coeff = float(ham_coeff_per_qubit)
            self.use_full_ham = False

    @property
    def total_encode_params(self):
        return self._shape.E

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError(f"expected x of shape (batch, {self._shape.E}), got {tuple(x.shape)}")
        # The reference walks a column cursor over x and applies an encoding gate only while the cursor is inside x
        # (`if param_col < x.shape[1]`, core/quantum_circuits_tq.py:83): missing columns are skipped gates, surplus
        # columns are never read.  RX(0) is the identity, so a skipped gate is a zero angle; the C ABI itself takes
        # exactly E columns (include/quanonet_hea.h: qhea_forward).
        E = self._shape.E
        if x.shape[1] > E:
            x = x[:, :E]
        elif x.shape[1] < E:
            x = torch.nn.functional.pad(x, (0, E - x.shape[1]))
        diag = self.ham_diag if self.use_full_ham else None
        out = _HEAFunction.apply(x, self.ansatz_weights, self._shape, self.ham_offset, self.ham_coeff, diag,
                                 self.ham_pauli)
        return out.unsqueeze(-1)


def build_quanonet_hip(num_qubits, branch_input_size, trunk_input_size, net_size,
                       ham_bound=(-5.0, 5.0), ham_diag=None, dtype=torch.float64, ham_pauli='Z'):
    """Mirror of build_quanonet_tq (core/quantum_circuits_tq.py:149-176)."""
    bd, bl, td, tl = net_size
    cfgs = _make_block_configs(num_qubits, td, tl, bd, bl)
    if ham_diag is not None:
        return HEACircuitHIP(num_qubits, cfgs, ham_diag=ham_diag, dtype=dtype)
    off, co = _ham_params(num_qubits, ham_bound[0], ham_bound[1])
    return HEACircuitHIP(num_qubits, cfgs, ham_offset=off, ham_coeff_per_qubit=co, dtype=dtype,
                         ham_pauli=ham_pauli)


def build_heaqnn_hip(num_qubits, input_size, net_size, ham_bound=(-5.0, 5.0), ham_diag=None,
                     dtype=torch.float64, ham_pauli='Z'):
    """Mirror of build_heaqnn_tq (core/quantum_circuits_tq.py:179-202)."""
    cfgs = [(num_qubits, net_size[1])] * net_size[0]
    if ham_diag is not None:
        return HEACircuitHIP(num_qubits, cfgs, ham_diag=ham_diag, dtype=dtype)
    off, co = _ham_params(num_qubits, ham_bound[0], ham_bound[1])
    return HEACircuitHIP(num_qubits, cfgs, ham_offset=off, ham_coeff_per_qubit=co, dtype=dtype,
                         ham_pauli=ham_pauli)
