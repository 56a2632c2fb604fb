"""
QuanONetPT / HEAQNNPT with the HIP quantum layer: host-side mirror of the reference's
core/models_pt.py:14-213 (same constructor arguments, same state_dict keys, same forward
semantics), so solver code and checkpoints written against the reference keep working.

state_dict keys:
  QuanONetPT: branch_freq.weights, branch_freq.bias, trunk_freq.weights, trunk_freq.bias,
              quantum_layer.ansatz_weights, bias           (+ quantum_layer.ham_diag buffer)
  HEAQNNPT:   freq.weights, freq.bias, quantum_layer.ansatz_weights
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .circuit import build_quanonet_hip, build_heaqnn_hip

HIP_BACKENDS = ('hip', 'mi355x', 'torchquantum')   # 'torchquantum' accepted as an alias: drop-in


class _TiledElementWise(nn.Module):
    """y[:,k] = x[:, k mod in] * weights[k] + bias[k]   (core/models_pt.py:14-41)."""

    def __init__(self, in_features, out_features, init_scale=0.1, dtype=torch.float64):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.repeats = int(np.ceil(out_features / in_features))
        self.weights = nn.Parameter(torch.full((out_features,), float(init_scale), dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=dtype))

    def forward(self, x):
        tiled = x.repeat(1, self.repeats)[:, :self.out_features]
        return tiled * self.weights + self.bias


class _ScaleRepeat(nn.Module):
    """y[:,k] = scale * x[:, k mod in]   (core/models_pt.py:44-68)."""

    def __init__(self, in_features, out_features, scale=0.01):
        super().__init__()
        self.scale = scale
        self.in_features = in_features
        self.out_features = out_features
        self.repeats = int(np.ceil(out_features / in_features))

    def forward(self, x):
        return (x * self.scale).repeat(1, self.repeats)[:, :self.out_features]


def _build_quantum_layer(quantum_backend, num_qubits, total_input_size, net_size, ham_bound, ham_diag,
                         branch_input_size=None, trunk_input_size=None, dtype=torch.float64, ham_pauli='Z'):
    """String-keyed plug-in dispatch (core/models_pt.py:71-100); only the HIP backend lives here."""
    if quantum_backend in HIP_BACKENDS:
        if branch_input_size is not None:
            return build_quanonet_hip(num_qubits, branch_input_size, trunk_input_size, net_size,
                                      ham_bound=ham_bound, ham_diag=ham_diag, dtype=dtype, ham_pauli=ham_pauli)
        return build_heaqnn_hip(num_qubits, total_input_size, net_size,
                                ham_bound=ham_bound, ham_diag=ham_diag, dtype=dtype, ham_pauli=ham_pauli)
    raise ValueError(f"Unknown quantum_backend for PyTorch models: '{quantum_backend}'")


class QuanONetPT(nn.Module):
    """out = Q(cat[T(trunk), Br(branch)]) + bias   (core/models_pt.py:103-166)."""

    def __init__(self, num_qubits, branch_input_size, trunk_input_size, net_size,
                 scale_coeff=1.0, if_trainable_freq=False, quantum_backend='hip',
                 ham_bound=(-5.0, 5.0), ham_diag=None, dtype=torch.float64, ham_pauli='Z'):
        super().__init__()
        branch_depth, branch_linear_depth, trunk_depth, trunk_linear_depth = net_size
        self.num_qubits = num_qubits
        self.net_size = tuple(net_size)
        self.if_trainable_freq = if_trainable_freq
        self.branch_enc_size = branch_depth * num_qubits
        self.trunk_enc_size = trunk_depth * num_qubits
        if if_trainable_freq:
            self.branch_freq = _TiledElementWise(branch_input_size, self.branch_enc_size, scale_coeff, dtype)
            self.trunk_freq = _TiledElementWise(trunk_input_size, self.trunk_enc_size, scale_coeff, dtype)
        else:
            self.branch_freq = _ScaleRepeat(branch_input_size, self.branch_enc_size, scale_coeff)
            self.trunk_freq = _ScaleRepeat(trunk_input_size, self.trunk_enc_size, scale_coeff)
        self.quantum_layer = _build_quantum_layer(
            quantum_backend, num_qubits,
            total_input_size=self.trunk_enc_size + self.branch_enc_size,
            net_size=net_size, ham_bound=ham_bound, ham_diag=ham_diag,
            branch_input_size=branch_input_size, trunk_input_size=trunk_input_size, dtype=dtype,
            ham_pauli=ham_pauli)
        self.bias = nn.Parameter(torch.zeros(1, dtype=dtype))

    def fused_desc(self):
        """Descriptor for the model-level C ABI (qhea_model_*); parameter order == self.parameters()."""
        q = self.quantum_layer
        return _lib.make_model_desc(_lib.MODEL_QUANONET, self.num_qubits, self.net_size,
                                    self.branch_freq.in_features, self.trunk_freq.in_features,
                                    self.if_trainable_freq, getattr(self.branch_freq, 'scale', 0.0),
                                    q.ham_offset, q.ham_coeff, getattr(q, 'ham_pauli', 0))

    def forward(self, branch_input, trunk_input):
        branch_enc = self.branch_freq(branch_input)
        trunk_enc = self.trunk_freq(trunk_input)
        x = torch.cat([trunk_enc, branch_enc], dim=1)      # trunk first (models_pt.py:164)
        return self.quantum_layer(x) + self.bias


class HEAQNNPT(nn.Module):
    """out = Q(F(x)), no bias   (core/models_pt.py:169-213)."""

    def __init__(self, num_qubits, input_size, net_size, scale_coeff=1.0, if_trainable_freq=False,
                 quantum_backend='hip', ham_bound=(-5.0, 5.0), ham_diag=None, dtype=torch.float64,
                 ham_pauli='Z'):
        super().__init__()
        depth = net_size[0]
        enc_size = depth * num_qubits
        self.num_qubits = num_qubits
        self.net_size = tuple(net_size)
        self.if_trainable_freq = if_trainable_freq
        if if_trainable_freq:
            self.freq = _TiledElementWise(input_size, enc_size, scale_coeff, dtype)
        else:
            self.freq = _ScaleRepeat(input_size, enc_size, scale_coeff)
        self.quantum_layer = _build_quantum_layer(
            quantum_backend, num_qubits, total_input_size=enc_size,
            net_size=net_size, ham_bound=ham_bound, ham_diag=ham_diag, dtype=dtype, ham_pauli=ham_pauli)

    def fused_desc(self):
        q = self.quantum_layer
        return _lib.make_model_desc(_lib.MODEL_HEAQNN, self.num_qubits, self.net_size[:2], self.freq.in_features, 0,
                                    self.if_trainable_freq, getattr(self.freq, 'scale', 0.0),
                                    q.ham_offset, q.ham_coeff, getattr(q, 'ham_pauli', 0))

    def forward(self, x):
        return self.quantum_layer(self.freq(x))
