"""
ctypes loader for oracle/libhea_oracle.so (the C restatement, OpenMP over the batch).
TEST INFRASTRUCTURE ONLY -- see oracle/hea_oracle.c.  Same call shape as the numpy oracle.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'libhea_oracle.so')
    src = os.path.join(_HERE, 'hea_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libhea_oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'libhea_oracle.so')
        if not os.path.exists(so):
            so = build()
        _LIB = ctypes.CDLL(so)
        d = ctypes.POINTER(ctypes.c_double)
        i32 = ctypes.POINTER(ctypes.c_int32)
        _LIB.qhea_oracle_forward.restype = ctypes.c_int
        _LIB.qhea_oracle_forward.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, ctypes.c_int64,
                                             d, d, ctypes.c_double, ctypes.c_double, d, ctypes.c_int, d, d]
        _LIB.qhea_oracle_backward.restype = ctypes.c_int
        _LIB.qhea_oracle_backward.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, ctypes.c_int64,
                                              d, d, ctypes.c_double, ctypes.c_double, d, ctypes.c_int, d, d, d, d]
        _LIB.qhea_oracle_threads.restype = ctypes.c_int
        _LIB.qhea_oracle_set_threads.restype = ctypes.c_int
        _LIB.qhea_oracle_set_threads.argtypes = [ctypes.c_int]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _cfg(block_configs):
    enc = np.ascontiguousarray([c[0] for c in block_configs], dtype=np.int32)
    ld = np.ascontiguousarray([c[1] for c in block_configs], dtype=np.int32)
    i32 = ctypes.POINTER(ctypes.c_int32)
    return enc, ld, enc.ctypes.data_as(i32), ld.ctypes.data_as(i32)


def _pauli(p):
    return {'Z': 0, 'X': 1, 'Y': 2}[p.upper()] if isinstance(p, str) else int(p)


def threads():
    return lib().qhea_oracle_threads()


def set_threads(n):
    """OpenMP threads of the following calls; returns the previous maximum."""
    return lib().qhea_oracle_set_threads(int(n))


def _fit_columns(x, block_configs):
    """The reference applies an encoding gate only while its column cursor is inside x (core/quantum_circuits_tq.py:83):
    missing columns are skipped gates (= zero angles, RX(0) = 1), surplus columns are never read.  The C routine takes
    exactly E columns."""
    E = sum(c[0] for c in block_configs)
    x = np.asarray(x, dtype=np.float64)
    if x.shape[1] > E:
        x = x[:, :E]
    elif x.shape[1] < E:
        x = np.concatenate([x, np.zeros((x.shape[0], E - x.shape[1]))], axis=1)
    return np.ascontiguousarray(x)


def hea_forward(num_qubits, block_configs, x, w, offset=0.0, coeff=1.0, ham_diag=None,
                return_state=False, ham_pauli='Z'):
    x = _fit_columns(x, block_configs)
    w = np.ascontiguousarray(w, dtype=np.float64)
    diag = None if ham_diag is None else np.ascontiguousarray(ham_diag, dtype=np.float64)
    B = x.shape[0]
    enc, ld, pe, pl = _cfg(block_configs)
    out = np.empty(B)
    st = np.empty((B, 1 << num_qubits, 2)) if return_state else None
    rc = lib().qhea_oracle_forward(num_qubits, len(block_configs), pe, pl, B, _p(x), _p(w),
                                   float(offset), float(coeff), _p(diag), _pauli(ham_pauli), _p(out), _p(st))
    if rc:
        raise ValueError(f"qhea_oracle_forward failed ({rc})")
    return (out, st) if return_state else out


def hea_backward(num_qubits, block_configs, x, w, g, offset=0.0, coeff=1.0, ham_diag=None, ham_pauli='Z'):
    width = np.asarray(x).shape[1]
    x = _fit_columns(x, block_configs)
    w = np.ascontiguousarray(w, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1)
    diag = None if ham_diag is None else np.ascontiguousarray(ham_diag, dtype=np.float64)
    B = x.shape[0]
    enc, ld, pe, pl = _cfg(block_configs)
    out = np.empty(B)
    gx = np.zeros_like(x)
    gw = np.zeros_like(w)
    rc = lib().qhea_oracle_backward(num_qubits, len(block_configs), pe, pl, B, _p(x), _p(w),
                                    float(offset), float(coeff), _p(diag), _pauli(ham_pauli), _p(g),
                                    _p(out), _p(gx), _p(gw))
    if rc:
        raise ValueError(f"qhea_oracle_backward failed ({rc})")
    if width != x.shape[1]:                      # gradient in the caller's own width (surplus columns: zero)
        full = np.zeros((B, width))
        full[:, :min(width, x.shape[1])] = gx[:, :min(width, x.shape[1])]
        gx = full
    return out, gx, gw
