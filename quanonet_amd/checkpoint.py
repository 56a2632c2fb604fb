"""
Checkpoint / weight-format interop.

* ``read_mindspore_ckpt``: dependency-free reader for MindSpore ``.ckpt`` files
  (protobuf ``Checkpoint{repeated Value value=1}``, ``Value{string tag=1;
  TensorProto tensor=2}``, ``TensorProto{repeated int64 dims=1; string
  tensor_type=2; bytes tensor_content=3}``).  Nothing in the file is executed.
* ``ms_to_pt_state``: MindSpore key names -> QuanONetPT / HEAQNNPT state_dict keys
  with the ``(P,) -> (blk,3,n)`` reshape.  Mapping follows the reference's
  ``utils/weight_transfer.py:37-98`` (trunk sub-layers first in the flat vector).
* ``parse_experiment_dir``: hyper-parameters from the experiment directory name,
  same grammar as the reference's ``infer.py:37-86`` / ``utils/logger.py:55-118``.
"""
import os
import re
import numpy as np

_DTYPES = {
    'Float32': np.dtype('<f4'), 'Float64': np.dtype('<f8'), 'Float16': np.dtype('<f2'),
    'Int32': np.dtype('<i4'), 'Int64': np.dtype('<i8'),
}


def _varint(buf, pos):
    val = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not (b & 0x80):
            return val, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one protobuf message."""
    pos = 0
    end = len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def read_mindspore_ckpt(path):
    """Return {tag: ndarray} for every tensor in a MindSpore .ckpt file."""
    with open(path, 'rb') as f:
        buf = f.read()
    out = {}
    for fno, wt, val in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        tag = None
        tensor = None
        for f2, w2, v2 in _fields(val):
            if f2 == 1 and w2 == 2:
                tag = bytes(v2).decode('utf-8')
            elif f2 == 2 and w2 == 2:
                tensor = v2
        if tag is None or tensor is None:
            continue
        dims = []
        ttype = 'Float32'
        content = b''
        for f3, w3, v3 in _fields(tensor):
            if f3 == 1:
                if w3 == 0:
                    dims.append(v3)
                else:                      # packed repeated int64
                    p = 0
                    while p < len(v3):
                        d, p = _varint(v3, p)
                        dims.append(d)
            elif f3 == 2:
                ttype = bytes(v3).decode('utf-8')
            elif f3 == 3:
                content = bytes(v3)
        if ttype not in _DTYPES:
            raise ValueError(f"{path}: tensor '{tag}' has unsupported type {ttype}")
        arr = np.frombuffer(content, dtype=_DTYPES[ttype]).copy()
        shape = [d for d in dims if d > 0]
        if shape and int(np.prod(shape)) == arr.size:
            arr = arr.reshape(shape)
        out[tag] = arr
    return out


def load_weight_file(path):
    """Load .ckpt (MindSpore protobuf) / .npz / .pt into {name: ndarray}."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.ckpt':
        return read_mindspore_ckpt(path)
    if ext == '.npz':
        d = np.load(path, allow_pickle=False)
        return {k: d[k] for k in d.files}
    if ext in ('.pt', '.pth'):
        import torch
        sd = torch.load(path, map_location='cpu', weights_only=True)
        return {k: v.detach().cpu().numpy() for k, v in sd.items()}
    raise ValueError(f"unknown checkpoint extension: {path}")


_MS_QUANONET = {
    'bias': 'bias',
    'branch_LinearLayer.Net2.weights': 'branch_freq.weights',
    'branch_LinearLayer.Net2.bias': 'branch_freq.bias',
    'trunk_LinearLayer.Net2.weights': 'trunk_freq.weights',
    'trunk_LinearLayer.Net2.bias': 'trunk_freq.bias',
}
_MS_HEAQNN = {
    'LinearLayer.Net2.weights': 'freq.weights',
    'LinearLayer.Net2.bias': 'freq.bias',
}


def is_mindspore_state(state):
    return 'QuanONet.weight' in state or 'HEAQNN.weight' in state


def ms_to_pt_state(state, num_qubits, net_size, model_type='QuanONet', dtype=np.float64):
    """
    MindSpore-named arrays -> PT-named arrays (utils/weight_transfer.py:46-98).
    Frequency-layer keys are mapped when present (trainable-frequency checkpoints).
    """
    n = int(num_qubits)
    out = {}
    if model_type == 'QuanONet':
        bd, bl, td, tl = net_size
        blk = bd * bl + td * tl
        raw = np.asarray(state['QuanONet.weight'], dtype=dtype).reshape(-1)
        names = _MS_QUANONET
    elif model_type == 'HEAQNN':
        blk = net_size[0] * net_size[1]
        raw = np.asarray(state['HEAQNN.weight'], dtype=dtype).reshape(-1)
        names = _MS_HEAQNN
    else:
        raise ValueError(f"unsupported model_type {model_type}")
    if raw.size != blk * 3 * n:
        raise ValueError(f"circuit weight has {raw.size} elements but expected {blk * 3 * n} "
                         f"({blk}x3x{n}); check net_size and num_qubits")
    out['quantum_layer.ansatz_weights'] = raw.reshape(blk, 3, n)
    for ms_key, pt_key in names.items():
        if ms_key in state:
            v = np.asarray(state[ms_key], dtype=dtype)
            out[pt_key] = v.reshape(1) if pt_key == 'bias' else v.reshape(-1)
    return out


def pt_to_ms_state(state, model_type='QuanONet'):
    """Inverse of ms_to_pt_state (flat circuit vector, MindSpore key names)."""
    inv = {v: k for k, v in (_MS_QUANONET if model_type == 'QuanONet' else _MS_HEAQNN).items()}
    out = {}
    for k, v in state.items():
        v = np.asarray(v)
        if k == 'quantum_layer.ansatz_weights':
            out[f'{model_type}.weight'] = v.reshape(-1)
        elif k in inv:
            out[inv[k]] = v.reshape(()) if k == 'bias' else v
    return out


_NUMBER = re.compile(r'-?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?')


def _signed_list(text):
    """'-5-5' / '0.5--1' / '1e-05-2' -> numbers, or None when the text is not such a list.  The writer joins str(v)
    with '-' (utils/logger.py:98-103): a number may carry its own sign and an exponent ('1e-05'), so the text is read
    number by number with one separating dash between two of them, not split on dashes."""
    vals, pos = [], 0
    while pos < len(text):
        if vals:
            if text[pos] != '-':
                return None
            pos += 1
        m = _NUMBER.match(text, pos)
        if m is None:
            return None
        vals.append(float(m.group()))
        pos = m.end()
    return vals or None


_BACKEND_TOKENS = {'TQ': 'torchquantum', 'Qiskit': 'qiskit', 'PL': 'pennylane', 'HIP': 'hip',
                   'torchquantum': 'torchquantum', 'qiskit': 'qiskit', 'pennylane': 'pennylane', 'hip': 'hip'}


def parse_experiment_dir(path):
    """
    Hyper-parameters encoded in an experiment directory name.  The grammar is the one the reference's logger WRITES
    (utils/logger.py:55-118): ``<Operator>_<Model>_Net<a-b[-c-d]>_Q<n>_<TF|FF>_S<scale>[_Pauli<P>][_Diag<..>|_Ham<lo-hi>]
    [_<backend>]_<num_train>x<num_points>_Seed<seed>``; it is read back here in ONE pass over the '_'-separated
    fields, each recognised by its prefix.  Accepts the directory or a checkpoint file inside it (as infer.py does
    with the same names, infer.py:60-86).  Keys absent from the name are absent from the result.
    """
    path = os.path.normpath(path)
    if os.path.splitext(path)[1].lower() in ('.ckpt', '.npz', '.pt', '.pth'):
        path = os.path.dirname(os.path.abspath(path))
    cfg = {}
    fields = os.path.basename(path).split('_')
    # everything left of the model token is the operator's name, which may itself contain '_' and look like a
    # hyper-parameter ('Q2', 'S1'): the recognisers below only see the fields to the right of the model
    start = next((i for i, t in enumerate(fields) if t in ('QuanONet', 'HEAQNN', 'DeepONet', 'FNN', 'FNO')), None)
    if start is None:
        return cfg
    cfg['model_type'] = fields[start]
    if start > 0:
        cfg['operator'] = '_'.join(fields[:start])
    for tok in fields[start + 1:]:
        head3, head1 = tok[:3], tok[:1]
        if head3 == 'Net' and tok[3:4].isdigit():
            if all(v.isdigit() for v in tok[3:].split('-')):
                cfg['net_size'] = [int(v) for v in tok[3:].split('-')]
        elif head1 == 'Q' and tok[1:].isdigit():
            cfg['num_qubits'] = int(tok[1:])
        elif tok in ('TF', 'FF', 'NTF'):
            cfg['if_trainable_freq'] = tok == 'TF'
        elif head1 == 'S' and _NUMBER.fullmatch(tok[1:]):
            cfg['scale_coeff'] = float(tok[1:])
        elif tok.startswith('Pauli') and tok[5:] in ('X', 'Y', 'Z'):
            cfg['ham_pauli'] = tok[5:]
        elif tok.startswith('Diag') and _signed_list(tok[4:]) is not None:
            cfg['ham_diag'] = _signed_list(tok[4:])
        elif head3 == 'Ham' and _signed_list(tok[3:]) is not None:
            cfg['ham_bound'] = _signed_list(tok[3:])
        elif tok.startswith('Seed') and tok[4:].isdigit():
            cfg['seed'] = int(tok[4:])
        elif tok.count('x') == 1 and all(part.isdigit() for part in tok.split('x')):
            cfg['num_train'], cfg['num_points'] = (int(v) for v in tok.split('x'))
        elif tok in _BACKEND_TOKENS:
            cfg['quantum_backend'] = _BACKEND_TOKENS[tok]
        # anything else (a field this grammar does not know, a malformed number) is skipped, never an error
    return cfg
