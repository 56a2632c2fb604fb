"""
Host-side training logic on CPU: shard arithmetic, the flat-buffer data-parallel step over gloo
(world_size 2) against the single-process full-batch step, checkpoint interop.  The circuit
arithmetic in these tests is the oracle test double (tests/helpers.py); no GPU is touched.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import helpers as H
from quanonet_amd.solver import DataParallelTrainer, regression_metrics, shard_slice


def test_shard_slice_partitions_every_batch():
    for n in (0, 1, 7, 100, 1023):
        for w in (1, 2, 3, 8):
            cuts = [shard_slice(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


N, NET, B_IN, T_IN, GB, STEPS = 3, (2, 1, 2, 2), 6, 2, 23, 3     # odd global batch -> uneven shards


def _data():
    rng = np.random.default_rng(5)
    return (torch.tensor(rng.normal(size=(STEPS, GB, B_IN))), torch.tensor(rng.uniform(size=(STEPS, GB, T_IN))),
            torch.tensor(rng.normal(size=(STEPS, GB, 1))))


def _single_process():
    model = H.quanonet_with_oracle_layer(N, B_IN, T_IN, NET)
    tr = DataParallelTrainer(model, lr=1e-2, fused=False)
    br, tk, y = _data()
    grads = []
    for s in range(STEPS):
        tr.train_step(br[s], tk[s], y[s], global_batch=GB)
        grads.append(tr.flat.clone())
    return tr.pflat.clone(), grads


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = H.quanonet_with_oracle_layer(N, B_IN, T_IN, NET, seed=rank)      # different init per rank on purpose
    tr = DataParallelTrainer(model, lr=1e-2, world_size=world, dist=dist, fused=False)   # broadcasts rank 0's
    br, tk, y = _data()
    grads = []
    for s in range(STEPS):
        lo, hi = shard_slice(GB, rank, world)
        tr.train_step(br[s, lo:hi], tk[s, lo:hi], y[s, lo:hi], global_batch=GB)
        grads.append(tr.flat.clone().numpy())
    q.put((rank, tr.pflat.clone().numpy(), grads))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_step_equals_single_process():
    ref_params, ref_grads = _single_process()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0), (_, p1, g1) = res
    np.testing.assert_array_equal(p0, p1)                         # replicas stay bit-identical
    for s in range(STEPS):
        np.testing.assert_array_equal(g0[s], g1[s])               # all-reduced buffer identical on both ranks
        np.testing.assert_allclose(g0[s], ref_grads[s].numpy(), rtol=0, atol=1e-12)   # == full-batch gradient (+sse, sum y^2)
    np.testing.assert_allclose(p0, ref_params.numpy(), rtol=0, atol=1e-12)


def test_flat_views_and_state_dict_order():
    model = H.quanonet_with_oracle_layer(N, B_IN, T_IN, NET)
    tr = DataParallelTrainer(model, lr=1e-2, fused=False)
    names = [k for k, _ in model.named_parameters()]
    assert names[0] == 'bias' and names[-1] == 'quantum_layer.ansatz_weights'
    off = 0
    for k, p in model.named_parameters():
        assert p.data.data_ptr() == tr.pflat[off:].data_ptr()      # parameters alias the flat vector
        assert p.grad.data_ptr() == tr.flat[off:].data_ptr()
        off += p.numel()
    assert off == tr.numel


def test_checkpoint_roundtrip_and_name_parsing(tmp_path):
    from quanonet_amd import checkpoint as ck
    st = dict(np.load(os.path.join(H.GOLDEN, 'advection_q5.npz')))
    pt = ck.ms_to_pt_state(st, 5, (40, 2, 20, 2))
    assert pt['quantum_layer.ansatz_weights'].shape == (120, 3, 5)
    np.testing.assert_array_equal(pt['quantum_layer.ansatz_weights'].reshape(-1), st['QuanONet.weight'].astype(np.float64))
    back = ck.pt_to_ms_state(pt)
    for k in st:
        np.testing.assert_array_equal(np.asarray(back[k]).reshape(-1), st[k].astype(np.float64).reshape(-1))
    cfg = ck.parse_experiment_dir('x/Advection_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0/best_model.ckpt')
    assert cfg == {'operator': 'Advection', 'model_type': 'QuanONet', 'net_size': [40, 2, 20, 2], 'num_qubits': 5,
                   'scale_coeff': 0.1, 'if_trainable_freq': True, 'num_train': 1000, 'num_points': 100, 'seed': 0}
    cfg2 = ck.parse_experiment_dir('RDiffusion_HEAQNN_Net64-2_Q8_FF_S0.01_TQ_1000x100_Seed3')
    assert cfg2['model_type'] == 'HEAQNN' and cfg2['net_size'] == [64, 2] and cfg2['if_trainable_freq'] is False
    assert cfg2['quantum_backend'] == 'torchquantum' and cfg2['seed'] == 3 and cfg2['scale_coeff'] == 0.01
    # every optional field the reference's logger can write (utils/logger.py:55-118), incl. negative bounds
    cfg3 = ck.parse_experiment_dir('out/Darcy_QuanONet_Net160-2-90-2_Q5_TF_S0.001_PauliX_Ham-2-6_PL_1000x25_Seed4/')
    assert cfg3 == {'operator': 'Darcy', 'model_type': 'QuanONet', 'net_size': [160, 2, 90, 2], 'num_qubits': 5,
                    'if_trainable_freq': True, 'scale_coeff': 0.001, 'ham_pauli': 'X', 'ham_bound': [-2.0, 6.0],
                    'quantum_backend': 'pennylane', 'num_train': 1000, 'num_points': 25, 'seed': 4}
    cfg4 = ck.parse_experiment_dir('Antideriv_QuanONet_Net5-1-5-1_Q2_TF_S0.001_Diag-5--2.5-2.5-5_1000x100_Seed0')
    assert cfg4['ham_diag'] == [-5.0, -2.5, 2.5, 5.0] and 'ham_bound' not in cfg4 and cfg4['num_qubits'] == 2
    # the names of the four shipped checkpoints' directories (pretrained_weights/**)
    for nm in ('Antideriv_QuanONet_Net5-1-5-1_Q2_TF_S0.001_1000x100_Seed0', 'Advection_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0'):
        c = ck.parse_experiment_dir(nm)
        assert c['model_type'] == 'QuanONet' and len(c['net_size']) == 4 and c['if_trainable_freq'] is True
    assert ck.parse_experiment_dir('weights') == {}
    # all four directory names under the reference's pretrained_weights/
    for nm, n, pts in (('Antideriv_QuanONet_Net5-1-5-1_Q2_TF_S0.001_1000x100_Seed0', 2, 100),
                       ('Advection_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0', 5, 100),
                       ('RDiffusion_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x100_Seed0', 5, 100),
                       ('Darcy_QuanONet_Net40-2-20-2_Q5_TF_S0.1_1000x25_Seed0', 5, 25)):
        c = ck.parse_experiment_dir(nm)
        assert (c['operator'], c['num_qubits'], c['num_points'], c['seed']) == (nm.split('_')[0], n, pts, 0)
    # str(v) of small / negative values: exponent forms inside the dash-joined lists and the scale
    c = ck.parse_experiment_dir('Op_QuanONet_Net2-1-2-1_Q2_TF_S1e-05_Diag1e-05--2.5-3.0-4e+02_10x5_Seed1')
    assert c['scale_coeff'] == 1e-05 and c['ham_diag'] == [1e-05, -2.5, 3.0, 400.0]
    assert ck.parse_experiment_dir('Op_HEAQNN_Net2-1_Q3_FF_S0.1_Ham-1e-05-2.5_10x5_Seed1')['ham_bound'] == [-1e-05, 2.5]
    # an operator whose name has '_' parts that look like hyper-parameters does not overwrite them
    c = ck.parse_experiment_dir('My_Q9_S7_Op_QuanONet_Net2-1-2-1_Q2_TF_S0.5_10x5_Seed1')
    assert c['operator'] == 'My_Q9_S7_Op' and c['num_qubits'] == 2 and c['scale_coeff'] == 0.5
    # malformed fields are skipped, not raised
    c = ck.parse_experiment_dir('Op_QuanONet_Net2-1-2-1_Q2_TF_Sx_Diag1e-_Hamfoo_10x5_Seed1')
    assert 'scale_coeff' not in c and 'ham_diag' not in c and 'ham_bound' not in c and c['seed'] == 1
    # MindSpore .ckpt reader against a protobuf written here (same wire layout as SURVEY.md 8c)
    def varint(v):
        out = b''
        while True:
            b = v & 0x7F
            v >>= 7
            out += bytes([b | (0x80 if v else 0)])
            if not v:
                return out
    def field(no, payload):
        return varint((no << 3) | 2) + varint(len(payload)) + payload
    arr = np.arange(6, dtype='<f4')
    tensor = varint((1 << 3) | 0) + varint(2) + varint((1 << 3) | 0) + varint(3) + field(2, b'Float32') + field(3, arr.tobytes())
    value = field(1, b'QuanONet.weight') + field(2, tensor)
    path = tmp_path / 'm.ckpt'
    path.write_bytes(field(1, value))
    got = ck.read_mindspore_ckpt(str(path))
    np.testing.assert_array_equal(got['QuanONet.weight'], arr.reshape(2, 3))
    np.savez(tmp_path / 'w.npz', **st)
    assert set(ck.load_weight_file(str(tmp_path / 'w.npz'))) == set(st)


def _metrics_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    pred, true = rng.normal(size=(1001, 1)), rng.normal(size=(1001, 1))
    lo, hi = shard_slice(1001, rank, world)                       # uneven shards on purpose
    m = regression_metrics(torch.tensor(pred[lo:hi]), torch.tensor(true[lo:hi]), dist, world)
    q.put((rank, m))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_metrics_equal_the_unsharded_ones():
    """PTSolver.evaluate shards the test set over the ranks; metrics as utils/metrics.py:6-29 and
    the relative L2 of solvers/solver_pt.py."""
    rng = np.random.default_rng(5)
    pred, true = rng.normal(size=(1001, 1)), rng.normal(size=(1001, 1))
    d = (pred - true).ravel()
    ref = {'MSE': np.mean(d ** 2), 'MAE': np.mean(np.abs(d)), 'Max_Error': np.max(np.abs(d)),
           'rel_l2': np.linalg.norm(d) / (np.linalg.norm(true) + 1e-8)}
    one = regression_metrics(torch.tensor(pred), torch.tensor(true))
    for k in ref:
        assert abs(one[k] - ref[k]) < 1e-12, k
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, m in res:
        for k in ref:
            assert abs(m[k] - ref[k]) < 1e-12, k


def test_lbfgs_is_rejected_like_the_reference():
    model = H.quanonet_with_oracle_layer(N, B_IN, T_IN, NET)
    with pytest.raises(NotImplementedError):            # solvers/solver_pt.py:154-155
        DataParallelTrainer(model, lr=1e-2, fused=False, optimizer='lbfgs')


def test_completed_run_is_skipped_like_the_reference(tmp_path):
    """solvers/solver_pt.py:192-194: a run whose directory already holds metric.json is not trained again (the reference
    exits the process there; PTSolver.train returns None when config['skip_completed'] is set)."""
    from quanonet_amd.solver import PTSolver, set_random_seed
    cfg, data, a, order, seed = H.trajectory_solver_inputs('quanonet_tf', str(tmp_path), skip_completed=True, num_epochs=1)
    set_random_seed(seed)
    from quanonet_amd.models import QuanONetPT
    model = H.quanonet_with_oracle_layer(cfg['num_qubits'], data['train_branch_input'].shape[1], data['train_trunk_input'].shape[1],
                                         tuple(cfg['net_size']))
    s = PTSolver(cfg, data, device=torch.device('cpu'), model=model, log=lambda *x, **k: None)
    assert not s.is_completed()
    hist = s.train()
    assert hist is not None and len(hist['loss_train']) == 1
    s.evaluate(hist)                                   # writes metric.json
    assert s.is_completed()
    before = {k: v.clone() for k, v in s.model.state_dict().items()}
    assert s.train() is None                           # skipped
    for k, v in s.model.state_dict().items():
        assert torch.equal(v, before[k])
