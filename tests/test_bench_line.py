"""
bench.py's N > 1 line explains itself: a same-device rehearsal of two ranks on the one GPU of the test
box, started as a child process exactly as a user would (`python bench.py --gpus 2 ...` spawns its ranks through
torch.distributed.run before touching the GPU), must name the exchange it ran, why, every rank's device, and the step
time through each exchange.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_same_device_bench_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--same-device', '--backend', 'gloo',
                        '--steps', '5', '--warmup', '2'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and c['world_size'] == 2 and c['same_device_rehearsal'] is True
    assert c["dp_exchange"].startswith("peer-mapped buffers"), c
    assert "calibrated on this step" in c["dp_exchange_reason"]
    assert 'self-check passed' in c['dp_exchange_reason']
    assert [v['rank'] for v in c['devices']] == [0, 1] and all(v['name'] for v in c['devices'])
    by = c['ms_per_step_by_exchange']
    assert len(by) >= 4 and all(v > 0 for v in by.values())
    assert any(k.startswith('all_reduce (gloo)') for k in by) and any(k.startswith('separate one-workgroup') for k in by)
    assert d['value'] > 0 and d['ms_per_step'] > 0 and d['scaling'] == 'weak'
