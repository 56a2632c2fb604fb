"""
quanonet_amd/csrc/hea_sincos.hpp -- the short sincos the kernels use for rotation angles -- against libm in long double, on the
HOST: the function is __host__ __device__ with every FMA written out, so the arithmetic is the one the device runs.
Bound: 2.5e-16 absolute (libm itself: ~1.1e-16), sin^2 + cos^2 = 1 to 5e-16, NaN in -> NaN out, |x| >= 1e5 through the library.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_fast_sincos_against_libm(tmp_path):
    exe = str(tmp_path / 'sincos_check')
    src = os.path.join(ROOT, 'tests', 'native', 'sincos_check.cpp')
    r = subprocess.run([HIPCC, '-O2', '-std=c++17', '-x', 'hip', '--offload-arch=gfx950', src, '-o', exe],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for n, span in ((2000000, 1000.0), (1000000, 10.0), (1000000, 99999.0)):
        out = subprocess.run([exe, str(n), str(span)], stdout=subprocess.PIPE, text=True, timeout=300).stdout.split()
        es, ec, en = float(out[0]), float(out[1]), float(out[2])
        assert es < 2.5e-16 and ec < 2.5e-16 and en < 5e-16, (span, out)
        assert out[4] == '1'
