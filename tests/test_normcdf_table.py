"""The tables of the gaussian's normal CDF in the specialised kernels (gendr_amd/csrc/gendr_math.h: kNormTab, norm_q_tab; round 6), without a GPU:
the constants are parsed from the header and the double evaluation is simulated in numpy (the device code's operations, FMAs apart) against
60-digit values of Phi(-x).  The exhaustive comparison with the library's normcdf(double) on the GPU is tests/test_gpu_exact_math.py."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_tables():
    src = open(os.path.join(ROOT, 'gendr_amd', 'csrc', 'gendr_math.h')).read()
    body = src[src.index('__device__ const double kNormTab'):]
    body = body[body.index('{') + 1:body.index('};')]
    rows = [[float(v) for v in r.split(',') if v.strip()] for r in re.findall(r'\{([^{}]+)\}', body)]
    m = re.search(r'kNormLn2_16Hi = ([^,]+), kNormLn2_16Lo = ([^,]+), kNorm16_Ln2 = ([^;]+);', src)
    return np.array(rows), float(m.group(1)), float(m.group(2)), float(m.group(3))


def test_table_form_of_the_normal_cdf_is_accurate_to_2_pow_minus_50():
    mp = pytest.importorskip('mpmath')
    mp.mp.dps = 50
    tab, l_hi, l_lo, inv_l = _header_tables()
    assert tab.shape == (16, 12)
    assert np.allclose(tab[:, 11], 2.0 ** (np.arange(16) / 16.0), rtol=1e-16, atol=0)         # row i ends with 2^(i/16)
    h = 45.0 / 128.0
    rs = np.random.RandomState(7)
    x = np.concatenate([rs.uniform(0, 5.625, 1500), np.arange(1, 16) * h + rs.uniform(-1e-6, 1e-6, 15), [0.0, 5.6249995]]).astype(np.float32).astype(np.float64)
    y = -0.5 * (x * x)
    k = np.rint(y * inv_l)
    r = (y - k * l_hi) - k * l_lo
    e = np.full_like(x, 1.0 / 5040)
    for c in (1.0 / 720, 1.0 / 120, 1.0 / 24, 1.0 / 6, 0.5, 1.0, 1.0):
        e = e * r + c
    ki = k.astype(np.int64)
    i = np.minimum((x * (128.0 / 45.0)).astype(np.int64), 15)
    t = x - (2 * i + 1) * (45.0 / 256.0)
    q = tab[i, 10].copy()
    for n in range(9, -1, -1):
        q = q * t + tab[i, n]
    got = np.ldexp(e * tab[ki & 15, 11], (ki >> 4).astype(np.int32)) * q
    worst = max(abs(mp.mpf(float(v)) / mp.ncdf(-mp.mpf(float(u))) - 1) for u, v in zip(x, got))
    assert worst < mp.mpf(2) ** -50, float(mp.log(worst, 2))
