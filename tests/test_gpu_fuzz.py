"""Randomised shapes the fixed suites do not enumerate (fixed seeds): batch size, face count around the 64-face chunk
boundaries, odd image sizes and sizes with empty 64x64 super-tiles, texel layouts, option sets -- the HIP path against the
oracle under the acceptance rule of criteria.py, and the culled traversal against the all-pairs one.  tools/fuzz_parity.py
runs the same draw for any number of cases."""
import numpy as np
import pytest

import criteria
import parity
import scenes

pytestmark = pytest.mark.gpu


def _draw(rs):
    name, opts = scenes.OPTION_MATRIX[rs.randint(len(scenes.OPTION_MATRIX))]
    opts = dict(opts)
    B = int(rs.choice([1, 2, 3, 5, 9]))
    nf = int(rs.choice([1, 2, 17, 63, 64, 65, 127, 130, 200]))
    isz = int(rs.choice([8, 13, 31, 64, 72, 100, 128, 136, 192, 200]))
    vertex = opts.get('texture_type') == 'vertex'
    T = 1 if vertex else int(rs.choice([1, 1, 4, 9]))
    scale = float(rs.choice([0.25, 0.5, 1.0]))
    fv, tex = scenes.soup(B=B, nf=max(nf, 9), seed=int(rs.randint(1 << 30)), T=T, vertex_tex=vertex)
    fv, tex = fv[:, :nf].copy(), tex[:, :nf].copy()
    fv[..., :2] *= scale
    opts['T'] = T
    return name, opts, fv, tex, isz


@pytest.mark.parametrize("seed", range(16))
def test_random_shape_and_option_set(oracle_mod, native_lib, seed):
    name, opts, fv, tex, isz = _draw(np.random.RandomState(1000 + seed))
    res, h, r = parity.compare(fv, tex, isz, opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    bad, _, _ = criteria.check_case(fv, tex, isz, opts, h, grad, oracle_f32=r)
    assert not bad, (name, fv.shape, isz, bad)
    h2 = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
    for k in ('rgba', 'aggrs_info'):
        assert np.array_equal(h[k], h2[k], equal_nan=True), (name, k)
