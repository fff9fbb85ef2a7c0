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


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_with_a_small_dist_eps_cull_exactly(native_lib, seed):
    """The regime in which the reference's border test (kernel.cu:747), not the distribution's tail, ends a face's reach: the same draw
    with dist_eps forced to 1 .. 30, forward only, several draws per seed.  Held to culled == all-pairs bit for bit -- the property the
    coverage kernel's box test has to deliver (round 4: tools/fuzz_parity.py case 255); the all-pairs traversal is what the pin tests hold
    to the reference's kernels.  (No comparison with the oracle here: on random slivers with a small dist_eps the reference's own two
    builds differ on half the pixels)"""
    rs = np.random.RandomState(5000 + seed)
    for _ in range(4):
        name, opts, fv, tex, isz = _draw(rs)
        opts['dist_eps'] = float(rs.choice([1.0, 1.5, 3.0, 10.0, 30.0]))
        opts['dist_scale'] = float(opts.get('dist_scale', 1e-2)) * float(rs.choice([1.0, 4.0, 10.0]))
        a = parity.run_hip(fv, tex, isz, opts, None)
        b = parity.run_hip(fv, tex, isz, dict(opts, cull=0), None)
        for k in ('rgba', 'aggrs_info'):
            assert np.array_equal(a[k], b[k], equal_nan=True), (name, opts, fv.shape, isz, k)
