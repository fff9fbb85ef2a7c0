"""north_star: "every dist_func x aggr_alpha_func option".  scenes.OPTION_MATRIX covers every id once; this file enumerates
the CROSS PRODUCT -- all 18 x 10 pairings (reference: functional/renderer.py:44-79, kernel.cu:243-363 x :474-563), which is
what selects between the specialised kernel rows, the light / full runtime-dispatch classes and the team rows
(csrc/gendr_capi.hip) -- on one scene at 32^2, soft and hard RGB alternating, the SHIPPED build, against the reference's own
kernels compiled for this GPU (oracle/_ref) under the flat 1e-5 gate of tests/pin.py.  Parameters per id
are valid ones by gendr_validate's rules (kernel.cu:296,491,501,512,522,534,552)."""
import numpy as np
import pytest

import criteria
import parity
import pin
import scenes

pytestmark = pytest.mark.gpu

DISTS, AGGRS = scenes.DISTS, scenes.AGGRS
assert len(DISTS) == 18 and len(AGGRS) == 10


def cross_product():
    """(id string, options) for the 180 pairings; RGB aggregation, squared distances and the sidedness alternate with the
    position so that each dist and each t-conorm meets both values of each."""
    out = []
    for i, (d, dopt) in enumerate(DISTS):
        for j, (a, p) in enumerate(AGGRS):
            o = dict(dopt, dist_func=d, aggr_alpha_func=a, aggr_rgb_func='hard' if (i + j) & 1 else 'softmax',
                     double_side=bool((i + 2 * j) & 2))
            if p is not None:
                o['aggr_alpha_t_conorm_p'] = p
            if (i + j) % 5 == 4 and d not in ('hard',):
                o['dist_squared'] = True
                o['dist_scale'] = o['dist_scale'] ** 2 * 4            # the same reach for the squared distance
            out.append(('%s-%s' % (d, a), o))
    return out


CASES = cross_product()
TABLE = pin.load_table()


@pytest.fixture(scope='module')
def ref_kernels():
    parity.require_reference()


@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_every_dist_func_times_aggr_alpha_func_against_the_reference_kernels(oracle_mod, native_lib, ref_kernels, name, opts):
    isz = pin.MATRIX_SIZE
    fv, tex = pin.matrix_inputs(opts, 'sphere')
    grad = pin.matrix_grad(fv, isz)
    r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
    c = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
    h = parity.run_hip(fv, tex, isz, opts, grad, variant='default')
    bad = pin.flat_failures('cross:' + name, pin.measure(h, r, c['abs_faces'], c['abs_textures']), TABLE, section='default')
    assert not bad, ('HIP product (shipped build) vs reference kernels', opts, bad)
    h2 = parity.run_hip(fv, tex, isz, dict(opts, cull=0), None, variant='default')
    for k in ('rgba', 'aggrs_info'):
        assert np.array_equal(h[k], h2[k], equal_nan=True), ('culled != all-pairs', name, k)


def test_cross_product_is_complete():
    from gendr_amd.functional import renderer as R
    assert {R.DIST_FUNC_IDS[d] for d, _ in DISTS} == set(range(18)) and {R.AGGR_ALPHA_FUNC_IDS[a] for a, _ in AGGRS} == set(range(10))
    assert len({n for n, _ in CASES}) == 180
