"""The acceptance rule itself (tests/criteria.py) on CPU: it must accept what differs from the fp32 oracle only by
last-ulp libm effects or threshold flips, and must reject a 1e-4 error on a well-conditioned element -- wherever it
sits, even in a case with ill-conditioned elements elsewhere (the round-2 rule compared tensor-wide percentiles and
let that through)."""
import numpy as np
import pytest

import criteria
import oracle
import parity
import scenes


def _case(opts):
    fv, tex = scenes.sphere()
    isz = 48
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    return fv, tex, isz, grad, criteria.references(fv, tex, isz, opts, grad)


@pytest.mark.parametrize("opts", [dict(), dict(dist_func='gumbel_min', dist_scale=2e-2, aggr_alpha_func='einstein'),
                                  dict(dist_func='gamma_rev', dist_shape=1.5, dist_scale=2e-2)], ids=['uniform', 'gumbel_min', 'gamma_rev'])
def test_rule_accepts_libm_noise_and_rejects_real_errors(oracle_mod, opts):
    fv, tex, isz, grad, refs = _case(opts)
    o32 = refs['o32']
    same = {k: o32[k].copy() for k in ('rgba', 'aggrs_info', 'grad_faces', 'grad_textures')}
    assert not criteria.failures(criteria.elementwise(same, refs))
    # a different, equally valid libm: an element-wise mix of two hashed jitter modes
    mix = {k: np.where(np.random.RandomState(3).rand(*o32[k].shape) < 0.5, refs['jit'][2][k], refs['jit'][5][k]) for k in same}
    assert not criteria.failures(criteria.elementwise(mix, refs))
    # threshold flips: the run with the thresholds 10 % lower
    flip = {k: refs['lo'][k] for k in same}
    assert not criteria.failures(criteria.elementwise(flip, refs))
    # a 1e-4 relative error on ONE well-conditioned element of each tensor must be caught
    rep = criteria.elementwise(same, refs)
    for k in same:
        a = same[k].astype(np.float64)
        moved = np.max([np.abs(j[k].astype(np.float64) - a) for j in refs['jit'] + [refs['lo'], refs['hi']]], axis=0)
        nbr = criteria._nbr_max_image if k in criteria.IMAGE_KEYS else criteria._per_face_max
        err_allowed = 8 * nbr(np.where(np.isnan(moved), np.inf, moved))            # what the rule tolerates around the element
        cand = np.argwhere((np.abs(a) > 0.1 * np.nanmax(np.abs(a))) & (err_allowed < 1e-6 * np.abs(a)))
        if len(cand) == 0:
            continue
        idx = tuple(cand[len(cand) // 2])
        broken = {kk: v.copy() for kk, v in same.items()}
        broken[k][idx] = broken[k][idx] * (1 + 1e-4)
        bad = criteria.failures(criteria.elementwise(broken, refs))
        assert bad and k in bad[0], (k, idx, rep[k])


def test_rule_is_tight_on_the_algebraic_path(oracle_mod):
    """uniform / probabilistic: no libm call before alpha -> the alpha plane is held to exactly 1e-5 everywhere, and so
    are nearly all gradient elements (the softmax's expf is the only jittered call)."""
    fv, tex, isz, grad, refs = _case(dict(aggr_rgb_func='hard'))
    rep = criteria.elementwise({k: refs['o32'][k] for k in ('rgba', 'aggrs_info', 'grad_faces', 'grad_textures')}, refs)
    for k, r in rep.items():
        assert r['loosened'] == 0.0, (k, r)


@pytest.mark.parametrize("name", ['uniform_prob_softmax', 'gumbelmin_einstein', 'gammarev_prob'])
def test_loosened_shares_are_capped_by_the_table(oracle_mod, name):
    """tests/golden/loosened_table.json records, per case and tensor, the share of elements the rule does not hold to 1e-5
    (a property of the oracle and of the rule's constants).  A wider rule -- a larger K, another noise source -- makes
    `loosened_failures` reject every report: the rule cannot be loosened silently."""
    opts = dict(scenes.OPTION_MATRIX)[name]
    fv, tex = scenes.sphere()
    isz = 64
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    refs = criteria.references(fv, tex, isz, opts, grad)
    rep = criteria.elementwise({k: refs['o32'][k] for k in ('rgba', 'aggrs_info', 'grad_faces', 'grad_textures')}, refs)
    key = 'sphere:' + name
    assert not criteria.loosened_failures(key, rep), criteria.loosened_failures(key, rep)
    row = criteria.loosened_table().get(key, {})
    for k, r in rep.items():
        assert abs(r['loosened'] - row.get(k, 0.0)) <= 1e-3, (k, r['loosened'], row.get(k))     # the table is current
    old = criteria.K_NOISE
    try:
        criteria.K_NOISE = 64.0
        wide = criteria.elementwise({k: refs['o32'][k] for k in ('rgba', 'aggrs_info', 'grad_faces', 'grad_textures')}, refs)
        if name != 'uniform_prob_softmax':
            assert criteria.loosened_failures(key, wide)
    finally:
        criteria.K_NOISE = old


def test_pin_table_is_present_and_small():
    """The exception table of the flat gate against the reference's kernels (tests/pin.py): present, and short -- at most a
    tenth of the cases may need an entry for the default build."""
    import pin
    t = pin.load_table()
    assert t is not None and 'default' in t
    n_cases = len(pin.MATRIX) * len(pin.SCENES) + len(pin.FULL)
    assert len(t['default']) <= n_cases // 10, sorted(t['default'])
    assert not any(k in t['default'] for k in ('C2', 'C4')), 'BASELINE configs 2 and 4 meet a flat 1e-5'


def test_flat_gate_logic():
    """tests/pin.py on synthetic numbers: an untabled case is held to 1e-5 flat; a tabled (case, tensor) to twice its tabulated
    maximum, 99th percentile and share above 1e-5 -- so a systematic error fails a tabled case too."""
    import pin
    table = dict(default={'X': {'grad_faces': dict(max=1e-3, p99=3e-5, frac=0.02)}})
    ok = dict(rgba=dict(max=0.0, p99=0.0, frac=0.0, n=100), grad_faces=dict(max=9e-6, p99=1e-6, frac=0.0, n=100))
    assert not pin.flat_failures('Y', ok, table)
    assert pin.flat_failures('Y', dict(ok, grad_faces=dict(max=2e-5, p99=1e-6, frac=0.01, n=100)), table)
    inside = dict(ok, grad_faces=dict(max=1.5e-3, p99=5e-5, frac=0.03, n=100))
    assert not pin.flat_failures('X', inside, table)
    assert pin.flat_failures('X', dict(ok, grad_faces=dict(max=1.5e-3, p99=1e-4, frac=0.9, n=100)), table)      # a 1e-4 error everywhere
    assert pin.flat_failures('X', dict(ok, rgba=dict(max=2e-5, p99=0.0, frac=0.01, n=100)), table)              # another tensor of the case: flat
    # the spread gate of the fast variant: quantile by quantile against the reference's own two builds
    spread = dict(rgba=dict(p50=0.0, p90=1e-6, p99=1e-4, p999=2e-3, frac=0.04, max=5.0, n=1000))
    m_in = dict(rgba=dict(p50=0.0, p90=5e-5, p99=3e-4, p999=5e-3, frac=0.05, max=9e5, n=1000))
    assert not pin.spread_failures('X', m_in, spread)
    assert pin.spread_failures('X', dict(rgba=dict(m_in['rgba'], p99=1e-3)), spread)
