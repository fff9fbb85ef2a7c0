"""The oracle's scalar functions against independent closed forms (scipy), and the product's host scalar
exports against the oracle.  CPU only.

The reference holds no tests or golden vectors for this path (the pin to outputs of its own kernels is
tests/golden/reference + tests/test_gpu_reference_pin.py); these checks additionally hold the
restated CDFs / pdfs to the published definitions of the distributions and the
t-conorms to their algebraic identities (appendix B)."""
import math

import numpy as np
import pytest
from scipy import special, stats

TAU = 0.05

# id -> (scipy frozen distribution in units of u = s*x/tau, asymmetric shift handled separately)
SCIPY = {
    1: stats.uniform(-1, 2),
    3: stats.semicircular(),
    4: stats.norm(),
    5: stats.laplace(),
    6: stats.logistic(),
    7: stats.hypsecant(scale=2 / math.pi * math.pi / 2),   # sech density 1/(pi cosh u): hypsecant with unit scale
    8: stats.cauchy(),
    10: stats.gumbel_r(),
    11: stats.gumbel_l(),
}


def _grid():
    xs = np.concatenate([np.linspace(0, 3 * TAU, 61), np.array([1e-7, 5 * TAU, 12 * TAU])])
    return [(s, float(x)) for s in (1.0, -1.0) for x in xs]


@pytest.mark.parametrize("fid", sorted(SCIPY))
def test_cdf_pdf_against_scipy(oracle_mod, fid):
    d = SCIPY[fid]
    for s, x in _grid():
        u = s * x / TAU
        got = oracle_mod.sigmoid_forward(fid, s, x, TAU, f64=True)
        want = d.cdf(u)
        tol = 2e-7 if fid == 8 else 1e-12           # cauchy goes through atanf (float) by the reference's choice
        assert abs(got - want) <= tol + 1e-12 * abs(want), (fid, s, x, got, want)
        gp = oracle_mod.sigmoid_backward(fid, s, x, TAU, f64=True)
        wp = d.pdf(u) / TAU
        if fid in (1, 3) and abs(abs(u) - 1) < 1e-9:
            continue                                 # support boundary
        if fid == 4:
            assert abs(gp - wp) <= 1e-9 * max(1, abs(wp))
        else:
            assert abs(gp - wp) <= 1e-9 * max(1, abs(wp)), (fid, s, x, gp, wp)


def test_hypsecant_definition(oracle_mod):
    # D(u) = 2/pi * atan(tanh(u/2)) + 1/2  ==  2/pi * atan(exp(u))   (closed form of the sech CDF)
    for u in np.linspace(-6, 6, 49):
        got = oracle_mod.sigmoid_forward(7, 1.0 if u >= 0 else -1.0, abs(u) * TAU, TAU, f64=True)
        assert abs(got - 2 / math.pi * math.atan(math.exp(u))) < 1e-12


def test_one_sided_families(oracle_mod):
    for shift in (0.0, 0.5, 1.5):
        for s, x in _grid():
            u = s * x / TAU + shift
            # exponential
            got = oracle_mod.sigmoid_forward(12, s, x, TAU, 0.0, shift, f64=True)
            want = 0.0 if u < 0 else 1 - math.exp(-u)
            assert abs(got - want) < 1e-12
            # gamma(p): regularised lower incomplete gamma, cut to 1 beyond 15 (kernel.cu:304)
            for p in (0.5, 1.0, 2.0, 3.5):
                got = oracle_mod.sigmoid_forward(14, s, x, TAU, p, shift, f64=True)
                want = 0.0 if u <= 0 else (1.0 if u > 15 else special.gammainc(p, u))
                assert abs(got - want) < (5e-9 if u <= 5 else 2e-4), (p, u, got, want)   # 32-term series: truncation shows beyond u ~ 6
                gp = oracle_mod.sigmoid_backward(14, s, x, TAU, p, shift, f64=True)
                wp = 0.0 if u <= 0 else stats.gamma(p).pdf(u) / TAU
                assert abs(gp - wp) <= 1e-9 * max(1, abs(wp))
            # levy: erfc(sqrt(1/(2u)))
            got = oracle_mod.sigmoid_forward(16, s, x, TAU, 0.0, shift, f64=True)
            want = 0.0 if u * TAU <= 1e-6 else math.erfc(math.sqrt(1 / (2 * u)))
            assert abs(got - want) < 1e-12


def test_reversed_families_mirror(oracle_mod):
    """X_rev(s, x; shift) == 1 - X(-s, x; shift) wherever both are in their smooth branch."""
    for fwd, rev in ((12, 13), (14, 15), (16, 17)):
        for s, x in _grid():
            for shift in (0.0, 0.7):
                a = oracle_mod.sigmoid_forward(rev, s, x, TAU, 2.0, shift, f64=True)
                b = oracle_mod.sigmoid_forward(fwd, -s, x, TAU, 2.0, shift, f64=True)
                assert abs(a - (1 - b)) < 1e-9, (fwd, s, x, shift, a, b)


def test_pdf_is_derivative_of_cdf(oracle_mod):
    h = 1e-7
    for fid in range(1, 18):
        if fid == 8:
            continue                                 # float atanf inside the f64 instantiation
        for s in (1.0, -1.0):
            for x in (0.3 * TAU, 0.77 * TAU, 1.9 * TAU):
                if fid in (1, 2, 3) and x > TAU:
                    continue
                f = lambda xx: oracle_mod.sigmoid_forward(fid, s, xx, TAU, 2.0, 0.25, f64=True)
                num = s * (f(x + h) - f(x - h)) / (2 * h)
                ana = oracle_mod.sigmoid_backward(fid, s, x, TAU, 2.0, 0.25, f64=True)
                assert abs(num - ana) <= 2e-5 * max(1.0, abs(ana)), (fid, s, x, num, ana)


T_CONORM_P = {4: 0.5, 5: 3.0, 6: 2.0, 7: 0.7, 8: 1.5, 9: -1.5}


def test_t_conorm_axioms_and_gradient(oracle_mod):
    rs = np.random.RandomState(0)
    for tid in range(1, 10):
        p = T_CONORM_P.get(tid, 0.0)
        T = lambda a, b: oracle_mod.t_conorm_forward(tid, a, b, 0, p, f64=True)
        for _ in range(50):
            a, b, c = rs.uniform(0.02, 0.9, 3)
            assert abs(T(a, b) - T(b, a)) < 1e-12                       # commutative
            assert abs(T(T(a, b), c) - T(a, T(b, c))) < 1e-9            # associative
            assert abs(T(a, 0.0) - a) < 1e-9                            # neutral element 0
            assert T(a, b) >= max(a, b) - 1e-12                         # >= max
            # closed-form d T(T(a,b),c) / d b  (kernel.cu:567-614) against a finite difference
            if tid == 1:
                continue
            A = T(T(a, b), c)
            h = 1e-7
            num = (T(T(a, b + h), c) - T(T(a, b - h), c)) / (2 * h)
            ana = oracle_mod.t_conorm_backward(tid, A, b, 3, p, f64=True)
            assert abs(num - ana) <= 1e-4 * max(1.0, abs(ana)), (tid, a, b, c, num, ana)


def test_product_scalars_equal_oracle(oracle_mod, native_lib):
    _product_scalars_equal_oracle(oracle_mod, native_lib)


@pytest.mark.gpu
def test_product_scalars_equal_oracle_on_the_gpu_box(oracle_mod, native_lib):
    """Same sweep inside the -m gpu set, so that the driver's GPU run sees SURVEY row a15 exhaustively."""
    _product_scalars_equal_oracle(oracle_mod, native_lib)


def _product_scalars_equal_oracle(oracle_mod, native_lib):
    """The host-callable exports of libgendr_hip.so (same source as the device code) against the oracle's
    float instantiation.  Both run on glibc here, so they must agree bit for bit."""
    rs = np.random.RandomState(1)
    n_bad = 0
    for fid in range(18):
        for s in (1.0, -1.0):
            for x in np.concatenate([rs.uniform(0, 4 * TAU, 40), [0.0, TAU, 20 * TAU]]):
                for shape, shift in ((2.0, 0.0), (0.5, 0.8)):
                    x = float(np.float32(x))
                    a = native_lib.gendr_sigmoid_forward(fid, s, x, TAU, shape, shift)
                    b = oracle_mod.sigmoid_forward(fid, s, x, TAU, shape, shift)
                    assert a == b or (a != a and b != b), ('cdf', fid, s, x, a, b)
                    a = native_lib.gendr_sigmoid_backward(fid, s, x, TAU, shape, shift)
                    b = oracle_mod.sigmoid_backward(fid, s, x, TAU, shape, shift)
                    assert a == b or (a != a and b != b), ('pdf', fid, s, x, a, b)
    for tid in range(1, 10):
        p = T_CONORM_P.get(tid, 0.0)
        for _ in range(200):
            a, b = (float(np.float32(v)) for v in rs.uniform(0, 1, 2))
            x = native_lib.gendr_t_conorm_forward(tid, a, b, 0, p)
            y = oracle_mod.t_conorm_forward(tid, a, b, 0, p)
            assert x == y or (x != x and y != y), ('fold', tid, a, b, x, y)
            x = native_lib.gendr_t_conorm_backward(tid, max(a, b), min(a, b), 2, p)
            y = oracle_mod.t_conorm_backward(tid, max(a, b), min(a, b), 2, p)
            assert x == y or (x != x and y != y), ('grad', tid, a, b, x, y)
    assert n_bad == 0


def test_invalid_parameters_give_nan(oracle_mod, native_lib):
    # kernel.cu:296,491,501,512,522,534,552: invalid p / shape -> NaN; unknown id -> NaN (:361,:561)
    for mod in (oracle_mod.sigmoid_forward, lambda *a: native_lib.gendr_sigmoid_forward(*a)):
        assert math.isnan(mod(14, 1.0, 0.1, TAU, -1.0, 0.0))
        assert math.isnan(mod(99, 1.0, 0.1, TAU, 0.0, 0.0))
    for tid, p in ((4, -1.0), (5, 1.0), (5, 0.0), (6, 0.0), (7, -2.0), (8, 0.0), (9, 1.0), (42, 1.0)):
        assert math.isnan(oracle_mod.t_conorm_forward(tid, 0.3, 0.4, 0, p))
        assert math.isnan(native_lib.gendr_t_conorm_forward(tid, 0.3, 0.4, 0, p))
