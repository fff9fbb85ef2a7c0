"""(1) The short correctly-rounded square root and reciprocal of the pair math (gendr_math.h: sqrt_rn, rcp_rn) equal
sqrtf(x) and 1.f / x for EVERY float in [2^-96, 2^96]: exhaustive check on the GPU (1.6e9 bit patterns each).
(2) The gaussian distribution's normal CDF (gendr_math.h: norm_cdf -- for u >= 0 a degree-30 polynomial in double, rounded
once) against what the reference's kernel.cu:293 computes when compiled for this platform, the library's normcdf(double)
rounded to float: every float of [0, 6] (1.09e9 bit patterns) -- for both device forms: the tables the kernels specialised for
the gaussian distribution hold in LDS (round 6: norm_cdf_tab, what BASELINE config 3 runs) and the degree-30 polynomial of the
runtime-dispatch kernels (norm_cdf)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("what,name", [(0, 'sqrt'), (1, 'rcp(+x)'), (2, 'rcp(-x)')])
def test_short_forms_are_correctly_rounded_everywhere(native_lib, what, name):
    rep = torch.zeros(16, dtype=torch.int64, device='cuda')
    rc = native_lib.gendr_selftest(what, ctypes.c_void_p(rep.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    r = rep.cpu().tolist()
    lo, hi = 0x0F800000, 0x6F800000                      # bit patterns of 2^-96 and 2^96
    assert r[1] == hi - lo + 1, r
    assert r[0] == 0, '%s: %d mismatches, e.g. bit patterns %s' % (name, r[0], [hex(v) for v in r[2:15] if v])


@pytest.mark.parametrize("what,side", [(3, '+u, table form'), (4, '-u, table form'), (5, '+u, polynomial form'), (6, '-u, polynomial form')])
def test_normal_cdf_is_the_librarys_double_one_rounded(native_lib, what, side):
    """Two double evaluations of Phi(u) with relative errors of 2^-50 and 2^-52 round to the same float unless Phi(u) lies that
    close to a rounding midpoint of the float grid: about 2^-25 of the inputs by measure.  Counted exhaustively on both sides
    (the D >= 1/2 side rounds 1 - Q on the 2^-24 grid of [1/2, 1), the D < 1/2 side rounds Q itself); every difference is ONE
    unit of the last place."""
    rep = torch.zeros(16, dtype=torch.int64, device='cuda')
    rc = native_lib.gendr_selftest(what, ctypes.c_void_p(rep.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    r = rep.cpu().tolist()
    assert r[1] == 0x40C00000 + 1, r                       # 0 .. 6.0f
    print('norm_cdf(%s) vs (float)normcdf((double)%s): %d of %d inputs differ, largest difference %d ulp, e.g. bit patterns %s'
          % (side, side, r[0], r[1], r[14], [hex(v) for v in r[2:14] if v]))
    assert r[0] <= 256 and r[14] <= 1, r
