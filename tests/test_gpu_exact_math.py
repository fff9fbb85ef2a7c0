"""The short correctly-rounded square root and reciprocal of the pair math (gendr_math.h: sqrt_rn, rcp_rn) equal
sqrtf(x) and 1.f / x for EVERY float in [2^-96, 2^96]: exhaustive check on the GPU (1.6e9 bit patterns each)."""
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
