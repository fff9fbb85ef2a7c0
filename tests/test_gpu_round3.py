"""Round-3 additions of the C ABI (gendr_params, ABI 5): the deterministic backward, the caller's limit on the coverage
pool (a pool that runs out must change nothing but speed), aggrs_info left unwritten for unlisted tiles."""
import ctypes

import numpy as np
import pytest
import torch

import criteria
import parity
import scenes

pytestmark = pytest.mark.gpu

CASES = [(n, o) for n, o in scenes.OPTION_MATRIX if n in (
    'uniform_prob_softmax', 'uniform_prob_hardrgb', 'gauss_sq_einstein', 'logistic_prob', 'gamma_yager_vertex', 'cauchy_dombi',
    'uniform_T4', 'uniform_T9_clamp', 'hard_hard_hard', 'uniform_singleside')]


def _inputs(opts, maker=scenes.sphere):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    return maker(**kw)


@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_deterministic_backward_is_bit_reproducible_and_correct(oracle_mod, native_lib, name, opts):
    """gendr_params.deterministic = 1: two calls return bit-identical gradients (the default path's float atomics do not
    promise that, experiments/train_reconstruction.py:582-586), the forward pass is untouched, and the gradients pass the
    same element-wise rule against the oracle as the default path's."""
    for maker, isz in ((scenes.sphere, 64), (scenes.soup, 48)):
        fv, tex = _inputs(opts, maker)
        grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        d1 = parity.run_hip(fv, tex, isz, dict(opts, deterministic=1), grad)
        d2 = parity.run_hip(fv, tex, isz, dict(opts, deterministic=1), grad)
        for k in ('grad_faces', 'grad_textures'):
            assert np.array_equal(d1[k], d2[k], equal_nan=True), (name, k)
        a = parity.run_hip(fv, tex, isz, opts, grad)
        assert np.array_equal(a['rgba'], d1['rgba'], equal_nan=True) and np.array_equal(a['aggrs_info'], d1['aggrs_info'], equal_nan=True)
        bad, rep, refs = criteria.check_case(fv, tex, isz, opts, d1, grad)
        assert not bad, (name, bad)
        o32 = refs['o32']
        for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
            s = parity.stats(d1[k], a[k], scale=o32[ak].reshape(a[k].shape))
            assert s['max_rel'] <= 5e-6, (name, k, s)        # the two paths differ by summation order only


def test_deterministic_backward_is_independent_of_the_batch(native_lib):
    """One writer and one summation order per gradient element: an item's gradients do not depend on what else is in the
    batch (with the clamped texel mode: the reference's texel-index overflow reads the next item's texel otherwise)."""
    fv, tex = scenes.sphere(B=2)
    grad = np.random.RandomState(1).randn(2, 4, 64, 64).astype(np.float32)
    o = dict(deterministic=1, texel_mode=1)
    both = parity.run_hip(fv, tex, 64, o, grad)
    for i in range(2):
        one = parity.run_hip(fv[i:i + 1], tex[i:i + 1], 64, o, grad[i:i + 1])
        assert np.array_equal(one['grad_faces'][0], both['grad_faces'][i])
        assert np.array_equal(one['grad_textures'][0], both['grad_textures'][i])


def test_deterministic_through_the_autograd_function(native_lib, monkeypatch):
    import gendr_amd
    from gendr_amd.synthetic import benchmark_scene
    monkeypatch.setenv('GENDR_DETERMINISTIC', '1')
    fv0, tex0 = benchmark_scene(4, subdivisions=2, device='cuda:0')
    g = torch.randn(4, 4, 96, 96, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    outs = []
    for _ in range(2):
        fv, tex = fv0.clone().requires_grad_(True), tex0.clone().requires_grad_(True)
        img = gendr_amd.functional.render(fv, tex, image_size=96)
        img.backward(g)
        outs.append((fv.grad.clone(), tex.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("limit", [8, 1000])
@pytest.mark.parametrize("name,opts", CASES[:5], ids=[n for n, _ in CASES[:5]])
def test_exhausted_entry_pool_changes_nothing_but_speed(native_lib, name, opts, limit):
    """gendr_params.pool_entries_max: tiles that find the pool exhausted take the render kernels' own walk over all faces;
    the allocation counter cannot wrap or hand out another region's slice (64-bit, advanced only while the request fits)."""
    fv, tex = _inputs(opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, 64, 64).astype(np.float32)
    a = parity.run_hip(fv, tex, 64, opts, grad)
    b = parity.run_hip(fv, tex, 64, dict(opts, pool_entries_max=limit), grad)
    assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True)
    assert np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True)
    for k in ('grad_faces', 'grad_textures'):
        scale = max(1e-30, float(np.nanmax(np.abs(a[k]))))
        assert float(np.nanmax(np.abs(a[k] - b[k]))) <= 2e-5 * scale, k


def test_unlisted_aux_is_left_alone_when_asked(native_lib):
    """skip_unlisted_aux = 1 (what the autograd Function sets): RGBA identical, aggrs_info identical wherever a face
    reaches the pixel's tile, untouched (the sentinel survives) where none does; gradients identical."""
    from gendr_amd.functional import renderer as R
    fv, tex = scenes.sphere()
    B, nf = fv.shape[:2]
    isz = 128
    o, extra = parity.split_options({})
    faces = torch.from_numpy(fv).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(tex).cuda()
    p0 = parity.hip_params(isz, o, extra)
    rgba0, aux0, rec0 = R.native_forward(faces, textures, p0)
    p1 = parity.hip_params(isz, o, dict(extra, skip_unlisted_aux=1))
    sentinel = torch.full((B, 2, isz, isz), -777.0, device='cuda')
    rgba1, aux1, rec1 = R.native_forward(faces, textures, p1, aggrs_info=sentinel.clone())
    assert torch.equal(rgba0, rgba1)
    untouched = aux1 == -777.0
    assert bool(untouched.any()) and bool((~untouched).any())
    assert torch.equal(aux0[~untouched], aux1[~untouched])
    assert bool((rgba0[:, 3][untouched[:, 0]] == 0).all())                  # only pixels no face reaches
    g = torch.randn(B, 4, isz, isz, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    gf0, gt0 = R.native_backward(faces, textures, rgba0, aux0, rec0, g, p0)
    gf1, gt1 = R.native_backward(faces, textures, rgba1, aux1, rec1, g, p1)
    assert float((gf0 - gf1).abs().max()) <= 2e-5 * float(gf0.abs().max())
    assert float((gt0 - gt1).abs().max()) <= 2e-5 * float(gt0.abs().max())
