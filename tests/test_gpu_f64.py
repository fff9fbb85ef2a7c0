"""float64 instantiation of the render kernels (reference: AT_DISPATCH_FLOATING_TYPES, kernel.cu:1102,1117,1189) against
the oracle's _f64 body: float64 tensors are computed in double on the device."""
import numpy as np
import pytest
import torch

import parity
import scenes

pytestmark = pytest.mark.gpu


def _run_hip_f64(fv, tex, isz, opts, grad):
    from gendr_amd.functional import renderer as R
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    B, nf = fv.shape[:2]
    faces = torch.from_numpy(np.ascontiguousarray(fv, np.float64)).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(np.ascontiguousarray(tex, np.float64)).cuda()
    rgba, aux, ws = R.native_forward(faces, textures, p)
    assert rgba.dtype == torch.float64
    g = torch.from_numpy(np.ascontiguousarray(grad, np.float64)).cuda()
    gf, gt = R.native_backward(faces, textures, rgba, aux, ws, g, p)
    return dict(rgba=rgba.cpu().numpy(), aggrs_info=aux.cpu().numpy(), grad_faces=gf.cpu().numpy(), grad_textures=gt.cpu().numpy())


CASES = [n for n in scenes.OPTION_MATRIX if n[0] in (
    'uniform_prob_softmax', 'uniform_prob_hardrgb', 'hard_hard_hard', 'gauss_sq_einstein', 'logistic_prob', 'gamma_yager_vertex',
    'cubic_max', 'wigner_hamacher', 'laplace_frank', 'guder_aczel', 'cauchy_dombi', 'reciprocal_ss', 'gumbelmax_prob',
    'exp_prob', 'levy_prob', 'uniform_T4', 'uniform_bg', 'uniform_singleside')]


@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_f64_matches_the_oracle_f64(oracle_mod, native_lib, name, opts):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    fv, tex = scenes.soup(**kw)
    fv, tex = fv.astype(np.float64), tex.astype(np.float64)
    isz = 40
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz)
    h = _run_hip_f64(fv, tex, isz, opts, grad)
    r = parity.run_oracle(fv, tex, isz, opts, grad, np.float64)
    # same double operations in the same order per pair; the libm calls (device vs glibc) differ by ulps of double,
    # the gradient sums by their order: 1e-12 relative to the sum of |contributions|
    # families whose formulas chain pow / log / exp amplify the last-ulp libm differences (1e-16) by the conditioning of
    # the formula; cauchy goes through atanf (float) by the reference's choice
    if 'cauchy' in name:
        tol = 5e-5        # atanf differs by an ulp of FLOAT between device and glibc; dombi's pow amplifies it
    elif any(t in name for t in ('aczel', 'dombi', 'frank', '_ss', 'yager', 'gamma', 'levy', 'gumbel')):
        tol = 1e-8
    else:
        tol = 1e-11
    s = parity.stats(h['rgba'], r['rgba'])
    assert s['max_rel'] <= tol, (name, s)
    s = parity.stats(h['aggrs_info'], r['aggrs_info'])
    assert s['max_rel'] <= tol, (name, s)
    for k, sc in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        s = parity.stats(h[k], r[k], scale=r[sc])
        assert s['max_rel'] <= 10 * tol, (name, k, s)


@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_f64_matches_the_torch_restatement(native_lib, name, opts):
    """The same comparison against oracle/torch_ref.py -- the second, differently structured restatement of kernel.cu
    (whole-image tensor operations per face, its own CDF / t-conorm / texel code), which shares neither structure nor helper
    functions with the device's per-pixel loop: the C oracle's _f64 body and csrc/compat/gendr_f64.h are both per-pixel
    transcriptions by the same hand, so their agreement alone says little."""
    from oracle import torch_ref
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    fv, tex = scenes.sphere(**kw)
    fv, tex = fv[:1].astype(np.float64), tex[:1].astype(np.float64)
    isz = 32
    grad = np.random.RandomState(1).randn(1, 4, isz, isz)
    h = _run_hip_f64(fv, tex, isz, opts, grad)
    t = torch_ref.render(torch.from_numpy(fv), torch.from_numpy(tex), isz, grad=torch.from_numpy(grad),
                         **{k: v for k, v in opts.items() if k != 'T'})
    t = {k: v.numpy() for k, v in t.items()}
    tol = 5e-5 if 'cauchy' in name else (1e-7 if any(x in name for x in ('aczel', 'dombi', 'frank', '_ss', 'yager', 'gamma', 'levy', 'gumbel')) else 1e-9)
    for k in ('rgba', 'aggrs_info'):
        bad = ~np.isclose(h[k], t[k], rtol=tol, atol=1e-12, equal_nan=True)
        assert bad.mean() <= 2e-3, (name, k, float(bad.mean()), float(np.nanmax(np.abs(h[k] - t[k]))))   # pixels on a skip threshold may flip
    for k in ('grad_faces', 'grad_textures'):
        got, want = h[k].reshape(t[k].shape), t[k]
        scale = np.maximum(np.maximum(np.abs(want), 1e-3 * np.abs(want).max()), 1e-300)   # heaviside: the xy gradient is exactly 0
        rel = np.abs(got - want) / scale
        assert np.percentile(rel, 99) <= 1e4 * tol, (name, k, float(np.percentile(rel, 99)))


def test_autograd_keeps_float64(native_lib):
    from gendr_amd.functional import render
    fv, tex = scenes.sphere(B=1)
    fv64 = torch.from_numpy(fv).double().cuda().requires_grad_(True)
    tex64 = torch.from_numpy(tex).double().cuda().requires_grad_(True)
    img = render(fv64, tex64, image_size=32, dist_func='logistic', dist_scale=2e-2)
    assert img.dtype == torch.float64
    img.sum().backward()
    assert fv64.grad.dtype == torch.float64 and tex64.grad.dtype == torch.float64
    img32 = render(fv64.detach().float(), tex64.detach().float(), image_size=32, dist_func='logistic', dist_scale=2e-2)
    assert img32.dtype == torch.float32
    # same picture, different arithmetic (single pixels on a grazing face may differ by more: SURVEY H1)
    assert float((img32.double() - img.detach()).abs().mean()) < 1e-5


def test_finite_differences_of_the_f64_forward(native_lib):
    """d/dx of sum(alpha * w) by central differences on the double forward == the double backward (alpha only: the
    reference's hand-derived RGB backward is not the full derivative, e.g. it ignores d depth / d xy)."""
    from gendr_amd.functional import render
    fv, tex = scenes.sphere(B=1)
    fv = torch.from_numpy(fv).double().cuda()
    tex = torch.from_numpy(tex).double().cuda()
    w = torch.randn(1, 4, 32, 32, dtype=torch.float64, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    w[:, :3] = 0
    opts = dict(image_size=32, dist_func='gaussian', dist_scale=3e-2, aggr_alpha_func='einstein')
    x = fv.clone().requires_grad_(True)
    (render(x, tex, **opts) * w).sum().backward()
    g = x.grad.reshape(-1)
    flat = fv.reshape(-1)
    idx = torch.argsort(g.abs(), descending=True)[:6]
    for i in idx.tolist():
        h = 1e-6
        a, b = flat.clone(), flat.clone()
        a[i] += h; b[i] -= h
        num = ((render(a.view_as(fv), tex, **opts) * w).sum() - (render(b.view_as(fv), tex, **opts) * w).sum()) / (2 * h)
        assert abs(float(num) - float(g[i])) <= 1e-4 * max(1.0, abs(float(g[i]))), (i, float(num), float(g[i]))
