"""The pin: outputs of the REFERENCE's own render kernels, run here, against the CPU restatement and the HIP product.

oracle/_ref/gendr_ref_render.co is the device half of /root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu
compiled for gfx950 (oracle/build_ref.py: PyTorch-ROCm's own CUDA -> HIP translator for the two includes, clang
--cuda-device-only, -ffp-contract=off, no file of the reference edited, nothing supplied in place of anything); it
travels to the GPU box as a built artefact.  oracle/ref_gpu.py launches its kernels with the reference's launch shapes.

  * float64: the restatement (oracle/gendr_oracle_body.inc, S = double) has to reproduce the reference kernels'
    double instantiation to rounding noise on the whole option matrix -- libm differences are 1e-16 there, so a
    misread formula, promotion, threshold or traversal order shows at full size;
  * float32: the restatement and the HIP product are both held to the reference kernels' float results by the same
    element-wise rule the product is held to against the restatement (tests/criteria.py), and the face preprocessing
    (kernel.cu:620) bit for bit.
"""
import numpy as np
import pytest

import criteria
import parity
import scenes

pytestmark = pytest.mark.gpu

MATRIX = [(n, o) for n, o in scenes.OPTION_MATRIX if o.get('texel_mode', 0) == 0]
IDS = [n for n, _ in MATRIX]


@pytest.fixture(scope='module')
def ref_kernels():
    if not parity.reference_available():
        pytest.skip('oracle/_ref is not built (python -m oracle.build_ref needs /root/reference)')


def _inputs(opts, scene):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    if scene == 'soup':
        return scenes.soup(B=2, nf=24, **kw)
    if scene == 'slivers':
        return scenes.slivers(B=1, nf=36, **kw)
    return scenes.sphere(B=2, **kw)


def _grad(fv, isz, dtype):
    return np.random.RandomState(5).randn(fv.shape[0], 4, isz, isz).astype(dtype)


GRAD_FLOOR = parity.GRAD_FLOOR
_rel = parity.rel_error


@pytest.mark.parametrize("scene", ['soup', 'sphere', 'slivers'])
@pytest.mark.parametrize("name,opts", MATRIX, ids=IDS)
def test_restatement_reproduces_reference_kernels_f64(oracle_mod, ref_kernels, name, opts, scene):
    isz = 32
    fv, tex = _inputs(opts, scene)
    grad = _grad(fv, isz, np.float64)
    r = parity.run_reference(fv, tex, isz, opts, grad, np.float64)
    c = parity.run_oracle(fv.astype(np.float64), tex.astype(np.float64), isz, opts, grad, np.float64)
    assert np.array_equal(r['faces_info'], c['faces_info'], equal_nan=True), 'face preprocessing (kernel.cu:620) must agree bit for bit'
    # cauchy evaluates atanf in FLOAT whatever scalar_t is (kernel.cu:258): the device's and glibc's float results differ in the last bit
    tol = 5e-5 if opts.get('dist_func') == 'cauchy' else 1e-9
    for k in ('rgba', 'aggrs_info'):
        e = _rel(r[k], c[k])
        assert e.max() <= tol, (k, float(e.max()), np.unravel_index(int(e.argmax()), e.shape))
    for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        e = _rel(r[k], c[k], scale=c[ak], floor=GRAD_FLOOR)
        assert e.max() <= max(tol, 1e-8), (k, float(e.max()), np.unravel_index(int(e.argmax()), e.shape))


def _against_reference(fv, tex, isz, opts, got, grad, ref_out, n_jitter):
    """`got` (restatement or product, float) against the reference kernels' float output under the element-wise rule:
    the noise and threshold terms come from the restatement's jittered / shifted evaluations, the value compared with is
    the reference's."""
    refs = criteria.references(fv, tex, isz, opts, grad, n_jitter=n_jitter)
    pinned = dict(refs, o32=dict(ref_out, abs_faces=refs['o32']['abs_faces'], abs_textures=refs['o32']['abs_textures'],
                                 grad_faces=ref_out['grad_faces'].reshape(refs['o32']['grad_faces'].shape)))
    return criteria.failures(criteria.elementwise(got, pinned)), refs


@pytest.mark.parametrize("scene", ['soup', 'sphere', 'slivers'])
@pytest.mark.parametrize("name,opts", MATRIX, ids=IDS)
def test_restatement_and_product_against_reference_kernels_f32(oracle_mod, native_lib, ref_kernels, name, opts, scene):
    isz = 32
    fv, tex = _inputs(opts, scene)
    grad = _grad(fv, isz, np.float32)
    r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
    c = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
    assert np.array_equal(r['faces_info'], c['faces_info'], equal_nan=True)
    if criteria.alpha_is_algebraic(name):
        assert np.array_equal(r['rgba'][:, 3], c['rgba'][:, 3], equal_nan=True), 'alpha without a libm call must agree bit for bit'
    bad, _ = _against_reference(fv, tex, isz, opts, c, grad, r, len(criteria.JITTER_MODES))
    assert not bad, ('restatement vs reference kernels', bad)
    h = parity.run_hip(fv, tex, isz, opts, grad)
    if criteria.alpha_is_algebraic(name):
        assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3], equal_nan=True)
    bad, _ = _against_reference(fv, tex, isz, opts, h, grad, r, len(criteria.JITTER_MODES))
    assert not bad, ('HIP product vs reference kernels', bad)


C2 = dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)
C3 = dict(dist_func='gaussian', dist_scale=1e-4, dist_squared=True, aggr_alpha_func='einstein', double_side=False)
C4 = dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)
C5 = dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
          aggr_rgb_func='softmax', texture_type='vertex', double_side=False)


@pytest.mark.parametrize("name,opts,isz", [('C2', C2, 256), ('C3', C3, 256), ('C4', C4, 512), ('C5', C5, 768)])
def test_product_against_reference_kernels_at_baseline_configs(oracle_mod, native_lib, ref_kernels, name, opts, isz):
    """BASELINE.json's configurations (C5's option set at 768^2), one frame of the benchmark mesh: the HIP product
    against the reference kernels' float output; float64: the restatement against the reference kernels."""
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(2, texture='vertex' if name == 'C5' else 'surface')
    fv, tex = fv.numpy()[1:2], tex.numpy()[1:2]
    grad = np.random.RandomState(1).randn(1, 4, isz, isz).astype(np.float32)
    r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
    h = parity.run_hip(fv, tex, isz, opts, grad)
    if name == 'C2':
        assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3])
    bad, _ = _against_reference(fv, tex, isz, opts, h, grad, r, 6)
    assert not bad, bad
    r64 = parity.run_reference(fv, tex, isz, opts, grad, np.float64)
    c64 = parity.run_oracle(fv.astype(np.float64), tex.astype(np.float64), isz, opts, grad.astype(np.float64), np.float64)
    assert np.array_equal(r64['faces_info'], c64['faces_info'], equal_nan=True)
    for k in ('rgba', 'aggrs_info'):
        assert _rel(r64[k], c64[k]).max() <= 1e-9, k
    for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        assert _rel(r64[k], c64[k], scale=c64[ak], floor=GRAD_FLOOR).max() <= 1e-8, k


def test_reference_builds_differ_by_contraction(oracle_mod, ref_kernels):
    """Context for every tolerance above: the reference's OWN results move when the compiler fuses a*b+c (nvcc's default;
    clang's default).  Reported, and asserted only to be non-trivial on the sliver scene where it is largest."""
    fv, tex = scenes.slivers(B=1, nf=36)
    grad = _grad(fv, 32, np.float32)
    a = parity.run_reference(fv, tex, 32, {}, grad, np.float32)
    b = parity.run_reference(fv, tex, 32, {}, grad, np.float32, variant='render_fma')
    d = _rel(b['rgba'], a['rgba'])
    print('reference, contraction on vs off: rgba max rel %.3g, differing elements %.2f %%' % (d.max(), 100 * (d > 0).mean()))
    assert np.isfinite(a['rgba']).all()
