"""HIP path (through the C ABI) against the CPU oracle on identical inputs.  Needs an MI355X."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import criteria
import parity
import scenes

pytestmark = pytest.mark.gpu

SCENES = {'soup': (scenes.soup, 48), 'sphere': (scenes.sphere, 64), 'slivers': (scenes.slivers, 64)}


def _scene(scene, opts):
    maker, isz = SCENES[scene]
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    fv, tex = maker(**kw)
    return fv, tex, isz


@pytest.mark.parametrize("scene", sorted(SCENES))
@pytest.mark.parametrize("name,opts", scenes.OPTION_MATRIX, ids=[n for n, _ in scenes.OPTION_MATRIX])
def test_option_matrix(oracle_mod, native_lib, scene, name, opts):
    fv, tex, isz = _scene(scene, opts)
    res, h, r = parity.compare(fv, tex, isz, opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    bad, rep, _ = criteria.check_case(fv, tex, isz, opts, h, grad, oracle_f32=r, key='%s:%s' % (scene, name))      # element-wise rule + the cap on its loosened share
    assert not bad, (scene, name, bad)
    if criteria.alpha_is_algebraic(name):
        assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3]), 'alpha must be bit-exact on algebraic paths'
    if opts.get('aggr_rgb_func') == 'hard' and criteria.alpha_is_algebraic(name):
        assert np.array_equal(h['rgba'], r['rgba']) and np.array_equal(h['aggrs_info'], r['aggrs_info'])


@pytest.mark.parametrize("scene", sorted(SCENES))
@pytest.mark.parametrize("name,opts", scenes.OPTION_MATRIX, ids=[n for n, _ in scenes.OPTION_MATRIX])
def test_culling_is_exact(native_lib, scene, name, opts):
    """Tile culling only removes pairs the reference itself skips: results are bit-identical to the
    all-pairs traversal (forward) and equal up to fp32 summation order (backward)."""
    fv, tex, isz = _scene(scene, opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    a = parity.run_hip(fv, tex, isz, opts, grad)
    b = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
    assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True)
    assert np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True)
    for k in ('grad_faces', 'grad_textures'):
        scale = max(1e-30, float(np.nanmax(np.abs(b[k]))))
        assert float(np.nanmax(np.abs(a[k] - b[k]))) <= 2e-5 * scale, k


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', '*.npz')))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_vectors(oracle_mod, native_lib, path):
    z = np.load(path)
    opts = json.loads(str(z['options']))
    isz = int(z['image_size'])
    h = parity.run_hip(z['fv'], z['tex'], isz, opts, z['grad'])
    # the committed vectors are the nominal fp32 oracle of the element-wise rule (the sensitivity runs are recomputed)
    golden = dict(rgba=z['rgba'], aggrs_info=z['aggrs_info'], grad_faces=z['grad_faces'], grad_textures=z['grad_textures'],
                  abs_faces=z['abs_faces'], abs_textures=z['abs_textures'])
    bad, rep, _ = criteria.check_case(z['fv'], z['tex'], isz, opts, h, z['grad'], oracle_f32=golden)
    assert not bad, (os.path.basename(path), bad)


def test_faces_info_kernel_matches_oracle(oracle_mod, native_lib):
    import ctypes
    fv, _ = scenes.soup(B=2, nf=48)
    want = oracle_mod.face_info(fv)
    faces = torch.from_numpy(fv).reshape(2, 48, 9).cuda()
    info = torch.empty(2, 48, 27, device='cuda')
    assert native_lib.gendr_face_info(ctypes.c_void_p(faces.data_ptr()), ctypes.c_void_p(info.data_ptr()), 2, 48,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(info.cpu().numpy(), want, equal_nan=True)     # pure +,-,*,/ : bit-exact


def test_many_faces_and_list_chunking(oracle_mod, native_lib):
    """nf > the 4096-entry scan range and (with a heavy-tailed distribution, no cull radius) more list
    entries than fit the LDS record window: exercises every chunking loop."""
    rs = np.random.RandomState(3)
    nf = 4500
    fv = np.zeros((1, nf, 3, 3), np.float32)
    c = rs.uniform(-0.9, 0.9, (nf, 1, 2))
    fv[0, :, :, :2] = c + 0.05 * rs.uniform(-1, 1, (nf, 3, 2))
    fv[0, :, :, 2] = rs.uniform(1.5, 5, (nf, 3))
    tex = rs.uniform(0, 1, (1, nf, 1, 3)).astype(np.float32)
    for opts in (dict(), dict(dist_func='cauchy', dist_scale=1e-3, aggr_alpha_func='einstein')):
        res, h, r = parity.compare(fv, tex, 32, opts)
        grad = np.random.RandomState(1).randn(1, 4, 32, 32).astype(np.float32)
        bad, _, _ = criteria.check_case(fv, tex, 32, opts, h, grad, oracle_f32=r)
        assert not bad, bad


@pytest.mark.parametrize("isz", [1, 7, 50])
def test_image_sizes_that_are_not_tile_multiples(oracle_mod, native_lib, isz):
    fv, tex = scenes.soup(B=2, nf=24, seed=4)
    res, h, r = parity.compare(fv, tex, isz, {})
    assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3])
    assert res['rgba']['max_rel'] <= 1e-5 and res['grad_faces_cond']['max_rel'] <= 1e-5


def test_no_faces_renders_background(native_lib):
    fv = np.zeros((2, 0, 3, 3), np.float32)
    tex = np.zeros((2, 0, 1, 3), np.float32)
    h = parity.run_hip(fv, tex, 20, dict(background=(0.25, 0.5, 0.75)))
    assert np.all(h['rgba'][:, 3] == 0)
    assert np.allclose(h['rgba'][:, :3].reshape(2, 3, -1).mean(-1), [[0.25, 0.5, 0.75]] * 2)
