"""The coverage kernel's team form (cover_kernel<REC, 8>, gendr_amd/csrc/gendr_kernels.h; chosen by team_cover() in gendr_capi.hip,
forced by gendr_params.team = 2, switched off by -1): one 8-wave workgroup per listed tile, wave w examines faces [16 w, 16 w + 16) of
every 128 the tile lists.  Held to: THE SAME coverage entries, entry counts and pair counts per tile as the one-wave form (read back
from the workspace), hence the same rgba / aggrs_info bit for bit and the same gradients up to the order of the atomics -- also where the
option set has no team render kernel (heavy distributions, vertex textures), with flagged (loose) faces, with more faces per tile than
one round of 128 holds, at odd image sizes and on tiles without a slice of the entry pool."""
import ctypes

import numpy as np
import pytest
import torch

import parity
import scenes

pytestmark = pytest.mark.gpu


def _params(isz, opts):
    o, extra = parity.split_options(opts)
    return parity.hip_params(isz, o, extra)


def _uses_team_cover(B, nf, T, isz, opts):
    from gendr_amd import _native
    return _native.lib().gendr_uses_team_cover(B, nf, T, ctypes.byref(_params(isz, opts)))


def _entries_per_tile(fv, tex, isz, opts):
    """{tile: (entry count, pair count, entries [count, 4] uint32)} of the forward call's workspace (queue records + entry pool;
    the layout of workspace_layout() in gendr_capi.hip, as tools/entrystats.py reads it)."""
    from gendr_amd.functional import renderer as R
    B, nf = fv.shape[:2]
    T = tex.shape[2]
    faces = torch.from_numpy(np.ascontiguousarray(fv, np.float32)).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(np.ascontiguousarray(tex, np.float32)).cuda()
    rgba, aux, ws = R.native_forward(faces, textures, _params(isz, opts))
    torch.cuda.synchronize()
    w = ws.cpu().numpy()
    a256 = lambda v: (v + 255) // 256 * 256
    tiles_x = (isz + 7) // 8
    tiles = B * tiles_x * tiles_x
    chunks = (nf + 63) // 64
    vertex = opts.get('texture_type') == 'vertex'
    rec = 60 if vertex else (56 if T == 1 else 48)
    off = a256(B * nf * 4 * 4) + a256(B * nf * rec * 4) + a256(tiles * chunks * 8) + a256(tiles * 4)
    info = w[off:off + tiles * 16].view(np.int32).reshape(tiles, 4)
    off += a256(tiles * 16)
    control = w[len(w) - 24 * 1024 * 4:].view(np.int32)            # kCtlInts = 24 x kCtlStride (1024); the last region of the layout
    ent = w[off:off + (len(w) - off) // 16 * 16].view(np.uint32).reshape(-1, 4)
    out = {}
    for x in range(8):
        qb = (x * tiles) >> 3                                        # queue_begin()
        for s in range(int(control[x * 1024])):                      # listed tiles of queue x
            tile, first, cnt, pairs = (int(v) for v in info[qb + s])
            assert 0 <= tile < tiles and tile not in out
            out[tile] = (-1, -1, None) if first < 0 else (cnt, pairs, ent[first:first + cnt].copy())
    return out, dict(rgba=rgba.cpu().numpy(), aggrs_info=aux.cpu().numpy())


def _listed(d):
    return {t: v for t, v in d.items() if v[0] > 0}


CASES = [
    # (name, scene, image size, options)
    ('optshape_logistic', lambda: _sphere1280(3), 64, dict(dist_func='logistic', dist_scale=1e-2, aggr_rgb_func='hard', dist_eps=100.)),
    ('gamma_no_team_render_kernel', lambda: _sphere1280(2), 64, dict(dist_func='gamma', dist_shape=2., dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.)),
    ('vertex_textures_gaussian', lambda: scenes.soup(B=2, nf=200, vertex_tex=True, seed=3), 72, dict(dist_func='gaussian', dist_scale=3e-2, texture_type='vertex')),
    ('slivers_loose_faces', lambda: scenes.slivers(B=2), 64, dict(dist_func='logistic', dist_scale=2e-2, double_side=True)),
    ('odd_size_T4', lambda: scenes.soup(B=3, nf=130, T=4, seed=5), 45, dict(dist_func='uniform', dist_scale=5e-2, T=4)),
    ('short_lists', lambda: scenes.sphere(B=2), 100, dict(dist_func='uniform', dist_scale=1e-3)),
]


def _sphere1280(B):
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(B)
    return fv.numpy(), tex.numpy()


@pytest.mark.parametrize("name,make,isz,opts", CASES, ids=[c[0] for c in CASES])
def test_team_cover_writes_the_same_entries(native_lib, name, make, isz, opts):
    fv, tex = make()
    B, nf = fv.shape[:2]
    assert _uses_team_cover(B, nf, tex.shape[2], isz, dict(opts, team=2)) == 1
    assert _uses_team_cover(B, nf, tex.shape[2], isz, dict(opts, team=-1)) == 0
    one, out_one = _entries_per_tile(fv, tex, isz, dict(opts, team=-1))
    team, out_team = _entries_per_tile(fv, tex, isz, dict(opts, team=2))
    lo, lt = _listed(one), _listed(team)
    assert len(lo) > 0 and set(lo) == set(lt), (len(lo), len(lt))
    for t in lo:
        assert lo[t][0] == lt[t][0] and lo[t][1] == lt[t][1], ('entry / pair count of tile', t, lo[t][:2], lt[t][:2])
        assert np.array_equal(lo[t][2], lt[t][2]), ('entries of tile', t)
    if name != 'short_lists':
        assert max(v[0] for v in lo.values()) > 16, 'the scene should list more than one step of faces in some tile'
    for k in ('rgba', 'aggrs_info'):
        assert np.array_equal(out_one[k], out_team[k], equal_nan=True), k


def test_more_faces_per_tile_than_a_round_holds(native_lib):
    """1280 faces at 32^2 with a long tail: the tiles under the object list several hundred faces -- three and more rounds of 128."""
    fv, tex = _sphere1280(2)
    opts = dict(dist_func='logistic', dist_scale=3e-2, aggr_rgb_func='softmax')
    one, out_one = _entries_per_tile(fv, tex, 32, dict(opts, team=-1))
    team, out_team = _entries_per_tile(fv, tex, 32, dict(opts, team=2))
    lo, lt = _listed(one), _listed(team)
    assert set(lo) == set(lt) and max(v[0] for v in lo.values()) > 256
    for t in lo:
        assert lo[t][:2] == lt[t][:2] and np.array_equal(lo[t][2], lt[t][2]), t
    assert np.array_equal(out_one['rgba'], out_team['rgba'])


def test_gradients_and_pool_exhaustion(native_lib):
    """Backward through the entries of the team form (gradients equal up to the order of the atomics), and a pool capped at a few
    entries: tiles without a slice are skipped by both forms and rendered by the render kernels' own walk."""
    fv, tex = _sphere1280(3)
    isz = 64
    grad = np.random.RandomState(2).randn(3, 4, isz, isz).astype(np.float32)
    for extra in (dict(), dict(pool_entries_max=4096)):
        opts = dict(dist_func='gamma', dist_shape=2., dist_scale=1e-2, **extra)
        a = parity.run_hip(fv, tex, isz, dict(opts, team=-1), grad)
        b = parity.run_hip(fv, tex, isz, dict(opts, team=2), grad)
        for k in ('rgba', 'aggrs_info'):
            assert np.array_equal(a[k], b[k], equal_nan=True), (k, extra)
        for k in ('grad_faces', 'grad_textures'):
            assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * max(1e-30, float(np.abs(a[k]).max())), (k, extra)


def test_the_rule(native_lib):
    """Automatic: calls of up to 8192 tiles whose tiles can expect to list 32 faces and more (team_cover() in gendr_capi.hip)."""
    soft = dict(dist_func='logistic', dist_scale=1e-2, aggr_rgb_func='hard', dist_eps=100.)
    assert _uses_team_cover(24, 1280, 1, 64, soft) == 1                                        # opt_shape.py
    assert _uses_team_cover(24, 1280, 1, 64, dict(soft, dist_scale=1e-4)) == 1                 # 1280 faces on 64 tiles: dense whatever the tail
    assert _uses_team_cover(24, 1280, 1, 64, dict(soft, dist_func='gamma', dist_shape=2.)) == 1   # no team render kernel needed
    assert _uses_team_cover(8, 1280, 1, 128, dict(soft, dist_scale=1e-3)) == 0                 # short lists
    assert _uses_team_cover(8, 1280, 1, 128, soft) == 1
    assert _uses_team_cover(4, 1280, 1, 256, dict()) == 0                                      # BASELINE config 2's regime
    assert _uses_team_cover(4, 1280, 1, 256, dict(dist_func='logistic', dist_scale=1e-2)) == 1
    assert _uses_team_cover(256, 1280, 1, 512, dict(dist_func='logistic', dist_scale=1e-2)) == 0   # BASELINE config 4: a million tiles
    assert _uses_team_cover(24, 1280, 1, 64, dict(soft, team=-1)) == 0
    assert _uses_team_cover(24, 1280, 1, 64, dict(soft, cull=0)) == 0
    assert _uses_team_cover(24, 1280, 1, 64, dict(soft, dist_func='cauchy')) == 0              # no cull radius, no entry pool
