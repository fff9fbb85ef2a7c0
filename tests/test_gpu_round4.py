"""Round-4 paths of the render kernels: dense entries, pixel mode and region tags (long-tailed distributions: an entry covers
most of its tile), on a scene small enough for the oracle -- exercised for certain (the workspace is read back: entries with
all 64 pixels, tiles over the pixel-mode threshold and tagged entries must all occur) and held to the all-pairs traversal bit
for bit, to the oracle by the element-wise rule, and to the reference's own kernels by the flat gate."""
import numpy as np
import pytest
import torch

import criteria
import parity
import pin
import scenes

pytestmark = pytest.mark.gpu

OPTS = dict(dist_func='logistic', dist_scale=3e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)


def _entries(ws, B, nf, isz, rec_floats):
    """(queue records [tiles,4], entries [n,4]) of a float32 workspace (layout: gendr_capi.hip workspace_layout)."""
    w = ws.cpu().numpy()
    a256 = lambda v: (v + 255) // 256 * 256
    tiles_x = (isz + 7) // 8
    tiles = B * tiles_x * tiles_x
    chunks = (nf + 63) // 64
    off = a256(B * nf * 4 * 4) + a256(B * nf * rec_floats * 4) + a256(tiles * chunks * 8) + a256(tiles * 4)
    info = w[off:off + tiles * 16].view(np.int32).reshape(tiles, 4)
    ents = w[off + a256(tiles * 16):].view(np.int32)
    # only the first `queue length` records of each of the 8 queues were written by this call (the rest of the region is whatever
    # the allocator handed out); the lengths sit in the control block at the end of the workspace
    control = w[len(w) - 24 * 1024 * 4:].view(np.int32)
    rows = [info[x * tiles // 8:x * tiles // 8 + int(control[x * 1024])] for x in range(8)]
    return np.concatenate(rows), ents


@pytest.mark.parametrize("team", [-1, 1])          # the one-wave kernels' paths, and the team kernels' (round 5: automatic at this size)
@pytest.mark.parametrize("rgb", ['softmax', 'hard'])
def test_pixel_mode_dense_entries_and_region_tags(oracle_mod, native_lib, rgb, team):
    from gendr_amd.functional import renderer as R
    fv, tex = scenes.sphere(B=2)
    isz = 64
    opts = dict(OPTS, aggr_rgb_func=rgb, team=team)
    grad = np.random.RandomState(2).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    # the paths are taken: read the coverage entries back
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    faces = torch.from_numpy(fv).reshape(2, -1, 9).cuda().contiguous()
    t = torch.from_numpy(tex).cuda().contiguous()
    _, _, ws = R.native_forward(faces, t, p)
    torch.cuda.synchronize()
    info, ents = _entries(ws, 2, fv.shape[1], isz, 56)
    listed = info[(info[:, 1] >= 0) & (info[:, 2] > 0)]
    assert len(listed) > 0
    pixel_mode = listed[listed[:, 3] >= 36 * listed[:, 2]]
    assert len(pixel_mode) > len(listed) // 4, 'the scene should put a good share of its tiles over the pixel-mode threshold'
    full = tagged = total = 0
    for tile, first, cnt, pairs in listed:
        e = ents[first * 4:(first + cnt) * 4].reshape(cnt, 4)
        full += int(((e[:, 1] & 255) == 64).sum())
        tagged += int((((e[:, 1] >> 8) & 3) != 0).sum())
        total += cnt
    assert full > 0 and tagged > total // 10, (full, tagged, total)
    # ... and change nothing: culled (entries, pixel mode, tags) == all pairs (no pool: the reference's own traversal)
    a = parity.run_hip(fv, tex, isz, opts, grad)
    b = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
    assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True) and np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True)
    for k in ('grad_faces', 'grad_textures'):
        assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * max(1e-30, float(np.abs(b[k]).max())), k
    bad, _, _ = criteria.check_case(fv, tex, isz, opts, a, grad)
    assert not bad, bad
    if parity.reference_available():
        r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
        c = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
        assert np.array_equal(a['rgba'], r['rgba']), 'logistic / probabilistic: rgba bit-identical to the reference\'s kernels'
        m = pin.measure(a, r, c['abs_faces'], c['abs_textures'])
        assert all(x['max'] <= 1e-5 for x in m.values()), m


def test_silhouette_kernels_take_the_same_paths(native_lib):
    """The alpha-only kernels share for_each_batch: alpha == render()[:, 3] bit for bit in the dense regime as well."""
    from gendr_amd.functional.silhouette import render_silhouette
    from gendr_amd.functional.renderer import render
    fv, tex = scenes.sphere(B=2)
    f = torch.from_numpy(fv).cuda().requires_grad_(True)
    t = torch.from_numpy(tex).cuda()
    kw = dict(image_size=64, dist_func='logistic', dist_scale=3e-2, aggr_alpha_func='probabilistic')
    full = render(f, t, aggr_rgb_func='hard', double_side=False, **kw)
    alpha = render_silhouette(f, **kw)
    assert torch.equal(alpha, full[:, 3])
    g = torch.randn_like(alpha)
    ga, = torch.autograd.grad(alpha, f, g, retain_graph=True)
    gb, = torch.autograd.grad(full[:, 3], f, g)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())


@pytest.mark.parametrize("opts", [dict(dist_func=0), dict(dist_func='gamma', dist_shape=3.5, dist_scale=1e-2)], ids=['hard_prob_softmax', 'gamma35_prob'])
def test_pair_hints_and_dense_entries_in_one_tile(oracle_mod, native_lib, opts):
    """Regression (tools/fuzz_parity.py, round 4): an image-filling face is a dense entry of every tile -- it runs outside the
    batches, so the batches' pair hints must not decide whether the tile has anything to differentiate.  With the shortcut that
    skipped tiles whose hinted batches were all dead, its texture gradient came out 40 % short."""
    fv, tex = scenes.soup(B=5, nf=127, seed=3)
    isz = 200
    grad = np.random.RandomState(1).randn(5, 4, isz, isz).astype(np.float32)
    o = parity.run_oracle(fv, tex, isz, opts, grad)
    with_hints = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=1), grad)
    without = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=-1), grad)
    for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        # hints change nothing but speed: the two calls differ by the order of their atomics only
        e = parity.rel_error(with_hints[k], without[k].reshape(with_hints[k].shape), scale=o[ak].reshape(with_hints[k].shape), floor=parity.GRAD_FLOOR)
        assert e.max() <= 2e-5, (k, float(e.max()), np.unravel_index(int(e.argmax()), e.shape))
    e = parity.rel_error(with_hints['grad_textures'], o['grad_textures'], scale=o['abs_textures'], floor=parity.GRAD_FLOOR)
    assert e.max() <= 1e-3, float(e.max())          # (against the oracle: the bug was 0.4)


@pytest.mark.parametrize("isz", [8, 64, 100])
def test_box_edge_next_to_a_pixel_centre(oracle_mod, native_lib, isz):
    """The reference's border test (kernel.cu:747: x > max + sqrt(eps * scale) || x < min - ... ) is what ends a face's reach when
    dist_eps is small -- no later stage repeats it for a listed pair, so the coverage kernel's box test has to be the per-pixel
    expression, not a widened estimate.  (Round 3's column intervals widened the box ends by 2^-8 column with the edge thresholds:
    tools/fuzz_parity.py case 255 had a box edge 0.0015 column outside a pixel centre and rendered that column; 16 of 2304 rgba elements
    off by up to 0.7.)  Faces whose box edges -- left, right, bottom, top -- lie a few ulps to 3e-3 column on either side of a pixel
    centre: culled == all-pairs bit for bit (the all-pairs traversal applies the reference's tests pixel by pixel; it is what the
    pin tests hold against the reference's kernels)."""
    opts = dict(dist_eps=1.5, dist_scale=0.2)
    sthr = float(np.sqrt(np.float32(np.float32(1.5) * np.float32(0.2))))
    pitch = 2.0 / isz
    centre = lambda i: (2 * i + 1 - isz) / isz
    faces = []
    rs = np.random.RandomState(5)
    for side in range(4):
        for delta in (-3e-3, -1e-3, -1e-4, -2e-7, 0.0, 2e-7, 1e-4, 1e-3, 3e-3):
            i = int(rs.randint(1, isz - 1))
            edge = centre(i) + delta * pitch                      # where the box edge is to lie
            j = centre(int(rs.randint(isz // 4, 3 * isz // 4)))  # the face's position along the other axis
            ext = 0.05
            # ZERO-AREA faces (two corners coincide): their error bound is infinite, so their cull box is the reference's border and
            # nothing else -- and their computed distances are whatever the clamped determinant (kernel.cu:653) makes of them, so a
            # pixel just outside the border is not caught by the distance test :769 either (a well-conditioned face hides the
            # defect: beyond its border the squared distance exceeds dist_eps * dist_scale by construction)
            if side == 0:   lo_x, lo_y = edge + sthr, j                  # left border  = min x - sthr
            elif side == 1: lo_x, lo_y = edge - sthr - ext, j            # right border = max x + sthr
            elif side == 2: lo_x, lo_y = j, edge + sthr                  # bottom
            else:           lo_x, lo_y = j, edge - sthr - ext            # top
            for kind in range(2):
                if kind == 0: tri = [[lo_x, lo_y + ext, 2.0], [lo_x, lo_y + ext, 2.0], [lo_x + ext, lo_y, 4.3]]
                else:         tri = [[lo_x, lo_y, 2.0], [lo_x + ext, lo_y + ext, 3.0], [lo_x + 0.5 * ext, lo_y + 0.5 * ext, 2.5]]    # three corners on a line
                faces.append(tri)
    fv = np.asarray(faces, np.float32)[None]                                                                  # [1, nf, 3, 3]
    fv = np.concatenate([fv, fv[:, ::-1] * np.float32([1, -1, 1])], 0)                                        # a second image, mirrored
    nf = fv.shape[1]
    tex = np.random.RandomState(6).rand(2, nf, 4, 3).astype(np.float32)
    grad = np.random.RandomState(7).randn(2, 4, isz, isz).astype(np.float32)
    for one in (None, 0, 17, 40, nf - 1):                          # all faces together, and single faces (nothing else to hide behind)
        f1, t1 = (fv, tex) if one is None else (fv[:, one:one + 1], tex[:, one:one + 1])
        a = parity.run_hip(f1, t1, isz, opts, grad)
        b = parity.run_hip(f1, t1, isz, dict(opts, cull=0), grad)
        assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True) and np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True), one
        for k in ('grad_faces', 'grad_textures'):
            assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * max(1e-30, float(np.abs(b[k]).max())), (one, k)
    if isz == 8:
        # the face of the fuzz case itself (image 7, face 1 of tools/fuzz_parity.py 500 0, case 255): border at x = -0.37462, the
        # centres of pixel column 2 at -0.375; before the fix the culled traversal rendered rows 2-5 of that column (rgba off by 0.5)
        f1 = np.float32([[[[0.17310063540935516, -0.044718850404024124, 2.000278949737549],
                           [0.17310063540935516, -0.044718850404024124, 2.000278949737549],
                           [0.2192317247390747, -0.0495893619954586, 4.278071880340576]]]])
        for shift in (0.0, 0.25, -0.5):                       # ... and the same face in front of other columns
            f2 = f1 + np.float32([shift, 0, 0])
            t2 = np.random.RandomState(8).rand(1, 1, 4, 3).astype(np.float32)
            a = parity.run_hip(f2, t2, isz, opts, None)
            b = parity.run_hip(f2, t2, isz, dict(opts, cull=0), None)
            assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True) and np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True), shift
    # every face alone, forward only
    for k in range(nf):
        a = parity.run_hip(fv[:, k:k + 1], tex[:, k:k + 1], isz, opts, None)
        b = parity.run_hip(fv[:, k:k + 1], tex[:, k:k + 1], isz, dict(opts, cull=0), None)
        assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True), k
