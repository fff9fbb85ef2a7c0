"""The reference's Python surface on the GPU: autograd Function, render(), GenDR, the pybind-shaped shim."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import parity
import scenes

pytestmark = pytest.mark.gpu


def _inputs(B=2, nf=32, vertex=False, seed=0):
    fv, tex = scenes.soup(B=B, nf=nf, seed=seed, vertex_tex=vertex)
    return torch.from_numpy(fv).cuda(), torch.from_numpy(tex).cuda()


def test_autograd_matches_native_calls(native_lib):
    import gendr_amd
    fv, tex = _inputs()
    fv.requires_grad_(True)
    tex.requires_grad_(True)
    img = gendr_amd.functional.render(fv, tex, image_size=40, dist_func='logistic', dist_scale=2e-2)
    assert img.shape == (2, 4, 40, 40) and img.dtype == torch.float32
    g = torch.randn_like(img)
    img.backward(g)
    assert fv.grad.shape == fv.shape and tex.grad.shape == tex.shape
    ref = parity.run_hip(fv.detach().cpu().numpy(), tex.detach().cpu().numpy(), 40,
                         dict(dist_func='logistic', dist_scale=2e-2), g.cpu().numpy())
    assert np.array_equal(img.detach().cpu().numpy(), ref['rgba'])
    assert np.allclose(fv.grad.cpu().numpy().reshape(2, 32, 9), ref['grad_faces'], rtol=1e-4, atol=1e-6)


def test_ids_shapes_and_none_parameters(native_lib):
    from gendr_amd.functional import render
    fv, tex = _inputs()
    a = render(fv, tex, image_size=32, dist_func='gaussian', aggr_alpha_func='einstein', aggr_rgb_func='hard')
    b = render(fv.reshape(2, 32, 9), tex, 32, [0, 0, 0], 4, 1e-2, False, None, None, 1e4, 3, None, 0)
    assert torch.equal(a, b)
    c = render(fv, tex, image_size=32, dist_scale=np.float64(1e-2))      # numpy scalar (opt_shape.py:327)
    d = render(fv, tex, image_size=32)
    assert torch.equal(c, d)


def test_invalid_options_raise_value_error(native_lib):
    from gendr_amd.functional import render
    fv, tex = _inputs()
    with pytest.raises(ValueError):
        render(fv, tex, image_size=32, aggr_alpha_func='frank')           # p = None -> 0: invalid for frank
    with pytest.raises(ValueError):
        render(fv, tex, image_size=32, dist_func='gamma', dist_shape=-1.0)
    with pytest.raises(ValueError):
        render(fv, tex, image_size=32, dist_func=23)
    with pytest.raises(ValueError):
        render(fv, tex[:, :5], image_size=32)


def test_gradient_accepts_non_contiguous_and_partial_requires_grad(native_lib):
    from gendr_amd.functional import render
    fv, tex = _inputs()
    fv.requires_grad_(True)
    img = render(fv, tex, image_size=32)
    loss = img.permute(0, 2, 3, 1)[..., 3].sum()          # non-contiguous upstream gradient
    loss.backward()
    assert torch.isfinite(fv.grad).all() and tex.grad is None


def test_gendr_module_and_anti_aliasing(native_lib):
    import gendr_amd
    fv, tex = _inputs(vertex=True)
    r = gendr_amd.GenDR(image_size=24, anti_aliasing=True, texture_type='vertex', dist_func='logistic', dist_scale=3e-2)
    out = r.forward_tensors(fv, tex)
    big = gendr_amd.functional.render(fv, tex, image_size=48, texture_type='vertex', dist_func='logistic',
                                      dist_scale=3e-2, double_side=False)
    assert out.shape == (2, 4, 24, 24)
    assert torch.equal(out, F.avg_pool2d(big, 2, 2))
    mesh = type('MeshLike', (), dict(face_vertices=fv, face_textures=tex))()
    assert torch.equal(r(mesh), out)
    r.dist_scale = 1e-1                                    # attributes are read at call time
    assert not torch.equal(r(mesh), out)


def test_pybind_shaped_shim(native_lib):
    from gendr_amd.cuda import generalized_renderer as gr
    from gendr_amd.functional import render
    fv, tex = _inputs()
    B, nf, isz = 2, 32, 32
    faces = fv.reshape(B, nf, 9).clone().requires_grad_(True)
    bg = [0.1, 0.6, 0.3]
    faces_info = torch.zeros(B, nf, 27, device='cuda')
    aggrs = torch.zeros(B, 2, isz, isz, device='cuda')
    soft = torch.ones(B, 4, isz, isz, device='cuda')
    for k in range(3):
        soft[:, k] *= bg[k]
    args = (isz, 6, 2e-2, False, 0.0, 0.0, 1e4, 2, 0.0, 1, 1e-3, 1e-3, 1.0, 100.0, True, 0)
    fi, ag, sc = gr.forward_render(faces.detach(), tex, faces_info, aggrs, soft, *args)
    want = render(faces, tex, image_size=isz, background_color=bg, dist_func=6, dist_scale=2e-2)
    assert sc is soft and torch.equal(sc, want.detach())
    assert fi.abs().sum() > 0
    g = torch.randn_like(want)
    want.backward(g)
    gf = torch.zeros(B, nf, 9, device='cuda')
    gt = torch.zeros_like(tex)
    gr.backward_render(faces.detach(), tex, soft, faces_info, aggrs, gf, gt, g, *args)
    assert torch.allclose(gf, faces.grad, rtol=1e-4, atol=1e-6)
    with pytest.raises(RuntimeError):
        gr.forward_render(faces.detach().cpu(), tex, faces_info, aggrs, soft, *args)
    assert abs(gr.sigmoid_forward(6, 1.0, 0.0, 0.01, 0.0, 0.0) - 0.5) < 1e-7
    assert abs(gr.t_conorm_forward(2, 0.5, 0.5, 0, 0.0) - 0.75) < 1e-7


def test_runs_on_the_callers_stream_and_device(native_lib):
    from gendr_amd.functional import render
    fv, tex = _inputs()
    base = render(fv, tex, image_size=32)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        other = render(fv, tex, image_size=32)
    s.synchronize()
    assert torch.equal(base, other)


def test_forward_is_deterministic_and_batch_items_independent(native_lib):
    from gendr_amd.functional import render
    fv, tex = _inputs(B=4)
    a = render(fv, tex, image_size=48)
    b = render(fv, tex, image_size=48)
    assert torch.equal(a, b)
    one = render(fv[2:3], tex[2:3], image_size=48)
    assert torch.equal(a[2:3, 3], one[:, 3])               # RGB can differ through the texel-overflow quirk


def test_shape_optimisation_loop_through_the_alias_package(native_lib):
    """The call pattern of experiments/opt_shape.py:259-311: Mesh -> Lighting -> LookAt -> GenDR (soft, hard RGB),
    IoU loss on alpha, Adam step; plus the script's hard renderer under no_grad (opt_shape.py:148-159)."""
    import gendr
    from gendr_amd.synthetic import icosphere
    v, f = icosphere(2)
    target = gendr.Mesh(torch.from_numpy(v * 0.9).cuda()[None].repeat(4, 1, 1), torch.from_numpy(f).int().cuda()[None].repeat(4, 1, 1))
    verts = torch.nn.Parameter(torch.from_numpy(v * 0.6).cuda())
    faces = torch.from_numpy(f).int().cuda()
    cam = gendr.LookAt(viewing_angle=15)
    cam.set_eyes_from_angles(torch.full((4,), 2.732), torch.full((4,), 30.0), torch.tensor([0.0, 90.0, 180.0, 270.0]))
    lights = gendr.Lighting()
    soft = gendr.GenDR(image_size=64, dist_func='uniform', dist_scale=None, dist_eps=1e4, aggr_alpha_func='probabilistic',
                       aggr_rgb_func='hard')
    soft.dist_scale = np.float64(10 ** -1.5)                # set by attribute, numpy scalar (opt_shape.py:289,327)
    hard = gendr.GenDR(image_size=64, dist_func=0, dist_scale=0.0, aggr_alpha_func=0, aggr_rgb_func='hard')
    with torch.no_grad():
        ref = hard(cam(lights(target)))[:, 3]
    assert set(ref.unique().tolist()) <= {0.0, 1.0}
    opt = torch.optim.Adam([verts], lr=0.02)
    first = None
    for _ in range(15):
        mesh = gendr.Mesh(verts[None].repeat(4, 1, 1), faces[None].repeat(4, 1, 1))
        pred = soft(cam(lights(mesh)))[:, 3]
        iou = (pred * ref).sum((1, 2)) / ((pred + ref - pred * ref).sum((1, 2)) + 1e-6)
        loss = (1 - iou).mean()
        first = loss.item() if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert torch.isfinite(verts).all() and loss.item() < first
