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


def test_multiview_reconstruction_step_through_the_alias_package(native_lib):
    """The call pattern of experiments/train_reconstruction.py:211-241,179-200: a batch of predicted meshes, duplicated
    ([Ma, Mb, Ma, Mb]) and seen from [Va, Va, Vb, Vb] (4 x batch views), Lighting -> LookAt(viewing_angle=15) ->
    GenDR(image_size=64, dist_eps=300, aggr_rgb_func='hard'), multiview IoU on the alpha channel, Laplacian + Flatten
    regularisers, one optimiser step; then the evaluation path: face_vertices -> voxelization(32) -> IoU with a grid."""
    import gendr
    from gendr_amd.synthetic import icosphere
    from oracle import iou_ref
    v, f = icosphere(2)
    Bm = 4                                                       # meshes per half batch; 4 * 2 * Bm = 32 rendered views
    base = torch.from_numpy(v).cuda()
    faces1 = torch.from_numpy(f).int().cuda()
    offs = torch.nn.Parameter(torch.zeros(2 * Bm, v.shape[0], 3, device='cuda'))
    scale = torch.linspace(0.5, 0.8, 2 * Bm, device='cuda')[:, None, None]
    transform = gendr.LookAt(viewing_angle=15)
    lighting = gendr.Lighting()
    renderer = gendr.GenDR(image_size=64, dist_func='uniform', dist_scale=10 ** -1.5, dist_squared=False, dist_shape=0,
                           dist_shift=0, dist_eps=300., aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=0,
                           aggr_rgb_func='hard')
    lap = gendr.LaplacianLoss(base, faces1.long(), average=True).cuda()      # the script moves the whole Model to the GPU
    flat = gendr.FlattenLoss(faces1.long(), average=True).cuda()
    va = gendr.functional.get_points_from_angles(torch.full((Bm,), 2.732), torch.full((Bm,), 30.0), torch.arange(Bm) * -15.0).cuda()
    vb = gendr.functional.get_points_from_angles(torch.full((Bm,), 2.732), torch.full((Bm,), 30.0), torch.arange(Bm) * -15.0 - 90.0).cuda()
    with torch.no_grad():                                        # targets: the unit-scale sphere from the same views
        tm = gendr.Mesh(base[None].repeat(4 * Bm, 1, 1) * 0.9, faces1[None].repeat(4 * Bm, 1, 1))
        transform.set_eyes(torch.cat((va, va, vb, vb), 0))
        targets = renderer(transform(lighting(tm))).chunk(4, dim=0)
    opt = torch.optim.Adam([offs], lr=1e-2)
    losses = []
    for _ in range(6):
        vertices = base[None] * scale + offs                     # [2 Bm, nv, 3] = [Ma, Mb]
        faces = faces1[None].repeat(2 * Bm, 1, 1)
        laplacian_loss, flatten_loss = lap(vertices), flat(vertices)
        transform.set_eyes(torch.cat((va, va, vb, vb), 0))
        mesh = gendr.Mesh(torch.cat((vertices, vertices), 0), torch.cat((faces, faces), 0))
        sil = renderer(transform(lighting(mesh))).chunk(4, dim=0)
        assert sil[0].shape == (Bm, 4, 64, 64)

        def iou_loss(p, t):
            dims = (1, 2)
            return 1 - ((p * t).sum(dims) / ((p + t - p * t).sum(dims) + 1e-6)).sum() / p.shape[0]
        loss = (iou_loss(sil[0][:, 3], targets[0][:, 3]) + iou_loss(sil[1][:, 3], targets[0][:, 3]) +
                iou_loss(sil[2][:, 3], targets[2][:, 3]) + iou_loss(sil[3][:, 3], targets[2][:, 3])) / 4
        want = iou_ref.multiview_iou_loss([s.detach().cpu().numpy() for s in sil], targets[0].cpu().numpy(), targets[2].cpu().numpy())
        assert abs(float(loss.detach()) - want) < 1e-5
        total = loss + 5e-3 * laplacian_loss + 5e-4 * flatten_loss
        opt.zero_grad()
        total.backward()
        opt.step()
        losses.append(float(loss))
    assert torch.isfinite(offs).all() and losses[-1] < losses[0]
    # evaluation path (train_reconstruction.py:233-241)
    with torch.no_grad():
        vertices = base[None] * scale + offs
        faces_ = gendr.functional.face_vertices(vertices, faces1[None].repeat(2 * Bm, 1, 1)).data
        faces_norm = faces_ * 1. * (32. - 1) / 32. + 0.5
        vox = gendr.functional.voxelization(faces_norm, 32, False).cpu().numpy()
        vox = vox.transpose(0, 2, 1, 3)[:, :, :, ::-1]
        assert vox.shape == (2 * Bm, 32, 32, 32) and 0 < vox.sum() < vox.size
        iou = (vox * vox).sum((1, 2, 3)) / (0 < (vox + vox)).sum((1, 2, 3))
        assert np.allclose(iou, 1.0)


def test_gradients_cleared_by_the_forward_call_and_a_second_backward(native_lib):
    """The Function hands gendr_forward the gradient buffers to clear (gendr_params.clear_ptr) and uses them once: a
    second backward through the same graph must give the same gradients again, not their double."""
    import scenes
    from gendr_amd.functional import render
    fv, tex = scenes.soup()
    a = torch.from_numpy(fv).cuda().requires_grad_(True)
    t = torch.from_numpy(tex).cuda().requires_grad_(True)
    g = torch.randn(fv.shape[0], 4, 48, 48, device='cuda')
    out = render(a, t, image_size=48)
    ga1, gt1 = torch.autograd.grad(out, (a, t), g, retain_graph=True)
    ga2, gt2 = torch.autograd.grad(out, (a, t), g)
    # float atomics: the order of the per-tile sums differs from launch to launch
    assert torch.allclose(ga1, ga2, rtol=1e-4, atol=1e-5 * float(ga1.abs().max()))
    assert torch.allclose(gt1, gt2, rtol=1e-4, atol=1e-5 * float(gt1.abs().max()))
    # inference: nothing allocated for gradients, same image
    with torch.no_grad():
        out2 = render(a, t, image_size=48)
    assert torch.equal(out, out2)


def test_clear_ptr_of_the_c_abi(native_lib):
    import ctypes
    import parity
    import scenes
    from gendr_amd.functional import renderer as R
    fv, tex = scenes.soup()
    o, extra = parity.split_options({})
    p = parity.hip_params(48, o, extra)
    faces = torch.from_numpy(fv).reshape(fv.shape[0], -1, 9).cuda()
    textures = torch.from_numpy(tex).cuda()
    buf = torch.full((4096 + 8,), float('nan'), device='cuda')
    p.clear_ptr = buf.data_ptr() + 16
    p.clear_floats = 4096
    R.native_forward(faces, textures, p)
    torch.cuda.synchronize()
    b = buf.cpu()
    assert torch.isnan(b[:4]).all() and torch.isnan(b[4100:]).all() and (b[4:4100] == 0).all()
    p.clear_ptr = buf.data_ptr() + 4                       # not 16-byte aligned
    with pytest.raises(Exception):
        R.native_forward(faces, textures, p)


def test_cpp_autograd_node_equals_the_python_function(native_lib):
    """`render()` takes float32 CUDA inputs through the C++ autograd node (csrc/gendr_torch.cpp: the same host logic as
    GenDRFunction, two C-ABI calls per step, no Python frames); switched off it goes through GenDRFunction.  Same images bit
    for bit, same gradients up to the atomics' order, the same exception types, graph-capturable, retain_graph works."""
    from gendr_amd import _native
    from gendr_amd.functional import renderer as R
    assert _native.torch_ext() is not None, 'gendr_amd/_gendr_torch.so is not built (gendr_amd.build.build_torch_ext())'
    fv, tex = _inputs(B=3, nf=40)
    g = None
    out = {}
    before = _native.torch_ext().stats()
    for mode in (True, False):
        R._CPP_AUTOGRAD = mode
        try:
            a, t = fv.clone().requires_grad_(True), tex.clone().requires_grad_(True)
            img = R.render(a, t, image_size=48, dist_func='logistic', dist_scale=2e-2, background_color=[0.1, 0.2, 0.3])
            assert ('GenDRFunction' in img.grad_fn.name()) == (not mode), img.grad_fn.name()
            g = torch.randn_like(img) if g is None else g
            img.backward(g, retain_graph=True)
            first = a.grad.clone()
            a.grad = None; t.grad = None
            img.backward(g)                                   # a second backward through the same graph: its own, filled buffers
            assert torch.allclose(a.grad, first, rtol=1e-4, atol=1e-6)
            out[mode] = (img.detach(), a.grad.clone(), t.grad.clone())
        finally:
            R._CPP_AUTOGRAD = True
    # the differentiated forward ran WITH pair hints and handed its cleared gradient buffer to the first backward; the second backward
    # through the same graph filled its own (round 5: a first version of the node read the grad mode inside forward -- always off there
    # -- and ran every step without hints and with a fill launch: backward kernel 100 instead of 88 us, found in the kernel trace)
    after = _native.torch_ext().stats()
    d = {k: after[k] - before[k] for k in after}
    assert d == dict(forward_with_grad=1, forward_with_hints_off=0, backward_prefilled=1, backward_filled_here=1), d
    assert torch.equal(out[True][0], out[False][0])
    assert torch.allclose(out[True][1], out[False][1], rtol=1e-4, atol=1e-6) and torch.allclose(out[True][2], out[False][2], rtol=1e-4, atol=1e-6)
    with pytest.raises(ValueError):
        R.render(fv, tex, image_size=32, dist_func=23)
    with pytest.raises(ValueError):
        R.render(fv, tex[:, :5], image_size=32)
    with pytest.raises(TypeError):
        R.render(fv.cpu(), tex, image_size=32)
    # no differentiable input, inference mode, [B, nf, 9] faces
    with torch.no_grad():
        b = R.render(fv.reshape(3, 40, 9), tex, image_size=48, dist_func='logistic', dist_scale=2e-2, background_color=[0.1, 0.2, 0.3])
    assert torch.equal(b, out[True][0])


def test_tensor_valued_options_are_read_by_value_on_every_call(native_lib):
    """a 0-d tensor / nn.Parameter passed as dist_scale and updated IN PLACE between two render() calls (sigma.mul_(),
    optimizer.step()) must render with the new value -- the option cache of the C++ node's path is keyed on values, not objects."""
    from gendr_amd.functional import renderer as R
    fv, tex = _inputs(B=2, nf=24)
    sigma = torch.nn.Parameter(torch.tensor(1e-2))
    a = R.render(fv, tex, image_size=40, dist_scale=sigma)
    with torch.no_grad():
        sigma.mul_(10)
    b = R.render(fv, tex, image_size=40, dist_scale=sigma)
    assert torch.equal(b, R.render(fv, tex, image_size=40, dist_scale=float(sigma)))
    assert not torch.equal(a, b)
    g = np.float64(3e-3)
    assert torch.equal(R.render(fv, tex, image_size=40, aggr_rgb_gamma=g), R.render(fv, tex, image_size=40, aggr_rgb_gamma=float(g)))


def test_double_backward_raises_on_both_paths(native_lib):
    """GenDRFunction is @once_differentiable; the C++ node must refuse create_graph=True as loudly."""
    from gendr_amd.functional import renderer as R
    fv, tex = _inputs(B=1, nf=16)
    for mode in (True, False):
        R._CPP_AUTOGRAD = mode
        try:
            a = fv.clone().requires_grad_(True)
            img = R.render(a, tex, image_size=32)
            with pytest.raises(RuntimeError):
                (ga,) = torch.autograd.grad(img.sum(), a, create_graph=True)
                ga.sum().backward()
        finally:
            R._CPP_AUTOGRAD = True
