"""Alpha-only rendering with the fused silhouette IoU (SURVEY f-4): the alpha plane equals channel 3 of the full
render bit for bit, its gradient equals the full render's gradient for an upstream gradient that lives in the alpha
channel, the fused IoU sums equal the numpy oracle (pinned to the reference's iou_loss), and the fused backward equals
autograd through the alpha image."""
import numpy as np
import pytest
import torch

import scenes
from oracle import iou_ref

pytestmark = pytest.mark.gpu

CASES = [
    ('uniform_prob', dict()),                                            # own kernel
    ('hard_hard', dict(dist_func='hard', aggr_alpha_func='hard')),       # own kernel, forward only matters
    ('logistic_prob', dict(dist_func='logistic', dist_scale=2e-2)),      # light x light runtime dispatch
    ('gauss_sq_einstein', dict(dist_func='gaussian', dist_squared=True, dist_scale=3e-3, aggr_alpha_func='einstein')),
    ('gamma_yager', dict(dist_func='gamma', dist_shape=2.0, dist_scale=2e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0)),
    ('cubic_max', dict(dist_func='cubic_hermite', dist_scale=5e-2, aggr_alpha_func='max')),
    ('laplace_frank', dict(dist_func='laplace', dist_scale=2e-2, aggr_alpha_func='frank', aggr_alpha_t_conorm_p=3.0)),
    ('uniform_smalleps', dict(dist_eps=1.5, dist_scale=2e-2)),
]


def _scene(name):
    maker, isz = {'soup': (scenes.soup, 48), 'sphere': (scenes.sphere, 64), 'slivers': (scenes.slivers, 64)}[name]
    fv, tex = maker()
    return torch.from_numpy(fv).cuda(), torch.from_numpy(tex).cuda(), isz


@pytest.mark.parametrize("scene", ['soup', 'sphere', 'slivers'])
@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_alpha_plane_and_gradient_equal_the_full_render(native_lib, scene, name, opts):
    from gendr_amd.functional import render, render_silhouette
    fv, tex, isz = _scene(scene)
    a = fv.clone().requires_grad_(True)
    b = fv.clone().requires_grad_(True)
    full = render(a, tex, image_size=isz, **opts)
    sil = render_silhouette(b, image_size=isz, **opts)
    assert sil.shape == (fv.shape[0], isz, isz)
    assert torch.equal(sil, full[:, 3]), 'alpha-only kernels must reproduce channel 3 bit for bit'
    g = torch.randn(fv.shape[0], isz, isz, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    g4 = torch.zeros_like(full)
    g4[:, 3] = g
    full.backward(g4)
    sil.backward(g)
    ref = a.grad
    scale = max(1e-30, float(ref.abs().max()))
    # same per-pair values (the depth stage only decides whether a pair is dropped); float atomics order differs
    assert float((b.grad - ref).abs().max()) <= 2e-5 * scale, (scene, name)
    assert float(b.grad[..., 2].abs().max()) == 0.0            # no depth gradient without the colour term


@pytest.mark.parametrize("name,opts", CASES[:4], ids=[n for n, _ in CASES[:4]])
def test_faces_outside_near_far_are_dropped_like_the_reference(native_lib, name, opts):
    """kernel.cu:994: a pair whose clipped depth leaves [near, far] folds into alpha but gets no gradient -- the
    alpha-only backward must still apply that test for faces that are not provably inside the range."""
    from gendr_amd.functional import render, render_silhouette
    fv, tex, isz = _scene('soup')
    o = dict(opts, near=2.0, far=3.5)                          # soup depths are 1.5 .. 5: many faces straddle the range
    a = fv.clone().requires_grad_(True)
    b = fv.clone().requires_grad_(True)
    full = render(a, tex, image_size=isz, **o)
    sil = render_silhouette(b, image_size=isz, **o)
    assert torch.equal(sil, full[:, 3])
    g = torch.randn(fv.shape[0], isz, isz, device='cuda', generator=torch.Generator('cuda').manual_seed(4))
    g4 = torch.zeros_like(full)
    g4[:, 3] = g
    full.backward(g4)
    sil.backward(g)
    assert float((b.grad - a.grad).abs().max()) <= 2e-5 * max(1e-30, float(a.grad.abs().max()))


def test_fused_iou_matches_oracle_and_autograd(native_lib):
    from gendr_amd.functional import render_silhouette, silhouette_iou, silhouette_iou_loss
    fv, tex, isz = _scene('sphere')
    B = fv.shape[0]
    target = (torch.rand(B, isz, isz, device='cuda', generator=torch.Generator('cuda').manual_seed(9)) > 0.5).float()
    target[0, :, : isz // 2] = 0
    opts = dict(image_size=isz, dist_scale=3e-2)
    x = fv.clone().requires_grad_(True)
    inter, union, alpha = silhouette_iou(x, target, return_alpha=True, **opts)
    assert torch.equal(alpha, render_silhouette(fv, **opts))
    wi, wu = iou_ref.iou_sums(alpha.cpu().numpy(), target.cpu().numpy())
    assert np.allclose(inter.detach().cpu().numpy(), wi, rtol=2e-6, atol=1e-5)
    assert np.allclose(union.detach().cpu().numpy(), wu, rtol=2e-6, atol=1e-5)
    loss = silhouette_iou_loss(x, target, **opts)
    assert abs(float(loss) - iou_ref.iou_loss_opt_shape(alpha.cpu().numpy(), target.cpu().numpy())) < 1e-6
    loss.backward()
    # the same loss through the alpha image and torch autograd
    y = fv.clone().requires_grad_(True)
    a = render_silhouette(y, **opts)
    i2 = (a * target).sum((1, 2))
    u2 = (a + target - a * target).sum((1, 2)) + 1e-6
    (1. - i2 / u2).mean().backward()
    assert float((x.grad - y.grad).abs().max()) <= 2e-5 * float(y.grad.abs().max())


def test_gendr_module_silhouette_paths(native_lib):
    import gendr_amd
    fv, tex, isz = _scene('sphere')
    mesh = type('M', (), dict(face_vertices=fv, face_textures=tex))()
    for aa in (False, True):
        ren = gendr_amd.GenDR(image_size=32, anti_aliasing=aa, dist_scale=3e-2)
        assert torch.equal(ren.silhouette(mesh), ren(mesh)[:, 3])
        target = (ren(mesh)[:, 3] > 0.4).float()
        want = iou_ref.iou_loss_opt_shape(ren(mesh)[:, 3].cpu().numpy(), target.cpu().numpy())
        assert abs(float(ren.silhouette_iou_loss(mesh, target)) - want) < 2e-6


def test_invalid_use(native_lib):
    from gendr_amd.functional import silhouette_iou, render_silhouette
    fv, tex, isz = _scene('soup')
    with pytest.raises(ValueError):
        silhouette_iou(fv, torch.zeros(1, 3, 3, device='cuda'), image_size=isz)
    with pytest.raises(TypeError):
        render_silhouette(fv.cpu(), image_size=isz)
