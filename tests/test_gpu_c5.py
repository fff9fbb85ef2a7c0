"""BASELINE config 5 as stated: 1024x1024 with anti_aliasing=True (2048^2 internal), dist=gamma(shape 2),
t-conorm=yager(p 2), texture_type=vertex -- one frame element-wise against the oracle (through the native calls and
through the GenDR module with its 2x2 average pooling), and a batch whose tensors pass 2^31 bytes through
size-independent properties (kernel.cu:1026-1063 for the gradient assembly under test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import criteria
import parity

pytestmark = pytest.mark.gpu

C5 = dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
          aggr_rgb_func='softmax', texture_type='vertex', double_side=False)
IS = 2048
FRAME = 2          # the batch item checked element-wise


def _scene(B):
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(B, texture='vertex')
    return fv, tex


@pytest.fixture(scope='module')
def oracle_frame(oracle_mod):
    """fp32 and fp64 oracle of ONE 2048^2 frame, forward and backward; the upstream gradient is what the 2x2 average
    pooling hands down from a random 1024^2 gradient (so the same run also serves the GenDR(anti_aliasing) test)."""
    fv, tex = _scene(4)
    fv1, tex1 = fv[FRAME:FRAME + 1].numpy(), tex[FRAME:FRAME + 1].numpy()
    g_small = torch.from_numpy(np.random.RandomState(5).randn(4, 4, IS // 2, IS // 2).astype(np.float32))
    # d avg_pool2d: every pixel of a 2x2 block receives a quarter of the pooled pixel's gradient
    g_full = (g_small[FRAME:FRAME + 1].repeat_interleave(2, 2).repeat_interleave(2, 3) * 0.25).numpy()
    r32 = parity.run_oracle(fv1, tex1, IS, C5, g_full)
    # the sensitivity runs of the element-wise rule (tests/criteria.py); four libm-jitter modes: a 2048^2 oracle run takes
    # half a minute
    refs = criteria.references(fv1, tex1, IS, C5, g_full, oracle_f32=r32, n_jitter=4)
    return dict(fv=fv, tex=tex, fv1=fv1, tex1=tex1, g_small=g_small, g_full=g_full, r32=r32, refs=refs)


def test_c5_full_frame_native_against_oracle(native_lib, oracle_frame):
    o = oracle_frame
    h = parity.run_hip(o['fv1'], o['tex1'], IS, C5, o['g_full'])
    bad = criteria.failures(criteria.elementwise(h, o['refs']))
    assert not bad, bad


def test_c5_through_gendr_with_anti_aliasing(native_lib, oracle_frame):
    """gendr.GenDR(image_size=1024, anti_aliasing=True, ...)(mesh) on B = 4: 2048^2 internally, pooled to 1024^2;
    frame FRAME against the pooled oracle image, its input gradients against the oracle's backward."""
    import gendr_amd
    o = oracle_frame
    fv = o['fv'].cuda().requires_grad_(True)
    tex = o['tex'].cuda().requires_grad_(True)
    ren = gendr_amd.GenDR(image_size=IS // 2, anti_aliasing=True, dist_func='gamma', dist_shape=2.0, dist_scale=1e-2,
                          aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0, texture_type='vertex')
    img = ren.forward_tensors(fv, tex)
    assert img.shape == (4, 4, IS // 2, IS // 2)
    img.backward(o['g_small'].cuda())
    # the pooled image against the pooled oracle images: the element-wise rule on 2x2 averages of every reference run
    def pooled(run):
        return dict(rgba=F.avg_pool2d(torch.from_numpy(np.asarray(run['rgba'], np.float64)), 2, 2).numpy())
    refs = o['refs']
    prefs = dict(o32=pooled(refs['o32']), o64=pooled(refs['o64']), lo=pooled(refs['lo']), hi=pooled(refs['hi']),
                 jit=[pooled(j) for j in refs['jit']])
    bad = criteria.failures(criteria.elementwise(dict(rgba=img[FRAME:FRAME + 1].detach().cpu().numpy()), prefs))
    assert not bad, bad
    gf = fv.grad[FRAME].reshape(1, -1, 9).cpu().numpy()
    gt = tex.grad[FRAME:FRAME + 1].cpu().numpy()
    bad = criteria.failures(criteria.elementwise(dict(grad_faces=gf, grad_textures=gt), refs))
    assert not bad, bad


def test_c5_batch_past_2GiB_properties(native_lib):
    """B = 40 at 2048^2: rgba and its gradient are 2.7 GB each, so the last items sit at byte offsets above 2^31
    (and the last pixel's element index above 2^29).  Items are independent, so every item equals itself rendered
    alone -- compared on the device; the culled traversal equals the all-pairs traversal; everything is finite."""
    from gendr_amd.functional import renderer as R
    B = 40
    fv, tex = _scene(B)
    o, extra = parity.split_options(C5)
    p = parity.hip_params(IS, o, extra)
    faces = fv.reshape(B, -1, 9).cuda().contiguous()
    textures = tex.cuda().contiguous()
    grad = torch.randn(B, 4, IS, IS, device='cuda', generator=torch.Generator('cuda').manual_seed(7))
    assert grad.numel() * 4 > 2 ** 31
    rgba, aux, rec = R.native_forward(faces, textures, p)
    gf, gt = R.native_backward(faces, textures, rgba, aux, rec, grad, p)
    assert bool(torch.isfinite(rgba).all()) and bool(torch.isfinite(aux).all())
    assert bool(torch.isfinite(gf).all()) and bool(torch.isfinite(gt).all())
    assert float(rgba[:, 3].max()) <= 1.0 and float(rgba[:, 3].min()) >= 0.0
    p0 = parity.hip_params(IS, o, dict(extra, cull=0))
    for i in (0, 17, 39):                                   # 39: beyond 2^31 bytes in rgba, aux-plane pair and grad
        f1, t1, g1 = faces[i:i + 1].contiguous(), textures[i:i + 1].contiguous(), grad[i:i + 1].contiguous()
        r1, a1, rec1 = R.native_forward(f1, t1, p)
        assert torch.equal(r1[0], rgba[i]) and torch.equal(a1[0], aux[i]), i
        gf1, gt1 = R.native_backward(f1, t1, r1, a1, rec1, g1, p)
        for a, b in ((gf1[0], gf[i]), (gt1[0], gt[i])):     # float atomics: summation order differs between launches
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), i
        if i != 17:
            r0, a0, rec0 = R.native_forward(f1, t1, p0)     # all-pairs traversal of the same item
            assert torch.equal(r0, r1) and torch.equal(a0, a1), i
            gf0, gt0 = R.native_backward(f1, t1, r0, a0, rec0, g1, p0)
            for a, b in ((gf0, gf1), (gt0, gt1)):
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), i
