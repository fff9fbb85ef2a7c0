"""The render op inside a HIP graph (torch.cuda.CUDAGraph): the native calls take the capturing stream, allocate
nothing themselves and never synchronise, so forward + backward can be captured once and replayed."""
import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu


def test_forward_backward_capture_and_replay(native_lib):
    from gendr_amd.functional import render
    fv0, tex0 = scenes.soup(B=2, nf=40, seed=7)
    fv = torch.from_numpy(fv0).cuda().requires_grad_(True)
    tex = torch.from_numpy(tex0).cuda().requires_grad_(True)
    g = torch.from_numpy(np.random.RandomState(0).randn(2, 4, 64, 64).astype(np.float32)).cuda()
    opts = dict(image_size=64, dist_func='logistic', dist_scale=2e-2)

    def step():
        img = render(fv, tex, **opts)
        gf, gt = torch.autograd.grad(img, (fv, tex), g)
        return img, gf, gt

    ref = [t.detach().clone() for t in step()]    # detached: an autograd graph kept alive from before the capture
                                                   # drags its AccumulateGrad nodes' stream into it (PyTorch rule)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                       # warm-up on the side stream, as torch's capture rules ask
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    # new inputs through the captured buffers
    fv1, tex1 = scenes.soup(B=2, nf=40, seed=8)
    with torch.no_grad():
        fv.copy_(torch.from_numpy(fv1))
        tex.copy_(torch.from_numpy(tex1))
    graph.replay()
    torch.cuda.synchronize()
    got = [t.detach().clone() for t in out]
    want = [t.detach() for t in step()]
    assert torch.equal(got[0], want[0])
    for a, b in zip(got[1:], want[1:]):          # float atomics: order differs between launches
        assert (a - b).abs().max() <= 1e-5 * max(1.0, b.abs().max().item())
    # and the first capture-time values were not garbage either
    with torch.no_grad():
        fv.copy_(torch.from_numpy(fv0))
        tex.copy_(torch.from_numpy(tex0))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0])
