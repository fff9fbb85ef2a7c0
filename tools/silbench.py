"""C2 geometry: full render -> [:, 3] -> iou_loss -> backward  vs  the alpha-only kernels with the fused IoU (SURVEY f-4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gendr_amd.functional import render, render_silhouette, silhouette_iou_loss
from gendr_amd.synthetic import benchmark_scene

B, isz = 64, 256
fv, tex = benchmark_scene(B)
fv, tex = fv.cuda().requires_grad_(True), tex.cuda()
with torch.no_grad():
    target = (render(fv, tex, image_size=isz, double_side=False)[:, 3] > 0.5).float()


def iou_loss(p, t):
    dims = (1, 2)
    return (1. - (p * t).sum(dims) / ((p + t - p * t).sum(dims) + 1e-6)).mean()


def full():
    fv.grad = None
    iou_loss(render(fv, tex, image_size=isz, double_side=False)[:, 3], target).backward()


def alpha_plane():
    fv.grad = None
    iou_loss(render_silhouette(fv, image_size=isz), target).backward()


def fused():
    fv.grad = None
    silhouette_iou_loss(fv, target, image_size=isz).backward()


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (('full render [:,3] + torch iou_loss', full), ('alpha-only kernels + torch iou_loss', alpha_plane),
                 ('alpha-only kernels, fused IoU', fused)):
    ms = timeit(fn)
    print('%-38s %.3f ms/step  %8.0f frames/s' % (name, ms, B / ms * 1e3), flush=True)
