"""Kernel-level timing of the native forward / backward calls (HIP events), with ablation scenes.
    python tools/kbench.py [--config c2] [--iters 20]
"""
import argparse, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench as B
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene
import parity


def time_calls(faces, tex, p, grad, iters):
    rgba, aux, rec = R.native_forward(faces, tex, p)
    gf, gt = R.native_backward(faces, tex, rgba, aux, rec, grad, p)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for e in ev:
        e[0].record()
        rgba, aux, rec = R.native_forward(faces, tex, p, rgba=rgba, aggrs_info=aux)
        e[1].record()
        gf.zero_(); gt.zero_()
        R.native_backward(faces, tex, rgba, aux, rec, grad, p, grad_faces=gf, grad_textures=gt)
        e[2].record()
    torch.cuda.synchronize()
    f = sorted(e[0].elapsed_time(e[1]) for e in ev)[len(ev) // 2]
    b = sorted(e[1].elapsed_time(e[2]) for e in ev)[len(ev) // 2]
    return f, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--modes', default='normal,offscreen(scan only),nocull')
    ap.add_argument('--deterministic', action='store_true', help='gendr_params.deterministic = 1 (per-face backward, no atomics)')
    ap.add_argument('--loose', type=int, default=0, help='gendr_params.loose_faces: 0 automatic, 1 on, -1 off')
    ap.add_argument('--hints', type=int, default=0, help='gendr_params.pair_hints: 0 automatic, 1 on, -1 off')
    args = ap.parse_args()
    cfg = B.CONFIGS[args.config]
    Bn = args.batch or cfg['batch']
    isz = cfg['image_size']
    opts = dict(cfg['opts']); opts.setdefault('double_side', False)
    fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
    dev = 'cuda:0'
    grad = torch.randn(Bn, 4, isz, isz, device=dev)
    o, extra = parity.split_options(opts)
    res = {}
    for name, shift, cull in (('normal', 0.0, 1), ('offscreen(scan only)', 10.0, 1), ('nocull', 0.0, 0)):
        if name == 'nocull' and args.config not in ('c2', 'c3'):
            continue
        if name not in args.modes.split(','):
            continue
        p = parity.hip_params(isz, o, dict(extra, cull=cull, deterministic=1 if args.deterministic else 0, loose_faces=args.loose, pair_hints=args.hints))
        f = fv.clone(); f[..., 0] += shift
        faces = f.reshape(Bn, -1, 9).to(dev).contiguous()
        t = tex.to(dev).contiguous()
        it = 3 if name == 'nocull' else args.iters
        fm, bm = time_calls(faces, t, p, grad, it)
        res[name] = (fm, bm)
        print('%-22s fwd %8.3f ms  bwd %8.3f ms   -> %.0f frames/s' % (name, fm, bm, Bn / ((fm + bm) * 1e-3)), flush=True)
    return res


if __name__ == '__main__':
    main()
