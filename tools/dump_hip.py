"""Dumps the HIP path's outputs (both build variants) for the parity matrix so that the acceptance rule can be
studied offline against the CPU oracle:   python tools/dump_hip.py  ->  gpurun_out/hipdump/<case>.npz
Inputs are regenerated from tests/scenes.py (deterministic), so only the outputs travel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

import parity
import scenes
from gpu_report import FULL


def cases(full=('C2', 'C3', 'C4')):
    for scene_name, maker, isz in (("soup", scenes.soup, 48), ("sphere", scenes.sphere, 64), ("slivers", scenes.slivers, 64)):
        for name, opts in scenes.OPTION_MATRIX:
            kw = {}
            if opts.get('texture_type') == 'vertex':
                kw['vertex_tex'] = True
            if 'T' in opts:
                kw['T'] = opts['T']
            fv, tex = maker(**kw)
            yield scene_name + '__' + name, fv, tex, isz, opts
    from gendr_amd.synthetic import benchmark_scene
    for name in full:
        isz, texture, opts = FULL[name]
        fv, tex = benchmark_scene(3, texture=texture)
        yield 'full__' + name, fv.numpy()[2:3], tex.numpy()[2:3], isz, opts


def main():
    out = os.path.join(ROOT, 'gpurun_out', 'hipdump')
    os.makedirs(out, exist_ok=True)
    for name, fv, tex, isz, opts in cases():
        grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        d = {}
        for variant in ('default', 'exact'):
            h = parity.run_hip(fv, tex, isz, opts, grad, variant=variant)
            for k, v in h.items():
                d[variant + '__' + k] = v
        # the images do not depend on the variant (only backward kernels differ): keep one copy
        assert np.array_equal(d['default__rgba'], d['exact__rgba'], equal_nan=True)
        del d['exact__rgba'], d['exact__aggrs_info']
        np.savez_compressed(os.path.join(out, name + '.npz'), **d)
        print(name, flush=True)


if __name__ == '__main__':
    main()
