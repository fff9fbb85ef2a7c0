"""Regenerates tests/golden/reference/*.npz: OUTPUTS OF THE REFERENCE'S OWN KERNELS.

Unlike tests/golden/*.npz (this repo's oracle, kept against accidental change), these vectors are computed by the
device half of the reference's generalized_renderer_cuda_kernel.cu, compiled for gfx950 as it stands by
oracle/build_ref.py (oracle/_ref/gendr_ref_render.co, no contraction) and launched by oracle/ref_gpu.py with the
reference's launch shapes -- on an MI355X, in float64 and float32.  They are data: inputs, options, outputs.  The CPU
suite holds the restatement to them (tests/test_oracle_render.py::test_restatement_reproduces_reference_vectors), so the
oracle is pinned to the reference without a GPU.

Needs a GPU and oracle/_ref (built where /root/reference exists; travels to the GPU box as a built artefact):

    gpurun -- 'python tests/golden/make_reference_golden.py --out gpurun_out/reference_golden'
    cp gpurun_out/reference_golden/*.npz tests/golden/reference/
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import parity  # noqa: E402
import scenes  # noqa: E402

IMAGE_SIZE = 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(HERE, 'reference'))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    from oracle import build_ref
    man = json.load(open(build_ref.manifest_path()))
    for name, opts in scenes.OPTION_MATRIX:
        if opts.get('texel_mode', 0) != 0:
            continue                                   # the clamped texel mode is this repo's, not the reference's
        kw = dict(B=2, nf=20, seed=3)
        if opts.get('texture_type') == 'vertex':
            kw['vertex_tex'] = True
        if 'T' in opts:
            kw['T'] = opts['T']
        fv, tex = scenes.soup(**kw)
        grad = np.random.RandomState(7).randn(2, 4, IMAGE_SIZE, IMAGE_SIZE).astype(np.float32)
        r64 = parity.run_reference(fv, tex, IMAGE_SIZE, opts, grad, np.float64)
        r32 = parity.run_reference(fv, tex, IMAGE_SIZE, opts, grad, np.float32)
        path = os.path.join(args.out, name + '.npz')
        np.savez_compressed(path, fv=fv, tex=tex, grad=grad, image_size=IMAGE_SIZE, options=json.dumps(opts),
                            produced_by=json.dumps(dict(code_object=man['objects']['render']['file'],
                                                        flags=man['objects']['render']['flags'],
                                                        reference_sha256=man['reference_sha256'], device='MI355X (gfx950)')),
                            **{'f64_' + k: r64[k] for k in ('rgba', 'aggrs_info', 'faces_info', 'grad_faces', 'grad_textures')},
                            **{'f32_' + k: r32[k] for k in ('rgba', 'aggrs_info', 'faces_info', 'grad_faces', 'grad_textures')})
        print(name, os.path.getsize(path))


if __name__ == '__main__':
    main()
