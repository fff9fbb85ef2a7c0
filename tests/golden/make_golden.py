"""Regenerates tests/golden/*.npz.

IMPORTANT: these vectors come from THIS repo's CPU oracle (oracle/), not from the reference (vectors computed by the
reference's own kernels are in tests/golden/reference/, see make_reference_golden.py).  The files here pin the oracle
itself against accidental change and give the GPU tests fixed
known-answer inputs/outputs that do not depend on the oracle being rebuilt on the GPU box.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import parity  # noqa: E402  (imports torch lazily only for the HIP side)
import scenes  # noqa: E402

IMAGE_SIZE = 24
CASES = ['uniform_prob_softmax', 'uniform_prob_hardrgb', 'hard_hard_hard', 'gauss_sq_einstein', 'logistic_prob',
         'gamma_yager_vertex', 'cubic_max', 'wigner_hamacher', 'laplace_frank', 'guder_aczel', 'cauchy_dombi',
         'reciprocal_ss', 'gumbelmax_prob', 'exp_prob', 'levy_prob', 'uniform_hardalpha', 'uniform_smalleps',
         'uniform_singleside', 'uniform_bg', 'uniform_T4', 'uniform_T9_clamp']


def main():
    matrix = dict(scenes.OPTION_MATRIX)
    for name in CASES:
        opts = matrix[name]
        kw = dict(B=2, nf=20, seed=3)
        if opts.get('texture_type') == 'vertex':
            kw['vertex_tex'] = True
        if 'T' in opts:
            kw['T'] = opts['T']
        fv, tex = scenes.soup(**kw)
        grad = np.random.RandomState(7).randn(2, 4, IMAGE_SIZE, IMAGE_SIZE).astype(np.float32)
        r = parity.run_oracle(fv, tex, IMAGE_SIZE, opts, grad)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), fv=fv, tex=tex, grad=grad,
                            image_size=IMAGE_SIZE, options=json.dumps(opts),
                            rgba=r['rgba'], aggrs_info=r['aggrs_info'], faces_info=r['faces_info'],
                            grad_faces=r['grad_faces'], grad_textures=r['grad_textures'],
                            abs_faces=r['abs_faces'], abs_textures=r['abs_textures'])
        print(name, os.path.getsize(os.path.join(HERE, name + '.npz')))


if __name__ == '__main__':
    main()
