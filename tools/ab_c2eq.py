"""Equality of two library builds on BASELINE config 2's option set (what a -DGENDR_DEV_MIN=2 library holds): rgba / aggrs_info bit for bit, gradients to
the order of the atomics, at three shapes incl. an odd image size and a small batch (split tiles).   python tools/ab_c2eq.py a.so b.so   (files at the repo root)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(256, 6), (100, 3), (64, 1), (136, 9)]
if len(sys.argv) > 2 and sys.argv[1] == '--worker':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, hashlib
    import parity
    from gendr_amd.synthetic import benchmark_scene
    out = {}
    opts = dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)
    for isz, B in SHAPES:
        fv, tex = benchmark_scene(B)
        grad = np.random.RandomState(1).randn(B, 4, isz, isz).astype(np.float32)
        h = parity.run_hip(fv.numpy().reshape(B, -1, 3, 3), tex.numpy(), isz, opts, grad)
        np.save('/tmp/abc2_%s_%d_%d.npy' % (sys.argv[2], isz, B), h['grad_faces'])
        out['%dx%d' % (isz, B)] = dict(rgba=hashlib.sha1(h['rgba'].tobytes()).hexdigest()[:12], aux=hashlib.sha1(h['aggrs_info'].tobytes()).hexdigest()[:12],
                                      gmax=float(np.abs(h['grad_faces']).max()))
    print('RESULT ' + json.dumps(out)); sys.exit(0)
import shutil, numpy as np
libs = sys.argv[1:]
shutil.copy(os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'), '/tmp/abc2_base.so')
res = {}
try:
    for i, l in enumerate(libs):
        shutil.copy(os.path.join(ROOT, l), os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', str(i)], capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if x.startswith('RESULT ')]
        if not line:
            print(l, 'FAILED', r.stdout[-1500:], r.stderr[-2500:]); continue
        res[l] = json.loads(line[0][7:])
finally:
    shutil.copy('/tmp/abc2_base.so', os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'))
ok = True
for isz, B in SHAPES:
    k = '%dx%d' % (isz, B)
    row = [res[l][k] for l in libs if l in res]
    same = all(r['rgba'] == row[0]['rgba'] and r['aux'] == row[0]['aux'] for r in row)
    g = [np.load('/tmp/abc2_%d_%d_%d.npy' % (i, isz, B)) for i in range(len(libs))]
    gd = max(float(np.abs(g[0].astype(np.float64) - x).max()) for x in g[1:]) if len(g) > 1 else 0.0
    print('%-8s forward %s   max |grad_faces difference| %.3g of max %.3g' % (k, 'IDENTICAL' if same else 'DIFFERENT', gd, row[0]['gmax']))
    ok &= same and gd <= 2e-5 * max(row[0]['gmax'], 1e-30)
print('EQUAL' if ok else 'NOT EQUAL'); sys.exit(0 if ok else 1)
