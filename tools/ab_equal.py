"""A/B of two library builds for EQUALITY of results (and time) at a few shapes: the forward outputs, the coverage entries' effect
(rgba, aggrs_info) bit for bit, the gradients to the order of the atomics.   python tools/ab_equal.py a.so b.so   (files at the repo root)
Each library runs in its own subprocess (the library is bound at import)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ('optshape 64x24 logistic hard', 64, 24, dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='hard', dist_eps=100.)),
    ('70x5 logistic softmax sigma 3e-2', 70, 5, dict(dist_func='logistic', dist_scale=3e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax')),
    ('33x3 gaussian max (runtime-dispatch team kernel)', 33, 3, dict(dist_func='gaussian', dist_scale=2e-2, aggr_alpha_func='max', aggr_rgb_func='hard')),
    ('128x8 logistic hard sigma 1e-4', 128, 8, dict(dist_func='logistic', dist_scale=1e-4, aggr_alpha_func='probabilistic', aggr_rgb_func='hard')),
]

if len(sys.argv) > 2 and sys.argv[1] == '--worker':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, torch, hashlib
    import parity, scenes
    from gendr_amd.synthetic import benchmark_scene
    from tools.kbench import time_calls
    out = {}
    for name, isz, B, opts in CASES:
        fv, tex = benchmark_scene(B)
        fvn, texn = fv.numpy().reshape(B, -1, 3, 3), tex.numpy()
        grad = np.random.RandomState(1).randn(B, 4, isz, isz).astype(np.float32)
        h = parity.run_hip(fvn, texn, isz, opts, grad)
        o, extra = parity.split_options(opts)
        p = parity.hip_params(isz, o, extra)
        faces = fv.reshape(B, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
        f, b = time_calls(faces, t, p, torch.from_numpy(grad).cuda(), 20)
        out[name] = dict(rgba=hashlib.sha1(h['rgba'].tobytes()).hexdigest(), aux=hashlib.sha1(h['aggrs_info'].tobytes()).hexdigest(),
                         gf=[float(np.abs(h['grad_faces']).max()), float(h['grad_faces'].astype(np.float64).sum())], fwd_ms=f, bwd_ms=b)
        np.save('/tmp/abeq_%s_%d.npy' % (sys.argv[2], len(out)), h['grad_faces'])
    print('RESULT ' + json.dumps(out))
    sys.exit(0)

import shutil
libs = sys.argv[1:]
shutil.copy(os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'), '/tmp/abeq_base.so')
res = {}
try:
    for i, l in enumerate(libs):
        shutil.copy(os.path.join(ROOT, l), os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', str(i)], capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if x.startswith('RESULT ')]
        if not line:
            print(l, 'FAILED'); print(r.stdout[-2000:]); print(r.stderr[-3000:]); continue
        res[l] = json.loads(line[0][7:])
finally:
    shutil.copy('/tmp/abeq_base.so', os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so'))
import numpy as np
for name, _, _, _ in CASES:
    print('==', name)
    for l in libs:
        if l in res:
            r = res[l][name]
            print('   %-22s fwd %.3f ms bwd %.3f ms  rgba %s aux %s  grad max %.6g sum %.9g' % (l, r['fwd_ms'], r['bwd_ms'], r['rgba'][:10], r['aux'][:10], r['gf'][0], r['gf'][1]))
for k in range(1, len(CASES) + 1):
    g = [np.load('/tmp/abeq_%d_%d.npy' % (i, k)) for i in range(len(libs)) if os.path.exists('/tmp/abeq_%d_%d.npy' % (i, k))]
    if len(g) >= 2:
        print('case %d: max |grad_faces difference| %.3g of max %.3g' % (k, float(np.abs(g[0].astype(np.float64) - g[1]).max()), float(np.abs(g[0]).max())))
