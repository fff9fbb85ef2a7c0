"""Kernel-level fwd+bwd step time of the headline config over the batch sizes a strong-scaling run hands to one GPU
(64 / N frames): python tools/batch_sweep.py [config] -> gpurun_out/<tag>_batch_sweep.json.  The step is replayed from a
HIP graph (no host launch gaps), which is how bench.py runs small per-GPU batches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench as B
import parity
from gendr_amd import build
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c2'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r03'
cfg = B.CONFIGS[cfgname]
isz = cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
o, extra = parity.split_options(opts)
out = dict(config=cfgname, kernel_sha=build.source_sha(), what='ms per forward+backward of the native calls, HIP-graph replay, median of 30', batches={})
for Bn in (1, 2, 4, 8, 16, 32, 64):
    fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
    faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
    grad = torch.randn(Bn, 4, isz, isz, device='cuda')
    p = parity.hip_params(isz, o, dict(extra, skip_unlisted_aux=1))
    flat, gf, gt = R.gradient_buffers(faces, t, fill=False)
    p.clear_ptr, p.clear_floats = flat.data_ptr(), flat.numel()
    rgba, aux, rec = R.native_forward(faces, t, p)
    R.native_backward(faces, t, rgba, aux, rec, grad, p, grad_faces=gf, grad_textures=gt)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            R.native_forward(faces, t, p, rgba=rgba, aggrs_info=aux)
            R.native_backward(faces, t, rgba, aux, rec, grad, p, grad_faces=gf, grad_textures=gt)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _, _, rec2 = R.native_forward(faces, t, p, rgba=rgba, aggrs_info=aux)
        R.native_backward(faces, t, rgba, aux, rec2, grad, p, grad_faces=gf, grad_textures=gt)
    for _ in range(3):
        g.replay()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[15]
    out['batches'][str(Bn)] = ms
    print('batch %2d: %.4f ms  -> %.0f frames/s' % (Bn, ms, Bn / ms * 1e3), flush=True)
t64 = out['batches']['64']
out['strong_projection'] = {str(n): t64 / out['batches'][str(64 // n)] for n in (1, 2, 4, 8)}
out['note'] = ('strong_projection[N] = t(batch 64) / t(batch 64 / N): the speed-up N GPUs would give on the fixed global batch of 64 if '
               'nothing but this op ran (no collective on the data path); a PROJECTION from one GPU, no multi-GPU run behind it')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', '%s_%s_batch_sweep.json' % (tag, cfgname)), 'w'), indent=1)
print(json.dumps(out['strong_projection']))
