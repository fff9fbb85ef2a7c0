"""Times arbitrary option sets at the C2 geometry (256^2, 1280 faces, batch 64) -- specialised vs generic kernels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity
from tools.kbench import time_calls
from gendr_amd.synthetic import benchmark_scene
Bn, isz = 64, 256
fv, tex = benchmark_scene(Bn)
dev = 'cuda:0'
grad = torch.randn(Bn, 4, isz, isz, device=dev)
faces = fv.reshape(Bn, -1, 9).to(dev).contiguous(); t = tex.to(dev).contiguous()
CASES = [
    ('uniform/prob/softmax (spec)', dict()),
    ('uniform/einstein/softmax (generic)', dict(aggr_alpha_func='einstein')),
    ('uniform/prob/hard (spec)', dict(aggr_rgb_func='hard')),
    ('uniform/einstein/hard (generic)', dict(aggr_rgb_func='hard', aggr_alpha_func='einstein')),
    ('logistic/prob/softmax tau=3e-3 (spec)', dict(dist_func='logistic', dist_scale=3e-3)),
    ('logistic/einstein/softmax tau=3e-3 (generic)', dict(dist_func='logistic', dist_scale=3e-3, aggr_alpha_func='einstein')),
    ('gaussian/prob/hard tau=1e-2 (generic)', dict(dist_func='gaussian', aggr_rgb_func='hard')),
    ('gudermannian/prob/softmax tau=3e-3 (full)', dict(dist_func='gudermannian', dist_scale=3e-3)),
    ('uniform/frank(2)/softmax (full)', dict(aggr_alpha_func='frank', aggr_alpha_t_conorm_p=2.0)),
    ('gudermannian/einstein/softmax tau=3e-3 (full)', dict(dist_func='gudermannian', dist_scale=3e-3, aggr_alpha_func='einstein')),
    ('gamma(2)/yager(2)/softmax surface (full)', dict(dist_func='gamma', dist_shape=2.0, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0)),
]
for name, o in CASES:
    oo, extra = parity.split_options(dict(o, double_side=False))
    p = parity.hip_params(isz, oo, extra)
    f, b = time_calls(faces, t, p, grad, 10)
    print('%-48s fwd %7.3f ms bwd %7.3f ms -> %8.0f frames/s' % (name, f, b, Bn / ((f + b) * 1e-3)), flush=True)
