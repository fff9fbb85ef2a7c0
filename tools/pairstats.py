"""How many (pixel, face) pairs reach phase B, and how many of them contribute (pass the reference's skip tests)?
Needs the GENDR_ABLATE=1 build as gpurun_ablate_1.so (its forward leaves the per-tile pair count in alpha)."""
import sys, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, 'gendr_amd', 'libgendr_hip.so')
shutil.copy(lib, '/tmp/full.so')
shutil.copy(os.path.join(ROOT, 'gpurun_ablate_1.so'), lib)
try:
    import numpy as np, torch, bench, oracle
    from gendr_amd.functional import renderer as R
    from gendr_amd.synthetic import benchmark_scene
    cfg = bench.CONFIGS['c2']
    B, isz = 8, cfg['image_size']
    o = dict(background_color=[0, 0, 0], dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None,
             dist_eps=1e4, aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax', aggr_rgb_eps=1e-3,
             aggr_rgb_gamma=1e-3, near=1, far=100, double_side=False, texture_type='surface')
    fv, tex = benchmark_scene(B, subdivisions=cfg['subdiv'], texture=cfg['texture'], seed=0)
    fvc, texc = fv.cuda().reshape(B, -1, 9).contiguous(), tex.cuda().contiguous()
    rgba, aux, ws = R.native_forward(fvc, texc, R.make_params(isz, **o))
    alpha = rgba[:, 3].cpu().numpy()
    per_tile = alpha.reshape(B, isz // 8, 8, isz // 8, 8)[:, :, 0, :, 0]
    total = per_tile.sum()
    opts = oracle.make_opts(image_size=isz, **{k: v for k, v in o.items() if k != 'background_color'})
    info = oracle.face_info(fv.numpy().reshape(B, -1, 9)) if hasattr(oracle, 'face_info') else None
    print('pairs reaching phase B: %d  (%.2f per pixel, %.1f per listed tile)' % (total, total / (B * isz * isz), total / max((per_tile > 0).sum(), 1)))
    if hasattr(oracle, 'count_pairs'):
        c = oracle.count_pairs(fv.numpy().reshape(B, -1, 9), opts)
        print('pairs that contribute (oracle): %d  -> %.1f %% of phase B lanes' % (c, 100.0 * c / total))
finally:
    shutil.copy('/tmp/full.so', lib)
