"""Test scenes (numpy).  Kept small enough that the CPU oracle finishes in seconds."""
import numpy as np


def soup(B=2, nf=48, seed=0, T=1, vertex_tex=False):
    """Random triangle soup in NDC with the degenerate / edge cases the path has to survive:
    zero-area face, duplicated vertex, obtuse and sliver triangles, back-facing faces,
    faces outside [near, far], a face covering the whole image, a face outside the viewport."""
    rs = np.random.RandomState(seed)
    fv = np.zeros((B, nf, 3, 3), np.float32)
    for b in range(B):
        c = rs.uniform(-0.9, 0.9, (nf, 1, 2))
        size = rs.uniform(0.05, 0.45, (nf, 1, 1))
        fv[b, :, :, :2] = c + size * rs.uniform(-1, 1, (nf, 3, 2))
        fv[b, :, :, 2] = rs.uniform(1.5, 5.0, (nf, 3))
        k = 0
        fv[b, k, 2, :2] = 0.5 * (fv[b, k, 0, :2] + fv[b, k, 1, :2]); k += 1          # zero area (collinear)
        fv[b, k, 1] = fv[b, k, 0]; k += 1                                               # duplicated vertex
        fv[b, k, :, :2] = [[-0.6, -0.1], [0.6, -0.1], [0.0, -0.05]]; k += 1             # very obtuse
        fv[b, k, :, :2] = [[-0.5, 0.3], [0.5, 0.3001], [0.0, 0.30005]]; k += 1          # sliver
        fv[b, k, :, 2] = 0.5; k += 1                                                    # nearer than near=1
        fv[b, k, :, 2] = 250.0; k += 1                                                  # farther than far=100
        fv[b, k, :, :2] = [[-3, -3], [3, -3], [0, 4]]; k += 1                           # covers the image
        fv[b, k, :, :2] += 5.0; k += 1                                                  # outside the viewport
        fv[b, k] = fv[b, k][::-1].copy(); k += 1                                        # flipped winding
    if vertex_tex:
        tex = rs.uniform(0, 1, (B, nf, 3, 3)).astype(np.float32)
    else:
        tex = rs.uniform(0, 1, (B, nf, T, 3)).astype(np.float32)
    return fv, tex


def slivers(B=2, nf=72, seed=0, T=1, vertex_tex=False):
    """Grazing / nearly degenerate triangles of every thinness (what a closed mesh shows at its silhouette):
    the float inverse of such faces is wrong by percents, which is exactly where a culling margin can fail."""
    rs = np.random.RandomState(seed + 11)
    fv = np.zeros((B, nf, 3, 3), np.float32)
    for b in range(B):
        a = rs.uniform(-0.8, 0.8, (nf, 2))
        ang = rs.uniform(0, 2 * np.pi, nf)
        length = rs.uniform(0.02, 0.6, nf)
        d = np.stack([np.cos(ang), np.sin(ang)], -1)
        n = np.stack([-d[:, 1], d[:, 0]], -1)
        thin = 10.0 ** rs.uniform(-7.5, -2.0, nf)             # height of the third vertex over the long edge
        bpt = a + d * length[:, None]
        cpt = a + d * (length * rs.uniform(0.05, 0.95, nf))[:, None] + n * (thin * rs.choice([-1, 1], nf))[:, None]
        fv[b, :, 0, :2], fv[b, :, 1, :2], fv[b, :, 2, :2] = a, bpt, cpt
        fv[b, :, :, 2] = rs.uniform(1.5, 5.0, (nf, 3))
        fv[b, ::7, :, :2] += 1.5                                # some partly / fully outside the viewport
    tex = rs.uniform(0, 1, (B, nf, 3 if vertex_tex else T, 3)).astype(np.float32)
    return fv, tex


def sphere(B=2, subdivisions=1, vertex_tex=False, T=1, seed=0):
    """Closed icosphere seen from the benchmark's camera ring (grazing faces at the silhouette)."""
    import torch
    from gendr_amd.synthetic import benchmark_scene
    fv, _ = benchmark_scene(B, subdivisions=subdivisions, seed=seed)
    fv = fv.numpy()
    rs = np.random.RandomState(seed + 1)
    nf = fv.shape[1]
    tex = rs.uniform(0, 1, (B, nf, 3 if vertex_tex else T, 3)).astype(np.float32)
    return fv, tex


# (name, options) matrix covering every branch family of SURVEY.md 2.2
OPTION_MATRIX = [
    ("uniform_prob_softmax", dict()),
    ("uniform_prob_hardrgb", dict(aggr_rgb_func='hard')),
    ("hard_hard_hard", dict(dist_func='hard', aggr_alpha_func='hard', aggr_rgb_func='hard')),
    ("hard_prob_softmax", dict(dist_func=0)),
    ("gauss_sq_einstein", dict(dist_func='gaussian', dist_squared=True, dist_scale=3e-3, aggr_alpha_func='einstein')),
    ("logistic_prob", dict(dist_func='logistic', dist_scale=2e-2)),
    ("gamma_yager_vertex", dict(dist_func='gamma', dist_shape=2.0, dist_scale=2e-2, aggr_alpha_func='yager',
                                aggr_alpha_t_conorm_p=2.0, texture_type='vertex')),
    ("cubic_max", dict(dist_func='cubic_hermite', dist_scale=5e-2, aggr_alpha_func='max')),
    ("wigner_hamacher", dict(dist_func='wigner_semicircle', dist_scale=5e-2, aggr_alpha_func='hamacher', aggr_alpha_t_conorm_p=0.5)),
    ("laplace_frank", dict(dist_func='laplace', dist_scale=2e-2, aggr_alpha_func='frank', aggr_alpha_t_conorm_p=3.0)),
    ("guder_aczel", dict(dist_func='gudermannian', dist_scale=2e-2, aggr_alpha_func='aczel_alsina', aggr_alpha_t_conorm_p=0.7)),
    ("cauchy_dombi", dict(dist_func='cauchy', dist_scale=1e-2, aggr_alpha_func='dombi', aggr_alpha_t_conorm_p=1.5)),
    ("reciprocal_ss", dict(dist_func='reciprocal', dist_scale=1e-2, aggr_alpha_func='schweizer_sklar', aggr_alpha_t_conorm_p=-1.5)),
    ("gumbelmax_prob", dict(dist_func='gumbel_max', dist_scale=2e-2)),
    ("gumbelmin_einstein", dict(dist_func='gumbel_min', dist_scale=2e-2, aggr_alpha_func='einstein')),
    ("exp_prob", dict(dist_func='exponential', dist_scale=3e-2, dist_shift=0.5)),
    ("exprev_prob", dict(dist_func='exponential_rev', dist_scale=2e-2)),
    ("gammarev_prob", dict(dist_func='gamma_rev', dist_shape=1.5, dist_scale=2e-2)),
    ("gamma1_shift_prob", dict(dist_func='gamma', dist_shape=1.0, dist_shift=0.5, dist_scale=2e-2)),       # shape 1 and 2: the device's
    ("gammarev2_einstein", dict(dist_func='gamma_rev', dist_shape=2.0, dist_scale=2e-2, aggr_alpha_func='einstein')),   # power-free branches
    ("gamma35_prob", dict(dist_func='gamma', dist_shape=3.5, dist_scale=1e-2)),
    ("levy_prob", dict(dist_func='levy', dist_scale=2e-2, dist_shift=1.0)),
    ("levyrev_prob", dict(dist_func='levy_rev', dist_scale=1e-2)),
    ("uniform_hardalpha", dict(aggr_alpha_func='hard')),
    ("uniform_smalleps", dict(dist_eps=1.5, dist_scale=2e-2)),
    ("uniform_singleside", dict(double_side=False)),
    ("logistic_hardrgb_single", dict(dist_func='logistic', dist_scale=2e-2, aggr_rgb_func='hard', double_side=False)),
    ("uniform_bg", dict(background=(0.2, 0.5, 0.9))),
    ("uniform_T4", dict(T=4)),
    ("uniform_T9_clamp", dict(T=9, texel_mode=1)),
    ("uniform_T1_clamp", dict(texel_mode=1)),
]


# every dist_func / aggr_alpha_func id with parameters that are valid by gendr_validate's rules (kernel.cu:296,491,501,512,522,534,552):
# the axes of tests/test_gpu_cross_product.py and of the fuzz draws' independent picks
DISTS = [
    ('hard', {}), ('uniform', dict(dist_scale=2e-2)), ('cubic_hermite', dict(dist_scale=5e-2)),
    ('wigner_semicircle', dict(dist_scale=5e-2)), ('gaussian', dict(dist_scale=2e-2)), ('laplace', dict(dist_scale=2e-2)),
    ('logistic', dict(dist_scale=2e-2)), ('gudermannian', dict(dist_scale=2e-2)), ('cauchy', dict(dist_scale=1e-2)),
    ('reciprocal', dict(dist_scale=1e-2)), ('gumbel_max', dict(dist_scale=2e-2)), ('gumbel_min', dict(dist_scale=2e-2)),
    ('exponential', dict(dist_scale=3e-2, dist_shift=0.5)), ('exponential_rev', dict(dist_scale=2e-2)),
    ('gamma', dict(dist_scale=2e-2, dist_shape=2.0)), ('gamma_rev', dict(dist_scale=2e-2, dist_shape=1.5)),
    ('levy', dict(dist_scale=2e-2, dist_shift=1.0)), ('levy_rev', dict(dist_scale=1e-2)),
]
AGGRS = [
    ('hard', None), ('max', None), ('probabilistic', None), ('einstein', None), ('hamacher', 0.5), ('frank', 3.0),
    ('yager', 2.0), ('aczel_alsina', 0.7), ('dombi', 1.5), ('schweizer_sklar', -1.5),
]


def independent_options(rs, opts):
    """The fuzz draws used to pick one of OPTION_MATRIX's 31 named option sets (dist and aggr never varied
    independently).  Half of the draws now replace the set's t-conorm, half its distribution, by an independent pick."""
    opts = dict(opts)
    if rs.randint(2):
        a, p = AGGRS[rs.randint(len(AGGRS))]
        opts['aggr_alpha_func'] = a
        opts.pop('aggr_alpha_t_conorm_p', None)
        if p is not None:
            opts['aggr_alpha_t_conorm_p'] = p
    if rs.randint(2):
        d, dopt = DISTS[rs.randint(len(DISTS))]
        for k in ('dist_scale', 'dist_shape', 'dist_shift', 'dist_squared'):
            opts.pop(k, None)
        opts.update(dopt, dist_func=d)
    return opts
