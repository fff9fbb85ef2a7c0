"""Second, independent restatement of the hot path: vectorised pure PyTorch on the CPU.

TEST INFRASTRUCTURE ONLY (same rules as the C oracle; held to the C oracle, which is pinned to the reference's own kernels, see gendr_oracle.h).
Purpose: (i) cross-check the C oracle with a differently structured implementation (whole-image tensor ops per
face instead of scalar loops), (ii) BASELINE.json config 1 ("pure-PyTorch CPU per-pixel reference"), (iii) the
"pure-PyTorch CPU evaluation" bench.py can time next to the GPU number.

Covers the whole option matrix: all 18 distributions (CDF kernel.cu:243-363, density :367-459), all 10 alpha
aggregators (fold :474-563, closed-form partial :567-614), hard and softmax RGB, dist_squared, surface textures
with any T = R*R (both texel modes, incl. the texel-index overflow quirk :179-184) and vertex textures, forward
and backward.  Same operation order and the same float<->double promotions as kernel.cu (cited inline: an
expression is evaluated in double exactly where a double literal or a double-returning call promotes it in the
reference's scalar_t = float instantiation); works in float32 or float64.
"kernel.cu" = /root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu.
"""
import math

import torch

DIST = {'hard': 0, 'heaviside': 0, 'uniform': 1, 'cubic_hermite': 2, 'wigner_semicircle': 3, 'gaussian': 4,
        'laplace': 5, 'logistic': 6, 'gudermannian': 7, 'hyperbolic_secant': 7, 'cauchy': 8, 'reciprocal': 9,
        'gumbel_max': 10, 'gumbel_min': 11, 'exponential': 12, 'exponential_rev': 13, 'gamma': 14, 'gamma_rev': 15,
        'levy': 16, 'levy_rev': 17}
ALPHA = {'hard': 0, 'max': 1, 'probabilistic': 2, 'einstein': 3, 'hamacher': 4, 'frank': 5, 'yager': 6,
         'aczel_alsina': 7, 'dombi': 8, 'schweizer_sklar': 9}
RGB = {'hard': 0, 'softmax': 1}
PI = math.pi


def _f(x, dt):
    return torch.tensor(x, dtype=dt)


def _sqrt(x):
    """IEEE-correct square root.  torch.sqrt on float32 CPU tensors is NOT correctly rounded (vectorised
    approximation, 1 ulp off in places), sqrtf in the reference is: go through float64 (a correctly rounded
    double sqrt rounded to float is the correctly rounded float sqrt)."""
    return torch.sqrt(x.double()).to(x.dtype)


def face_info(fv):
    """kernel.cu:620-676 for fv [nf,3,3] -> inv [nf,9], sym [nf,9], obt [nf,3]."""
    dt = fv.dtype
    x, y = fv[:, :, 0], fv[:, :, 1]
    x0, x1, x2, y0, y1, y2 = x[:, 0], x[:, 1], x[:, 2], y[:, 0], y[:, 1], y[:, 2]
    adj = torch.stack([y1 - y2, x2 - x1, x1 * y2 - x2 * y1,
                       y2 - y0, x0 - x2, x2 * y0 - x0 * y2,
                       y0 - y1, x1 - x0, x0 * y1 - x1 * y0], 1)
    det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0)
    d64 = det.double()
    det = torch.where(det > 0, torch.clamp(d64, min=1e-10), torch.clamp(d64, max=-1e-10)).to(dt)   # :653
    inv = adj / det[:, None]
    sym = (x[:, :, None] * x[:, None, :] + y[:, :, None] * y[:, None, :] + 1).reshape(-1, 9)        # :659-665
    obt = torch.zeros(fv.shape[0], 3, dtype=dt)
    found = torch.zeros(fv.shape[0], dtype=torch.bool)
    for k in range(3):                                                                               # :667-675
        k1, k2 = (k + 1) % 3, (k + 2) % 3
        neg = ((x[:, k1] - x[:, k]) * (x[:, k2] - x[:, k]) + (y[:, k1] - y[:, k]) * (y[:, k2] - y[:, k])) < 0
        obt[:, k] = (neg & ~found).to(dt)
        found |= neg
    return inv, sym, obt


class _Pair:
    pass


def _eval_face(f, inv, sym, obt, xp, yp, o, dt):
    """All pixels against one face: kernel.cu:747-786.  Returns a _Pair of [P] tensors plus `live`."""
    q = _Pair()
    xs, ys = f[:, 0], f[:, 1]
    thr = (_f(o['dist_eps'], torch.float32) * _f(o['dist_scale'], torch.float32)).to(dt)                # :725 float * float
    sthr = _sqrt(thr)                                                                                   # :747
    live = ~((xp > xs.max() + sthr) | (xp < xs.min() - sthr) | (yp > ys.max() + sthr) | (yp < ys.min() - sthr))
    w = [inv[3 * k] * xp + inv[3 * k + 1] * yp + inv[3 * k + 2] for k in range(3)]                   # :39-43
    q.w = w
    one, zero = _f(1.0, dt), _f(0.0, dt)
    if o['dist_func'] == 0:
        inside = (w[0] <= 1) & (w[0] >= 0) & (w[1] <= 1) & (w[1] >= 0) & (w[2] <= 1) & (w[2] >= 0)
        q.frag = torch.where(inside, one, zero)
        q.sign = torch.zeros_like(xp); q.dx = torch.zeros_like(xp); q.dy = torch.zeros_like(xp)
        q.dis = torch.zeros_like(xp); q.t = [torch.zeros_like(xp)] * 3
    else:
        strict_in = (w[0] > 0) & (w[1] > 0) & (w[2] > 0) & (w[0] < 1) & (w[1] < 1) & (w[2] < 1)      # :83-84
        # ---- inside: nearest of the three edges (:86-123)
        best = torch.full_like(xp, 100000000.0)
        bx = torch.zeros_like(xp); by = torch.zeros_like(xp)
        bt = [torch.zeros_like(xp) for _ in range(3)]
        tv_edge = []
        for k in range(3):
            v1, v2 = (k + 1) % 3, (k + 2) % 3
            a0 = [sym[3 * k + j] - sym[3 * v1 + j] for j in range(3)]
            tv = (w[0] * a0[0] + w[1] * a0[1] + w[2] * a0[2] - a0[v1]) / (a0[k] - a0[v1])
            tv_edge.append(tv)
            t0 = [None] * 3
            t0[k] = tv; t0[v1] = 1 - tv; t0[v2] = torch.zeros_like(tv)
            t0 = [t0[j] - w[j] for j in range(3)]
            dx = t0[0] * xs[0] + t0[1] * xs[1] + t0[2] * xs[2]
            dy = t0[0] * ys[0] + t0[1] * ys[1] + t0[2] * ys[2]
            d = dx * dx + dy * dy
            better = d < best
            best = torch.where(better, d, best)
            bx = torch.where(better, dx, bx); by = torch.where(better, dy, by)
            bt = [torch.where(better, t0[j], bt[j]) for j in range(3)]
        # ---- outside: region logic (:125-139), one clamped edge (:141-163)
        n = [w[k] <= 0 for k in range(3)]
        v0 = torch.full(xp.shape, -1, dtype=torch.long)
        c0, c1, c2 = n[1] & n[2], n[2] & n[0], n[0] & n[1]
        dot0 = (xp - xs[0]) * (xs[2] - xs[0]) + (yp - ys[0]) * (ys[2] - ys[0])
        dot1 = (xp - xs[1]) * (xs[0] - xs[1]) + (yp - ys[1]) * (ys[0] - ys[1])
        dot2 = (xp - xs[2]) * (xs[1] - xs[2]) + (yp - ys[2]) * (ys[1] - ys[2])
        e0 = torch.where((obt[0] == 1) & (dot0 > 0), 2, 0)
        e1 = torch.where((obt[1] == 1) & (dot1 > 0), 0, 1)
        e2 = torch.where((obt[2] == 1) & (dot2 > 0), 1, 2)
        single = torch.where(n[0], 1, torch.where(n[1], 2, torch.where(n[2], 0, -1)))
        v0 = torch.where(c0, e0, torch.where(c1, e1, torch.where(c2, e2, single)))
        nan_w = (w[0] != w[0]) | (w[1] != w[1]) | (w[2] != w[2])
        wstack = torch.stack(w, 0)
        argmin = torch.where(torch.isnan(wstack), torch.full_like(wstack, float('inf')), wstack).argmin(0)
        v0 = torch.where(v0 < 0, (argmin + 1) % 3, v0)        # decision for the reference's v0 = -1 indexing
        tv = torch.where(v0 == 0, tv_edge[0], torch.where(v0 == 1, tv_edge[1], tv_edge[2]))
        ta = torch.clamp(tv, 0, 1)
        tb = torch.clamp(1 - tv, 0, 1)
        ta = torch.where(torch.isnan(tv), torch.zeros_like(tv), ta)
        tb = torch.where(torch.isnan(tv), torch.zeros_like(tv), tb)
        z = torch.zeros_like(tv)
        t_out = [torch.where(v0 == 0, ta, torch.where(v0 == 1, z, tb)) - w[0],
                 torch.where(v0 == 0, tb, torch.where(v0 == 1, ta, z)) - w[1],
                 torch.where(v0 == 0, z, torch.where(v0 == 1, tb, ta)) - w[2]]
        ox = t_out[0] * xs[0] + t_out[1] * xs[1] + t_out[2] * xs[2]
        oy = t_out[0] * ys[0] + t_out[1] * ys[1] + t_out[2] * ys[2]
        q.sign = torch.where(strict_in, one, -one)
        q.dx = torch.where(strict_in, bx, ox); q.dy = torch.where(strict_in, by, oy)
        q.t = [torch.where(strict_in, bt[j], t_out[j]) for j in range(3)]
        live &= ~(nan_w & ~strict_in)
        dis = q.dx * q.dx + q.dy * q.dy                                                              # :768
        live &= ~((q.sign < 0) & (dis >= thr))                                                      # :769
        if not o['dist_squared']:
            dis = _sqrt(dis)
        q.dis = dis
        q.frag = cdf(o['dist_func'], q.sign, dis, o, dt)
    live &= ~(q.frag.double() <= 0.000001)                                                          # :784
    q.live = live
    return q


def _params(o, dt):
    """dist_scale / dist_shape / dist_shift / t-conorm p: float32 kernel arguments (kernel.cu:1077-1092) read as scalar_t."""
    return [_f(0.0 if o.get(k) is None else o[k], torch.float32).to(dt) for k in ('dist_scale', 'dist_shape', 'dist_shift', 't_conorm_p')]


def cdf(fid, sign, x, o, dt):
    """sigmoid_forward_cuda, kernel.cu:243-363: D(sign, x) for tensors sign (+-1) and x >= 0 of dtype dt."""
    scale, shape, shift, _ = _params(o, dt)
    S = lambda t: t.to(dt)
    one, zero = _f(1.0, dt), _f(0.0, dt)
    u = sign * x / scale
    if fid == 0:                                                                                     # :251-252
        return torch.where(sign > 0, one, zero)
    if fid == 6:                                                                                     # :254-255
        return S(1. / (1. + torch.exp(-sign * x / scale).double()))
    if fid == 8:                                                                                     # :257-258 explicit atanf
        return S(torch.atan(u.float()).double() / PI + 0.5)
    if fid == 9:                                                                                     # :260-261
        return S((sign * x / scale / (1 + x / scale)).double() / 2. + 0.5)
    if fid == 5:                                                                                     # :263-268
        e = torch.exp(-x / scale).double()
        return torch.where(sign < 0, S(0.5 * e), S(1. - 0.5 * e))
    if fid in (1, 2):                                                                                # :270-277, :282-290
        y = S((sign * x).double() * 0.5 / scale.double() + 0.5)
        if fid == 2:
            y = 3 * y * y - 2 * y * y * y
        return torch.where(u < -1, zero, torch.where(u < 1, y, one))
    if fid == 7:                                                                                     # :279-280
        return S(torch.atan(torch.tanh(u.double() / 2.)) * 2. / PI + 0.5)
    if fid == 4:                                                                                     # :292-293 normcdf(scalar_t)
        return S(0.5 * torch.erfc(-u * _f(0.70710678118654752440, dt)))
    if fid in (14, 15):                                                                              # :295-319
        if float(shape) < 0:
            return torch.full_like(x, float('nan'))
        sx = sign * x + shift * scale if fid == 14 else sign * x - shift * scale
        xs = sx if fid == 14 else -sx
        xr = xs / scale
        kummers = S(_f(1. / math.gamma(float(shape.double()) + 1.), torch.float64)).expand_as(x).clone()
        factor = kummers.clone()
        for i in range(1, 32):
            factor = factor * (xs / scale / (shape + i))
            kummers = kummers + factor
        y = torch.pow(xr, shape) * torch.exp(-xs / scale) * kummers
        cut = xr.double() > 15.
        if fid == 14:
            return torch.where(sx <= 0, zero, torch.where(cut, one, y))
        return torch.where(sx >= 0, one, torch.where(cut, zero, S(1. - y.double())))
    if fid == 3:                                                                                     # :320-327
        mid = S(0.5 + (sign * x * _sqrt(scale * scale - x * x)).double() / (PI * scale.double() * scale.double())
                + torch.asin(u).double() / PI)
        return torch.where(u < -1, zero, torch.where(u < 1, mid, one))
    if fid == 10:                                                                                    # :329-331
        return torch.exp(-torch.exp(-sign * x / scale))
    if fid == 11:                                                                                    # :333-335
        return S(1. - torch.exp(-torch.exp(u)).double())
    if fid in (16, 17):                                                                              # :337-347
        sx = sign * x + shift * scale if fid == 16 else sign * x - shift * scale
        xs = sx if fid == 16 else -sx
        y = S(torch.erfc(torch.sqrt(scale.double() / 2. / xs.double())))
        if fid == 16:
            return torch.where(sx.double() <= 1e-6, zero, y)
        return torch.where(sx.double() >= -1e-6, one, S(1. - y.double()))
    if fid in (12, 13):                                                                              # :349-359
        sx = sign * x + shift * scale if fid == 12 else sign * x - shift * scale
        xs = sx if fid == 12 else -sx
        y = S(1. - torch.exp(-xs / scale).double())
        if fid == 12:
            return torch.where(sx < 0, zero, y)
        return torch.where(sx > 0, one, S(1. - y.double()))
    raise ValueError('unknown dist_func id %r' % (fid,))


def pdf(fid, sign, x, o, dt):
    """sigmoid_backward_cuda, kernel.cu:367-459."""
    scale, shape, shift, _ = _params(o, dt)
    S = lambda t: t.to(dt)
    zero = _f(0.0, dt)
    sd = scale.double()
    u = sign * x / scale
    if fid == 0:                                                                                     # :375-376
        return torch.zeros_like(x)
    if fid == 6:                                                                                     # :378-380
        y = S(1. / (1. + torch.exp(-sign * x / scale).double()))
        return y * (1 - y) / scale
    if fid == 8:                                                                                     # :382-383
        return S(1. / (PI * sd + PI / sd * x.double() * x.double()))
    if fid == 9:                                                                                     # :385-386
        s1 = (scale + x).double()
        return S(sd / (2. * s1 * s1))
    if fid == 5:                                                                                     # :388-389
        return S(0.5 / sd * torch.exp(-x / scale).double())
    if fid == 1:                                                                                     # :391-392
        return torch.where((u > -1) & (u < 1), S(0.5 / sd), zero)
    if fid == 7:                                                                                     # :394-395
        return S(1. / torch.cosh(u).double() / PI / sd)
    if fid == 2:                                                                                     # :397-402
        v = S(0.75 / sd - 0.75 * (x * x).double() / torch.pow(sd, 3.))
        return torch.where((u.double() < -1.) | (u.double() > 1.), zero, v)
    if fid == 4:                                                                                     # :404-405
        q = (x / scale).double()
        return S(1. / sd / math.sqrt(2. * PI) * torch.exp(-0.5 * q * q))
    if fid in (14, 15):                                                                              # :407-423, explicit double
        if float(shape) < 0:
            return torch.full_like(x, float('nan'))
        pd = shape.double()
        if fid == 14:
            dead = (sign * x + shift * scale) <= 0
            xs = sign.double() * x.double() + shift.double() * sd
        else:
            dead = (sign * x - shift * scale) >= 0
            xs = -(sign.double() * x.double() - shift.double() * sd)
        g = math.gamma(float(pd)) if float(pd) > 0 else float('inf')                                 # tgamma(0) = +inf
        v = torch.pow(1. / sd, pd) / g * torch.pow(xs, pd - 1.) * torch.exp(-xs / sd)
        return torch.where(dead, zero, S(v))
    if fid == 3:                                                                                     # :425-427
        v = S(2. / PI / sd / sd * _sqrt(scale * scale - x * x).double())
        return torch.where(x / scale > 1, zero, v)
    if fid == 10:                                                                                    # :429-430
        return torch.exp(-(u + torch.exp(-u))) / scale
    if fid == 11:                                                                                    # :432-433
        return torch.exp(-((-sign * x / scale) + torch.exp(u))) / scale
    if fid in (16, 17):                                                                              # :435-444
        sx = sign * x + shift * scale if fid == 16 else sign * x - shift * scale
        xs = (sx if fid == 16 else -sx).double()
        v = S(torch.sqrt(sd / 2. / PI) * torch.exp(-sd / 2. / xs) / torch.pow(xs, 1.5))
        dead = sx.double() <= 1e-6 if fid == 16 else sx.double() >= -1e-6
        return torch.where(dead, zero, v)
    if fid in (12, 13):                                                                              # :446-455
        sx = sign * x + shift * scale if fid == 12 else sign * x - shift * scale
        xs = sx if fid == 12 else -sx
        dead = sx < 0 if fid == 12 else sx > 0
        return torch.where(dead, zero, S(1. / sd * torch.exp(-xs / scale).double()))
    raise ValueError('unknown dist_func id %r' % (fid,))


def t_conorm_fold(tid, a_ex, b_new, o, dt):
    """t_conorm_forward_cuda, kernel.cu:474-563: T(alpha so far, new fragment)."""
    p = _params(o, dt)[3]
    S = lambda t: t.to(dt)
    pd = p.double()
    if tid == 1:
        return torch.maximum(a_ex, b_new)
    if tid == 2:
        return a_ex + b_new - a_ex * b_new
    if tid == 3:
        return (a_ex + b_new) / (1 + a_ex * b_new)
    a, b = S(1. - a_ex.double()), S(1. - b_new.double())
    one = _f(1.0, dt)
    if tid == 4:                                                                                     # :490-498
        c = S((a * b).double() / torch.clamp(pd + (1. - pd) * (a + b - a * b).double(), min=1e-6))
    elif tid == 5:                                                                                   # :500-509
        c = S(torch.log1p((torch.pow(p, a).double() - 1.) * (torch.pow(p, b).double() - 1.) / (pd - 1.)) / torch.log(p).double())
    elif tid == 6:                                                                                   # :511-519
        c = S(torch.clamp(1. - torch.pow(torch.pow(1. - a.double(), pd) + torch.pow(1. - b.double(), pd), 1. / pd), min=0.))
    elif tid == 7:                                                                                   # :521-531
        c = S(torch.exp(-torch.pow((torch.pow(-torch.log(a), p) + torch.pow(-torch.log(b), p)).double(), 1. / pd)))
        return torch.where((a.double() < 1e-8) | (b.double() < 1e-8), one, S(1. - c.double()))
    elif tid == 8:                                                                                   # :533-549
        c = S(1. / (1. + torch.pow(torch.pow((1. - a.double()) / a.double(), pd) + torch.pow((1. - b.double()) / b.double(), pd), 1. / pd)))
        return torch.where((a.double() < 1e-8) | (b.double() < 1e-8), one, S(1. - c.double()))
    elif tid == 9:                                                                                   # :551-559
        c = S(torch.pow((torch.pow(a, p) + torch.pow(b, p)).double() - 1., 1. / pd))
    else:
        raise ValueError('unknown t-conorm id %r' % (tid,))
    return S(1. - c.double())


def t_conorm_grad(tid, A, b, o, dt):
    """t_conorm_backward_cuda, kernel.cu:567-614: d alpha_final / d D_f from the final alpha A and b = D_f."""
    p = _params(o, dt)[3]
    S = lambda t: t.to(dt)
    pd, Ad, bd = p.double(), A.double(), b.double()
    if tid == 1:
        return torch.where(A == b, _f(1.0, dt), _f(0.0, dt))
    if tid == 2:
        return S((1. - Ad) / torch.clamp(1. - bd, min=1e-6))
    if tid == 3:
        return S((1. - (A * A).double()) / torch.clamp(1. - (b * b).double(), min=1e-6))
    if tid == 4:
        return S((1.0 - Ad) * (-Ad - pd * (1.0 - Ad) + pd + 1.0) / torch.clamp((1.0 - bd) * (-bd - pd * (1.0 - bd) + pd + 1.0), min=1e-6))
    if tid == 5:
        d = S(torch.pow(pd, 1.0 - bd) - 1.0)
        return S(torch.pow(p, A - b).double() * (torch.pow(pd, 1.0 - Ad) - 1.0) / (d.double() + torch.copysign(torch.full_like(bd, 1e-6), d.double())))
    if tid == 6:
        v = S(torch.pow(bd, pd - 1.) * torch.pow(Ad, 1. - pd))
        return torch.where(A == 1., _f(0.0, dt), v)
    if tid == 7:
        return S((1. - Ad) * torch.pow(-torch.log1p(torch.clamp(-bd, min=-1. + 1e-6)), pd - 1.)
                 * torch.pow(-torch.log1p(torch.clamp(-Ad, min=-1. + 1e-6)), 1. - pd) / torch.clamp(1. - bd, min=1e-6))
    if tid == 8:
        return S((1. - Ad) * (1. - Ad) * torch.pow(bd / torch.clamp(1. - bd, min=1e-6), pd - 1.)
                 * torch.pow(Ad / torch.clamp(1. - Ad, min=1e-6), 1. - pd) / torch.clamp(1. - bd, min=1e-6) / torch.clamp(1. - bd, min=1e-6))
    if tid == 9:
        a1, b1 = S(torch.clamp(1. - Ad, min=1e-6)), S(torch.clamp(1. - bd, min=1e-6))
        inner = torch.pow((-torch.pow(b1, p) + torch.pow(a1, p)).double() + 1., 1. / pd)
        return S(torch.pow(b1.double(), pd - 1.) * torch.pow(torch.pow(b1, p).double() + torch.pow(inner, pd) - 1., (1. - pd) / pd))
    raise ValueError('unknown t-conorm id %r' % (tid,))


def _clip_depth(q, f, dt):
    wc = [torch.clamp(q.w[k], 0, 1) for k in range(3)]                                              # :68-72
    s = wc[0] + wc[1] + wc[2]
    s = torch.where(s.double() > 1e-5, s, _f(1e-5, dt))
    wc = [c / s for c in wc]
    zp = 1 / (wc[0] / f[0, 2] + wc[1] / f[1, 2] + wc[2] / f[2, 2])                                   # :809
    return wc, zp


def _colour(tex_all, face_lin, wc, o, dt):
    """forward_sample_texture, kernel.cu:174-191.  tex_all [B*nf, T, 3]; returns (colour [3][P], own [P] int64): `own`
    is the texel of the face's own block that receives gradient (backward_sample_texture :194-213 only matches
    j < T), -1 if none.  Surface texel index :179-184: for w = (1,0,0) / (0,1,0) it runs past the face's block into
    the following texels (quirk); reads that would leave the tensor use the clamped index and get no gradient."""
    if o['texture_type'] == 1:
        tex = tex_all[face_lin]
        return [wc[0] * tex[0, k] + wc[1] * tex[1, k] + wc[2] * tex[2, k] for k in range(3)], torch.zeros(wc[0].shape, dtype=torch.int64)
    T = tex_all.shape[1]
    R = int(math.sqrt(T))
    flat = tex_all.reshape(-1, 3)

    def index(clamp):
        wx, wy = (wc[0] * R).to(torch.int64), (wc[1] * R).to(torch.int64)
        if clamp:
            wx, wy = torch.clamp(wx, max=R - 1), torch.clamp(wy, max=R - 1)
        lower = (wc[0] + wc[1]) * R - wx.to(dt) - wy.to(dt) <= 1
        return torch.where(lower, wy * R + wx, (R - 1 - wy) * R + (R - 1 - wx))

    clamped = torch.clamp(index(True), 0, T - 1)
    if o['texel_mode'] == 1:
        idx, own = clamped, clamped
    else:
        idx = index(False)
        at = face_lin * T + idx
        oob = (at >= flat.shape[0]) | (at < 0)
        own = torch.where(oob | (idx < 0) | (idx >= T), torch.full_like(idx, -1), idx)
        idx = torch.where(oob, clamped, idx)
    at = face_lin * T + idx
    return [flat[at, k] for k in range(3)], own


def render(fv, tex, image_size, grad=None, background=(0., 0., 0.), dist_func='uniform', dist_scale=1e-2,
           dist_squared=False, dist_shape=None, dist_shift=None, dist_eps=1e4, aggr_alpha_func='probabilistic',
           aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax',
           aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3, near=1, far=100, double_side=True, texture_type='surface',
           texel_mode=0):
    """fv [B,nf,3,3], tex [B,nf,T,3] CPU tensors (float32 or float64).  Returns dict(rgba, aggrs_info[, grad_faces,
    grad_textures]) following kernel.cu:680-862 and :866-1065."""
    dt = fv.dtype
    o = dict(dist_func=DIST[dist_func] if isinstance(dist_func, str) else dist_func, dist_scale=dist_scale,
             dist_squared=dist_squared, dist_eps=dist_eps, dist_shape=dist_shape, dist_shift=dist_shift,
             t_conorm_p=aggr_alpha_t_conorm_p,
             alpha=ALPHA[aggr_alpha_func] if isinstance(aggr_alpha_func, str) else aggr_alpha_func,
             rgb=RGB[aggr_rgb_func] if isinstance(aggr_rgb_func, str) else aggr_rgb_func,
             texture_type={'surface': 0, 'vertex': 1}[texture_type] if isinstance(texture_type, str) else texture_type,
             texel_mode=texel_mode)
    B, nf = fv.shape[:2]
    isz = image_size
    idx = torch.arange(isz, dtype=torch.float64)
    coord = ((2. * idx + 1. - isz) / isz).to(dt)                                                     # :718-719
    xp = coord[None, :].expand(isz, isz).reshape(-1)
    yp = coord.flip(0)[:, None].expand(isz, isz).reshape(-1)                                         # row 0 = top
    P = isz * isz
    gam = _f(aggr_rgb_gamma, torch.float32).to(dt)
    nearf, farf = _f(near, torch.float32).to(dt), _f(far, torch.float32).to(dt)
    zrange = (_f(far, torch.float32) - _f(near, torch.float32)).to(dt)
    rgba = torch.zeros(B, 4, P, dtype=dt)
    aux = torch.zeros(B, 2, P, dtype=dt)
    gfv = torch.zeros(B, nf, 3, 3, dtype=dt) if grad is not None else None
    gtex = torch.zeros_like(tex) if grad is not None else None
    tex_all = tex.reshape(B * nf, tex.shape[2], 3)
    T = tex.shape[2]
    for b in range(B):
        inv, sym, obt = face_info(fv[b])
        alpha = torch.zeros(P, dtype=dt)
        ssum = torch.exp(_f(aggr_rgb_eps, torch.float32) / _f(aggr_rgb_gamma, torch.float32)).to(dt).expand(P).clone()   # :729
        smax = _f(aggr_rgb_eps, torch.float32).to(dt).expand(P).clone()
        bg = [_f(background[k], torch.float32).to(dt).expand(P) for k in range(3)]
        col = [bg[k] * ssum if o['rgb'] == 1 else bg[k].clone() for k in range(3)]
        depth_min = torch.full((P,), 10000000.0, dtype=dt)
        face_min = torch.full((P,), -1, dtype=torch.long)
        saved = []
        for fn in range(nf):
            f = fv[b, fn]
            q = _eval_face(f, inv[fn], sym[fn], obt[fn], xp, yp, o, dt)
            live = q.live
            if o['alpha'] == 0:                                                                      # :791-803
                alpha = torch.where(live & (q.frag.double() > 0.5), _f(1.0, dt), alpha)
            else:
                alpha = torch.where(live, t_conorm_fold(o['alpha'], alpha, q.frag, o, dt), alpha)
            wc, zp = _clip_depth(q, f, dt)
            ok = live & ~((zp < nearf) | (zp > farf))                                                # :810
            front = bool((f[2, 1] - f[0, 1]) * (f[1, 0] - f[0, 0]) < (f[1, 1] - f[0, 1]) * (f[2, 0] - f[0, 0]))   # :56-58
            cc, own = _colour(tex_all, b * nf + fn, wc, o, dt)
            inside = (q.w[0] <= 1) & (q.w[0] >= 0) & (q.w[1] <= 1) & (q.w[1] >= 0) & (q.w[2] <= 1) & (q.w[2] >= 0)
            if o['rgb'] == 0:                                                                        # :815-822
                win = ok & (zp < depth_min) & inside & bool(double_side or front)
                depth_min = torch.where(win, zp, depth_min)
                face_min = torch.where(win, torch.full_like(face_min, fn), face_min)
                col = [torch.where(win, cc[k], col[k]) for k in range(3)]
            elif front or double_side:                                                               # :824-838
                zn = (farf - zp) / zrange
                deeper = ok & (zn > smax)
                edz = torch.where(deeper, torch.exp((smax - zn) / gam), _f(1.0, dt))
                smax_new = torch.where(deeper, zn, smax)
                ez = torch.exp((zn - smax_new) / gam)
                ssum = torch.where(ok, edz * ssum + ez * q.frag, ssum)
                col = [torch.where(ok, edz * col[k] + ez * q.frag * cc[k], col[k]) for k in range(3)]
                smax = torch.where(ok, smax_new, smax)
            if grad is not None:
                saved.append((q, wc, zp, ok, front, cc, own))
        rgba[b, 3] = alpha                                                                           # :845-861
        if o['rgb'] == 0:
            for k in range(3):
                rgba[b, k] = torch.where(face_min != -1, col[k], bg[k])
            aux[b, 0] = depth_min; aux[b, 1] = face_min.to(dt)
        else:
            for k in range(3):
                rgba[b, k] = col[k] / ssum
            aux[b, 0] = ssum; aux[b, 1] = smax

        if grad is None:
            continue
        g = grad[b].reshape(4, P).to(dt)
        out = rgba[b]
        for fn in range(nf):                                                                         # :919-1064
            q, wc, zp, ok, front, cc, own = saved[fn]
            f = fv[b, fn]
            C_alpha = g[3]
            if o['alpha'] != 0:                                                                      # :973-987
                C_alpha = C_alpha * t_conorm_grad(o['alpha'], out[3], q.frag, o, dt)
            C_xy = C_alpha
            gz = [torch.zeros(P, dtype=dt) for _ in range(3)]
            if o['rgb'] == 0:                                                                        # :997-1004
                win = ok & (aux[b, 1] == float(fn))
                if o['texture_type'] == 1:
                    for k in range(3):
                        for j in range(3):
                            gtex[b, fn, j, k] = torch.where(win, wc[j] * g[k], _f(0.0, dt)).double().sum().to(dt)
                else:
                    for k in range(3):
                        for j in range(T):
                            gtex[b, fn, j, k] = torch.where(win & (own == j), g[k], _f(0.0, dt)).double().sum().to(dt)
            elif front or double_side:                                                               # :1006-1030
                zn = (farf - zp) / zrange
                zs = q.frag * torch.exp((zn - aux[b, 1]) / gam) / aux[b, 0]
                C_rgb = torch.zeros(P, dtype=dt)
                for k in range(3):
                    if o['texture_type'] == 1:
                        for j in range(3):
                            gtex[b, fn, j, k] = torch.where(ok, zs * (wc[j] * g[k]), _f(0.0, dt)).double().sum().to(dt)
                    else:
                        for j in range(T):
                            gtex[b, fn, j, k] = torch.where(ok & (own == j), zs * g[k], _f(0.0, dt)).double().sum().to(dt)
                    C_rgb = C_rgb + g[k] * (cc[k] - out[k])
                C_rgb = C_rgb * zs
                C_xy = C_xy + C_rgb / q.frag
                nf_range = (_f(near, torch.float32) - _f(far, torch.float32)).to(dt)
                C_z = C_rgb / gam / nf_range * zp * zp
                gz = [C_z * wc[k] / f[k, 2] / f[k, 2] for k in range(3)]
            gxy = [[torch.zeros(P, dtype=dt)] * 2 for _ in range(3)]
            if o['dist_func'] != 0:                                                                  # :1034-1052
                C_xy = C_xy * pdf(o['dist_func'], q.sign, q.dis, o, dt)
                for k in range(3):
                    tw = q.t[k] + q.w[k]
                    if o['dist_squared']:
                        gxy[k] = [2 * q.sign * C_xy * tw * q.dx, 2 * q.sign * C_xy * tw * q.dy]
                    else:
                        nrm = torch.clamp(_sqrt(q.dx * q.dx + q.dy * q.dy).double(), min=1e-6)
                        gxy[k] = [((q.sign * C_xy * tw * q.dx).double() / nrm).to(dt), ((q.sign * C_xy * tw * q.dy).double() / nrm).to(dt)]
            zero = _f(0.0, dt)
            for k in range(3):
                gfv[b, fn, k, 0] = torch.where(ok, gxy[k][0], zero).double().sum().to(dt)
                gfv[b, fn, k, 1] = torch.where(ok, gxy[k][1], zero).double().sum().to(dt)
                gfv[b, fn, k, 2] = torch.where(ok, gz[k], zero).double().sum().to(dt)
    res = dict(rgba=rgba.reshape(B, 4, isz, isz), aggrs_info=aux.reshape(B, 2, isz, isz))
    if grad is not None:
        res['grad_faces'] = gfv
        res['grad_textures'] = gtex
    return res
