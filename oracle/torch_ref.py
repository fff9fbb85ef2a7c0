"""Second, independent restatement of the hot path: vectorised pure PyTorch on the CPU.

TEST INFRASTRUCTURE ONLY (same rules as the C oracle; PARITY UNPINNED by the reference, see gendr_oracle.h).
Purpose: (i) cross-check the C oracle with a differently structured implementation (whole-image tensor ops per
face instead of scalar loops), (ii) BASELINE.json config 1 ("pure-PyTorch CPU per-pixel reference"), (iii) the
"pure-PyTorch CPU evaluation" bench.py can time next to the GPU number.

Covers the option sets of the headline configs: dist_func in {hard, uniform, logistic, gaussian},
aggr_alpha_func in {hard, max, probabilistic, einstein}, aggr_rgb_func in {hard, softmax}, dist_squared,
surface textures with T == 1 (both texel modes) and vertex textures.  Same operation order and the same
float<->double promotions as kernel.cu (cited inline); works in float32 or float64.
"kernel.cu" = /root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu.
"""
import math

import torch

DIST = {'hard': 0, 'heaviside': 0, 'uniform': 1, 'gaussian': 4, 'logistic': 6}
ALPHA = {'hard': 0, 'max': 1, 'probabilistic': 2, 'einstein': 3}
RGB = {'hard': 0, 'softmax': 1}


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
        scale = _f(o['dist_scale'], torch.float32).to(dt)
        u = q.sign * dis / scale
        if o['dist_func'] == 1:                                                                      # uniform :270-277
            mid = ((q.sign * dis).double() * 0.5 / scale.double() + 0.5).to(dt)
            q.frag = torch.where(u < -1, zero, torch.where(u < 1, mid, one))
        elif o['dist_func'] == 6:                                                                    # logistic :254-255
            q.frag = (1. / (1. + torch.exp(-q.sign * dis / scale).double())).to(dt)
        elif o['dist_func'] == 4:                                                                    # gaussian :292-293
            q.frag = (0.5 * torch.erfc(-u * _f(0.70710678118654752440, dt))).to(dt)
        else:
            raise ValueError('dist_func not covered by the PyTorch restatement')
    live &= ~(q.frag.double() <= 0.000001)                                                          # :784
    q.live = live
    return q


def _pdf(q, o, dt):
    """kernel.cu:367-459 for the covered distributions."""
    scale = _f(o['dist_scale'], torch.float32).to(dt)
    if o['dist_func'] == 0:
        return torch.zeros_like(q.dis)
    u = q.sign * q.dis / scale
    if o['dist_func'] == 1:
        return torch.where((u > -1) & (u < 1), (0.5 / scale.double()).to(dt), _f(0.0, dt))
    if o['dist_func'] == 6:
        y = (1. / (1. + torch.exp(-q.sign * q.dis / scale).double())).to(dt)
        return y * (1 - y) / scale
    if o['dist_func'] == 4:
        qq = (q.dis / scale).double()
        return (1. / scale.double() / math.sqrt(2. * math.pi) * torch.exp(-0.5 * qq * qq)).to(dt)
    raise ValueError


def _clip_depth(q, f, dt):
    wc = [torch.clamp(q.w[k], 0, 1) for k in range(3)]                                              # :68-72
    s = wc[0] + wc[1] + wc[2]
    s = torch.where(s.double() > 1e-5, s, _f(1e-5, dt))
    wc = [c / s for c in wc]
    zp = 1 / (wc[0] / f[0, 2] + wc[1] / f[1, 2] + wc[2] / f[2, 2])                                   # :809
    return wc, zp


def _colour(tex, tex_next, wc, o, last_face):
    if o['texture_type'] == 1:
        return [wc[0] * tex[0, k] + wc[1] * tex[1, k] + wc[2] * tex[2, k] for k in range(3)], None
    # surface, R == 1 (kernel.cu:179-185): index 1 reads the next face's texel
    wx, wy = wc[0].to(torch.int64), wc[1].to(torch.int64)
    idx = torch.where((wc[0] + wc[1]) - wx - wy <= 1, wy + wx, -wy - wx)
    own = idx == 0
    if o['texel_mode'] == 1:
        return [tex[0, k].expand_as(wc[0]) for k in range(3)], torch.ones_like(own)
    use_next = (~own) & (not last_face)
    return [torch.where(use_next, tex_next[0, k], tex[0, k]) for k in range(3)], own


def render(fv, tex, image_size, grad=None, background=(0., 0., 0.), dist_func='uniform', dist_scale=1e-2,
           dist_squared=False, dist_eps=1e4, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax',
           aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3, near=1, far=100, double_side=True, texture_type='surface',
           texel_mode=0):
    """fv [B,nf,3,3], tex [B,nf,T,3] CPU tensors (float32 or float64).  Returns dict(rgba, aggrs_info[, grad_faces,
    grad_textures]) following kernel.cu:680-862 and :866-1065."""
    dt = fv.dtype
    o = dict(dist_func=DIST[dist_func] if isinstance(dist_func, str) else dist_func, dist_scale=dist_scale,
             dist_squared=dist_squared, dist_eps=dist_eps,
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
            elif o['alpha'] == 1:
                alpha = torch.where(live, torch.maximum(alpha, q.frag), alpha)
            elif o['alpha'] == 2:
                alpha = torch.where(live, alpha + q.frag - alpha * q.frag, alpha)
            elif o['alpha'] == 3:
                alpha = torch.where(live, (alpha + q.frag) / (1 + alpha * q.frag), alpha)
            wc, zp = _clip_depth(q, f, dt)
            ok = live & ~((zp < nearf) | (zp > farf))                                                # :810
            front = bool((f[2, 1] - f[0, 1]) * (f[1, 0] - f[0, 0]) < (f[1, 1] - f[0, 1]) * (f[2, 0] - f[0, 0]))   # :56-58
            last = (b == B - 1 and fn == nf - 1)
            nxt = tex[b, fn + 1] if fn + 1 < nf else (tex[b + 1, 0] if b + 1 < B else tex[b, fn])
            cc, own = _colour(tex[b, fn], nxt, wc, o, last)
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
            if o['alpha'] == 1:
                C_alpha = C_alpha * torch.where(out[3] == q.frag, _f(1.0, dt), _f(0.0, dt))
            elif o['alpha'] == 2:
                C_alpha = C_alpha * ((1. - out[3].double()) / torch.clamp(1. - q.frag.double(), min=1e-6)).to(dt)
            elif o['alpha'] == 3:
                C_alpha = C_alpha * ((1. - (out[3] * out[3]).double()) / torch.clamp(1. - (q.frag * q.frag).double(), min=1e-6)).to(dt)
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
                        gtex[b, fn, 0, k] = torch.where(win & own, g[k], _f(0.0, dt)).double().sum().to(dt)
            elif front or double_side:                                                               # :1006-1030
                zn = (farf - zp) / zrange
                zs = q.frag * torch.exp((zn - aux[b, 1]) / gam) / aux[b, 0]
                C_rgb = torch.zeros(P, dtype=dt)
                for k in range(3):
                    if o['texture_type'] == 1:
                        for j in range(3):
                            gtex[b, fn, j, k] = torch.where(ok, zs * (wc[j] * g[k]), _f(0.0, dt)).double().sum().to(dt)
                    else:
                        gtex[b, fn, 0, k] = torch.where(ok & own, zs * g[k], _f(0.0, dt)).double().sum().to(dt)
                    C_rgb = C_rgb + g[k] * (cc[k] - out[k])
                C_rgb = C_rgb * zs
                C_xy = C_xy + C_rgb / q.frag
                nf_range = (_f(near, torch.float32) - _f(far, torch.float32)).to(dt)
                C_z = C_rgb / gam / nf_range * zp * zp
                gz = [C_z * wc[k] / f[k, 2] / f[k, 2] for k in range(3)]
            gxy = [[torch.zeros(P, dtype=dt)] * 2 for _ in range(3)]
            if o['dist_func'] != 0:                                                                  # :1034-1052
                C_xy = C_xy * _pdf(q, o, dt)
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
