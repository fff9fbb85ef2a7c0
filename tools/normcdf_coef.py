"""Coefficients of gendr_math.h: norm_q() -- g(x) = Phi(-x) e^(x^2/2) on [0, 5.625] as a polynomial of degree 30 in t = 2 x / 5.625 - 1.
Chebyshev interpolant (degree + 1 nodes) in 70-digit arithmetic, converted to the monomial basis in the same arithmetic, rounded to
double; then a simulation of the double Horner evaluation on random float arguments against 70-digit values.
    python tools/normcdf_coef.py 30          (needs mpmath; a few seconds)"""
import mpmath as mp, numpy as np, sys
mp.mp.dps = 70
A, B = mp.mpf(0), mp.mpf('5.625')
def g(x): return mp.ncdf(-x) * mp.exp(x * x / 2)          # Q(x) e^{x^2/2}
def cheb_coeffs(f, n):
    N = n + 1
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1)/2) / N) for k in range(N)]
    fv = [f((B - A) / 2 * t + (A + B) / 2) for t in nodes]
    c = []
    for j in range(N):
        s = mp.fsum(fv[k] * mp.cos(mp.pi * j * (k + mp.mpf(1)/2) / N) for k in range(N))
        c.append(2 * s / N)
    c[0] /= 2
    return c
def cheb2mono(c):
    n = len(c)
    T0 = [mp.mpf(1)]; T1 = [mp.mpf(0), mp.mpf(1)]
    out = [mp.mpf(0)] * n
    for k in range(n):
        if k == 0: T = T0
        elif k == 1: T = T1
        else:
            T = [mp.mpf(0)] + [2 * v for v in T1]
            for i, v in enumerate(T0): T[i] -= v
            T0, T1 = T1, T
        for i, v in enumerate(T): out[i] += c[k] * v
    return out
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
co = np.array([float(v) for v in cheb2mono(cheb_coeffs(g, deg))])
rs = np.random.RandomState(1)
u = rs.uniform(0, 5.625, 3000).astype(np.float32).astype(np.float64)
t = u * (2.0 / 5.625) - 1.0
acc = np.zeros_like(t)
for k in co[::-1]: acc = acc * t + k
worst = 0
for i in range(len(u)):
    tr = g(mp.mpf(float(u[i])))
    worst = max(worst, abs(mp.mpf(float(acc[i])) / tr - 1))
print('deg', deg, 'worst rel err of g 2^%.1f' % float(mp.log(worst, 2)), 'max|c| %.3g' % np.abs(co).max())
for i in range(0, len(co), 3):
    print('        ' + ', '.join('%.17g' % v for v in co[i:i + 3]) + ',')
