"""Tables of gendr_math.h: norm_q_tab() -- Q(x) = Phi(-x) = e^(-x^2/2) g(x) on [0, 5.625] with TABLES instead of the degree-30 / degree-13
polynomials of norm_q() (round 6):
  * g(x) = Phi(-x) e^(x^2/2) on 16 intervals of width 45/128, one polynomial of degree DEG in t = x - centre per interval
    (Chebyshev interpolant on the slightly widened interval in 60-digit arithmetic, converted to the monomial basis);
  * e^y, y = -x^2/2: y = k ln2/16 + r, 2^(k/16) = 2^(k >> 4) T[k & 15], e^r by a degree-7 Taylor polynomial (|r| <= ln2/32).
Prints the C tables and simulates the double evaluation on random float arguments against 60-digit values.
    python tools/normcdf_table.py [DEG]"""
import sys
import mpmath as mp
import numpy as np
mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 9
NI, H = 16, mp.mpf(45) / 128
def g(x): return mp.ncdf(-x) * mp.exp(x * x / 2)
def cheb_mono(f, a, b, n):
    N = n + 1
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / N) for k in range(N)]
    fv = [f((b - a) / 2 * t + (a + b) / 2) for t in nodes]
    c = [2 * mp.fsum(fv[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / N) for k in range(N)) / N for j in range(N)]
    c[0] /= 2
    T0, T1 = [mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]
    out = [mp.mpf(0)] * N
    for k in range(N):
        if k == 0: T = T0
        elif k == 1: T = T1
        else:
            T = [mp.mpf(0)] + [2 * v for v in T1]
            for i, v in enumerate(T0): T[i] -= v
            T0, T1 = T1, T
        for i, v in enumerate(T): out[i] += c[k] * v
    return out      # in s = (x - mid) / halfwidth
tab = []
for i in range(NI):
    c0 = (2 * i + 1) * H / 2
    hw = H / 2 * (1 + mp.mpf(1) / 4096)                  # widened: the interval index comes from a rounded product
    m = cheb_mono(g, c0 - hw, c0 + hw, DEG)
    tab.append([float(m[j] / hw ** j) for j in range(DEG + 1)])      # monomial in t = x - c0
T = [float(mp.mpf(2) ** (mp.mpf(j) / 16)) for j in range(16)]
L16 = mp.log(2) / 16
l_hi = float(int(L16 * 2 ** 48) / mp.mpf(2) ** 48)       # 43 significant bits: k * l_hi is exact for |k| < 2^10
l_lo = float(L16 - mp.mpf(l_hi))
inv_l = float(1 / L16)
fact = [1.0 / float(mp.factorial(n)) for n in range(8)]
tabn = np.array(tab); Tn = np.array(T)
def norm_q(x):                                            # numpy double simulation of the device code
    y = -0.5 * (x * x)
    k = np.rint(y * inv_l)
    r = (y - k * l_hi) - k * l_lo                         # (fma on the device: k * l_hi is exact anyway)
    e = np.full_like(x, fact[7])
    for n in range(6, -1, -1): e = e * r + fact[n]
    ki = k.astype(np.int64)
    e = np.ldexp(e * Tn[ki & 15], (ki >> 4).astype(np.int32))
    i = np.minimum((x * float(1 / H)).astype(np.int64), NI - 1)
    t = x - (2 * i + 1) * float(H / 2)
    q = tabn[i, DEG].copy()
    for n in range(DEG - 1, -1, -1): q = q * t + tabn[i, n]
    return e * q
rs = np.random.RandomState(1)
x = np.concatenate([rs.uniform(0, 5.625, 20000), np.arange(1, NI) * float(H) + rs.uniform(-1e-6, 1e-6, NI - 1), [0.0, 5.6249995]]).astype(np.float32).astype(np.float64)
got = norm_q(x)
worst = max(abs(mp.mpf(float(v)) / mp.ncdf(-mp.mpf(float(u))) - 1) for u, v in zip(x, got))
print('// degree %d: worst relative error of Q over %d float arguments: 2^%.2f' % (DEG, len(x), float(mp.log(worst, 2))))
print('constexpr int kNormDeg = %d;' % DEG)
print('constexpr double kNormLn2_16Hi = %r, kNormLn2_16Lo = %r, kNorm16_Ln2 = %r;' % (l_hi, l_lo, inv_l))
print('// row i: 2^(i/16), then the coefficients of g on [i, i + 1] * 45/128 in t = x - (i + 1/2) * 45/128, constant term first')
print('__device__ const double kNormTab[16][%d] = {' % (DEG + 3 - (DEG + 2) % 2 if False else DEG + 2 + (DEG + 2) % 2))
for i in range(NI):
    row = [T[i]] + tab[i] + [0.0] * ((DEG + 2) % 2)
    print('    {' + ', '.join('%.17g' % v for v in row) + '},')
print('};')
