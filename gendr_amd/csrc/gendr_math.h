// gendr_math.h -- scalar math of the generalized soft rasterizer (gfx950 device code and
// the host scalar exports share it).
//
// What is computed: the 18 distribution CDFs D(s, x) and their x-derivatives, the 9
// t-conorm fold steps T(a, b) and the closed-form d alpha_final / d D_f, as the reference
// defines them in gendr/cuda/generalized_renderer_cuda_kernel.cu ("kernel.cu") :243-363,
// :367-459, :474-563, :567-614.
//
// Parity policy: fp32 with the reference's double sub-expressions kept where
// two or more double operations are chained before the value is rounded back to float
// (a single double +,-,*,/ or sqrt on float operands rounds to the same float as the float
// operation, so those are written in float).  Compiled with -ffp-contract=off: no FMA
// contraction anywhere in this header.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define GENDR_HD __host__ __device__ __forceinline__

// Build variant "fast" (gendr_amd/build.py VARIANTS: -DGENDR_FAST_MATH=1 -ffp-contract=on; libgendr_hip_fast.so).  The
// default build reproduces the reference's ROUNDING operation by operation (correctly rounded quotients through double
// reciprocals, correctly rounded sqrt / reciprocal, the library's expf, no contraction); the fast build keeps the
// reference's FORMULAS, operation order, fold order, skip tests and culling, and computes the per-pair arithmetic the way a
// compiler with contraction on (nvcc's default for the reference, /root/reference/setup.py:10) and 1-ulp hardware
// estimates would: a * (1/b) with a float reciprocal (<= 1 ulp), v_sqrt_f32, v_rcp_f32 + one Newton step, 2^x-based exp
// (~2 ulp), float instead of double sub-expressions.  It is gated on the GPU by the spread of the reference's OWN two
// builds (oracle/_ref render vs render_fma: tests/test_gpu_fast_variant.py) and never substituted silently
// (GENDR_VARIANT=fast / _native.use_variant('fast')).  Device code only: the host scalar exports stay exact.
#ifndef GENDR_FAST_MATH
#define GENDR_FAST_MATH 0
#endif
#if GENDR_FAST_MATH && defined(__HIP_DEVICE_COMPILE__)
#define GENDR_FAST_DEV 1
#else
#define GENDR_FAST_DEV 0
#endif

namespace gendr {

// the type a wave-uniform (or per-face) divisor's reciprocal is kept in: see div_by()
#if GENDR_FAST_DEV
typedef float rcp_t;
#else
typedef double rcp_t;
#endif

constexpr double kPi = 3.14159265358979323846;
constexpr double kProbThreshold = 0.000001;   // kernel.cu:13
constexpr int    kGammaSteps = 32;            // kernel.cu:16
constexpr double kGammaCut = 15.;             // kernel.cu:17

enum DistId : int {
    kHeaviside = 0, kUniform = 1, kCubicHermite = 2, kWigner = 3, kGaussian = 4, kLaplace = 5,
    kLogistic = 6, kGudermannian = 7, kCauchy = 8, kReciprocal = 9, kGumbelMax = 10, kGumbelMin = 11,
    kExponential = 12, kExponentialRev = 13, kGamma = 14, kGammaRev = 15, kLevy = 16, kLevyRev = 17,
    kNumDist = 18
};
enum AlphaId : int {
    kAlphaHard = 0, kMax = 1, kProbabilistic = 2, kEinstein = 3, kHamacher = 4, kFrank = 5,
    kYager = 6, kAczelAlsina = 7, kDombi = 8, kSchweizerSklar = 9, kNumAlpha = 10
};

struct DistParams {
    float scale;   // tau
    float shape;   // p of gamma
    float shift;   // shift (in units of tau) of the one-sided families
    rcp_t rscale;  // RN_double(1 / (double)scale), see div_by()
    // gamma family: what does not depend on the pair, computed once per call on the host (kernel.cu:309,:420-421)
    float  gamma_k0;       // (float)(1. / tgamma(shape + 1.)), first Kummer term
    double gamma_pdf_c;    // pow(1. / scale, shape) / tgamma(shape)
    const rcp_t* gamma_r;  // gamma_r[i - 1] = RN_double(1 / (double)(shape + i)), i = 1..31, or NULL: divide
    const double* norm_tab; // kNormTab in LDS (the kernels specialised for the gaussian distribution: norm_cdf_tab), or NULL
};

GENDR_HD DistParams make_dist_params(float scale, float shape, float shift)
{
    DistParams d = {scale, shape, shift, (rcp_t)(1. / (double)scale),
                    (float)(1. / tgamma((double)shape + 1.)), pow(1. / (double)scale, (double)shape) / tgamma((double)shape), nullptr, nullptr};
    return d;
}

// a / b for floats, given rb = RN_double(1 / (double)b):  RN_float(a / b) == (float)((double)a * rb) exactly
// (a float quotient is never within 2^-49 relative of a float rounding midpoint; the double product is within
// 2^-52 of a / b; 0, inf and NaN divisors behave as IEEE division).  Used for divisors that are uniform over
// a wavefront: three fp64-rate instructions instead of the IEEE f32 division expansion.
// Fast build: rb is the float reciprocal (rounded from the double one, or v_rcp_f32 + one Newton step): one multiply,
// <= 1 ulp.
GENDR_HD float div_by(float a, rcp_t rb)
{
#if GENDR_FAST_DEV
    return a * rb;
#else
    return (float)((double)a * rb);
#endif
}

// RN_double(n / b) for doubles, given rb = RN_double(1 / b) -- the CORRECTLY ROUNDED reciprocal (the host's division) -- without a
// division: q = n rb is within two ulps of the quotient; one residual correction (r = n - q b is exact in an fma) makes it
// faithful, and by Markstein's theorem a second one with a correctly rounded reciprocal yields the correctly rounded quotient
// (the `provably exact five-instruction form` of DESIGN 3.6, in double; 4e8 random operand pairs against the IEEE division on
// the host: tools/div5_check.c).  Five full-rate-f64 operations against the ~15 of the compiler's IEEE f64 division expansion
// (v_div_scale x2, v_rcp_f64, v_div_fmas, v_div_fixup and the Newton steps): the uniform and cubic CDFs' `x * 0.5 / scale`
// (kernel.cu:274, :283) is a double division by a call-wide constant in every pair of BASELINE config 2.  Operands here are
// finite and normal (the branch is taken for |u| < 1 only); host: the division itself.
GENDR_HD double div_rn_f64(double n, double b, double rb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double q = n * rb;
    double r = __builtin_fma(-q, b, n);
    q = __builtin_fma(r, rb, q);
    r = __builtin_fma(-q, b, n);
    return __builtin_fma(r, rb, q);
#else
    (void)rb;
    return n / b;
#endif
}

// Reciprocal of a positive, finite, normal double for use with div_by(): the argument above leaves 2^-49 - 2^-52 of
// slack, so rb may be off by a few ulps.  On the device: v_rcp_f64 and two Newton steps (error < 2 ulp, 5
// instructions) instead of the IEEE f64 division expansion (~15); on the host the true quotient.
GENDR_HD rcp_t rcp_for_div_by(float bf)
{
#if GENDR_FAST_DEV
    const float y = __builtin_amdgcn_rcpf(bf);
    return __builtin_fmaf(__builtin_fmaf(-bf, y, 1.f), y, y);
#elif defined(__HIP_DEVICE_COMPILE__)
    const double b = (double)bf;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
#else
    return 1. / (double)bf;
#endif
}

// ---- correctly rounded square root and reciprocal, short forms ----------------------------------------------------
// hipcc's expansions of sqrtf() and of 1.f / x are correctly rounded for EVERY input (denormals, huge values) and pay
// for it: 54 and 43 issue cycles per wave on MI355X (tools/micro/opbench.hip).  The pair math only ever sees
// arguments in the normal range, where the classic fused-multiply-add refinements of the hardware estimates
// (v_rsq_f32 / v_rcp_f32, 1 ulp) are enough: 8 and 5 instructions.  Their results equal sqrtf(x) and 1.f / x for
// every float in [2^-96, 2^96] -- verified EXHAUSTIVELY, all 1.6e9 bit patterns, by gendr_selftest() on the GPU
// (tests/test_gpu_exact_math.py); outside that range, and for 0 / inf / NaN, the compiler's expansion is used.
GENDR_HD float sqrt_rn(float x)
{
#if GENDR_FAST_DEV
    return __builtin_amdgcn_sqrtf(x);                        // v_sqrt_f32: 1 ulp
#elif defined(__HIP_DEVICE_COMPILE__)
    if (!(x >= 0x1p-96f && x <= 0x1p+96f)) return sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;                                         // ~ sqrt(x)
    float h = 0.5f * y;                                      // ~ 1 / (2 sqrt(x))
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);                // exact residual of a faithful g
    return __builtin_fmaf(d, h, g);
#else
    return sqrtf(x);
#endif
}
GENDR_HD float rcp_rn(float x)
{
#if GENDR_FAST_DEV
    // v_rcp_f32 (1 ulp), one Newton step, and v_div_fixup_f32 for what the step turns into NaN (0, inf): 1 / x as IEEE has it
    const float y0 = __builtin_amdgcn_rcpf(x);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-x, y0, 1.f), y0, y0), x, 1.f);
#elif defined(__HIP_DEVICE_COMPILE__)
    if (!(fabsf(x) >= 0x1p-96f && fabsf(x) <= 0x1p+96f)) return 1.f / x;
    float y = __builtin_amdgcn_rcpf(x);
    y = __builtin_fmaf(__builtin_fmaf(-x, y, 1.f), y, y);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.f), y, y);
#else
    return 1.f / x;
#endif
}

// ---- gradient-side arithmetic (backward kernel only) ----------------------------------------------------------
// The partials of a (pixel, face) pair are summed per face in an order that already differs from the reference's
// (and from run to run: float atomics), so a gradient can only be compared within a tolerance.  What feeds ONLY
// those sums is therefore computed to fp32 accuracy (<= 1 ulp per quotient) instead of with the reference's exact
// rounding: v_rcp_f32 plus one correction step instead of the IEEE division expansion (measured on MI355X: 15 vs 43
// issue cycles per wave; an exact double-reciprocal quotient costs 13 + 41 for its reciprocal).  Everything the
// backward pass RECOMPUTES from the forward pass -- barycentrics, distance, CDF, depth: whatever decides which pairs
// contribute -- keeps the forward kernel's exact arithmetic.  -DGENDR_EXACT_GRADIENT=1 restores the reference's
// rounding everywhere (diagnostic).
#ifndef GENDR_EXACT_GRADIENT
#define GENDR_EXACT_GRADIENT 0
#endif
// The forward-side forms of the gaussian and gamma families (norm_cdf, the power of gamma's CDF) and their densities: see
// norm_cdf().  GENDR_EXACT_CDF / GENDR_EXACT_PDF = 1: what the reference's kernel compiled for this platform calls.
#ifndef GENDR_EXACT_CDF
#define GENDR_EXACT_CDF GENDR_EXACT_GRADIENT
#endif
#ifndef GENDR_EXACT_PDF
#define GENDR_EXACT_PDF GENDR_EXACT_GRADIENT
#endif
GENDR_HD float grad_rcp(float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !GENDR_EXACT_GRADIENT
    float y = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y, 1.f), y, y);
#else
    return 1.f / b;
#endif
}
GENDR_HD float grad_div(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !GENDR_EXACT_GRADIENT
    const float y = __builtin_amdgcn_rcpf(b);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), y, q);
#else
    return a / b;
#endif
}

// exp of the pair math.  Fast build: exp(x) = 2^(x log2 e) with the product in two pieces -- ph, and pl = the rounding
// error of ph plus the tail of log2 e -- the hardware 2^ph (v_exp_f32, 1 ulp) and a first-order correction by pl: ~2 ulp
// for every argument the path meets (the plain 2^(x * log2e) loses |x| * 6e-8 relative), 8 instructions against the
// library's 47 issue cycles.  Arguments beyond the float exponent range (and +-inf) take the bare 2^ph: 0 resp. inf.
GENDR_HD float exp_f(float x)
{
#if GENDR_FAST_DEV
    const float ph = x * 0x1.715476p+0f;
    const float pl = __builtin_fmaf(x, 0x1.4ae0c0p-26f, __builtin_fmaf(x, 0x1.715476p+0f, -ph));
    const float e = __builtin_amdgcn_exp2f(ph);
    return __builtin_fabsf(ph) < 126.f ? __builtin_fmaf(e, pl * 0x1.62e430p-1f, e) : e;
#else
    return expf(x);
#endif
}
// a / b where the reference writes a float division whose divisor is NOT uniform (t-conorm folds, the colour normalisation):
// IEEE division in the parity builds; fast build: v_rcp_f32 and one correction step on the quotient (<= 1 ulp)
GENDR_HD float div_f(float a, float b)
{
#if GENDR_FAST_DEV
    const float y = __builtin_amdgcn_rcpf(b);
    const float q = a * y;
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-q, b, a), y, q), b, a);   // (0, inf, NaN operands: as IEEE division)
#else
    return a / b;
#endif
}

GENDR_HD float quiet_nan() { return __builtin_nanf(""); }

// normal CDF of a float argument (kernel.cu:293: `normcdf(sign * x / scale)` on a float).  CUDA resolves that call to its
// float overload; HIP has no normcdf(float), so the reference's kernel compiled for THIS platform (oracle/build_ref.py)
// promotes to normcdf(double) and rounds the result -- the pin.  Round 4's default build evaluated 0.5 erfcf(-u / sqrt 2) in
// float.  A last-bit difference of ANY fragment of a pixel -- also a tiny one of a face that only grazes it -- can move the last
// bit of the pixel's folded alpha A, and the reference's saturated partials ((1 - A^2) / (1 - D^2) with A one or two units of
// the last place below 1) turn that into a factor: BASELINE config 3 missed the flat 1e-5 against the reference's kernels on
// 2.5 percent of its face-gradient elements (max 1.3e-3), and a first round-5 attempt that made only the D >= 1/2 side exact
// still did.  Calling the library's normcdf(double) per pair fixes it and costs the forward kernel a factor 2.4 (measured:
// 0.194 -> 0.459 ms at config 3; ~1000 wave instructions per batch, all of the library's argument ranges under divergence).
// Device code of the default build instead, ONE branch-free path for |u| < 5.625:
//     Q(x) = Phi(-x) = e^(-x^2 / 2) g(x),   x = |u|,      D = u < 0 ? Q : 1 - Q,   rounded to float once,
//   * x^2 / 2 is exact in double (x is a float); e^y by y = k ln 2 + r, |r| <= ln 2 / 2, a degree-13 Taylor polynomial and
//     v_ldexp_f64;
//   * g(x) = Phi(-x) e^(x^2 / 2) (Mills' ratio over sqrt(2 pi): smooth, 0.5 ... 0.07) by one polynomial of degree 30 on
//     [0, 5.625] (Chebyshev interpolant converted to the monomial basis in 70-digit arithmetic, tools/normcdf_coef.py; Horner in
//     double is stable here).  Relative error of Q < 2^-50 against 60-digit values.
//   A float result differs from the library's double result rounded only where Phi lies within ~2^-50 (relative) of a
//   rounding midpoint: gendr_selftest(3) compares the two for EVERY float of [-6, 6] on the GPU
//   (tests/test_gpu_exact_math.py holds the count of differing inputs -- each by one unit of the last place).
//   * u >= 5.625: Q < 2^-25, exactly 1.  u <= -5.625: Phi < 1e-8, far below the 1e-6 at which the pair is skipped (:784):
//     the float form.
//   The 45 double constants are materialised in SCALAR registers where they are used (sconst(): a volatile asm the loop
//   optimiser cannot hoist): left to itself the compiler keeps them in 90 vector registers across the batch loop and spills
//   (measured: 30 scratch reloads per batch, forward 0.194 -> 0.401 ms).
// `exact` build (GENDR_EXACT_CDF): the library's double function rounded, for every argument.  Host: the float form.
#ifndef GENDR_NORMCDF_POLY
#define GENDR_NORMCDF_POLY 1
#endif
constexpr double kNormQEnd = 5.625;
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double sconst(double c) { asm volatile("" : "+s"(c)); return c; }
#else
inline double sconst(double c) { return c; }
#endif
GENDR_HD double norm_q(double x)             // Q(x) = Phi(-x) on [0, kNormQEnd], relative error < 2^-50
{
    constexpr double g[31] = {
        0.12830188067911721, -0.10713724028835606, 0.083707806232990997,
        -0.061776828075667381, 0.043369365984936591, -0.029121180260832224,
        0.018784229226290805, -0.011680985402171174, 0.0070234710631495905,
        -0.0040935184923098222, 0.0023176382985891928, -0.0012770442848394634,
        0.00068593819084243537, -0.00035967374710264654, 0.00018434426003539211,
        -9.245390998209764e-05, 4.5424960960400369e-05, -2.1896313100973583e-05,
        1.0350492914507299e-05, -4.7799949982060067e-06, 2.1831155246446762e-06,
        -1.0166169632747425e-06, 4.4682677084580456e-07, -1.5632209228437453e-07,
        6.8643198950753235e-08, -5.6374486597669725e-08, 2.2602849241977347e-08,
        3.8511174054904546e-09, -1.29523774689231e-09, -3.2409135017486482e-09,
        1.2443686235586019e-09,
    };
    constexpr double kInvFact[14] = {1., 1., 1. / 2, 1. / 6, 1. / 24, 1. / 120, 1. / 720, 1. / 5040, 1. / 40320, 1. / 362880,
                                     1. / 3628800, 1. / 39916800, 1. / 479001600, 1. / 6227020800.};
    const double y = -0.5 * (x * x);                                     // exact
    const double k = __builtin_rint(y * sconst(1.4426950408889634));     // log2(e)
    double r = __builtin_fma(-k, sconst(0.6931471805599453094), y);
    r = __builtin_fma(-k, sconst(2.3190468138462996e-17), r);
    double e = sconst(kInvFact[13]);
#pragma unroll
    for (int n = 12; n >= 0; n--) e = __builtin_fma(e, r, sconst(kInvFact[n]));
    const double t = __builtin_fma(x, sconst(2. / kNormQEnd), -1.);
    double q = sconst(g[30]);
#pragma unroll
    for (int n = 29; n >= 0; n--) q = __builtin_fma(q, t, sconst(g[n]));
    return __builtin_ldexp(e, (int)k) * q;
}
// Round 6: the same Q(x) = e^(-x^2 / 2) g(x) from TABLES (tools/normcdf_table.py) -- what the kernels specialised for the gaussian distribution
// (BASELINE config 3) evaluate: 18 double FMAs and 7 LDS reads per pair instead of 43 FMAs on scalar constants:
//   * g on sixteen intervals of width 45/128, one polynomial of degree 10 in t = x - centre per interval (Chebyshev interpolants in
//     60-digit arithmetic on intervals widened by 2^-12: the interval index comes from a rounded product);
//   * e^y, y = -x^2 / 2 (exact): y = k ln2/16 + r, |r| <= ln2/32, e^r by a degree-7 Taylor polynomial, 2^(k/16) = 2^(k >> 4) T[k & 15].
// Relative error of Q < 2^-51 against 60-digit values (20 000 random float arguments and the interval ends).  Row i of the table:
// the eleven coefficients of interval i, constant term first, then T[i] = 2^(i/16).  The kernels copy it into LDS (norm_table()).
constexpr int kNormRow = 12, kNormRows = 16;
constexpr double kNormLn2_16Hi = 0.043321698784993146, kNormLn2_16Lo = 3.436201886692732e-15, kNorm16_Ln2 = 23.083120654223414;
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ const double kNormTab[kNormRows][kNormRow] = {
    {0.43693140490055954, -0.32213793188376272, 0.1901527982808108, -0.096237545099447847, 0.043309010575230816, -0.017724927379106997, 0.0066988836797126543, -0.002363856377000447, 0.00078540348898180638, -0.00024913453600126088, 7.4695486193459816e-05, 1},
    {0.34357789834435598, -0.21775862307140328, 0.11437212472952392, -0.052481732639003513, 0.021674052757761009, -0.0082104115210894588, 0.0028907240119249791, -0.0009551256301397116, 0.00029837461870584688, -8.9228541337659114e-05, 2.5316536962902442e-05, 1.0442737824274138},
    {0.2791695527198545, -0.15357841570624881, 0.07209426164526761, -0.030071439518569408, 0.011416071376259407, -0.0040075566904269316, 0.0013156341585925701, -0.00040731349638098733, 0.00011970326110734182, -3.3766621964396342e-05, 9.0697235090130877e-06, 1.0905077326652577},
    {0.23293616803301631, -0.11232160489205739, 0.047363971631746507, -0.0180139059742708, 0.0062996058161522497, -0.0020524876058942696, 0.00062901400608775899, -0.00018264143885079163, 5.0534065102882098e-05, -1.3455971607177274e-05, 3.4234066084388384e-06, 1.1387886347566916},
    {0.19860566115621259, -0.084741918025393326, 0.032270649327551124, -0.011229580777076588, 0.0036262754034156242, -0.0010985399647981786, 0.00031472514712291001, -8.5804160884025307e-05, 2.2372183424093617e-05, -5.6280861495651065e-06, 1.3571606235107387e-06, 1.189207115002721},
    {0.17236420204051464, -0.065659936612156383, 0.022702279490926479, -0.0072543169592247632, 0.0021688443894955171, -0.00061213060503208694, 0.00016420541528683062, -4.2088820489151983e-05, 1.0352692086229128e-05, -2.4629396042218772e-06, 5.6336963705045865e-07, 1.241857812073484},
    {0.15180146102454875, -0.052052222982053711, 0.016426999175357559, -0.0048379877159034157, 0.0013428603272265737, -0.00035386841108253576, 8.903595373205243e-05, -2.1486628269413688e-05, 4.9918900104448469e-06, -1.1243078650330186e-06, 2.4416339756756753e-07, 1.2968395546510096},
    {0.13534071474962242, -0.042086880182701666, 0.012184724321444768, -0.0033197297002514016, 0.00085788269396129655, -0.00021154686394275852, 5.0015519048947921e-05, -1.1381371451677492e-05, 2.5007249308868635e-06, -5.3384310783587177e-07, 1.1017619694594463e-07, 1.3542555469368927},
    {0.12191833961535134, -0.034615992097746059, 0.0092380097397543135, -0.0023367402683716837, 0.00056379315241300164, -0.00013039355277858674, 2.9023424160954891e-05, -6.2376024033051704e-06, 1.2979496944962209e-06, -2.6294690147846116e-07, 5.1626586521538736e-08, 1.4142135623730951},
    {0.11079686535493637, -0.028898062126156907, 0.0071409265878897583, -0.0016828276974605932, 0.00038013625504886711, -8.2646400540019955e-05, 1.7351698577003854e-05, -3.5277649065932105e-06, 6.9618221508824576e-07, -1.3401389873973592e-07, 2.5058836534341368e-08, 1.4768261459394993},
    {0.10145225353860561, -0.024440797612439287, 0.0056156702385310774, -0.0012370257986617963, 0.00026232636848445032, -5.3734520577578675e-05, 1.0661737291786573e-05, -2.0539537988383873e-06, 3.849662868620553e-07, -7.0507052268039593e-08, 1.2570119502668261e-08, 1.5422108254079407},
    {0.093503847674904314, -0.020909146247034379, 0.0044844114044822678, -0.00092627035885610822, 0.00018488232239368459, -3.5759381436715371e-05, 6.7180434865172664e-06, -1.2283602322428714e-06, 2.1897577814280515e-07, -3.8208566445642432e-08, 6.5022433735953418e-09, 1.6104903319492543},
    {0.08666962815403631, -0.01806989105264031, 0.0036304636195565341, -0.00070523507483689516, 0.00013282151114729152, -2.4309358705199489e-05, 4.3322124590747322e-06, -7.5304364529789054e-07, 1.2786630477786665e-07, -2.1283726977290887e-08, 3.4613881253808612e-09, 1.681792830507429},
    {0.080736905972542697, -0.015757355570810094, 0.0029755095906966182, -0.00054510269977988183, 9.7100268540722769e-05, -1.6851144437364709e-05, 2.8538595507033227e-06, -4.723505915438799e-07, 7.6504355292875934e-08, -1.2163888929344312e-08, 1.8926924247723499e-09, 1.7562521603732995},
    {0.075542684520175846, -0.013851642515380009, 0.0024658862394416133, -0.00042713403836712731, 7.2125934792878444e-05, -1.1892163222149815e-05, 1.9172957750680814e-06, -3.0263508156577673e-07, 4.682045083025285e-08, -7.1199425609461804e-09, 1.061183720129842e-09, 1.8340080864093424},
    {0.070960222219157085, -0.012264506980635265, 0.0020641204103867584, -0.00033888777936599807, 5.4361692229899524e-05, -8.5318053600651057e-06, 1.311669751600357e-06, -1.9774691139247489e-07, 2.926276228191101e-08, -4.2614744957419819e-09, 6.0908187955212908e-10, 1.9152065613971474},
};
#endif
#ifndef GENDR_NORMCDF_TAB
#define GENDR_NORMCDF_TAB 1
#endif
GENDR_HD double norm_q_tab(double x, const double* tab)      // Q(x) = Phi(-x) on [0, kNormQEnd]; tab: kNormTab in LDS
{
    const double y = -0.5 * (x * x);                                     // exact
    const double k = __builtin_rint(y * sconst(kNorm16_Ln2));
    double r = __builtin_fma(-k, sconst(kNormLn2_16Hi), y);              // (k * hi is exact: hi has 43 significant bits, |k| < 2^9)
    r = __builtin_fma(-k, sconst(kNormLn2_16Lo), r);
    double e = sconst(1. / 5040);
    e = __builtin_fma(e, r, sconst(1. / 720));
    e = __builtin_fma(e, r, sconst(1. / 120));
    e = __builtin_fma(e, r, sconst(1. / 24));
    e = __builtin_fma(e, r, sconst(1. / 6));
    e = __builtin_fma(e, r, 0.5);
    e = __builtin_fma(e, r, 1.);
    e = __builtin_fma(e, r, 1.);
    const int ki = (int)k;
    int i = (int)(x * sconst(128. / 45.));
    i = i > kNormRows - 1 ? kNormRows - 1 : i;
    const double* row = tab + i * kNormRow;
    const double t = x - (double)(2 * i + 1) * (45. / 256.);
    double q = row[10];
#pragma unroll
    for (int n = 9; n >= 0; n--) q = __builtin_fma(q, t, row[n]);
    return __builtin_ldexp(e * tab[(ki & 15) * kNormRow + 11], ki >> 4) * q;
}
GENDR_HD float norm_cdf(float u);
// norm_cdf() for the kernels that hold the table in LDS (`tab`, see norm_table() in gendr_kernels.h)
GENDR_HD float norm_cdf_tab(float u, const double* tab)
{
#if defined(__HIP_DEVICE_COMPILE__) && !GENDR_EXACT_CDF && GENDR_NORMCDF_POLY && GENDR_NORMCDF_TAB
    if (u >= (float)kNormQEnd) return 1.f;
    if (u > -(float)kNormQEnd) {
        const double q = norm_q_tab((double)__builtin_fabsf(u), tab);
        return (float)(u < 0.f ? q : 1. - q);
    }
    return 0.5f * erfcf(-u * 0.70710678118654752440f);       // (NaN ends up here and stays NaN)
#else
    (void)tab;
    return norm_cdf(u);
#endif
}
GENDR_HD float norm_cdf(float u)
{
#if defined(__HIP_DEVICE_COMPILE__) && GENDR_EXACT_CDF
    return (float)normcdf((double)u);
#elif defined(__HIP_DEVICE_COMPILE__) && GENDR_NORMCDF_POLY
    if (u >= (float)kNormQEnd) return 1.f;
    if (u > -(float)kNormQEnd) {
        const double q = norm_q((double)__builtin_fabsf(u));
        return (float)(u < 0.f ? q : 1. - q);
    }
    return 0.5f * erfcf(-u * 0.70710678118654752440f);       // (NaN ends up here and stays NaN)
#else
    return 0.5f * erfcf(-u * 0.70710678118654752440f);
#endif
}

// shifted abscissa of the one-sided families; `rev` mirrors it (kernel.cu:301-308, :339-345, :351-357)
template <bool REV>
GENDR_HD float shifted(float sign, float x, const DistParams& d)
{
    return REV ? -(sign * x - d.shift * d.scale) : (sign * x + d.shift * d.scale);
}

// ------------------------------------------------------------------------------------------
// CDF.  One specialisation per distribution id so a kernel templated on the id carries only
// its own branch; cdf_rt() below is the runtime-id dispatcher used by the generic kernels.
// ------------------------------------------------------------------------------------------
template <int ID> struct Dist;

template <> struct Dist<kHeaviside> {
    static GENDR_HD float cdf(float sign, float, const DistParams&) { return sign > 0 ? 1.f : 0.f; }   // :251-252
    static GENDR_HD float pdf(float, float, const DistParams&) { return 0.f; }                          // :375-376
};

template <> struct Dist<kUniform> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :270-277
        const float u = div_by(sign * x, d.rscale);
        if (u < -1) return 0.f;
#if GENDR_FAST_DEV
        if (u < 1) return __builtin_fmaf(0.5f, u, 0.5f);
#else
        if (u < 1) return (float)(div_rn_f64((double)(sign * x) * 0.5, (double)d.scale, d.rscale) + 0.5);
#endif
        return 1.f;
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :391-392
        const float u = div_by(sign * x, d.rscale);
        return (u > -1 && u < 1) ? div_by(0.5f, d.rscale) : 0.f;
    }
};

template <> struct Dist<kCubicHermite> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :282-290
        const float u = div_by(sign * x, d.rscale);
        if (u < -1) return 0.f;
        if (u < 1) {
#if GENDR_FAST_DEV
            const float y = (float)((double)(sign * x) * 0.5 / (double)d.scale + 0.5);
#else
            const float y = (float)(div_rn_f64((double)(sign * x) * 0.5, (double)d.scale, d.rscale) + 0.5);
#endif
            return 3 * y * y - 2 * y * y * y;
        }
        return 1.f;
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :397-402
        const float u = div_by(sign * x, d.rscale);
        if (u < -1.f || u > 1.f) return 0.f;
        return (float)(0.75 / (double)d.scale - 0.75 * (double)(x * x) / pow((double)d.scale, 3.));
    }
};

template <> struct Dist<kWigner> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :320-327
        const float u = sign * x / d.scale;
        if (u < -1) return 0.f;
        if (u < 1)
            return (float)(0.5 + (double)(sign * x * sqrtf(d.scale * d.scale - x * x)) / (kPi * (double)d.scale * (double)d.scale)
                               + (double)asinf(u) / kPi);
        return 1.f;
    }
    static GENDR_HD float pdf(float, float x, const DistParams& d) {                                     // :425-427
        if (x / d.scale > 1) return 0.f;
        return (float)(2. / kPi / (double)d.scale / (double)d.scale * (double)sqrtf(d.scale * d.scale - x * x));
    }
};

template <> struct Dist<kGaussian> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) { return norm_cdf(div_by(sign * x, d.rscale)); }   // :292-293
    static GENDR_HD float pdf(float, float x, const DistParams& d) {                                     // :404-405 (exp in double)
#if defined(__HIP_DEVICE_COMPILE__) && !GENDR_EXACT_PDF
        // gradient side (see grad_div): the density to fp32 accuracy, exp in float instead of double
        const float q = div_by(x, d.rscale);
        return (float)(d.rscale * 0.3989422804014327) * exp_f(-0.5f * q * q);
#else
        const double q = (double)div_by(x, d.rscale);
        return (float)(1. / (double)d.scale / sqrt(2. * kPi) * exp(-0.5 * q * q));
#endif
    }
};

template <> struct Dist<kLaplace> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :263-268
        const float e = 0.5f * exp_f(div_by(-x, d.rscale));  // 0.5 * e is exact in either precision
        return sign < 0 ? e : 1.f - e;
    }
    static GENDR_HD float pdf(float, float x, const DistParams& d) {                                     // :388-389
        return (float)(0.5 / (double)d.scale * (double)exp_f(div_by(-x, d.rscale)));
    }
};

template <> struct Dist<kLogistic> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :254-255
#if GENDR_FAST_DEV
        return rcp_rn(1.f + exp_f(div_by(-sign * x, d.rscale)));
#else
        return (float)(1. / (1. + (double)exp_f(div_by(-sign * x, d.rscale))));
#endif
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :378-380
        const float y = cdf(sign, x, d);
        return div_by(y * (1 - y), d.rscale);
    }
};

template <> struct Dist<kGudermannian> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :279-280
        return (float)(atan(tanh((double)(sign * x / d.scale) / 2.)) * 2. / kPi + 0.5);
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :394-395
        return (float)(1. / (double)coshf(sign * x / d.scale) / kPi / (double)d.scale);
    }
};

template <> struct Dist<kCauchy> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :257-258
        return (float)((double)atanf(sign * x / d.scale) / kPi + 0.5);
    }
    static GENDR_HD float pdf(float, float x, const DistParams& d) {                                     // :382-383
        return (float)(1. / (kPi * (double)d.scale + kPi / (double)d.scale * (double)x * (double)x));
    }
};

template <> struct Dist<kReciprocal> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {                                // :260-261
        return sign * x / d.scale / (1 + x / d.scale) * 0.5f + 0.5f;   // "/2. + 0.5": exact halving, one rounding
    }
    static GENDR_HD float pdf(float, float x, const DistParams& d) {                                     // :385-386
        const double s = (double)(d.scale + x);
        return (float)((double)d.scale / (2. * s * s));
    }
};

template <> struct Dist<kGumbelMax> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) { return exp_f(-exp_f(-sign * x / d.scale)); }   // :329-331
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :429-430
        const float u = sign * x / d.scale;
        return exp_f(-(u + exp_f(-u))) / d.scale;
    }
};

template <> struct Dist<kGumbelMin> {
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) { return 1.f - exp_f(-exp_f(sign * x / d.scale)); }   // :333-335
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // :432-433
        return exp_f(-((-sign * x / d.scale) + exp_f(sign * x / d.scale))) / d.scale;
    }
};

template <bool REV> struct ExponentialFamily {                                                           // :349-359, :446-455
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {
        if (!REV) { if (sign * x + d.shift * d.scale < 0.f) return 0.f; }
        else      { if (sign * x - d.shift * d.scale > 0.f) return 1.f; }
        const float xs = shifted<REV>(sign, x, d);
        const float y = 1.f - exp_f(div_by(-xs, d.rscale));
        return REV ? 1.f - y : y;
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {
        if (!REV) { if (sign * x + d.shift * d.scale < 0.f) return 0.f; }
        else      { if (sign * x - d.shift * d.scale > 0.f) return 0.f; }
        const float xs = shifted<REV>(sign, x, d);
        return (float)(1. / (double)d.scale * (double)exp_f(div_by(-xs, d.rscale)));
    }
};
template <> struct Dist<kExponential> : ExponentialFamily<false> {};
template <> struct Dist<kExponentialRev> : ExponentialFamily<true> {};

template <bool REV> struct GammaFamily {                                                                 // :295-319, :407-423
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {
        if (d.shape < 0.f) return quiet_nan();
        if (!REV) { if (sign * x + d.shift * d.scale <= 0.f) return 0.f; }
        else      { if (sign * x - d.shift * d.scale >= 0.f) return 1.f; }
        const float xs = shifted<REV>(sign, x, d);
        const float xr = div_by(xs, d.rscale);               // xs / scale, used 34 times below
        if ((double)xr > kGammaCut) return REV ? 0.f : 1.f;
        float kummers = d.gamma_k0;                          // (float)(1. / tgamma(shape + 1.))
        float factor = kummers;
        for (int i = 1; i < kGammaSteps; i++) {              // 32-term Kummer series, float
            // xr / (shape + i): the 31 divisors are the same for every pair -> exact quotients by their reciprocals
            factor *= d.gamma_r ? div_by(xr, d.gamma_r[i - 1]) : xr / (d.shape + i);
            kummers += factor;
        }
#if GENDR_FAST_DEV
        // fast variant only: shape 1 and 2 without a power
        const float xp = d.shape == 2.f ? xr * xr : (d.shape == 1.f ? xr : powf(xr, d.shape));
#else
        // powf for every shape, as the reference's kernel calls it (:309).  Round 4's default build wrote xr * xr for shape 2 (the
        // correctly rounded square, which the library's powf only reaches to within an ulp): that last bit, through gamma_rev's
        // 1 - y and the saturated t-conorm partials, kept BASELINE config 5 from a flat 1e-5 against the reference's kernels on
        // 15 percent of its face-gradient elements (max 4.7e-3).  Measured cost of the call at config 5: forward +5 %, backward +6 %.
        const float xp = powf(xr, d.shape);
#endif
        const float y = xp * exp_f(div_by(-xs, d.rscale)) * kummers;
        return REV ? 1.f - y : y;
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {                                // explicit double in the reference
        if (d.shape < 0.f) return quiet_nan();
#if defined(__HIP_DEVICE_COMPILE__) && !GENDR_EXACT_PDF
        // gradient side (see grad_div): shape 1 and 2 (2: BASELINE config 5) need no power -- xs^(shape - 1) is 1 resp.
        // xs -- and the density to fp32 accuracy only a float exponential, instead of ~400 double-precision
        // instructions per pair.  Other shapes keep the double evaluation (a float power would underflow where the
        // constant in front is large).
        if (d.shape == 2.f || d.shape == 1.f) {
            if (!REV) { if (sign * x + d.shift * d.scale <= 0.f) return 0.f; }
            else      { if (sign * x - d.shift * d.scale >= 0.f) return 0.f; }
            const float xf = shifted<REV>(sign, x, d);
            return (float)d.gamma_pdf_c * (d.shape == 2.f ? xf : 1.f) * exp_f(div_by(-xf, d.rscale));
        }
#endif
        double xs;
        if (!REV) {
            if (sign * x + d.shift * d.scale <= 0.f) return 0.f;
            xs = (double)sign * (double)x + (double)d.shift * (double)d.scale;
        } else {
            if (sign * x - d.shift * d.scale >= 0.f) return 0.f;
            xs = -((double)sign * (double)x - (double)d.shift * (double)d.scale);
        }
        return (float)(d.gamma_pdf_c * pow(xs, (double)d.shape - 1.) * exp(-xs / (double)d.scale));
    }
};
template <> struct Dist<kGamma> : GammaFamily<false> {};
template <> struct Dist<kGammaRev> : GammaFamily<true> {};

template <bool REV> struct LevyFamily {                                                                  // :337-347, :435-444
    static GENDR_HD float cdf(float sign, float x, const DistParams& d) {
        if (!REV) { if ((double)(sign * x + d.shift * d.scale) <= 1e-6) return 0.f; }
        else      { if ((double)(sign * x - d.shift * d.scale) >= -1e-6) return 1.f; }
        const float xs = shifted<REV>(sign, x, d);
        const float y = (float)erfc(sqrt((double)d.scale / 2. / (double)xs));
        return REV ? 1.f - y : y;
    }
    static GENDR_HD float pdf(float sign, float x, const DistParams& d) {
        if (!REV) { if ((double)(sign * x + d.shift * d.scale) <= 1e-6) return 0.f; }
        else      { if ((double)(sign * x - d.shift * d.scale) >= -1e-6) return 0.f; }
        const float xs = shifted<REV>(sign, x, d);
        return (float)(sqrt((double)d.scale / 2. / kPi) * exp((double)(-d.scale) / 2. / (double)xs) / pow((double)xs, 1.5));
    }
};
template <> struct Dist<kLevy> : LevyFamily<false> {};
template <> struct Dist<kLevyRev> : LevyFamily<true> {};

#define GENDR_FOR_EACH_DIST(X) \
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17)

GENDR_HD float cdf_rt(int id, float sign, float x, const DistParams& d)
{
    switch (id) {
#define X(i) case i: return Dist<i>::cdf(sign, x, d);
        GENDR_FOR_EACH_DIST(X)
#undef X
        default: return quiet_nan();
    }
}
GENDR_HD float pdf_rt(int id, float sign, float x, const DistParams& d)
{
    switch (id) {
#define X(i) case i: return Dist<i>::pdf(sign, x, d);
        GENDR_FOR_EACH_DIST(X)
#undef X
        default: return quiet_nan();
    }
}

// "Light" distributions: everything except the ones whose device code needs long double-precision libm
// expansions (gudermannian: tanh/atan in double; gamma: tgamma + 32-term series + pow; levy: erfc/exp/pow in
// double).  Runtime-dispatch kernels exist in a light and a full flavour so that the heavy branches do not
// dictate the register allocation (hence the occupancy) of every non-specialised option set.
#define GENDR_FOR_EACH_LIGHT_DIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(8) X(9) X(10) X(11) X(12) X(13)
GENDR_HD bool is_light_dist(int id) { return id >= 0 && id <= 13 && id != 7; }

GENDR_HD float cdf_light_rt(int id, float sign, float x, const DistParams& d)
{
    switch (id) {
#define X(i) case i: return Dist<i>::cdf(sign, x, d);
        GENDR_FOR_EACH_LIGHT_DIST(X)
#undef X
        default: return quiet_nan();
    }
}
GENDR_HD float pdf_light_rt(int id, float sign, float x, const DistParams& d)
{
    switch (id) {
#define X(i) case i: return Dist<i>::pdf(sign, x, d);
        GENDR_FOR_EACH_LIGHT_DIST(X)
#undef X
        default: return quiet_nan();
    }
}

// ------------------------------------------------------------------------------------------
// t-conorms.  fold(a, b, p): alpha <- T(alpha, D_f) (kernel.cu:474-563);
// grad(A, b, p): d alpha_final / d D_f from the final alpha A (kernel.cu:567-614).
// ------------------------------------------------------------------------------------------
template <int ID> struct TConorm;

template <> struct TConorm<kMax> {
    static GENDR_HD float fold(float a, float b, float) { return fmaxf(a, b); }                          // :481-482
    static GENDR_HD float grad(float A, float b, float) { return A == b ? 1.f : 0.f; }                   // :574-575
};
template <> struct TConorm<kProbabilistic> {
    static GENDR_HD float fold(float a, float b, float) { return a + b - a * b; }                        // :484-485
    static GENDR_HD float grad(float A, float b, float) {                                                // :577-578
        return (float)((1. - (double)A) / fmax(1. - (double)b, 1e-6));
    }
    static GENDR_HD float grad_fp32(float A, float b, float) { return grad_div(1.f - A, fmaxf(1.f - b, 1e-6f)); }
};
template <> struct TConorm<kEinstein> {
    static GENDR_HD float fold(float a, float b, float) { return div_f(a + b, 1 + a * b); }              // :487-488
    static GENDR_HD float grad(float A, float b, float) {                                                // :580-581
        return (float)((1. - (double)(A * A)) / fmax(1. - (double)(b * b), 1e-6));
    }
    static GENDR_HD float grad_fp32(float A, float b, float) { return grad_div(1.f - A * A, fmaxf(1.f - b * b, 1e-6f)); }
};
template <> struct TConorm<kHamacher> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :490-498
        if (p < 0.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
        const float c = (float)((double)(a * b) / fmax((double)p + (1. - (double)p) * (double)(a + b - a * b), 1e-6));
        return 1.f - c;
    }
    static GENDR_HD float grad(float A_, float b_, float p_) {                                           // :583-584
        const double A = A_, b = b_, p = p_;
        return (float)((1.0 - A) * (-A - p * (1.0 - A) + p + 1.0) / fmax((1.0 - b) * (-b - p * (1.0 - b) + p + 1.0), 1e-6));
    }
};
template <> struct TConorm<kFrank> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :500-509
        if (p <= 0.f || p == 1.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
        const float c = (float)(log1p(((double)powf(p, a) - 1.) * ((double)powf(p, b) - 1.) / ((double)p - 1.)) / (double)logf(p));
        return 1.f - c;
    }
    static GENDR_HD float grad(float A, float b, float p) {                                              // :586-588
        const float d = (float)(pow((double)p, 1.0 - (double)b) - 1.0);
        return (float)((double)powf(p, A - b) * (pow((double)p, 1.0 - (double)A) - 1.0) / ((double)d + copysign(1e-6, (double)d)));
    }
};
template <> struct TConorm<kYager> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :511-519
        if (p <= 0.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
#if GENDR_FAST_DEV
        if (p == 2.f) { const float fa = 1.f - a, fb = 1.f - b; return 1.f - fmaxf(0.f, 1.f - sqrt_rn(fa * fa + fb * fb)); }
#endif
        const double xa = 1. - (double)a, xb = 1. - (double)b;
        // p = 2 (the setting of the reference's benchmark table, train_reconstruction.py:551): pow(x, 2.) is x * x and
        // pow(s, .5) is sqrt(s) up to the last bit of a double -- two multiplies and a square root instead of three pow()
        const float c = p == 2.f ? (float)fmax(0., 1. - sqrt(xa * xa + xb * xb))
                                 : (float)fmax(0., 1. - pow(pow(xa, (double)p) + pow(xb, (double)p), 1. / (double)p));
        return 1.f - c;
    }
    static GENDR_HD float grad(float A, float b, float p) {                                              // :590-592
        if (A == 1.f) return 0.f;
#if GENDR_FAST_DEV
        if (p == 2.f) return b * rcp_rn(A);
#endif
        if (p == 2.f) return (float)((double)b * (1. / (double)A));            // pow(b, 1.) * pow(A, -1.)
        return (float)(pow((double)b, (double)p - 1.) * pow((double)A, 1. - (double)p));
    }
};
template <> struct TConorm<kAczelAlsina> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :521-531
        if (p <= 0.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
        if ((double)a < 1e-8 || (double)b < 1e-8) return 1.f;
        const float c = (float)exp(-pow((double)(powf(-logf(a), p) + powf(-logf(b), p)), 1. / (double)p));
        return 1.f - c;
    }
    static GENDR_HD float grad(float A_, float b_, float p_) {                                           // :594-598
        const double A = A_, b = b_, p = p_;
        return (float)((1. - A) * pow(-log1p(fmax(-b, -1. + 1e-6)), p - 1.) * pow(-log1p(fmax(-A, -1. + 1e-6)), 1. - p)
                       / fmax(1. - b, 1e-6));
    }
};
template <> struct TConorm<kDombi> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :533-549
        if (p <= 0.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
        if ((double)a < 1e-8 || (double)b < 1e-8) return 1.f;
        const float c = (float)(1. / (1. + pow(pow((1. - (double)a) / (double)a, (double)p)
                                               + pow((1. - (double)b) / (double)b, (double)p), 1. / (double)p)));
        return 1.f - c;
    }
    static GENDR_HD float grad(float A_, float b_, float p_) {                                           // :600-604
        const double A = A_, b = b_, p = p_;
        return (float)((1. - A) * (1. - A) * pow(b / fmax(1. - b, 1e-6), p - 1.) * pow(A / fmax(1. - A, 1e-6), 1. - p)
                       / fmax(1. - b, 1e-6) / fmax(1. - b, 1e-6));
    }
};
template <> struct TConorm<kSchweizerSklar> {
    static GENDR_HD float fold(float a_ex, float b_new, float p) {                                       // :551-559
        if (p >= 0.f) return quiet_nan();
        const float a = 1.f - a_ex, b = 1.f - b_new;
        const float c = (float)pow((double)(powf(a, p) + powf(b, p)) - 1., 1. / (double)p);
        return 1.f - c;
    }
    static GENDR_HD float grad(float A, float b, float p) {                                              // :606-610
        const float a1 = (float)fmax(1. - (double)A, 1e-6);
        const float b1 = (float)fmax(1. - (double)b, 1e-6);
        const double pd = p;
        return (float)(pow((double)b1, pd - 1.)
                       * pow((double)powf(b1, p) + pow(pow((double)(-powf(b1, p) + powf(a1, p)) + 1., 1. / pd), pd) - 1., (1. - pd) / pd));
    }
};

#define GENDR_FOR_EACH_TCONORM(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)

GENDR_HD float tconorm_fold_rt(int id, float a, float b, float p)
{
    switch (id) {
#define X(i) case i: return TConorm<i>::fold(a, b, p);
        GENDR_FOR_EACH_TCONORM(X)
#undef X
        default: return quiet_nan();
    }
}
GENDR_HD float tconorm_grad_rt(int id, float A, float b, float p)
{
    switch (id) {
#define X(i) case i: return TConorm<i>::grad(A, b, p);
        GENDR_FOR_EACH_TCONORM(X)
#undef X
        default: return quiet_nan();
    }
}

// light t-conorms: max, probabilistic, einstein, hamacher (no pow / log)
GENDR_HD bool is_light_alpha(int id) { return id >= 0 && id <= 4; }
GENDR_HD float tconorm_fold_light_rt(int id, float a, float b, float p)
{
    switch (id) {
        case 1: return TConorm<1>::fold(a, b, p);
        case 2: return TConorm<2>::fold(a, b, p);
        case 3: return TConorm<3>::fold(a, b, p);
        case 4: return TConorm<4>::fold(a, b, p);
        default: return quiet_nan();
    }
}
GENDR_HD float tconorm_grad_light_rt(int id, float A, float b, float p)
{
    switch (id) {
        case 1: return TConorm<1>::grad(A, b, p);
        case 2: return TConorm<2>::grad(A, b, p);
        case 3: return TConorm<3>::grad(A, b, p);
        case 4: return TConorm<4>::grad(A, b, p);
        default: return quiet_nan();
    }
}

}  // namespace gendr
