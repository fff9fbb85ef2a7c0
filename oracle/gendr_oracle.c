/*
 * gendr_oracle.c -- CPU oracle (test infrastructure only; pinned to the reference's
 * own kernels, see gendr_oracle.h).  Instantiates gendr_oracle_body.inc for float and
 * double, the two types AT_DISPATCH_FLOATING_TYPES generates in the reference
 * (kernel.cu:1102,1117,1189), and exports the scalar functions the reference
 * exports through pybind (generalized_renderer_cuda.cpp:233-236).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp, no -march).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "gendr_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- libm jitter (sensitivity analysis of the parity tests only) ---------------------------------
 * The float instantiation calls its single-precision libm functions through these wrappers.  With
 * g_libm_jitter == 0 (always, except inside tests/criteria.py's sensitivity runs) they return the
 * library's value unchanged.  With +1 / -1 every result is moved one ulp up / down: what a different
 * but equally valid libm (the GPU's: <= 1-2 ulp, glibc: <= 1 ulp) can do to every element of the
 * output.  Correctly rounded operations (+ - * / sqrt) are never touched. */
static int g_libm_jitter = 0;
void gendr_oracle_set_libm_jitter(int j) { g_libm_jitter = j > 7 ? 7 : (j < -7 ? -7 : j); }
static inline float jit(float v, float arg)
{
    if (g_libm_jitter == 0 || !(v == v) || v == INFINITY || v == -INFINITY) return v;
    int up = g_libm_jitter > 0;
    if (g_libm_jitter >= 2 || g_libm_jitter <= -2) {
        /* +-2 ... +-7: the direction is one of six bits of a hash of the result's bit pattern (and its complement): neighbouring pairs move
         * AGAINST each other, which is what changes a ratio of two fragments (softmax weights) */
        unsigned u, a;                     /* result AND argument: equal results of different pairs still move apart */
        memcpy(&u, &v, sizeof u);
        memcpy(&a, &arg, sizeof a);
        u = (u ^ (a * 0x9e3779b9u)) * 2654435761u;
        up ^= (int)((u >> (13 + (g_libm_jitter > 0 ? g_libm_jitter : -g_libm_jitter))) & 1u);
    }
    return nextafterf(v, up ? INFINITY : -INFINITY);
}
static inline float j_expf(float x) { return jit(expf(x), x); }
static inline float j_logf(float x) { return jit(logf(x), x); }
static inline float j_powf(float x, float y) { return jit(powf(x, y), x + y); }
static inline float j_asinf(float x) { return jit(asinf(x), x); }
static inline float j_coshf(float x) { return jit(coshf(x), x); }
static inline float j_erfcf(float x) { return jit(erfcf(x), x); }
static inline float j_atanf(float x) { return jit(atanf(x), x); }

/* ---- float instantiation ------------------------------------------------ */
#define S float
#define FN(name) name##_f32
#define S_exp j_expf
#define S_log j_logf
#define S_pow j_powf
#define S_sqrt sqrtf
#define S_asin j_asinf
#define S_cosh j_coshf
#define S_atanf j_atanf
/* CUDA normcdff(x); restated as erfc form (published definition of the normal CDF) */
#define S_ncdf(x) (0.5f * j_erfcf(-(x) * 0.70710678118654752440f))
#include "gendr_oracle_body.inc"
#undef S
#undef FN
#undef S_exp
#undef S_log
#undef S_pow
#undef S_sqrt
#undef S_asin
#undef S_cosh
#undef S_ncdf
#undef S_atanf

/* ---- double instantiation ----------------------------------------------- */
#define S double
#define FN(name) name##_f64
#define S_exp exp
#define S_log log
#define S_pow pow
#define S_sqrt sqrt
#define S_asin asin
#define S_cosh cosh
#define S_atanf atanf
#define S_ncdf(x) (0.5 * erfc(-(x) * 0.70710678118654752440))
#include "gendr_oracle_body.inc"
#undef S
#undef FN

/* ---- scalar exports (kernel.cu:1230-1270) -------------------------------- */
float gendr_oracle_sigmoid_forward(int id, float sign, float x, float scale, float shape, float shift)
{ return cdf_f32(id, sign, x, scale, shape, shift); }
float gendr_oracle_sigmoid_backward(int id, float sign, float x, float scale, float shape, float shift)
{ return pdf_f32(id, sign, x, scale, shape, shift); }
float gendr_oracle_t_conorm_forward(int id, float a, float b, int face_id, float p)
{ (void)face_id; return tconorm_f32(id, a, b, p); }
float gendr_oracle_t_conorm_backward(int id, float a_all, float b_cur, int nf, float p)
{ (void)nf; return tconorm_grad_f32(id, a_all, b_cur, p); }

double gendr_oracle_sigmoid_forward_f64(int id, double sign, double x, double scale, double shape, double shift)
{ return cdf_f64(id, sign, x, scale, shape, shift); }
double gendr_oracle_sigmoid_backward_f64(int id, double sign, double x, double scale, double shape, double shift)
{ return pdf_f64(id, sign, x, scale, shape, shift); }
double gendr_oracle_t_conorm_forward_f64(int id, double a, double b, int face_id, double p)
{ (void)face_id; return tconorm_f64(id, a, b, p); }
double gendr_oracle_t_conorm_backward_f64(int id, double a_all, double b_cur, int nf, double p)
{ (void)nf; return tconorm_grad_f64(id, a_all, b_cur, p); }

long long gendr_oracle_count_pairs_f32(const float* faces, const float* faces_info,
                                       int B, int nf, const gendr_oracle_opts* o)
{
    const int is = o->image_size;
    const float thr = o->dist_eps * o->dist_scale;
    const float sqrt_thr = sqrtf(thr);
    long long total = 0;
    const int nthreads = o->num_threads > 0 ? o->num_threads : omp_get_max_threads();
#pragma omp parallel for collapse(2) reduction(+ : total) num_threads(nthreads)
    for (int bn = 0; bn < B; bn++)
        for (int row = 0; row < is; row++)
            for (int xi = 0; xi < is; xi++) {
                const int yi = is - 1 - row;
                const float yp = (float)((2. * yi + 1. - is) / is);
                const float xp = (float)((2. * xi + 1. - is) / is);
                for (int fn = 0; fn < nf; fn++) {
                    pair_t_f32 q;
                    const long fl = (long)bn * nf + fn;
                    total += eval_pair_f32(&q, faces + 9 * fl, faces_info + 27 * fl, xp, yp, thr, sqrt_thr, o);
                }
            }
    return total;
}

int gendr_oracle_max_threads(void) { return omp_get_max_threads(); }
