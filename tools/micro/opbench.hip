// Micro-benchmark: issue cost (SIMD cycles per wave-instruction sequence) of the division / sqrt / exp forms the pair
// math can choose from.  Every lane runs ITER independent evaluations per form on 8 interleaved chains; the grid fills
// the chip with 8 waves per SIMD, so the result is throughput, not latency.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/opbench.hip -o tools/micro/bin/opbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

constexpr int ITER = 2048, CH = 8;

__device__ __forceinline__ float div_by(float a, double rb) { return (float)((double)a * rb); }
__device__ __forceinline__ float mark3(float a, float b, float y) { const float q = a * y; const float r = __builtin_fmaf(-q, b, a); return __builtin_fmaf(r, y, q); }
__device__ __forceinline__ float mark5(float a, float b, float y)
{
    float q = a * y; float r = __builtin_fmaf(-q, b, a); q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-q, b, a); return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ double rcp64(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}

template <int FORM>
__global__ __launch_bounds__(256) void k(float* out, float b, double rb, float y, int iters)
{
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) v[c] = 1.0f + 0.001f * (threadIdx.x + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float x = v[c];
            if (FORM == 0) x = x * b + 0.5f;                              // 2 plain f32 ops (reference point)
            else if (FORM == 1) x = div_by(x, rb) + 0.5f;                 // cvt, mul_f64, cvt (+ add)
            else if (FORM == 2) x = mark3(x, b, y) + 0.5f;
            else if (FORM == 3) x = mark5(x, b, y) + 0.5f;
            else if (FORM == 4) x = x / b + 0.5f;                         // IEEE f32 division
            else if (FORM == 5) x = sqrtf(x) + 0.5f;                      // correctly rounded sqrt
            else if (FORM == 6) x = expf(-x) + 0.5f;
            else if (FORM == 7) x = (float)rcp64((double)x) + 0.5f;       // v_rcp_f64 + 2 Newton steps (+2 cvt)
            else if (FORM == 8) x = (float)(1.0 / (double)x) + 0.5f;      // IEEE f64 division
            else if (FORM == 9) x = __builtin_amdgcn_sqrtf(x) + 0.5f;     // bare v_sqrt_f32
            else if (FORM == 10) x = __expf(-x) + 0.5f;                   // v_exp_f32 path
            else if (FORM == 11) x = (float)((double)x + 0.5);            // cvt, add_f64, cvt
            v[c] = x;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) s += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FORM>
void run(const char* name, float* out, int extra_ops)
{
    const float b = 1.37f; const double rb = 1.0 / (double)b; const float y = 1.0f / b;
    const int blocks = 256 * 8;            // 8 waves per SIMD
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, b, rb, y, ITER);
    hipEventRecord(s);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, b, rb, y, ITER);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double evals = 5.0 * blocks * 4 /*waves*/ * (double)ITER * CH;       // wave-level evaluations
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;                    // at the nominal clock
    printf("%-34s %7.2f cycles per evaluation (nominal 2.4 GHz), %8.1f us\n", name, simd_cycles / evals, ms * 1000 / 5);
    (void)extra_ops;
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("mul + add (2 f32 ops)", out, 0);
    run<11>("cvt + add_f64 + cvt", out, 0);
    run<1>("div_by (cvt, mul_f64, cvt) + add", out, 0);
    run<2>("markstein 3 ops + add", out, 0);
    run<3>("markstein 5 ops + add", out, 0);
    run<4>("IEEE f32 division + add", out, 0);
    run<5>("sqrtf (correctly rounded) + add", out, 0);
    run<9>("v_sqrt_f32 + add", out, 0);
    run<6>("expf + add", out, 0);
    run<10>("__expf + add", out, 0);
    run<7>("rcp_f64 + 2 Newton + 2 cvt + add", out, 0);
    run<8>("IEEE f64 division + 2 cvt + add", out, 0);
    return 0;
}
