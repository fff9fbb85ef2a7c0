// Where does the library's powf(x, 2.f) differ from the correctly rounded square x * x?  For every float x of [2^-20, 16]:
// mismatches, and how close the exact square lies to a rounding midpoint in those cases (|x*x - RN(x*x)| / (ulp / 2): 1 = on the
// midpoint).  Decides the margin of the fast form of GammaFamily::cdf's power (gendr_math.h).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pow2probe.hip -o tools/micro/bin/pow2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>

__global__ void probe(unsigned lo, unsigned hi, unsigned long long* out, float yexp)
{
    unsigned long long bad = 0;
    float minratio = 2.f;
    for (unsigned long long u = (unsigned long long)lo + blockIdx.x * 256ull + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * 256ull) {
        const float x = __uint_as_float((unsigned)u);
        const float p = x * x;
        const float want = powf(x, yexp);
        if (__float_as_uint(p) != __float_as_uint(want)) {
            bad++;
            const float r = __builtin_fmaf(x, x, -p);
            int e; frexpf(p, &e);
            const float half_ulp = ldexpf(1.f, e - 25);            // p in [2^(e-1), 2^e): ulp = 2^(e-24)
            minratio = fminf(minratio, fabsf(r) / half_ulp);
        }
    }
    atomicAdd(out, bad);
    atomicMin((unsigned*)(out + 1), __float_as_uint(minratio));
}

int main()
{
    unsigned long long* out;
    hipMalloc(&out, 16);
    unsigned long long h[2] = {0, 0x7f7fffffull};
    hipMemcpy(out, h, 16, hipMemcpyHostToDevice);
    const float a = 0x1p-20f, b = 16.f;
    unsigned lo, hi; memcpy(&lo, &a, 4); memcpy(&hi, &b, 4);
    hipLaunchKernelGGL(probe, dim3(256 * 16), dim3(256), 0, 0, lo, hi, out, 2.f);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    float mr; unsigned m = (unsigned)h[1]; memcpy(&mr, &m, 4);
    printf("powf(x, 2) != x * x on %llu of %u floats in [2^-20, 16]; smallest |residual| / (ulp / 2) among them: %.9g (1 - that = %.3g)\n",
           h[0], hi - lo + 1, mr, 1.0 - mr);
    return 0;
}
