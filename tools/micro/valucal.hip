// Calibration of the VALU-occupation figure bench.py quotes: a kernel whose vector ALUs are occupied 100 % by
// construction -- every SIMD of the chip holds 8 waves, each of which issues nothing but independent v_fma_f32 -- run under the
// same rocprofv3 --pmc pass as the render kernels (profiles/run_pmc.sh).  Whatever
//     SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel duration * 2.4 GHz)
// reads for it is the factor the same expression over-reads by for every other kernel (clock below nominal under counters, the
// counter's unit); tools/pmc_finalize.py divides by it.  SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) must read 1.00 here.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valucal.hip -o tools/micro/bin/valucal
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int CH = 16;

__global__ __launch_bounds__(256) void valu_calibration_kernel(float* out, float a, float b, int iters)
{
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) v[c] = 1.0f + 0.001f * (threadIdx.x + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = __builtin_fmaf(v[c], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) s += v[c];
    if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the chains alive
}

int main()
{
    float* out = nullptr;
    if (hipMalloc(&out, 1 << 20) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 8;                               // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; rep++) {
        const int iters = 1 << 14;                            // 16 x 16384 FMAs per lane: ~0.9 ms
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_calibration_kernel, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_insts = (double)blocks * 4 * CH * iters;
        printf("valucal: %d blocks, %.3f ms, %.3f cycles per wave instruction and SIMD at 2.4 GHz\n", blocks, ms,
               ms * 1e-3 * 2.4e9 * (cus * 4) / wave_insts);
    }
    return 0;
}
