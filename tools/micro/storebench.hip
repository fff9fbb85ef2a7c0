// Micro-benchmark: what does the (tile shape -> store pattern) cost for the 6 output planes of the forward?
// B=64, 256x256, planes: 4 rgba + 2 aux.  One wave per tile of TW x TH = 64 pixels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int TW>
__global__ __launch_bounds__(256) void fill(float* __restrict__ rgba, float* __restrict__ aux, int is, int tiles_per_image)
{
    constexpr int TH = 64 / TW;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int b = wave / tiles_per_image, t = wave - b * tiles_per_image;
    const int tiles_x = is / TW;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int x = tx * TW + (lane % TW), y = ty * TH + (lane / TW);
    const long P = (long)is * is;
    const long pix = (long)y * is + x;
    float* o = rgba + (long)b * 4 * P + pix;
    o[0] = 0.1f; o[P] = 0.2f; o[2 * P] = 0.3f; o[3 * P] = 0.f;
    float* a = aux + (long)b * 2 * P + pix;
    a[0] = 1.f; a[P] = 2.f;
}

template <int TW>
float run(float* rgba, float* aux, int B, int is)
{
    const int tiles = (is * is) / 64;
    const int waves = B * tiles;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(fill<TW>, dim3(waves / 4), dim3(256), 0, 0, rgba, aux, is, tiles);
    hipEventRecord(s);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(fill<TW>, dim3(waves / 4), dim3(256), 0, 0, rgba, aux, is, tiles);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / 50 * 1000;
}

int main()
{
    const int B = 64, is = 256;
    float *rgba, *aux;
    hipMalloc(&rgba, (size_t)B * 4 * is * is * 4); hipMalloc(&aux, (size_t)B * 2 * is * is * 4);
    const double mb = (double)B * 6 * is * is * 4 / 1e6;
    printf("bytes %.1f MB\n", mb);
    printf("8x8   %.1f us\n", run<8>(rgba, aux, B, is));
    printf("16x4  %.1f us\n", run<16>(rgba, aux, B, is));
    printf("32x2  %.1f us\n", run<32>(rgba, aux, B, is));
    printf("64x1  %.1f us\n", run<64>(rgba, aux, B, is));
    hipMemsetAsync(rgba, 0, (size_t)B * 4 * is * is * 4, 0);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s);
    for (int i = 0; i < 20; i++) { hipMemsetAsync(rgba, 0, (size_t)B * 4 * is * is * 4, 0); hipMemsetAsync(aux, 0, (size_t)B * 2 * is * is * 4, 0); }
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("memset %.1f us\n", ms / 20 * 1000);
    return 0;
}
