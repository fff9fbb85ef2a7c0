// Micro-test: does gfx950 execute scalar stores (s_store_dwordx4 + s_dcache_wb), and does a later kernel see the data through
// scalar loads?   hipcc --offload-arch=gfx950 -O3 tools/micro/sstore.hip -o tools/micro/bin/sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void writer(unsigned* out, int per_wave, int interleaved)
{
    const unsigned w = blockIdx.x;
    for (int k = 0; k < per_wave; k++) {
        const unsigned long long m = __ballot((threadIdx.x + k + w) % 3 == 0);
        u4 v; v.x = (unsigned)m; v.y = (unsigned)(m >> 32); v.z = w; v.w = (unsigned)k;
        // slot layout: interleaved != 0 puts neighbouring 16-byte slots (one 64-byte line) into the hands of DIFFERENT waves
        unsigned* p = out + (interleaved ? ((size_t)k * gridDim.x + w) : ((size_t)w * per_wave + k)) * 4;
        asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(v), "s"(p) : "memory");
    }
    asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void reader(const unsigned* in, unsigned* bad, int n, int waves, int interleaved)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned w = interleaved ? i % waves : i / 16, k = interleaved ? i / waves : i % 16;
    if (in[(size_t)i * 4 + 2] != w || in[(size_t)i * 4 + 3] != k) atomicAdd(bad, 1u);
}
int main()
{
    const int waves = 4096, per = 16, n = waves * per;
    unsigned *out, *bad;
    hipMalloc(&out, (size_t)n * 16); hipMalloc(&bad, 4);
    int total_wrong = 0;
    for (int interleaved = 0; interleaved < 2; interleaved++) {
    hipMemset(out, 0xff, (size_t)n * 16); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(writer, dim3(waves), dim3(64), 0, 0, out, per, interleaved);
    hipLaunchKernelGGL(reader, dim3((n + 255) / 256), dim3(256), 0, 0, out, bad, n, waves, interleaved);
    unsigned hb = 0; hipError_t e = hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> h((size_t)n * 4); hipMemcpy(h.data(), out, (size_t)n * 16, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int i = 0; i < n; i++) {
        unsigned long long want = 0; const unsigned w = interleaved ? i % waves : i / per, k = interleaved ? i / waves : i % per;
        for (unsigned l = 0; l < 64; l++) if ((l + k + w) % 3 == 0) want |= 1ull << l;
        const unsigned long long got = h[(size_t)i * 4] | ((unsigned long long)h[(size_t)i * 4 + 1] << 32);
        if (got != want || h[(size_t)i * 4 + 2] != w || h[(size_t)i * 4 + 3] != k) wrong++;
    }
    printf("scalar stores (%s slots): err=%d device-side mismatches=%u host-side mismatches=%d of %d\n", interleaved ? "interleaved" : "contiguous", (int)e, hb, wrong, n);
    total_wrong += wrong + (int)hb;
    }
    const int wrong = total_wrong; const unsigned hb = 0;
    return wrong != 0 || hb != 0;
}
