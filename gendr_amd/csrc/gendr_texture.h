// gendr_texture.h -- SURVEY.md row f-3: per-face texel blocks <-> texture atlas image (OBJ load / save).
//
// Reference: load_textures_cuda_kernel (gendr/cuda/load_textures_cuda_kernel.cu:14-72) samples an image bilinearly
// at the barycentric centre of every texel of every face; create_texture_image_cuda_kernel
// (gendr/cuda/create_texture_image_cuda_kernel.cu:16-75) paints one tile per face into an atlas.  One-off,
// bandwidth-trivial kernels: one lane per texel / per atlas pixel, written to follow the reference's arithmetic
// expression by expression, including where double literals promote a sub-expression (SURVEY.md note P).
#pragma once

#include <hip/hip_runtime.h>

namespace gendr {

constexpr int kTexThreads = 256;

// image [H,W,3], uv [nf,3,2] in [0,1], is_update [nf], textures [nf,R*R,3] (only faces with is_update != 0 are written)
__global__ __launch_bounds__(kTexThreads) void load_textures_kernel(
    const float* __restrict__ image, const float* __restrict__ uv, const int* __restrict__ is_update,
    float* __restrict__ textures, long texels, int R, int H, int W)
{
    const long i = (long)blockIdx.x * kTexThreads + threadIdx.x;
    if (i >= texels) return;
    const int fn = (int)(i / (R * R));
    const int w_y = (int)(i % (R * R)) / R, w_x = (int)(i % R);
    float w0, w1, w2;                                                     // kernel.cu:33-41: evaluated in double, stored in float
    if (w_x + w_y < R) {
        w0 = (float)((w_x + 1. / 3.) / R);
        w1 = (float)((w_y + 1. / 3.) / R);
    } else {
        w0 = (float)(((R - 1. - w_x) + 2. / 3.) / R);
        w1 = (float)(((R - 1. - w_y) + 2. / 3.) / R);
    }
    w2 = (float)(1. - w0 - w1);
    if (is_update[fn] == 0) return;
    const float* f = uv + (long)fn * 6;
    const float pos_x = (f[0] * w0 + f[2] * w1 + f[4] * w2) * (float)(W - 1);
    const float pos_y = (f[1] * w0 + f[3] * w1 + f[5] * w2) * (float)(H - 1);
    const int xi = (int)pos_x, yi = (int)pos_y;
    const float wx1 = pos_x - (float)xi, wx0 = 1.f - wx1;
    const float wy1 = pos_y - (float)yi, wy0 = 1.f - wy1;
    // the reference reads row yi+1 / column xi+1 even when their weight is exactly 0 at the last row / column
    // (a read past the image); the index is clamped here, which cannot change a finite result
    const int x0 = min(max(xi, 0), W - 1), x1 = min(max(xi + 1, 0), W - 1);
    const int y0 = min(max(yi, 0), H - 1), y1 = min(max((int)(pos_y + 1.f), 0), H - 1);
    float* out = textures + i * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float c = 0.f;
        c += image[((long)y0 * W + x0) * 3 + k] * (wx0 * wy0);
        c += image[((long)y1 * W + x0) * 3 + k] * (wx0 * wy1);
        c += image[((long)y0 * W + x1) * 3 + k] * (wx1 * wy0);
        c += image[((long)y1 * W + x1) * 3 + k] * (wx1 * wy1);
        out[k] = c;
    }
}

// uv [nf,3,2] in atlas pixels, textures [nf,R_in*R_in,3], image [rows,cols,3]; cols = tile_width * R_out
__global__ __launch_bounds__(kTexThreads) void create_texture_image_kernel(
    const float* __restrict__ uv, const float* __restrict__ textures, float* __restrict__ image,
    long pixels, int nf, int R, int R_out, int tile_width, float eps)
{
    const long i = (long)blockIdx.x * kTexThreads + threadIdx.x;
    if (i >= pixels) return;
    const int cols = tile_width * R_out;
    const int x = (int)(i % cols), y = (int)(i / cols);
    const int fn = x / R_out + (y / R_out) * tile_width;                 // kernel.cu:28-30
    if (fn >= nf) return;
    const float* p = uv + (long)fn * 6;
    const float p0x = p[0], p0y = p[1], p1x = p[2], p1y = p[3], p2x = p[4], p2y = p[5];
    const float den = (p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y)) + eps;
    const float inv[9] = {(p1y - p2y) / den, (p2x - p1x) / den, (p1x * p2y - p2x * p1y) / den,
                          (p2y - p0y) / den, (p0x - p2x) / den, (p2x * p0y - p0x * p2y) / den,
                          (p0y - p1y) / den, (p1x - p0x) / den, (p0x * p1y - p1x * p0y) / den};
    float w[3], w_sum = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float wk = inv[3 * k] * (float)x + inv[3 * k + 1] * (float)y + inv[3 * k + 2];
        w[k] = (float)fmax(fmin((double)wk, 1.), 0.);                     // max(min(w, 1.), 0.) in double
        w_sum += w[k];
    }
    const float wn0 = w[0] / (w_sum + eps), wn1 = w[1] / (w_sum + eps);
    const int w_x = (int)(wn0 * (float)R), w_y = (int)(wn1 * (float)R);
    const float* tex = textures + (long)fn * R * R * 3;
    const bool lower = (wn0 + wn1) * (float)R - (float)w_x - (float)w_y <= 1.f;
    int texel = lower ? w_y * R + w_x : (R - 1 - w_y) * R + (R - 1 - w_x);
    texel = min(max(texel, 0), R * R - 1);                                // w = 1 exactly gives index R: stay in the block
#pragma unroll
    for (int k = 0; k < 3; k++) image[i * 3 + k] = tex[texel * 3 + k];
}

}  // namespace gendr
