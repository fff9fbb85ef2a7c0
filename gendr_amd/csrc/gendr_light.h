// gendr_light.h -- the lighting half of SURVEY.md row f-1: surface textures times (ambient + directional) light.
//
// Reference: gendr/lighting.py:48-71 builds a zero light tensor [B,nf,3], adds the ambient term and one term per
// directional light (gendr/functional/lighting.py:11-48: intensity * colour * relu(<normal, direction>)) with the face
// normals of Mesh.surface_normals (gendr/mesh.py:109-117: normalize(cross(v2 - v1, v0 - v1)), eps 1e-6), and
// multiplies the textures [B,nf,T,3] by it: about fifteen tensor kernels forward and as many backward.
// Here one kernel each way, one lane per face; the same float expressions in the same order.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gendr_hip.h"
#include "gendr_project.h"      // V3 helpers

namespace gendr {

constexpr int kLightThreads = 256;

__device__ __forceinline__ V3 clamped_unit(V3 v, float eps, float& raw)
{
    raw = sqrtf(dot(v, v));
    const float n = fmaxf(raw, eps);
    return {v.x / n, v.y / n, v.z / n};
}

__device__ __forceinline__ void face_light(const float* vertices, const int* face_index, long face, long vbase, int nv,
                                           const gendr_light_params& lp, V3& a, V3& b2, V3& raw, float& len, float* light,
                                           float* cosines, bool& ok)
{
    const int i0 = face_index[face * 3 + 0], i1 = face_index[face * 3 + 1], i2 = face_index[face * 3 + 2];
    ok = (unsigned)i0 < (unsigned)nv && (unsigned)i1 < (unsigned)nv && (unsigned)i2 < (unsigned)nv;
    if (!ok) return;
    const V3 v0 = v3(vertices + (vbase + i0) * 3), v1 = v3(vertices + (vbase + i1) * 3), v2 = v3(vertices + (vbase + i2) * 3);
    a = v2 - v1;
    b2 = v0 - v1;
    raw = cross(a, b2);
    const V3 n = clamped_unit(raw, 1e-6f, len);
    light[0] = lp.ambient_intensity * lp.ambient_color[0];
    light[1] = lp.ambient_intensity * lp.ambient_color[1];
    light[2] = lp.ambient_intensity * lp.ambient_color[2];
    for (int i = 0; i < lp.n_directional; i++) {
        const float c = n.x * lp.direction[i][0] + n.y * lp.direction[i][1] + n.z * lp.direction[i][2];
        cosines[i] = c;
        const float r = fmaxf(c, 0.f);
        for (int k = 0; k < 3; k++) light[k] += lp.intensity[i] * (lp.color[i][k] * r);
    }
}

__global__ __launch_bounds__(kLightThreads) void light_faces_kernel(
    const float* __restrict__ vertices, const int* __restrict__ face_index, const float* __restrict__ textures,
    float* __restrict__ out, int B, int nv, int nf, int T, int index_batched, const gendr_light_params lp)
{
    const long i = (long)blockIdx.x * kLightThreads + threadIdx.x;
    if (i >= (long)B * nf) return;
    const int b = (int)(i / nf);
    const long face = index_batched ? i : i - (long)b * nf;
    V3 a, b2, raw;
    float len, light[3], cosines[GENDR_MAX_DIRECTIONAL];
    bool ok;
    face_light(vertices, face_index, face, (long)b * nv, nv, lp, a, b2, raw, len, light, cosines, ok);
    const float nan = __int_as_float(0x7fc00000);
    for (long t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) {
            const long at = (i * T + t) * 3 + k;
            out[at] = ok ? textures[at] * light[k] : nan;        // a bad face index shows as NaN, nothing is read out of range
        }
}

// grad_vertices [B,nv,3] must be zero-filled (or NULL); grad_textures [B,nf,T,3] is written (or NULL)
__global__ __launch_bounds__(kLightThreads) void light_faces_backward_kernel(
    const float* __restrict__ vertices, const int* __restrict__ face_index, const float* __restrict__ textures,
    const float* __restrict__ grad_out, float* __restrict__ grad_textures, float* __restrict__ grad_vertices,
    int B, int nv, int nf, int T, int index_batched, const gendr_light_params lp)
{
    const long i = (long)blockIdx.x * kLightThreads + threadIdx.x;
    if (i >= (long)B * nf) return;
    const int b = (int)(i / nf);
    const long face = index_batched ? i : i - (long)b * nf;
    V3 a, b2, raw;
    float len, light[3], cosines[GENDR_MAX_DIRECTIONAL];
    bool ok;
    face_light(vertices, face_index, face, (long)b * nv, nv, lp, a, b2, raw, len, light, cosines, ok);
    float g_light[3] = {0.f, 0.f, 0.f};
    for (long t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) {
            const long at = (i * T + t) * 3 + k;
            const float g = grad_out[at];
            if (grad_textures) grad_textures[at] = ok ? g * light[k] : 0.f;
            if (ok) g_light[k] += g * textures[at];
        }
    if (!grad_vertices || !ok) return;
    V3 g_n = {0.f, 0.f, 0.f};
    for (int l = 0; l < lp.n_directional; l++) {
        if (!(cosines[l] > 0.f)) continue;                                   // relu
        const float g_cos = lp.intensity[l] * (lp.color[l][0] * g_light[0] + lp.color[l][1] * g_light[1] + lp.color[l][2] * g_light[2]);
        g_n = g_n + V3{lp.direction[l][0], lp.direction[l][1], lp.direction[l][2]} * g_cos;
    }
    // normalize: n = raw / max(|raw|, eps)
    V3 g_raw;
    if (len < 1e-6f) g_raw = g_n * (1.0f / 1e-6f);
    else {
        const V3 u = raw * (1.0f / len);
        g_raw = (g_n - u * dot(u, g_n)) * (1.0f / len);
    }
    // raw = a x b2:  g_a = b2 x g_raw,  g_b2 = g_raw x a ;  a = v2 - v1, b2 = v0 - v1
    const V3 g_a = cross(b2, g_raw), g_b2 = cross(g_raw, a);
    const int i0 = face_index[face * 3 + 0], i1 = face_index[face * 3 + 1], i2 = face_index[face * 3 + 2];
    float* gv = grad_vertices + (long)b * nv * 3;
    unsafeAtomicAdd(gv + i2 * 3 + 0, g_a.x);  unsafeAtomicAdd(gv + i2 * 3 + 1, g_a.y);  unsafeAtomicAdd(gv + i2 * 3 + 2, g_a.z);
    unsafeAtomicAdd(gv + i0 * 3 + 0, g_b2.x); unsafeAtomicAdd(gv + i0 * 3 + 1, g_b2.y); unsafeAtomicAdd(gv + i0 * 3 + 2, g_b2.z);
    unsafeAtomicAdd(gv + i1 * 3 + 0, -(g_a.x + g_b2.x)); unsafeAtomicAdd(gv + i1 * 3 + 1, -(g_a.y + g_b2.y));
    unsafeAtomicAdd(gv + i1 * 3 + 2, -(g_a.z + g_b2.z));
}

}  // namespace gendr
