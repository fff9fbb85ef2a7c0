// gendr_project.h -- SURVEY.md row f-1: the step right before the hot path, fused.
//
// The reference materialises, per render call, (vertices - eye) [B,nv,3], the rotated vertices (torch.matmul,
// gendr/functional/look_at.py:59-67), three perspective / orthogonal temporaries + a stack
// (gendr/transform.py:14-47) and finally the gathered face_vertices [B,nf,3,3]
// (gendr/functional/face_vertices.py:24-27): about ten small kernels and five passes over the vertex data.
// Here: one kernel forward (gather -> translate -> rotate -> project -> store face_vertices) and one backward
// (the transposed chain, scatter-add into grad_vertices, reduction into the 12 camera parameters).
// The 3x3 rotation itself is tiny ([B,3] tensors) and stays in PyTorch so that autograd reaches eye / angles.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gendr_hip.h"

namespace gendr {

constexpr int kProjThreads = 256;

// camera [B,12]: R row-major (x_axis, y_axis, z_axis), then eye.
__device__ __forceinline__ void project_point(const float* v, const float* cam, int perspective, float ws,
                                              float& d0, float& d1, float& d2, float& c0, float& c1, float& c2,
                                              float& ox, float& oy, float& oz)
{
    d0 = v[0] - cam[9]; d1 = v[1] - cam[10]; d2 = v[2] - cam[11];                 // vertices - eye
    c0 = d0 * cam[0] + d1 * cam[1] + d2 * cam[2];                                  // matmul(v, R^T): out_i = sum_j v_j R_ij
    c1 = d0 * cam[3] + d1 * cam[4] + d2 * cam[5];
    c2 = d0 * cam[6] + d1 * cam[7] + d2 * cam[8];
    if (perspective) { ox = c0 / c2 / ws; oy = c1 / c2 / ws; }                     // x / z / width (transform.py:25-27)
    else             { ox = c0 * ws;      oy = c1 * ws; }                          // x * scale   (transform.py:43-45)
    oz = c2;
}

__global__ __launch_bounds__(kProjThreads) void project_faces_kernel(
    const float* __restrict__ vertices, const int* __restrict__ face_index, const float* __restrict__ camera,
    float* __restrict__ face_vertices, int B, int nv, int nf, int index_batched, int perspective, float ws)
{
    const long i = (long)blockIdx.x * kProjThreads + threadIdx.x;       // one lane per (b, face, corner)
    const long per_item = (long)nf * 3;
    if (i >= per_item * B) return;
    const int b = (int)(i / per_item);
    const long fk = i - (long)b * per_item;
    const int idx = face_index[(index_batched ? (long)b * per_item : 0) + fk];
    float* o = face_vertices + i * 3;
    if ((unsigned)idx >= (unsigned)nv) {                   // never read outside the vertex tensor: a bad index shows as NaN
        o[0] = o[1] = o[2] = __int_as_float(0x7fc00000);
        return;
    }
    const float* v = vertices + ((long)b * nv + idx) * 3;
    float d0, d1, d2, c0, c1, c2, ox, oy, oz;
    project_point(v, camera + (long)b * 12, perspective, ws, d0, d1, d2, c0, c1, c2, ox, oy, oz);
    o[0] = ox; o[1] = oy; o[2] = oz;
}

__device__ __forceinline__ float proj_wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true));
    return v;       // total in lane 63
}

// grad_vertices [B,nv,3] and grad_camera [B,12] must be zero-filled by the caller.
// Blocks never straddle batch items (grid = B * blocks_per_item) so that the camera reduction is per item.
__global__ __launch_bounds__(kProjThreads) void project_faces_backward_kernel(
    const float* __restrict__ vertices, const int* __restrict__ face_index, const float* __restrict__ camera,
    const float* __restrict__ grad_face_vertices, float* __restrict__ grad_vertices, float* __restrict__ grad_camera,
    int B, int nv, int nf, int index_batched, int perspective, float ws, int blocks_per_item)
{
    const int b = blockIdx.x / blocks_per_item;
    const long per_item = (long)nf * 3;
    const long fk = (long)(blockIdx.x - b * blocks_per_item) * kProjThreads + threadIdx.x;
    const bool live = fk < per_item;
    const float* cam = camera + (long)b * 12;
    float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f, gR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int idx = live ? face_index[(index_batched ? (long)b * per_item : 0) + fk] : 0;
    if (live && (unsigned)idx < (unsigned)nv) {            // bad indices contribute nothing (forward already shows NaN)
        const float* v = vertices + ((long)b * nv + idx) * 3;
        float d0, d1, d2, c0, c1, c2, ox, oy, oz;
        project_point(v, cam, perspective, ws, d0, d1, d2, c0, c1, c2, ox, oy, oz);
        const float* g = grad_face_vertices + ((long)b * per_item + fk) * 3;
        float gc0, gc1, gc2;
        if (perspective) {
            // x = c0 / c2 / w  ->  dx/dc0 = 1/(c2 w), dx/dc2 = -x / c2
            gc0 = g[0] / c2 / ws;
            gc1 = g[1] / c2 / ws;
            gc2 = g[2] - (g[0] * ox + g[1] * oy) / c2;
        } else {
            gc0 = g[0] * ws; gc1 = g[1] * ws; gc2 = g[2];
        }
        gd0 = gc0 * cam[0] + gc1 * cam[3] + gc2 * cam[6];       // R^T gc
        gd1 = gc0 * cam[1] + gc1 * cam[4] + gc2 * cam[7];
        gd2 = gc0 * cam[2] + gc1 * cam[5] + gc2 * cam[8];
        float* gv = grad_vertices + ((long)b * nv + idx) * 3;
        unsafeAtomicAdd(gv + 0, gd0); unsafeAtomicAdd(gv + 1, gd1); unsafeAtomicAdd(gv + 2, gd2);
        gR[0] = gc0 * d0; gR[1] = gc0 * d1; gR[2] = gc0 * d2;   // dL/dR_ij = gc_i d_j
        gR[3] = gc1 * d0; gR[4] = gc1 * d1; gR[5] = gc1 * d2;
        gR[6] = gc2 * d0; gR[7] = gc2 * d1; gR[8] = gc2 * d2;
    }
    if (grad_camera) {
        const int lane = threadIdx.x & 63;
        float* gc = grad_camera + (long)b * 12;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float s = proj_wave_sum(gR[k]);
            if (lane == 63 && s != 0.f) unsafeAtomicAdd(gc + k, s);
        }
        const float e0 = proj_wave_sum(-gd0), e1 = proj_wave_sum(-gd1), e2 = proj_wave_sum(-gd2);   // d = v - eye
        if (lane == 63) {
            if (e0 != 0.f) unsafeAtomicAdd(gc + 9, e0);
            if (e1 != 0.f) unsafeAtomicAdd(gc + 10, e1);
            if (e2 != 0.f) unsafeAtomicAdd(gc + 11, e2);
        }
    }
}

// ---- camera rotation (look_at.py:52-59 / look.py): z = normalize(at - eye | direction), x = normalize(up x z),
// y = normalize(z x x), F.normalize semantics v / max(|v|, 1e-5).  One lane per batch item.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float clamped_norm(V3 v) { return fmaxf(sqrtf(dot(v, v)), 1e-5f); }
__device__ __forceinline__ V3 normalize(V3 v) { const float n = clamped_norm(v); return {v.x / n, v.y / n, v.z / n}; }
// gradient of normalize at v for upstream g
__device__ __forceinline__ V3 normalize_grad(V3 v, V3 g)
{
    const float raw = sqrtf(dot(v, v));
    if (raw < 1e-5f) return g * (1.0f / 1e-5f);                  // clamped: the denominator is a constant
    const V3 u = v * (1.0f / raw);
    return (g - u * dot(u, g)) * (1.0f / raw);
}
__device__ __forceinline__ void store3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__global__ void camera_rotation_kernel(const float* __restrict__ eye, const float* __restrict__ target,
                                       const float* __restrict__ up, float* __restrict__ camera, int B, int target_is_direction)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const V3 e = v3(eye + b * 3), t = v3(target + b * 3), u = v3(up + b * 3);
    const V3 z = normalize(target_is_direction ? t : t - e);
    const V3 x = normalize(cross(u, z));
    const V3 y = normalize(cross(z, x));
    float* c = camera + (long)b * 12;
    store3(c, x); store3(c + 3, y); store3(c + 6, z); store3(c + 9, e);
}

// grad_camera [B,12] -> grad_eye / grad_target / grad_up [B,3] (each optional)
__global__ void camera_rotation_backward_kernel(const float* __restrict__ eye, const float* __restrict__ target,
                                                const float* __restrict__ up, const float* __restrict__ grad_camera,
                                                float* __restrict__ grad_eye, float* __restrict__ grad_target,
                                                float* __restrict__ grad_up, int B, int target_is_direction)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const V3 e = v3(eye + b * 3), t = v3(target + b * 3), u = v3(up + b * 3);
    const V3 z_raw = target_is_direction ? t : t - e;
    const V3 z = normalize(z_raw);
    const V3 x_raw = cross(u, z);
    const V3 x = normalize(x_raw);
    const V3 y_raw = cross(z, x);
    const float* g = grad_camera + (long)b * 12;
    // for c = a x b:  ga = b x gc,  gb = gc x a
    const V3 gy_raw = normalize_grad(y_raw, v3(g + 3));
    V3 gz = v3(g + 6) + cross(x, gy_raw);
    const V3 gx = v3(g) + cross(gy_raw, z);
    const V3 gx_raw = normalize_grad(x_raw, gx);
    gz = gz + cross(gx_raw, u);
    const V3 gz_raw = normalize_grad(z_raw, gz);
    if (grad_up) store3(grad_up + b * 3, cross(z, gx_raw));
    if (grad_target) store3(grad_target + b * 3, gz_raw);
    if (grad_eye) store3(grad_eye + b * 3, target_is_direction ? v3(g + 9) : v3(g + 9) - gz_raw);
}

}  // namespace gendr
