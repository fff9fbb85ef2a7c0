// gendr_voxel.h -- SURVEY.md row f-2: mesh -> occupancy grid (the evaluation step after the hot path).
//
// Reference: gendr/functional/voxelization.py:11-62 drives four kernels
// (gendr/cuda/voxelization_cuda_kernel.cu:36-194): sub1 x3 (axis-parallel rays through the grid lines, one launch
// per axis with the face tensor permuted on the host), sub2 (vertex voxels), sub3 (boundary seed) and sub4
// (one sweep of the flood fill, relaunched from a host loop with a device->host sum per sweep until nothing changes).
// Here: two launches, no host round trips.
//   voxel_surface_kernel : all three ray axes and the vertex voxels in one grid (blockIdx.y selects), the face
//                          loop is wave-uniform so the nine floats of a face arrive by scalar loads;
//   voxel_fill_kernel    : one workgroup per batch item floods the whole grid to its fixpoint on 64-bit row
//                          bitmasks (one word = 64 voxels along the last axis), in LDS up to 64^3, in a global
//                          workspace above that; inside a word the flood is a Kogge-Stone fill, across words and
//                          rows it is an OR of neighbours, iterated until __syncthreads_or says nothing changed.
// The fixpoint of the reference's sweep (sub4, kernel.cu:148-194) does not depend on the sweep order: it is the
// set of empty voxels 6-connected to an empty boundary voxel, so results are identical, not just close.
// The arithmetic that decides which voxel a ray hits (kernel.cu:56-75) is kept operation by operation in fp32
// (no contraction; -ffp-contract=off for the whole library).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gendr {

constexpr int kVoxSurfaceThreads = 256;
constexpr int kVoxFillThreads = 1024;
typedef unsigned long long u64;

// s_y, s_x, s_z: element strides of the kernel's (y, x, z) in the [vs,vs,vs] output grid
__device__ __forceinline__ void set_voxel(int* __restrict__ grid, int vs, int s_y, int s_x, int s_z, int yi, int xi, int zi)
{
    if (0 <= yi && yi < vs && 0 <= xi && xi < vs && 0 <= zi && zi < vs)
        grid[(long)yi * s_y + (long)xi * s_x + (long)zi * s_z] = 1;
}

__device__ __forceinline__ int axis_stride(int axis, int vs) { return axis == 0 ? vs * vs : axis == 1 ? vs : 1; }

// grid (ceil(vs^2/256) or ceil(nf/256), 4, B); blockIdx.y = 0,1,2: rays along that axis; 3: vertex voxels.
__global__ __launch_bounds__(kVoxSurfaceThreads) void voxel_surface_kernel(
    const float* __restrict__ faces, int* __restrict__ voxels, int nf, int vs)
{
    const int b = blockIdx.z, mode = blockIdx.y;
    const int t = blockIdx.x * kVoxSurfaceThreads + threadIdx.x;
    int* grid = voxels + (long)b * vs * vs * vs;
    const float* fb = faces + (long)b * nf * 9;
    if (mode == 3) {                                                        // sub2, kernel.cu:114-124
        if (t >= nf) return;
        const float* f = fb + (long)t * 9;
        for (int k = 0; k < 3; k++)
            set_voxel(grid, vs, vs * vs, vs, 1, (int)floorf(f[3 * k]), (int)floorf(f[3 * k + 1]), (int)floorf(f[3 * k + 2]));
        return;
    }
    // the host-side permutations of voxelization.py:14-17, as an axis map: kernel (y, x, z) = original axes
    const int a_y = mode == 0 ? 2 : 0, a_x = mode == 1 ? 2 : 1, a_z = mode;
    const int s_y = axis_stride(a_y, vs), s_x = axis_stride(a_x, vs), s_z = axis_stride(a_z, vs);
    if (t >= vs * vs) return;
    const int y = t % vs, x = t / vs;                                       // kernel.cu:50-51
    for (int fn = 0; fn < nf; fn++) {
        const float* f = fb + (long)fn * 9;                                 // wave-uniform -> scalar loads
        const float y1d = f[3 + a_y] - f[a_y], x1d = f[3 + a_x] - f[a_x], z1d = f[3 + a_z] - f[a_z];
        const float y2d = f[6 + a_y] - f[a_y], x2d = f[6 + a_x] - f[a_x], z2d = f[6 + a_z] - f[a_z];
        const float ypd = (float)y - f[a_y], xpd = (float)x - f[a_x];
        const float det = x1d * y2d - x2d * y1d;
        if (det == 0.f) continue;
        const float t1 = (y2d * xpd - x2d * ypd) / det;
        const float t2 = (-y1d * xpd + x1d * ypd) / det;
        if (t1 < 0.f) continue;
        if (t2 < 0.f) continue;
        if (1.f < t1 + t2) continue;
        const int zi = (int)floorf(t1 * z1d + t2 * z2d + f[a_z]);
        set_voxel(grid, vs, s_y, s_x, s_z, y, x, zi);
        set_voxel(grid, vs, s_y, s_x, s_z, y - 1, x, zi);
        set_voxel(grid, vs, s_y, s_x, s_z, y, x - 1, zi);
        set_voxel(grid, vs, s_y, s_x, s_z, y - 1, x - 1, zi);
    }
}

// Flood of `gen` through `pro` inside one 64-bit word, both directions (Kogge-Stone occluded fill).
__device__ __forceinline__ u64 fill_word(u64 gen, u64 pro)
{
    u64 g = gen, p = pro;
    g |= p & (g << 1);  p &= p << 1;
    g |= p & (g << 2);  p &= p << 2;
    g |= p & (g << 4);  p &= p << 4;
    g |= p & (g << 8);  p &= p << 8;
    g |= p & (g << 16); p &= p << 16;
    g |= p & (g << 32);
    u64 h = gen; p = pro;
    h |= p & (h >> 1);  p &= p >> 1;
    h |= p & (h >> 2);  p &= p >> 2;
    h |= p & (h >> 4);  p &= p >> 4;
    h |= p & (h >> 8);  p &= p >> 8;
    h |= p & (h >> 16); p &= p >> 16;
    h |= p & (h >> 32);
    return g | h;
}

// One workgroup per batch item.  voxels [B,vs,vs,vs] int32: in = surface occupancy (0/1), out = 1 - visible
// (voxelization.py:28-44).  Rows are (y, x), W = ceil(vs / 64) words per row.
template <bool IN_LDS>
__global__ __launch_bounds__(kVoxFillThreads) void voxel_fill_kernel(int* __restrict__ voxels, u64* __restrict__ workspace, int vs, int W)
{
    extern __shared__ u64 s_rows[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = vs * vs, words = rows * W;
    u64* E = IN_LDS ? s_rows : workspace + (long)b * 2 * words;            // empty voxels
    u64* V = E + words;                                                     // visible voxels
    int* grid = voxels + (long)b * rows * vs;

    for (int idx = wave; idx < words; idx += kVoxFillThreads / 64) {        // one coalesced 64-voxel read per word
        const int row = idx / W, w = idx - row * W, z = w * 64 + lane;
        const bool valid = z < vs;
        const int occ = valid ? grid[(long)row * vs + z] : 1;
        const u64 e = __ballot(valid && occ == 0);
        if (lane == 0) {
            const int y = row / vs, x = row - y * vs;
            u64 edge = 0;                                                   // sub3, kernel.cu:140-142
            if (y == 0 || y == vs - 1 || x == 0 || x == vs - 1) edge = ~0ull;
            if (w == 0) edge |= 1ull;
            if (w == (vs - 1) / 64) edge |= 1ull << ((vs - 1) & 63);
            E[idx] = e;
            V[idx] = e & edge;
        }
    }
    __syncthreads();

    for (;;) {                                                              // sub4 to its fixpoint
        int changed = 0;
        for (int idx = tid; idx < words; idx += kVoxFillThreads) {
            const u64 e = E[idx], v = V[idx];
            if (e == v) continue;                                           // nothing left to reach in this word
            const int row = idx / W, w = idx - row * W;
            const int y = row / vs, x = row - y * vs;
            u64 n = v;
            if (y > 0) n |= V[idx - vs * W];
            if (y < vs - 1) n |= V[idx + vs * W];
            if (x > 0) n |= V[idx - W];
            if (x < vs - 1) n |= V[idx + W];
            if (w > 0) n |= V[idx - 1] >> 63;
            if (w < W - 1) n |= V[idx + 1] << 63;
            const u64 nv = fill_word(n & e, e);
            if (nv != v) { V[idx] = nv; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }

    for (int idx = wave; idx < words; idx += kVoxFillThreads / 64) {
        const int row = idx / W, w = idx - row * W, z = w * 64 + lane;
        if (z < vs) grid[(long)row * vs + z] = (int)(1ull ^ ((V[idx] >> lane) & 1ull));
    }
}

}  // namespace gendr
