// gendr_kernels.h -- CDNA4 (gfx950) kernels of the generalized soft rasterizer.
//
// Replaces the three __global__ kernels of the reference
// (gendr/cuda/generalized_renderer_cuda_kernel.cu = "kernel.cu"):
//   forward_render_inv_cuda_kernel :620-676  ->  face_setup_kernel  (+ face_info_kernel, reference layout)
//   forward_render_cuda_kernel     :680-862  ->  render_forward_kernel
//   backward_render_cuda_kernel    :866-1065 ->  render_backward_kernel
//
// Design (DESIGN.md has the long form):
//   * one 256-lane workgroup = one 16x16 pixel tile of one batch item; its 4 wavefronts own
//     the four 8x8 quadrants (lane = pixel, so the per-pixel alpha fold and online softmax keep
//     the reference's ascending-face order without any cross-lane combination);
//   * exact tile culling: the face-setup kernel writes, per face, a conservative box outside of
//     which the reference itself would skip the pair (kernel.cu:747,769,784).  The workgroup
//     ballots those boxes against its tile rectangle, compacts the surviving face indices in
//     ascending order into LDS, stages their face records through LDS in 16-byte bursts, and
//     each wavefront refines the list against its own 8x8 quadrant;
//   * inside the loop every lane still applies the reference's own three skip tests, so culling
//     only removes pairs that contribute exactly nothing;
//   * backward recomputes the pair (as the reference does), reduces the 9 (+3 / +9) partials over
//     the wavefront with DPP adds, accumulates per-tile sums in LDS and issues one hardware fp32
//     atomic per (tile, face, component) instead of 12..84 per (pixel, face).
//
// No MFMA: there is no dense contraction in this path.  Compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gendr_hip.h"
#include "gendr_math.h"

namespace gendr {

// ---------------------------------------------------------------------------------------------
// face record layout (floats).  Geometry part is common; the tail depends on the texture mode.
// ---------------------------------------------------------------------------------------------
constexpr int kRecBox   = 0;    // xlo, xhi, ylo, yhi : pixel centres outside are skipped
constexpr int kRecInv   = 4;    // inv[9]   (kernel.cu:645-657)
constexpr int kRecEdge  = 13;   // A[3][3]  A[k][j] = sym[k][j] - sym[(k+1)%3][j]  (kernel.cu:95-97,146-148)
constexpr int kRecDen   = 22;   // Dn[3]    Dn[k] = A[k][k] - A[k][(k+1)%3]        (denominator of :99,:150)
constexpr int kRecVert  = 25;   // the 9 input floats x0 y0 z0 x1 y1 z1 x2 y2 z2
constexpr int kRecBits  = 34;   // int bits: 1,2,4 = first obtuse corner 0,1,2 (:667-675); 8 = front side (:56-58)
constexpr int kRecSpare = 35;
constexpr int kRecTex   = 36;   // TEXM 0: own rgb, next-face rgb ; TEXM 1: 3 vertex colours ; TEXM 2: nothing

// texture modes of the kernels
constexpr int kTexSurface1 = 0;   // texture_type surface, T == 1 (default Mesh texture): texels staged in the record
constexpr int kTexVertex   = 1;   // texture_type vertex (T == 3): 9 floats staged in the record
constexpr int kTexSurfaceN = 2;   // texture_type surface, T = R*R > 1: texels read from HBM/L2 per pair

__host__ __device__ constexpr int record_floats(int texm) { return texm == kTexSurface1 ? 44 : (texm == kTexVertex ? 48 : 36); }

constexpr int kTile      = 16;    // tile edge in pixels
constexpr int kThreads   = 256;
constexpr int kListCap   = 4096;  // face indices per scan range (uint16 in LDS)
constexpr int kRecCap    = 160;   // face records resident in LDS at a time

struct RenderArgs {
    const float*  boxes;        // [B*nf][4]
    const float*  records;      // [B*nf][REC]
    const float*  textures;     // [B,nf,T,3]
    float*        rgba;         // [B,4,is,is]
    float*        aux;          // [B,2,is,is]
    const float*  grad_rgba;    // backward only
    float*        grad_faces;   // backward only
    float*        grad_textures;
    int B, nf, T, R, is;
    int tiles_x, tiles_per_image, total_tiles;
    gendr_params p;
    float thr;                  // dist_eps * dist_scale (kernel.cu:725)
    float softmax_sum0;         // exp(aggr_rgb_eps / aggr_rgb_gamma) (kernel.cu:729)
    float inv_unused;
};

// ---------------------------------------------------------------------------------------------
// per-face setup
// ---------------------------------------------------------------------------------------------
struct FaceGeom {
    float inv[9];
    float sym[9];
    int   obt;     // bit k set = corner k is the first obtuse one
    int   front;
};

// reference arithmetic of kernel.cu:637-675 (float, no contraction)
__device__ __forceinline__ void face_geometry(const float* f, FaceGeom& g)
{
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    const float adj[9] = {
        y1 - y2, x2 - x1, x1 * y2 - x2 * y1,
        y2 - y0, x0 - x2, x2 * y0 - x0 * y2,
        y0 - y1, x1 - x0, x0 * y1 - x1 * y0};
    float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    // clamp against the double literal 1e-10 (:653)
    det = det > 0 ? (float)fmax((double)det, 1e-10) : (float)fmin((double)det, -1e-10);
#pragma unroll
    for (int k = 0; k < 9; k++) g.inv[k] = adj[k] / det;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            g.sym[3 * j + k] = f[3 * j] * f[3 * k] + f[3 * j + 1] * f[3 * k + 1] + 1;
    const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    g.obt = 0;
#pragma unroll
    for (int k = 2; k >= 0; k--) {   // descending so that the lowest obtuse corner wins, as the `break` at :673 does
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        if ((px[k1] - px[k]) * (px[k2] - px[k]) + (py[k1] - py[k]) * (py[k2] - py[k]) < 0) g.obt = 1 << k;
    }
    g.front = ((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) ? 1 : 0;   // :56-58
}

__device__ __forceinline__ float round_up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
}
__device__ __forceinline__ float round_down(double v)
{
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
}

// One thread per face.  Writes boxes[i][4] and records[i][REC].
//   sthr   = sqrtf(dist_eps * dist_scale), the reference's border margin (:747)
//   cull_r = distance beyond which an outside pixel contributes nothing (gendr_cull_radius), or +inf
template <int TEXM>
__global__ __launch_bounds__(kThreads) void face_setup_kernel(
    const float* __restrict__ faces, const float* __restrict__ textures,
    float* __restrict__ boxes, float* __restrict__ records,
    long total_faces, float sthr, float cull_r)
{
    constexpr int REC = record_floats(TEXM);
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total_faces) return;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = faces[i * 9 + k];
    FaceGeom g;
    face_geometry(f, g);

    const float xmax = fmaxf(fmaxf(f[0], f[3]), f[6]), xmin = fminf(fminf(f[0], f[3]), f[6]);
    const float ymax = fmaxf(fmaxf(f[1], f[4]), f[7]), ymin = fminf(fminf(f[1], f[4]), f[7]);
    // the reference's own test: x > max + thr || x < min - thr ...  (same float operations)
    float xhi = xmax + sthr, xlo = xmin - sthr, yhi = ymax + sthr, ylo = ymin - sthr;

    if (cull_r < INFINITY) {
        // Bound E on |computed distance - true distance| for pixel centres in [-1,1]^2, from the measured
        // difference between the float inverse the loop will use and a double-precision inverse, plus the
        // rounding of the barycentric and distance evaluations.  A pixel farther than cull_r + E from the
        // face's bounding box cannot pass the reference's skip tests (DESIGN.md "exact culling").
        const double X0 = f[0], Y0 = f[1], X1 = f[3], Y1 = f[4], X2 = f[6], Y2 = f[7];
        const double det = X2 * (Y0 - Y1) + X0 * (Y1 - Y2) + X1 * (Y2 - Y0);
        const double adj[9] = {
            Y1 - Y2, X2 - X1, X1 * Y2 - X2 * Y1,
            Y2 - Y0, X0 - X2, X2 * Y0 - X0 * Y2,
            Y0 - Y1, X1 - X0, X0 * Y1 - X1 * Y0};
        const double eps = 1.1920928955078125e-07;
        const double vn[3] = {fabs(X0) + fabs(Y0), fabs(X1) + fabs(Y1), fabs(X2) + fabs(Y2)};
        double E = 0., wmax = 0.;
        for (int k = 0; k < 3; k++) {
            double dinv = 0., wk = 0.;
            for (int j = 0; j < 3; j++) {
                dinv += fabs((double)g.inv[3 * k + j] - adj[3 * k + j] / det);
                wk += fabs((double)g.inv[3 * k + j]);
            }
            E += (dinv + 4. * eps * wk) * vn[k];
            wmax = fmax(wmax, wk);
        }
        E = 2. * E + 8. * eps * (1. + wmax) * (vn[0] + vn[1] + vn[2]);
        const double Rf = (double)cull_r * (1. + 1. / 1024.) + E;
        if (Rf == Rf && Rf < 1e30) {   // finite: otherwise keep the reference box only
            xhi = fminf(xhi, round_up((double)xmax + Rf));
            xlo = fmaxf(xlo, round_down((double)xmin - Rf));
            yhi = fminf(yhi, round_up((double)ymax + Rf));
            ylo = fmaxf(ylo, round_down((double)ymin - Rf));
        }
    }

    float4* b4 = reinterpret_cast<float4*>(boxes + i * 4);
    *b4 = make_float4(xlo, xhi, ylo, yhi);

    float* r = records + i * REC;
    r[kRecBox + 0] = xlo; r[kRecBox + 1] = xhi; r[kRecBox + 2] = ylo; r[kRecBox + 3] = yhi;
#pragma unroll
    for (int k = 0; k < 9; k++) r[kRecInv + k] = g.inv[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3;
        float a[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { a[j] = g.sym[3 * k + j] - g.sym[3 * k1 + j]; r[kRecEdge + 3 * k + j] = a[j]; }
        r[kRecDen + k] = a[k] - a[k1];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) r[kRecVert + k] = f[k];
    r[kRecBits] = __int_as_float(g.obt | (g.front << 3));
    r[kRecSpare] = 0.f;
    if (TEXM == kTexSurface1) {
        const long nxt = (i + 1 < total_faces) ? i + 1 : i;   // reference reads the next face's texel (:179-182); none after the last
#pragma unroll
        for (int k = 0; k < 3; k++) { r[kRecTex + k] = textures[i * 3 + k]; r[kRecTex + 3 + k] = textures[nxt * 3 + k]; }
        r[kRecTex + 6] = 0.f; r[kRecTex + 7] = 0.f;
    } else if (TEXM == kTexVertex) {
#pragma unroll
        for (int k = 0; k < 9; k++) r[kRecTex + k] = textures[i * 9 + k];
        r[kRecTex + 9] = 0.f; r[kRecTex + 10] = 0.f; r[kRecTex + 11] = 0.f;
    }
}

// faces_info in the reference's own layout [B*nf][27] (kernel.cu:620-676)
__global__ __launch_bounds__(kThreads) void face_info_kernel(const float* __restrict__ faces, float* __restrict__ info, long total_faces)
{
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total_faces) return;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = faces[i * 9 + k];
    FaceGeom g;
    face_geometry(f, g);
    float* o = info + i * 27;
#pragma unroll
    for (int k = 0; k < 9; k++) { o[k] = g.inv[k]; o[9 + k] = g.sym[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) o[18 + k] = (g.obt >> k) & 1 ? 1.f : 0.f;
#pragma unroll
    for (int k = 21; k < 27; k++) o[k] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// tile bookkeeping shared by forward and backward
// ---------------------------------------------------------------------------------------------
struct TileCtx {
    int   b;            // batch item
    int   tx0, ty0;     // first pixel column / image row of the tile
    int   xi, row;      // this lane's pixel
    bool  valid;        // pixel inside the image
    float xp, yp;       // pixel centre, kernel.cu:716-719
    long  pix;          // row * is + xi
    // wave quadrant rectangle in pixel-centre coordinates (inclusive)
    float qx_lo, qx_hi, qy_lo, qy_hi;
    // whole-tile rectangle
    float tx_lo, tx_hi, ty_lo, ty_hi;
};

__device__ __forceinline__ float pixel_coord(int idx, int is)
{
    return (float)((2. * idx + 1. - is) / is);   // (2.*xi + 1. - is) / is, kernel.cu:718-719
}

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).  Remap so that each XCD
// walks a contiguous range of tiles: all 256 tiles of one image then hit the same 4 MiB L2 for that
// image's face records.  Affects speed only.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int xcd = b & 7, idx = b >> 3, per = n >> 3, rem = n & 7;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

__device__ __forceinline__ void tile_setup(TileCtx& t, const RenderArgs& a)
{
    const int tile = xcd_remap(blockIdx.x, a.total_tiles);
    t.b = tile / a.tiles_per_image;
    const int tl = tile - t.b * a.tiles_per_image;
    const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
    t.tx0 = txi * kTile;
    t.ty0 = tyi * kTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qx0 = t.tx0 + (wave & 1) * 8, qy0 = t.ty0 + (wave >> 1) * 8;
    t.xi = qx0 + (lane & 7);
    t.row = qy0 + (lane >> 3);
    t.valid = t.xi < a.is && t.row < a.is;
    const int is = a.is;
    t.xp = pixel_coord(t.xi, is);
    t.yp = pixel_coord(is - 1 - t.row, is);   // yi = is - 1 - row, kernel.cu:716
    t.pix = (long)t.row * is + t.xi;
    t.qx_lo = pixel_coord(qx0, is);
    t.qx_hi = pixel_coord(min(qx0 + 7, is - 1), is);
    t.qy_hi = pixel_coord(is - 1 - qy0, is);
    t.qy_lo = pixel_coord(is - 1 - min(qy0 + 7, is - 1), is);
    t.tx_lo = pixel_coord(t.tx0, is);
    t.tx_hi = pixel_coord(min(t.tx0 + kTile - 1, is - 1), is);
    t.ty_hi = pixel_coord(is - 1 - t.ty0, is);
    t.ty_lo = pixel_coord(is - 1 - min(t.ty0 + kTile - 1, is - 1), is);
}

// box = (xlo, xhi, ylo, yhi).  A rectangle of pixel centres misses the box iff every centre fails the
// per-pixel test "x > xhi || x < xlo || y > yhi || y < ylo".
__device__ __forceinline__ bool rect_hits_box(float rx_lo, float rx_hi, float ry_lo, float ry_hi, const float4& box)
{
    return !(rx_lo > box.y || rx_hi < box.x || ry_lo > box.w || ry_hi < box.z);
}

struct TileLds {
    uint16_t list[kListCap];        // face indices relative to the scan range, ascending
    uint16_t wlist[4][kRecCap];     // per-wave record slots, ascending
    int      wave_tot[2][4];
};

// Scans faces [range0, range1) of batch item t.b; leaves the ascending list of faces whose box meets the
// tile rectangle in lds.list and returns its length (identical in every thread).
__device__ __forceinline__ int scan_faces(const RenderArgs& a, const TileCtx& t, TileLds& lds, int range0, int range1)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* boxes = reinterpret_cast<const float4*>(a.boxes) + (long)t.b * a.nf;
    int count = 0, parity = 0;
    for (int base = range0; base < range1; base += kThreads) {
        const int fi = base + threadIdx.x;
        bool hit = false;
        if (fi < range1) {
            const float4 box = boxes[fi];
            hit = a.p.cull ? rect_hits_box(t.tx_lo, t.tx_hi, t.ty_lo, t.ty_hi, box) : true;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) lds.wave_tot[parity][wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = lds.wave_tot[parity][w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (hit) {
            const int pos = count + before + __popcll(m & ((1ull << lane) - 1ull));
            lds.list[pos] = (uint16_t)(fi - range0);
        }
        count += total;
        parity ^= 1;
    }
    __syncthreads();
    return count;
}

// Copies records of list[c0 .. c0+n) into LDS, 16 bytes per lane per step.
template <int REC>
__device__ __forceinline__ void stage_records(const RenderArgs& a, const TileCtx& t, const TileLds& lds,
                                              float* s_rec, int range0, int c0, int n)
{
    constexpr int Q = REC / 4;   // float4 per record
    const float4* src = reinterpret_cast<const float4*>(a.records);
    float4* dst = reinterpret_cast<float4*>(s_rec);
    for (int e = threadIdx.x; e < n * Q; e += kThreads) {
        const int slot = e / Q, q = e - slot * Q;
        const long face = (long)t.b * a.nf + range0 + lds.list[c0 + slot];
        dst[slot * Q + q] = src[face * Q + q];
    }
}

// Each wave keeps the slots whose box meets its own 8x8 quadrant (ascending).  Returns the count.
template <int REC>
__device__ __forceinline__ int wave_refine(const RenderArgs& a, const TileCtx& t, TileLds& lds, const float* s_rec, int n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wn = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        bool hit = false;
        if (s < n) {
            const float4 box = *reinterpret_cast<const float4*>(s_rec + s * REC + kRecBox);
            hit = a.p.cull ? rect_hits_box(t.qx_lo, t.qx_hi, t.qy_lo, t.qy_hi, box) : true;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) lds.wlist[wave][wn + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
        wn += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    return wn;
}

// ---------------------------------------------------------------------------------------------
// one (pixel, face) evaluation: kernel.cu:747-786 (forward) == :924-962 (backward)
// ---------------------------------------------------------------------------------------------
struct Pair {
    float w[3];        // barycentrics (:39-43)
    float t[3];        // t - w of the closest boundary point (:103-105,:157)
    float sign, dx, dy, dis, frag;
};

__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// kernel.cu:76-165 on the staged record.  Returns false when the pair must be dropped (NaN barycentrics:
// the reference indexes with v0 = -1 there; DESIGN.md quirk iv).
__device__ __forceinline__ bool point_to_face(Pair& q, const float* __restrict__ r, float xp, float yp)
{
    const float w0 = q.w[0], w1 = q.w[1], w2 = q.w[2];
    const float x0 = r[kRecVert + 0], y0 = r[kRecVert + 1], x1 = r[kRecVert + 3], y1 = r[kRecVert + 4],
                x2 = r[kRecVert + 6], y2 = r[kRecVert + 7];
    if (w0 > 0 && w1 > 0 && w2 > 0 && w0 < 1 && w1 < 1 && w2 < 1) {
        float best = 100000000.f, bx = 0.f, by = 0.f, bt0 = 0.f, bt1 = 0.f, bt2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int k1 = (k + 1) % 3;
            const float* A = r + kRecEdge + 3 * k;
            const float tv = (w0 * A[0] + w1 * A[1] + w2 * A[2] - A[k1]) / r[kRecDen + k];
            float t0[3];
            t0[k] = tv;
            t0[k1] = 1 - tv;
            t0[(k + 2) % 3] = 0;
            t0[0] -= w0; t0[1] -= w1; t0[2] -= w2;
            const float ddx = t0[0] * x0 + t0[1] * x1 + t0[2] * x2;
            const float ddy = t0[0] * y0 + t0[1] * y1 + t0[2] * y2;
            const float d = ddx * ddx + ddy * ddy;
            if (d < best) { best = d; bx = ddx; by = ddy; bt0 = t0[0]; bt1 = t0[1]; bt2 = t0[2]; }
        }
        q.dx = bx; q.dy = by; q.sign = 1.f;
        q.t[0] = bt0; q.t[1] = bt1; q.t[2] = bt2;
        return true;
    }
    const int bits = __float_as_int(r[kRecBits]);
    int v0 = -1;
    if (w1 <= 0 && w2 <= 0) {
        v0 = 0;
        if ((bits & 1) && (xp - x0) * (x2 - x0) + (yp - y0) * (y2 - y0) > 0) v0 = 2;
    } else if (w2 <= 0 && w0 <= 0) {
        v0 = 1;
        if ((bits & 2) && (xp - x1) * (x0 - x1) + (yp - y1) * (y0 - y1) > 0) v0 = 0;
    } else if (w0 <= 0 && w1 <= 0) {
        v0 = 2;
        if ((bits & 4) && (xp - x2) * (x1 - x2) + (yp - y2) * (y1 - y2) > 0) v0 = 1;
    } else if (w0 <= 0) v0 = 1;
    else if (w1 <= 0) v0 = 2;
    else if (w2 <= 0) v0 = 0;
    if (v0 < 0) {
        if (w0 != w0 || w1 != w1 || w2 != w2) return false;
        int m = 0; float wm = w0;
        if (w1 < wm) { m = 1; wm = w1; }
        if (w2 < wm) { m = 2; }
        v0 = (m + 1) % 3;
    }
    const int v1 = v0 == 2 ? 0 : v0 + 1;
    const float A0 = sel3(v0, r[kRecEdge + 0], r[kRecEdge + 3], r[kRecEdge + 6]);
    const float A1 = sel3(v0, r[kRecEdge + 1], r[kRecEdge + 4], r[kRecEdge + 7]);
    const float A2 = sel3(v0, r[kRecEdge + 2], r[kRecEdge + 5], r[kRecEdge + 8]);
    const float Av1 = sel3(v1, A0, A1, A2);
    const float den = sel3(v0, r[kRecDen + 0], r[kRecDen + 1], r[kRecDen + 2]);
    const float tv = (w0 * A0 + w1 * A1 + w2 * A2 - Av1) / den;
    const float tv0 = fminf(fmaxf(tv, 0.f), 1.f);          // min(max(t, 0.), 1.) : clamp is exact in either precision
    const float tv1 = fminf(fmaxf(1 - tv, 0.f), 1.f);
    // t[v0] = tv0, t[v1] = tv1, t[v2] = clamp(0) = 0 ; then t[k] -= w[k]
    const float t0 = (v0 == 0 ? tv0 : (v1 == 0 ? tv1 : 0.f)) - w0;
    const float t1 = (v0 == 1 ? tv0 : (v1 == 1 ? tv1 : 0.f)) - w1;
    const float t2 = (v0 == 2 ? tv0 : (v1 == 2 ? tv1 : 0.f)) - w2;
    q.t[0] = t0; q.t[1] = t1; q.t[2] = t2;
    q.dx = t0 * x0 + t1 * x1 + t2 * x2;
    q.dy = t0 * y0 + t1 * y1 + t2 * y2;
    q.sign = -1.f;
    return true;
}

__device__ __forceinline__ bool inside_closed(const float* w)
{
    return w[0] <= 1 && w[0] >= 0 && w[1] <= 1 && w[1] >= 0 && w[2] <= 1 && w[2] >= 0;   // :62-64
}

// Returns true if the pair contributes (none of the skips at :747, :769, :784 fires).
template <int DIST, int SQ>
__device__ __forceinline__ bool eval_pair(Pair& q, const float* __restrict__ r, float xp, float yp,
                                          const RenderArgs& a, const DistParams& dp)
{
    const float4 box = *reinterpret_cast<const float4*>(r + kRecBox);
    if (xp > box.y || xp < box.x || yp > box.w || yp < box.z) return false;
    q.w[0] = r[kRecInv + 0] * xp + r[kRecInv + 1] * yp + r[kRecInv + 2];
    q.w[1] = r[kRecInv + 3] * xp + r[kRecInv + 4] * yp + r[kRecInv + 5];
    q.w[2] = r[kRecInv + 6] * xp + r[kRecInv + 7] * yp + r[kRecInv + 8];
    const int dist = DIST >= 0 ? DIST : a.p.dist_func;
    if (dist == kHeaviside) {
        q.sign = 0.f; q.dx = 0.f; q.dy = 0.f; q.dis = 0.f; q.t[0] = q.t[1] = q.t[2] = 0.f;
        q.frag = inside_closed(q.w) ? 1.f : 0.f;                                    // :762-764
    } else {
        if (!point_to_face(q, r, xp, yp)) return false;
        float dis = q.dx * q.dx + q.dy * q.dy;                                      // :768
        if (q.sign < 0 && dis >= a.thr) return false;                               // :769
        const bool squared = SQ >= 0 ? (SQ != 0) : (a.p.dist_squared != 0);
        if (!squared) dis = sqrtf(dis);                                             // :770-772
        q.dis = dis;
        if constexpr (DIST >= 0) q.frag = Dist<(DIST >= 0 ? DIST : 0)>::cdf(q.sign, dis, dp);
        else                     q.frag = cdf_rt(dist, q.sign, dis, dp);
    }
    return !((double)q.frag <= kProbThreshold);                                     // :784
}

// barycentric_clip + depth, kernel.cu:68-72, :807-810
__device__ __forceinline__ float clip_and_depth(const Pair& q, const float* __restrict__ r, float* wc)
{
#pragma unroll
    for (int k = 0; k < 3; k++) wc[k] = fmaxf(fminf(q.w[k], 1.f), 0.f);
    float s = wc[0] + wc[1] + wc[2];
    s = ((double)s > 1e-5) ? s : (float)1e-5;              // max(sum, 1e-5) with a double literal, stored as float
#pragma unroll
    for (int k = 0; k < 3; k++) wc[k] /= s;
    return 1.f / (wc[0] / r[kRecVert + 2] + wc[1] / r[kRecVert + 5] + wc[2] / r[kRecVert + 8]);   // "1. /": one rounding
}

// surface texel index for clipped barycentrics (kernel.cu:179-185); may be >= T (reference quirk)
__device__ __forceinline__ int texel_index(const float* wc, int R, bool clamp)
{
    int wx = (int)(wc[0] * R), wy = (int)(wc[1] * R);
    if (clamp) { wx = min(wx, R - 1); wy = min(wy, R - 1); }
    if ((wc[0] + wc[1]) * R - wx - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}

// Resolves the texel a pair reads in surface mode with T > 1.  `own` receives the in-face texel index when
// the reference lets gradient flow to it (backward_sample_texture only matches j < T), else -1.
__device__ __forceinline__ long resolve_texel(const float* wc, const RenderArgs& a, long face_lin, int& own)
{
    const long total = (long)a.B * a.nf * a.T;
    if (a.p.texel_mode == 1) {
        int idx = texel_index(wc, a.R, true);
        idx = max(0, min(idx, a.T - 1));
        own = idx;
        return face_lin * a.T + idx;
    }
    int idx = texel_index(wc, a.R, false);
    long at = face_lin * a.T + idx;
    if (at >= total || at < 0) {
        idx = texel_index(wc, a.R, true);
        idx = max(0, min(idx, a.T - 1));
        own = -1;
        return face_lin * a.T + idx;
    }
    own = (idx >= 0 && idx < a.T) ? idx : -1;
    return at;
}

// colour sampled for a pair.  TEXM 0: own or next-face texel out of the record.
template <int TEXM>
__device__ __forceinline__ void sample_colour(float* c, int& own, const float* wc, const float* __restrict__ r,
                                              const RenderArgs& a, long face_lin)
{
    if (TEXM == kTexSurface1) {
        int idx = texel_index(wc, 1, a.p.texel_mode == 1);
        const bool last = face_lin + 1 >= (long)a.B * a.nf;
        own = (idx == 0) ? 0 : -1;
        if (a.p.texel_mode == 1) { idx = 0; own = 0; }
        if (idx != 0 && last) idx = 0;                      // nothing after the last face: own texel, still no gradient
        // idx is 0 (own) or 1 (next face) for R == 1; a negative index cannot occur here
        const float* tx = r + kRecTex + (idx != 0 ? 3 : 0);
        c[0] = tx[0]; c[1] = tx[1]; c[2] = tx[2];
    } else if (TEXM == kTexVertex) {
        own = 0;
        const float* tx = r + kRecTex;
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = wc[0] * tx[k] + wc[1] * tx[3 + k] + wc[2] * tx[6 + k];   // :187-189
    } else {
        const long at = resolve_texel(wc, a, face_lin, own);
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = a.textures[at * 3 + k];
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) void render_forward_kernel(const RenderArgs a)
{
    constexpr int REC = record_floats(TEXM);
    __shared__ __attribute__((aligned(16))) float s_rec[kRecCap * REC];
    __shared__ TileLds lds;

    TileCtx t;
    tile_setup(t, a);
    const int wave = threadIdx.x >> 6;
    const long P = (long)a.is * a.is;
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift};
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);
    const float gam = a.p.aggr_rgb_gamma;

    // per-pixel state, kernel.cu:728-740
    float bg[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        bg[k] = (a.p.background_from_buffer && t.valid) ? a.rgba[((long)t.b * 4 + k) * P + t.pix] : a.p.background[k];
    float alpha = 0.f;
    float ssum = a.softmax_sum0, smax = a.p.aggr_rgb_eps;
    float col[3];
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = rgb_soft ? bg[k] * ssum : bg[k];
    float depth_min = 10000000.f;
    int face_min = -1;

    for (int range0 = 0; range0 < a.nf; range0 += kListCap) {
        const int range1 = min(a.nf, range0 + kListCap);
        const int count = scan_faces(a, t, lds, range0, range1);
        for (int c0 = 0; c0 < count; c0 += kRecCap) {
            const int n = min(kRecCap, count - c0);
            if (c0 > 0) __syncthreads();                 // previous chunk fully consumed
            stage_records<REC>(a, t, lds, s_rec, range0, c0, n);
            __syncthreads();
            const int wn = wave_refine<REC>(a, t, lds, s_rec, n);

            for (int i = 0; i < wn; i++) {
                const int slot = lds.wlist[wave][i];
                const float* r = s_rec + slot * REC;
                Pair q;
                if (!t.valid) continue;
                if (!eval_pair<DIST, SQ>(q, r, t.xp, t.yp, a, dp)) continue;

                // alpha, kernel.cu:791-803
                if (alpha_func == kAlphaHard) {
                    if ((double)q.frag > 0.5) alpha = 1.f;
                } else if constexpr (ALPHA > 0) {
                    alpha = TConorm<(ALPHA > 0 ? ALPHA : 1)>::fold(alpha, q.frag, a.p.aggr_alpha_t_conorm_p);
                } else {
                    alpha = tconorm_fold_rt(alpha_func, alpha, q.frag, a.p.aggr_alpha_t_conorm_p);
                }

                float wc[3];
                const float zp = clip_and_depth(q, r, wc);
                if (zp < a.p.near_ || zp > a.p.far_) continue;                       // :810

                const int fn = range0 + lds.list[c0 + slot];
                const long face_lin = (long)t.b * a.nf + fn;
                const bool front = (__float_as_int(r[kRecBits]) & 8) != 0;
                if (!rgb_soft) {                                                     // :815-822
                    if (zp < depth_min && inside_closed(q.w) && (a.p.double_side || front)) {
                        depth_min = zp;
                        face_min = fn;
                        int own;
                        sample_colour<TEXM>(col, own, wc, r, a, face_lin);
                    }
                } else if (front || a.p.double_side) {                               // :824-838
                    const float zn = (a.p.far_ - zp) / (a.p.far_ - a.p.near_);
                    float edz = 1.f;
                    if (zn > smax) {
                        edz = expf((smax - zn) / gam);
                        smax = zn;
                    }
                    const float ez = expf((zn - smax) / gam);
                    ssum = edz * ssum + ez * q.frag;
                    float c[3]; int own;
                    sample_colour<TEXM>(c, own, wc, r, a, face_lin);
#pragma unroll
                    for (int k = 0; k < 3; k++) col[k] = edz * col[k] + ez * q.frag * c[k];
                }
            }
        }
        if (range1 < a.nf) __syncthreads();              // list is rebuilt by the next range
    }

    if (!t.valid) return;
    // epilogue, kernel.cu:845-861
    float* out = a.rgba + (long)t.b * 4 * P + t.pix;
    float* aux = a.aux + (long)t.b * 2 * P + t.pix;
    out[3 * P] = alpha;
    if (!rgb_soft) {
#pragma unroll
        for (int k = 0; k < 3; k++) out[k * P] = (face_min != -1) ? col[k] : bg[k];
        aux[0] = depth_min;
        aux[P] = (float)face_min;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) out[k * P] = col[k] / ssum;
        aux[0] = ssum;
        aux[P] = smax;
    }
}

// ---------------------------------------------------------------------------------------------
// wavefront sum with DPP adds (result valid in every lane via readlane 63)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
    // quad swaps, half-row mirror, row mirror, then the two cross-row broadcasts of GFX9
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true));   // row_bcast:15 -> rows 1,3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true));   // row_bcast:31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int TEXM> struct GradSlots { static constexpr int n = TEXM == kTexSurface1 ? 12 : (TEXM == kTexVertex ? 18 : 9); };

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) void render_backward_kernel(const RenderArgs a)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NG = GradSlots<TEXM>::n;       // 9 vertex components, then texture components kept in LDS
    __shared__ __attribute__((aligned(16))) float s_rec[kRecCap * REC];
    __shared__ float s_acc[kRecCap * NG];
    __shared__ TileLds lds;

    TileCtx t;
    tile_setup(t, a);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long P = (long)a.is * a.is;
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift};
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const int dist = DIST >= 0 ? DIST : a.p.dist_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);
    const bool squared = SQ >= 0 ? (SQ != 0) : (a.p.dist_squared != 0);
    const float gam = a.p.aggr_rgb_gamma;

    // per-pixel inputs, kernel.cu:916-917, :973, :980, :1013, :1021
    float g[4] = {0.f, 0.f, 0.f, 0.f}, out[4] = {0.f, 0.f, 0.f, 0.f}, ssum = 1.f, smax = 0.f;
    if (t.valid) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            g[k] = a.grad_rgba[((long)t.b * 4 + k) * P + t.pix];
            out[k] = a.rgba[((long)t.b * 4 + k) * P + t.pix];
        }
        ssum = a.aux[((long)t.b * 2 + 0) * P + t.pix];
        smax = a.aux[((long)t.b * 2 + 1) * P + t.pix];
    }

    for (int range0 = 0; range0 < a.nf; range0 += kListCap) {
        const int range1 = min(a.nf, range0 + kListCap);
        const int count = scan_faces(a, t, lds, range0, range1);
        for (int c0 = 0; c0 < count; c0 += kRecCap) {
            const int n = min(kRecCap, count - c0);
            if (c0 > 0) __syncthreads();
            stage_records<REC>(a, t, lds, s_rec, range0, c0, n);
            for (int e = threadIdx.x; e < n * NG; e += kThreads) s_acc[e] = 0.f;
            __syncthreads();
            const int wn = wave_refine<REC>(a, t, lds, s_rec, n);

            for (int i = 0; i < wn; i++) {
                const int slot = lds.wlist[wave][i];
                const float* r = s_rec + slot * REC;
                const int fn = range0 + lds.list[c0 + slot];
                const long face_lin = (long)t.b * a.nf + fn;

                float gv[9];                       // d loss / d (x,y,z) of the 3 vertices, kernel.cu:967
                float gt[NG > 9 ? NG - 9 : 1];     // texture partials kept in LDS
#pragma unroll
                for (int k = 0; k < 9; k++) gv[k] = 0.f;
#pragma unroll
                for (int k = 0; k < (NG > 9 ? NG - 9 : 1); k++) gt[k] = 0.f;

                Pair q;
                bool live = t.valid && eval_pair<DIST, SQ>(q, r, t.xp, t.yp, a, dp);
                if (live) {
                    // alpha partial, kernel.cu:973-987 (hard alpha leaves g[3] unscaled, as the reference does)
                    float C_xy = 0.f;
                    float C_alpha = g[3];
                    if (alpha_func != kAlphaHard) {
                        if constexpr (ALPHA > 0) C_alpha *= TConorm<(ALPHA > 0 ? ALPHA : 1)>::grad(out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
                        else                     C_alpha *= tconorm_grad_rt(alpha_func, out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
                    }
                    C_xy += C_alpha;

                    float wc[3];
                    const float zp = clip_and_depth(q, r, wc);
                    live = !(zp < a.p.near_ || zp > a.p.far_);                      // :994 drops the whole pair
                    if (live) {
                        const bool front = (__float_as_int(r[kRecBits]) & 8) != 0;
                        if (!rgb_soft) {                                            // :997-1004
                            if ((float)fn == smax) {
                                if constexpr (TEXM == kTexVertex) {
#pragma unroll
                                    for (int k = 0; k < 3; k++)
#pragma unroll
                                        for (int j = 0; j < 3; j++) gt[3 * j + k] = wc[j] * g[k];
                                } else {
                                    float c[3]; int own;
                                    sample_colour<TEXM>(c, own, wc, r, a, face_lin);
                                    if (own >= 0) {
                                        if constexpr (TEXM == kTexSurface1) {
#pragma unroll
                                            for (int k = 0; k < 3; k++) gt[k] = g[k];
                                        } else {
#pragma unroll
                                            for (int k = 0; k < 3; k++)
                                                unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + own) * 3 + k, g[k]);
                                        }
                                    }
                                }
                            }
                        } else if (front || a.p.double_side) {                      // :1006-1030
                            const float zn = (a.p.far_ - zp) / (a.p.far_ - a.p.near_);
                            const float zs = q.frag * expf((zn - smax) / gam) / ssum;   // :1010
                            float c[3]; int own;
                            sample_colour<TEXM>(c, own, wc, r, a, face_lin);
                            float C_rgb = 0.f;
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                if constexpr (TEXM == kTexVertex) {
#pragma unroll
                                    for (int j = 0; j < 3; j++) gt[3 * j + k] = zs * (wc[j] * g[k]);
                                } else if constexpr (TEXM == kTexSurface1) {
                                    if (own >= 0) gt[k] = zs * g[k];
                                } else {
                                    if (own >= 0) unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + own) * 3 + k, zs * g[k]);
                                }
                                C_rgb += g[k] * (c[k] - out[k]);                    // :1021
                            }
                            C_rgb *= zs;                                            // :1023
                            C_xy += C_rgb / q.frag;                                 // :1024
                            const float C_z = C_rgb / gam / (a.p.near_ - a.p.far_) * zp * zp;   // :1026
                            gv[2] = C_z * wc[0] / r[kRecVert + 2] / r[kRecVert + 2];
                            gv[5] = C_z * wc[1] / r[kRecVert + 5] / r[kRecVert + 5];
                            gv[8] = C_z * wc[2] / r[kRecVert + 8] / r[kRecVert + 8];
                        }

                        // distance gradient, kernel.cu:1034-1052.  Heaviside: D' = 0 times uninitialised
                        // values in the reference -> defined as exactly 0 here (DESIGN.md quirk i).
                        if (dist != kHeaviside) {
                            if constexpr (DIST >= 0) C_xy *= Dist<(DIST >= 0 ? DIST : 0)>::pdf(q.sign, q.dis, dp);
                            else                     C_xy *= pdf_rt(dist, q.sign, q.dis, dp);
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const float wk = q.t[k] + q.w[k];
                                if (squared) {
                                    gv[3 * k + 0] = 2 * q.sign * C_xy * wk * q.dx;
                                    gv[3 * k + 1] = 2 * q.sign * C_xy * wk * q.dy;
                                } else {
                                    const double nrm = fmax((double)sqrtf(q.dx * q.dx + q.dy * q.dy), 1e-6);
                                    gv[3 * k + 0] = (float)((double)(q.sign * C_xy * wk * q.dx) / nrm);
                                    gv[3 * k + 1] = (float)((double)(q.sign * C_xy * wk * q.dy) / nrm);
                                }
                            }
                        }
                    }
                }
                if (!live) {
#pragma unroll
                    for (int k = 0; k < 9; k++) gv[k] = 0.f;
#pragma unroll
                    for (int k = 0; k < (NG > 9 ? NG - 9 : 1); k++) gt[k] = 0.f;
                }
                if (!__any(live)) continue;
                // wavefront reduction, then one LDS atomic per component from lane 0
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const float s = wave_sum(gv[k]);
                    if (lane == 0 && s != 0.f) atomicAdd(&s_acc[slot * NG + k], s);
                }
#pragma unroll
                for (int k = 0; k < NG - 9; k++) {
                    const float s = wave_sum(gt[k]);
                    if (lane == 0 && s != 0.f) atomicAdd(&s_acc[slot * NG + 9 + k], s);
                }
            }

            __syncthreads();
            // flush the tile's partial sums: one hardware fp32 atomic per (face, component)
            for (int e = threadIdx.x; e < n * NG; e += kThreads) {
                const float v = s_acc[e];
                if (v != 0.f) {
                    const int slot = e / NG, k = e - slot * NG;
                    const long face_lin = (long)t.b * a.nf + range0 + lds.list[c0 + slot];
                    if (k < 9) unsafeAtomicAdd(a.grad_faces + face_lin * 9 + k, v);
                    else       unsafeAtomicAdd(a.grad_textures + face_lin * (NG - 9) + (k - 9), v);
                }
            }
        }
        if (range1 < a.nf) __syncthreads();
    }
}

}  // namespace gendr
