// gendr_kernels.h -- CDNA4 (gfx950) kernels of the generalized soft rasterizer.
//
// Replaces the three __global__ kernels of the reference
// (gendr/cuda/generalized_renderer_cuda_kernel.cu = "kernel.cu"):
//   forward_render_inv_cuda_kernel :620-676  ->  face_setup_kernel (+ face_info_kernel, reference layout)
//                                                bin_faces_kernel + cover_kernel (new: exact tile culling, per-pixel coverage),
//                                                order_tiles_kernel (heavy tiles first), loose_faces_kernel
//   forward_render_cuda_kernel     :680-862  ->  render_forward_kernel
//   backward_render_cuda_kernel    :866-1065 ->  render_backward_kernel
//
// Design (DESIGN.md has the long form and the measurements behind it):
//   * one wavefront = one 8x8 pixel tile.  Exact tile culling: the face-setup kernel writes, per face, a conservative
//     box outside of which the reference itself would skip the pair (kernel.cu:747,769,784); the binning kernel
//     tests those boxes against every tile rectangle, leaves one bit per (tile, face) in HBM -- ascending face order
//     for free, shared by forward and backward -- and queues the tiles that list anything (8 queues, one per XCD);
//   * the coverage kernel (one wave per listed tile, sixteen faces per step, lane = (face, two pixel rows)) applies the exact
//     per-pixel tests once and leaves, per tile, the list of (face, 64-bit pixel mask) entries that own at least one
//     pixel -- shared by forward and backward, so neither render kernel looks at a face record before it has to;
//   * the render kernels walk the tile queues.  Per tile they read the entry list (one coalesced load per 64 entries),
//     append the (pixel, face) pairs to a wave-private LDS list and run
//     phase B dense -- lane = pair -- over batches of 64 pairs: distance, CDF, depth, colour (+ gradients);
//     phase C (forward) folds the results per pixel in ascending face order, so the alpha fold and the online
//     softmax keep the reference's order without any cross-lane combination; backward instead sums each face's
//     partials over its pairs from a padded LDS matrix and issues one hardware fp32 atomic per (batch, face,
//     component) instead of 12..84 per (pixel, face);
//   * inside the loop every pair still passes the reference's own three skip tests, so culling only removes pairs
//     that contribute exactly nothing;
//   * pair hints (round 3): the forward kernel leaves two bits per evaluated pair -- the edge its closest-point search
//     selected, or "no gradient" -- and the backward kernel evaluates that one edge instead of searching again (same
//     operations for that edge: bit-identical values), skipping batches and tiles that hold no live pair;
//   * a face whose cull box is loose (seen edge-on: no error bound) is evaluated by the coverage kernel on the pixels of every
//     tile it is listed in and keeps exactly the pixels that can contribute; in images of 1024^2 and more loose_faces_kernel
//     first evaluates it on every pixel once and the binning kernel lists it by the box of the pixels that can contribute;
//   * everything is wave-local: no workgroup barriers in the render kernels, one wave-tile per workgroup.
//
// No MFMA: there is no dense contraction in this path.  Compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gendr_hip.h"
#include "gendr_math.h"

// register budgets of the occupancy-capped kernel variants (waves per SIMD)
#ifndef GENDR_FWD_WAVES
#define GENDR_FWD_WAVES 7
#endif
#ifndef GENDR_FULL_WAVES
#define GENDR_FULL_WAVES 2
#endif
#ifndef GENDR_HALPHA_WAVES
#define GENDR_HALPHA_WAVES 4
#endif
#ifndef GENDR_LIGHT_FWD_WAVES
#define GENDR_LIGHT_FWD_WAVES 5
#endif
#ifndef GENDR_LIGHT_BWD_WAVES
#define GENDR_LIGHT_BWD_WAVES 4
#endif
#ifndef GENDR_BWD_WAVES
#define GENDR_BWD_WAVES 4
#endif


#ifndef GENDR_TRACE
#define GENDR_TRACE 0          // 1: every wave of the backward kernel leaves time stamps (diagnostic build, tools/wave_trace.py)
#endif
#if GENDR_TRACE
// per wave: start, queue lengths known, pixel inputs parked, first batch entered, end (shader clock of its CU), batches,
// start and end on the chip-wide 100 MHz clock (s_memrealtime)
__device__ unsigned long long g_wave_trace[1 << 17][8];
#define GENDR_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); tr[i] = __builtin_readcyclecounter(); } while (0)
// start and end (chip-wide clock) of every wave of the coverage (0) and forward (2) kernels and of every workgroup of the
// binning kernel (1)
__device__ unsigned long long g_span_trace[3][1 << 16][2];
#define GENDR_SPAN_BEGIN const unsigned long long span0_ = __builtin_amdgcn_s_memrealtime()
#define GENDR_SPAN_END(k, idx) do { __builtin_amdgcn_s_waitcnt(0); \
        if ((threadIdx.x & 63) == 0 && (unsigned)(idx) < (1u << 16)) { g_span_trace[k][idx][0] = span0_; g_span_trace[k][idx][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GENDR_STAMP(i) do {} while (0)
#define GENDR_SPAN_BEGIN do {} while (0)
#define GENDR_SPAN_END(k, idx) do {} while (0)
#endif
#ifndef GENDR_TIMERS
#define GENDR_TIMERS 0         // 1: the backward kernel accumulates its wave-time per phase (diagnostic build, tools/phase_timers.py)
#endif
#if GENDR_TIMERS
#define GENDR_T(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define GENDR_T(i) do {} while (0)
#endif

#ifndef GENDR_BIN_WAVES
#define GENDR_BIN_WAVES 8     // waves per SIMD the binning kernel is compiled for: 8 = four 8-wave workgroups per CU (its 106 scalar registers allowed three)
#endif
#ifndef GENDR_LOOSE_AREA
#define GENDR_LOOSE_AREA 1.0f    // visible NDC area of a cull box from which face_setup_kernel calls it loose (the image has 4)
#endif
#ifndef GENDR_LOOSE_MIN_TILES
#define GENDR_LOOSE_MIN_TILES 16384   // tiles per image from which loose_faces_kernel narrows the boxes of flagged faces ahead of the binning kernel
#endif
#ifndef GENDR_LOOSE_FACES
#define GENDR_LOOSE_FACES 1    // 0: faces without a usable error bound keep the reference's cull box (A/B builds)
#endif
#ifndef GENDR_PAIR_HINTS
#define GENDR_PAIR_HINTS 1   // 0: no pair hints from the forward to the backward kernel (A/B builds)
#endif
#ifndef GENDR_RELOAD_ARGS
#define GENDR_RELOAD_ARGS 1
#endif
// The kernel's arguments re-read from the kernarg segment at the top of a tile iteration (scalar loads through a pointer the compiler cannot
// see through) instead of living in scalar registers from the kernel's first instruction on: the tile kernels need more scalar values than
// the 102 registers hold, and the allocator's answer -- v_writelane at the start, v_readlane per tile -- is VECTOR instructions in kernels
// bound by their vector issue.  Round 6: static vector instructions of C2's forward / backward kernel 976 -> 852 / 887 -> 809 (150 / 80 of
// them lane moves), forward phase 0.133 -> 0.130 ms, backward 0.095 -> 0.092 ms.  The struct is the kernel's only parameter: offset 0.
#if GENDR_RELOAD_ARGS
#define GENDR_RELOADED_ARGS(a) \
    const GENDR_CONST_AS RenderArgs* ap_ = (const GENDR_CONST_AS RenderArgs*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(ap_)); \
    const RenderArgs& a = *(const RenderArgs*)ap_
#else
#define GENDR_RELOADED_ARGS(a) do {} while (0)
#endif
#ifndef GENDR_ABLATE
#define GENDR_ABLATE 0   // diagnostic builds only (tools/): 1, 2, 6, 7 cut the forward batch body short after a stage, 5 drops the fill (tools/fwd_phases.sh)
#endif

namespace gendr {

// Set bits of a WAVE-UNIFORM 64-bit mask below this lane (plus `add`): v_mbcnt_lo / v_mbcnt_hi with the mask in scalar registers -- two vector
// instructions where __popcll(m & lanes_below) costs two ANDs and two bit counts on a lane-mask register pair; and "is this lane's bit set"
// as an exec mask straight from the scalar mask (no vector compare).  Round 6: 6 of the 13 vector instructions of the render kernels' append step.
__device__ __forceinline__ int bits_below(unsigned long long m, int add = 0)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)add));
}
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// wave-uniform read-only data: loads through this pointer type with a uniform address become s_load_*
#define GENDR_CONST_AS __attribute__((address_space(4)))
typedef const GENDR_CONST_AS float* RecPtr;
typedef float f16v __attribute__((ext_vector_type(16), aligned(4)));
typedef float f8v  __attribute__((ext_vector_type(8), aligned(4)));
typedef float f4v  __attribute__((ext_vector_type(4), aligned(4)));

typedef float f2v  __attribute__((ext_vector_type(2), aligned(4)));
typedef int   i4v  __attribute__((ext_vector_type(4)));
typedef int   i16v __attribute__((ext_vector_type(16), aligned(16)));

// Floats [BEGIN, END) of a face record -> dst[BEGIN..END) with the widest scalar loads that fit
// (s_load_dwordx16 / x8 / x4 / x2): one wait per stage instead of one per field.
template <int BEGIN, int END>
__device__ __forceinline__ void load_record(float* dst, RecPtr r)
{
    if constexpr (END - BEGIN >= 16) {
        const f16v v = *reinterpret_cast<const GENDR_CONST_AS f16v*>(r + BEGIN);
#pragma unroll
        for (int k = 0; k < 16; k++) dst[BEGIN + k] = v[k];
        load_record<BEGIN + 16, END>(dst, r);
    } else if constexpr (END - BEGIN >= 8) {
        const f8v v = *reinterpret_cast<const GENDR_CONST_AS f8v*>(r + BEGIN);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[BEGIN + k] = v[k];
        load_record<BEGIN + 8, END>(dst, r);
    } else if constexpr (END - BEGIN >= 4) {
        const f4v v = *reinterpret_cast<const GENDR_CONST_AS f4v*>(r + BEGIN);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[BEGIN + k] = v[k];
        load_record<BEGIN + 4, END>(dst, r);
    } else if constexpr (END - BEGIN >= 2) {
        const f2v v = *reinterpret_cast<const GENDR_CONST_AS f2v*>(r + BEGIN);
        dst[BEGIN] = v[0]; dst[BEGIN + 1] = v[1];
        load_record<BEGIN + 2, END>(dst, r);
    } else if constexpr (END - BEGIN == 1) {
        dst[BEGIN] = r[BEGIN];
    }
}

// a wave-uniform record address as a scalar-register pointer (the compiler cannot always prove the uniformity itself)
__device__ __forceinline__ RecPtr uniform_rec_ptr(const float* p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<RecPtr>(((unsigned long long)hi << 32) | lo);
}

// reciprocal of a per-face divisor in two consecutive floats of the face record (see div_by() in gendr_math.h): the
// correctly rounded double; fast build: its float rounding in the first of the two
__device__ __forceinline__ rcp_t rec_rcp(const float* r, int k)
{
#if GENDR_FAST_DEV
    return r[k];
#else
    return __hiloint2double(__float_as_int(r[k + 1]), __float_as_int(r[k]));
#endif
}
__device__ __forceinline__ void put_rcp(float* r, int k, double v)
{
#if GENDR_FAST_DEV
    r[k] = (float)v; r[k + 1] = 0.f;
#else
    r[k] = __int_as_float(__double2loint(v)); r[k + 1] = __int_as_float(__double2hiint(v));
#endif
}

// ---------------------------------------------------------------------------------------------
// face record layout (floats).  Geometry part is common; the tail depends on the texture mode.
// ---------------------------------------------------------------------------------------------
// Ordered by the stage of the inner loop that consumes it, so that each stage is one contiguous scalar load:
//   stage 1 [0,20)   box + barycentric matrix + flags + edge cull -> per-lane box / edge tests, barycentrics
//   stage 2 [20,42)  edge data + vertex x,y                     -> point-to-triangle distance
//   stage 3 [42,REC) reciprocal vertex depths + texels          -> depth, colour
// Divisors that are uniform over the wavefront are stored as correctly rounded DOUBLE reciprocals
// (see div_by()).
constexpr int kRecBox   = 0;    // xlo, xhi, ylo, yhi : pixel centres outside are skipped
constexpr int kRecInv   = 4;    // inv[9]   (kernel.cu:645-657)
constexpr int kRecBits  = 13;   // int bits: 1,2,4 = first obtuse corner 0,1,2 (:667-675); 8 = front side (:56-58)
constexpr int kRecWCull = 14;   // wcull[3]: a pixel whose computed barycentric w_k is below wcull_k lies farther than
                                // the cull radius beyond the edge opposite vertex k -> the reference skips the pair
constexpr int kRecEdge  = 20;   // A[3][3]  A[k][j] = sym[k][j] - sym[(k+1)%3][j]  (kernel.cu:95-97,146-148)
constexpr int kRecRDen  = 30;   // 3 doubles: 1 / Dn[k],  Dn[k] = A[k][k] - A[k][(k+1)%3]  (denominator of :99,:150)
constexpr int kRecXY    = 36;   // x0 y0 x1 y1 x2 y2
constexpr int kRecRZ    = 42;   // 3 doubles: 1 / z_k   (:809, :1027-1029)
constexpr int kRecTex   = 48;   // TEXM 0: own rgb, next-face rgb ; TEXM 1: 3 vertex colours ; TEXM 2: nothing
constexpr int kLooseList = 16;         // ints per image in RenderArgs::loose_image: stamped counter + up to 15 flagged faces
constexpr int kLooseWaves = 2048;      // grid of loose_faces_kernel (one-wave workgroups)
constexpr int kRecLoose = 17;  // int bits: 1 = the cull box is loose (no usable error bound): the coverage kernel finds the face's pixels by evaluation
constexpr int kRecStage1 = 20, kRecStage3 = 42;
constexpr int kBitFront = 8;      // record flag bits next to the obtuse-corner bits 1, 2, 4
constexpr int kBitDepthSafe = 16; // all three vertex depths well inside [near, far]: the clipped depth cannot fail :810 / :994
constexpr int kRgbNone = 2;       // RGB template value of the alpha-only kernels (SURVEY f-4): no colour, depth or softmax state

// texture modes of the kernels
constexpr int kTexSurface1 = 0;   // texture_type surface, T == 1 (default Mesh texture): texels staged in the record
constexpr int kTexVertex   = 1;   // texture_type vertex (T == 3): 9 floats staged in the record
constexpr int kTexSurfaceN = 2;   // texture_type surface, T = R*R > 1: texels read from HBM/L2 per pair

__host__ __device__ constexpr int record_floats(int texm) { return texm == kTexSurface1 ? 56 : (texm == kTexVertex ? 60 : 48); }

constexpr int kTile    = 8;     // one wavefront renders an 8x8 pixel tile
constexpr int kThreads = 64;    // one wave-tile per workgroup (measured: 64 > 128 > 256 > 512 threads, +6 % over 256)

// Control block (ints) at the end of the workspace, zeroed by face_setup_kernel on every call:
//   [x * kCtlStride], x = 0..7     : length of tile queue x,
//   [(8 + x) * kCtlStride]         : number of tiles of the same range that list no face.
// The 16 counters sit 4 KiB apart: device-scope atomics execute at the memory side and atomics to one line
// serialise (~13 ns each, measured), so counters that share a line would make the list kernel atomic-bound.
// Queue x holds the listed tiles among global tile ids [x*N/8, (x+1)*N/8), N = B * tiles_per_image -- whole images
// when B is a multiple of 8, bands of an image when B is small -- and occupies those slots of tile_list, growing
// from the front; the unlisted tiles of the range grow from the back.  Workgroups are dispatched round-robin over
// the 8 XCDs, so the waves of workgroups with blockIdx.x & 7 == x walk queue x: the tiles of an image (band) are
// rendered through one XCD's L2, which then holds that image's face records and mask rows once.  (If the dispatch
// order were different every tile would still be rendered exactly once; only the locality would suffer.)
//   [(16 + x) * kCtlStride]        : entries allocated so far in region x of the entry pool (bin_faces_kernel; 64-bit).
constexpr int kCtlStride = 1024, kCtlInts = 24 * kCtlStride;

// Per-face record of the binning kernel (floats): the cull box (xlo, xhi, ylo, yhi), 16 bytes per face, coalesced.  (Until round 6 the
// record also carried the rows of the barycentric matrix and wcull_k for a per-(face, tile) edge test that was measured in round 2 and
// never shipped: 48 bytes per face that face_setup_kernel wrote and every super-tile's workgroup fetched past.)
constexpr int kBinRec = 4;

// One entry of a tile's coverage list: face index and the ballot of the tile's pixels that pass the exact box / edge
// tests for it (bit p = pixel lane p).  Written by cover_kernel, read by both render kernels.
// npix: bits 0..7 the pixels of the mask; bits 8..9 the REGION TAG (round 4): 0 = nothing known; 1, 2, 3 = every pixel of the
// mask lies outside the face and the closest-point search of kernel.cu:120-142 selects the same edge for all of them (edge
// tag - 1) -- proven by the coverage kernel from the signs of the three barycentrics over the entry's pixel rows.  The dense /
// pixel-mode paths of the render kernels then evaluate that one edge with wave-uniform selects (point_to_face_edge()).
struct __attribute__((aligned(16))) CoverEnt { int fn; int npix; unsigned lo, hi; };

// Pair hints: what the forward kernel found out about the 64 pairs of a batch and the backward kernel would otherwise have to
// find out again -- two bits per pair (bit l of `lo` / `hi` = pair lane l): 0, 1, 2 = the edge of the face the closest-point
// search selected (the minimum over three candidates for a pixel inside the face, the corner logic of kernel.cu:120-142
// outside); kHintDead = the pair gets no gradient (it failed one of the skip tests :769 / :784, or its barycentrics are
// NaN; the depth test :810 / :994 is repeated by backward, which needs the depth anyway).  Backward evaluates one edge with point_to_face_edge() instead of the whole search -- the same
// float operations on the same operands for that edge, so every value downstream is bit for bit the forward kernel's.
// A tile's batches are windows of 64 consecutive codes of its pair list, so batch k is the same set of pairs in both
// kernels as long as neither splits the tile among several waves (see TileWalk: both use the same grades); the forward kernel says so per queue
// in the control block (kCtlHintFlag), and face_setup_kernel clears that word with the rest of the block on every call,
// so hints of an earlier call are never taken for this one's.  A tile has at most as many batches as entries (an entry
// holds 1..64 pairs), hence the parallel indexing.
struct __attribute__((aligned(16))) PairHints { unsigned long long lo, hi; };
constexpr int kHintDead = 3;
// control[x * kCtlStride + kCtlHintFlag]: OR of 1 (the forward kernel rendered queue x unsplit and wrote the hints of its
// tiles) and 2 (some pair of the queue cannot be described by a hint -- the inside branch selected no edge, kHintNone, which
// takes NaN or infinite candidates on a degenerate face -- backward then repeats the whole search for that queue)
constexpr int kCtlHintFlag = 1;
// control[x * kCtlStride + kCtlLive]: (order_tiles_kernel) the tiles of queue x with a non-empty coverage list; they come first
// in the heavy-first copy of the queue records (kCtlLive + 1..3: the split grades, see kCtlGrade)
constexpr int kCtlLive = 4;

__device__ __forceinline__ long queue_begin(int x, long n_tiles) { return ((long)x * n_tiles) >> 3; }
// the x with queue_begin(x) <= g < queue_begin(x + 1)
__device__ __forceinline__ int queue_of_tile(long g, long n_tiles) { return (int)min(7L, (8 * (g + 1) + n_tiles - 1) / n_tiles - 1); }

// n / d for 0 <= n < 2^30 with the host's magic pair (see RenderArgs::div_tpi_m)
__device__ __forceinline__ int fast_div(int n, unsigned m, int s) { return (int)(((unsigned long long)(unsigned)n * m) >> s); }

struct RenderArgs {
    const float*  records;      // [B*nf][REC]
    const unsigned long long* masks;   // [B*tiles][chunks] : bit f of chunk c set = face 64c+f may touch the tile (binning -> coverage)
    const float*  textures;     // [B,nf,T,3]
    float*        rgba;         // [B,4,is,is]
    float*        aux;          // [B,2,is,is]
    // silhouette (alpha-only) kernels, RGB == kRgbNone: `rgba` / `grad_rgba` are single planes [B,is,is]
    const float*  target;       // [B,is,is] target silhouettes of the fused IoU epilogue, or NULL
    float*        iou_sums;     // [B,2]: sum(alpha * target), sum(alpha * (1 - target)) per view (forward, accumulated)
    const float*  grad_iou;     // [B,2]: d loss / d (those two sums) per view (backward); NULL: grad_rgba is the alpha gradient
    const float*  grad_rgba;    // backward only
    float*        grad_faces;   // backward only
    float*        grad_textures;
    int*          tile_list;    // [B * tiles_per_image]: 8 queues of global tile ids, see kCtlInts
    int*          control;      // queue lengths
    CoverEnt*     entries;      // entry pool: 8 regions of ent_cap8 entries (one per tile queue)
    int4*         tile_info_raw;// the queue records in the order the binning kernel appended them (written by it and cover_kernel)
    int4*         tile_info;    // what the render kernels walk: the heavy-first copy of order_tiles_kernel, or tile_info_raw
                                // [B * tiles_per_image], parallel to tile_list: (tile, first entry, entries, pairs) of the queue
                                //   slot -- tile, first entry (-1: pool exhausted, the render
                                //   kernels then run the per-pixel tests themselves from the mask row), the entry count
                                //   pair count from cover_kernel: one scalar load tells a wave all it needs
    long          ent_cap8;     // capacity of one region of the entry pool
    PairHints*    hints;        // parallel to `entries`: slot (tile's first entry + k) = the hints of the tile's k-th batch of 64 pairs
    int B, nf, T, R, is;
    int tiles_x, tiles_per_image, total_tiles, total_blocks, chunks;
    // n / tiles_per_image and n / tiles_x for tile indices n < 2^30 (gendr_validate) as a multiply and a shift -- on a wave-uniform n two scalar
    // multiplies and a 64-bit shift -- where the compiler's expansion of an integer division converts to float and back on the vector pipes
    // (five vector instructions per divisor and tile): m = ceil(2^(30 + l) / d), l = ceil(log2 d) (Granlund-Montgomery; div_magic() on the host)
    unsigned div_tpi_m, div_tx_m;
    int      div_tpi_s, div_tx_s;
    int*   det_count;           // deterministic backward: number of deferred (large-box) faces, their list, their band sums
    int*   det_list;
    float* det_partial;
    // faces whose cull box is loose, images of kLooseMinTiles tiles and more (see loose_faces_kernel): per face a flag and the
    // box of its live pixels (columns lo / hi, rows lo / hi; empty: lo > hi), per image a list of kLooseList ints:
    // (loose_stamp << 4 | entries), then the faces.  NULL: no lists (smaller images: the coverage kernel alone deals with them)
    const int*  loose_flag;
    int4*       loose_box;
    int*        loose_image;
    int         loose_stamp;
    int         want_tags;      // the coverage kernel works out region tags (CoverEnt): the render kernels that follow have the dense path and the cull radius spans a tile
    int         rec_floats;     // floats per face record (record_floats(texture mode)): for the kernels that are not templated on it
    float       cull_r2;        // (cull radius)^2, rounded up: an outside pixel whose computed squared distance reaches it is dead
    gendr_params p;
    float thr;                  // dist_eps * dist_scale (kernel.cu:725)
    float softmax_sum0;         // exp(aggr_rgb_eps / aggr_rgb_gamma) (kernel.cu:729)
    float bg_soft[3];           // (background[k] * softmax_sum0) / softmax_sum0 in float, two roundings (kernel.cu:731-737, :857 for a pixel no face touches):
                                // what an unlisted tile's pixels hold under softmax RGB -- from the host: as an expression of uniform values the
                                // device evaluated an IEEE division per lane and fill (round 6)
    // correctly rounded double reciprocals of the per-call divisors (see div_by)
    double r_scale;             // 1 / dist_scale
    double r_gamma;             // 1 / aggr_rgb_gamma
    double r_zrange;            // 1 / (far - near)        (kernel.cu:826)
    double r_nzrange;           // 1 / (near - far)        (kernel.cu:1026)
    double r_is;                // 1 / image_size          (kernel.cu:718-719)
    float  gamma_k0;            // gamma family constants, see DistParams
    double gamma_pdf_c;
    float  rf_scale, rf_gamma, rf_zrange, rf_nzrange;   // the same reciprocals rounded to float (fast build, see rcp_t)
};
// the per-call reciprocals in the type div_by() takes in this build
#if GENDR_FAST_DEV
#define GENDR_R_SCALE(a)   ((a).rf_scale)
#define GENDR_R_GAMMA(a)   ((a).rf_gamma)
#define GENDR_R_ZRANGE(a)  ((a).rf_zrange)
#define GENDR_R_NZRANGE(a) ((a).rf_nzrange)
#else
#define GENDR_R_SCALE(a)   ((a).r_scale)
#define GENDR_R_GAMMA(a)   ((a).r_gamma)
#define GENDR_R_ZRANGE(a)  ((a).r_zrange)
#define GENDR_R_NZRANGE(a) ((a).r_nzrange)
#endif

// reciprocals of the 31 divisors of gamma's Kummer series -> LDS, once per wave (only the kernels that can meet a gamma
// distribution carry the table)
template <int DIST>
__device__ __forceinline__ const rcp_t* gamma_table(rcp_t* s_tab, const RenderArgs& a)
{
    if constexpr (DIST == kGamma || DIST == kGammaRev || DIST == -1) {
        const int lane = threadIdx.x & 63;
        if (lane < kGammaSteps - 1) s_tab[lane] = (rcp_t)(1. / (double)(a.p.dist_shape + (float)(lane + 1)));
        __builtin_amdgcn_wave_barrier();
        return s_tab;
    } else {
        return nullptr;
    }
}

// the normal CDF's tables (gendr_math.h: norm_q_tab) in LDS, for the kernels specialised for the gaussian distribution; every wave
// of a workgroup stores the same 192 doubles (three per lane) and reads them after its own stores
#ifndef GENDR_NORMTAB_LDS
#define GENDR_NORMTAB_LDS 1     // 0: the tables stay in global memory (read through the vector L1) -- A/B builds
#endif
template <int DIST>
__device__ __forceinline__ const double* norm_table(double* s_tab)
{
    if constexpr (DIST == kGaussian) {
#if !GENDR_NORMTAB_LDS
        return &kNormTab[0][0];
#endif
        const int lane = threadIdx.x & 63;
        const double* src = &kNormTab[0][0];
#pragma unroll
        for (int k = 0; k < kNormRows * kNormRow / 64; k++) s_tab[lane + 64 * k] = src[lane + 64 * k];
        __builtin_amdgcn_wave_barrier();
        return s_tab;
    } else {
        return nullptr;
    }
}

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
#pragma clang fp contract(off)      // in every build variant: faces_info stays bit for bit the reference's (per face, not per pair)
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

// One thread per face.  Writes boxes[i][kBinRec] (the binning kernel's record) and records[i][REC].
//   sthr   = sqrtf(dist_eps * dist_scale), the reference's border margin (:747)
//   cull_r = distance beyond which an outside pixel contributes nothing (gendr_cull_radius), or +inf
#ifndef GENDR_FS_ABLATE
#define GENDR_FS_ABLATE 0      // 1, 2: measurement builds (tools/fs_ablate.sh), wrong results
#endif
template <int TEXM>
__global__ __launch_bounds__(kThreads) void face_setup_kernel(
    const float* __restrict__ faces, const float* __restrict__ textures,
    float* __restrict__ boxes, float* __restrict__ records,
    long total_faces, float sthr, float cull_r, int* __restrict__ control, int ncontrol, float near_, float far_,
    float4* __restrict__ clear4, long clear_quads,
    int flag_loose, int* __restrict__ loose_flag, int4* __restrict__ loose_box, int* __restrict__ loose_image, int loose_stamp, int nf)
{
    constexpr int REC = record_floats(TEXM);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // launched with one wavefront per workgroup
    if (i < ncontrol / kCtlStride) { control[i * kCtlStride] = 0; control[i * kCtlStride + 1] = 0; }   // queue counters for this call (the pool counters are 64-bit)
    // the caller's buffer to clear (gendr_params::clear_ptr: the gradients of the coming backward call): every workgroup
    // zero-fills its slice with 16-byte stores while its loads are in flight
    if (clear_quads > 0) {
        const long per = (clear_quads + gridDim.x - 1) / gridDim.x;
        const long q0 = (long)blockIdx.x * per, q1 = min(q0 + per, clear_quads);
        for (long q = q0 + threadIdx.x; q < q1; q += blockDim.x) clear4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if ((long)blockIdx.x * blockDim.x >= total_faces) return;                           // whole wavefront past the end
    const bool in_range = i < total_faces;                                              // lanes past the end help with the stores
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = in_range ? faces[i * 9 + k] : 0.f;
    FaceGeom g;
    face_geometry(f, g);

    const float xmax = fmaxf(fmaxf(f[0], f[3]), f[6]), xmin = fminf(fminf(f[0], f[3]), f[6]);
    const float ymax = fmaxf(fmaxf(f[1], f[4]), f[7]), ymin = fminf(fminf(f[1], f[4]), f[7]);
    // the reference's own test: x > max + thr || x < min - thr ...  (same float operations)
    float xhi = xmax + sthr, xlo = xmin - sthr, yhi = ymax + sthr, ylo = ymin - sthr;
    float wcull[3] = {-INFINITY, -INFINITY, -INFINITY};

    if (cull_r < INFINITY && GENDR_FS_ABLATE != 2) {
        // Bound E on (true distance - computed distance) for pixel centres q = (x, y, 1), |x|,|y| <= 1.
        // The loop computes w = inv32 q (rounded) and the vector dis = sum_k (t_k - w_k) v_k with the closest
        // point c = sum_k t_k v_k on the triangle's boundary, i.e. dis = c - p_w with p_w = sum_k w_k v_k.
        // With V = [x_k; y_k; 1] and the exact inverse inv64 = V^-1:  p_w - p = V_xy (inv32 - inv64) q  + rounding.
        // The linear part is evaluated as a matrix product in double (the cancellation sum_k w_k v_k = p survives,
        // which matters for sliver faces whose float inverse is off by percents); rounding of every float
        // operation is bounded by u = 2^-24 relative to its result.  Likewise sum_k w_k = 1 + (1^T delta) q,
        // which bounds how far a pixel that the loop takes for an inside pixel can be from the triangle.
        // A pixel centre farther than cull_r + E from the face's bounding box cannot pass the reference's skip
        // tests (DESIGN.md "exact culling"); tests/test_gpu_parity.py::test_culling_is_exact checks it bit for bit.
        const double X0 = f[0], Y0 = f[1], X1 = f[3], Y1 = f[4], X2 = f[6], Y2 = f[7];
        const double det = X2 * (Y0 - Y1) + X0 * (Y1 - Y2) + X1 * (Y2 - Y0);
        const double adj[9] = {
            Y1 - Y2, X2 - X1, X1 * Y2 - X2 * Y1,
            Y2 - Y0, X0 - X2, X2 * Y0 - X0 * Y2,
            Y0 - Y1, X1 - X0, X0 * Y1 - X1 * Y0};
        const double u = 5.9604644775390625e-08;
        const double vx[3] = {X0, X1, X2}, vy[3] = {Y0, Y1, Y2};
        const double vn[3] = {fabs(X0) + fabs(Y0), fabs(X1) + fabs(Y1), fabs(X2) + fabs(Y2)};
        // one double reciprocal serves the nine quotients of the exact inverse and the three edge heights below: they
        // feed an error BOUND that carries a factor of two and a 1/1024 margin, not a result (twelve double divisions
        // were a third of this kernel's instructions, and its waves run alone on their SIMDs)
        const double rdet = 1. / det;
        double delta[9], W[3], dw[3], wmax = 0.;
        for (int k = 0; k < 3; k++) {
            W[k] = 0.; dw[k] = 0.;
            for (int j = 0; j < 3; j++) {
                delta[3 * k + j] = (double)g.inv[3 * k + j] - adj[3 * k + j] * rdet;
                W[k] += fabs((double)g.inv[3 * k + j]);
                dw[k] += fabs(delta[3 * k + j]);
            }
            dw[k] += 4. * u * W[k];                 // bound on |computed w_k - true w_k| over the image
            wmax = fmax(wmax, W[k]);
        }
        double gx = 0., gy = 0., gs = 0.;
        for (int j = 0; j < 3; j++) {
            double ax = 0., ay = 0., as = 0.;
            for (int k = 0; k < 3; k++) { ax += vx[k] * delta[3 * k + j]; ay += vy[k] * delta[3 * k + j]; as += delta[3 * k + j]; }
            gx += fabs(ax); gy += fabs(ay); gs += fabs(as);
        }
        const double vmax = fmax(vn[0], fmax(vn[1], vn[2]));
        double E = sqrt(gx * gx + gy * gy)                                           // |V_xy delta q|
                 + 4. * u * (W[0] * vn[0] + W[1] * vn[1] + W[2] * vn[2])              // rounding of w_k, carried by v_k
                 + 4. * u * ((1. + W[0]) * vn[0] + (1. + W[1]) * vn[1] + (1. + W[2]) * vn[2])   // rounding of (t_k - w_k) v_k sums
                 + (gs + 4. * u * (W[0] + W[1] + W[2])) * vmax;                       // |sum_k w_k - 1| scaling of an "inside" p_w
        E *= 2.;                                                                       // safety factor
        const double Rf = (double)cull_r * (1. + 1. / 1024.) + E;
        if (Rf == Rf && Rf < 1e30) {   // finite: otherwise keep the reference box only
            xhi = fminf(xhi, round_up((double)xmax + Rf));
            xlo = fmaxf(xlo, round_down((double)xmin - Rf));
            yhi = fminf(yhi, round_up((double)ymax + Rf));
            ylo = fmaxf(ylo, round_down((double)ymin - Rf));
            // Edge test in barycentric units: true w_k = -(distance beyond the edge opposite vertex k) / H_k,
            // H_k = |det| / |v_{k+1} - v_{k+2}|.  computed w_k < wcull_k  =>  true distance > Rf.
            const double ex[3] = {X1 - X2, X2 - X0, X0 - X1}, ey[3] = {Y1 - Y2, Y2 - Y0, Y0 - Y1};
            for (int k = 0; k < 3; k++) {
                const double len = sqrt(ex[k] * ex[k] + ey[k] * ey[k]);
                const double wc = -(Rf * len * fabs(rdet)) * (1. + 1. / 1024.) - dw[k];
                if (wc == wc && wc > -1e30) wcull[k] = round_down(wc);
            }
        }
    }

    // A LOOSE cull box: the error bound E dwarfs the cull radius (determinant clamped or tiny: the face is seen edge-on), so
    // the box is the reference's own margin or close to it -- the whole image at the default dist_eps -- and the face is listed
    // in every tile of its image, although next to none of those pairs passes the skip tests.  Such faces (rare; three of the 64
    // benchmark views hold two each) are flagged in their record; the coverage kernel then EVALUATES them on the pixels of
    // every tile, with the render kernels' own pair functions, and keeps exactly the pixels that can contribute.
    bool loose = false;
    if (in_range && cull_r < INFINITY && flag_loose) {
        const float bw = fminf(xhi, 1.f) - fmaxf(xlo, -1.f), bh = fminf(yhi, 1.f) - fmaxf(ylo, -1.f);         // visible part of the box
        const float fw = fmaxf(fminf(xmax, 1.f) - fmaxf(xmin, -1.f), 0.f) + 4.f * cull_r, fh = fmaxf(fminf(ymax, 1.f) - fmaxf(ymin, -1.f), 0.f) + 4.f * cull_r;
        loose = bw > 0.f && bh > 0.f && bw * bh >= GENDR_LOOSE_AREA && bw * bh > 4.f * fw * fh;     // a good part of the image (NDC area 4), four times what the face itself explains
        if (loose_flag) {
            // Large images (kLooseMinTiles tiles and more: a flagged face is listed in all 65 536 tiles of a 2048^2 image): the face
            // also goes on a short list of its image, loose_faces_kernel evaluates the listed faces on every pixel ONCE and leaves
            // the bounding box of the live ones, and the binning kernel lists the face by that box.
            loose_flag[i] = loose ? 1 : 0;
            if (loose) {
                // append to the image's list: [0] = (tag << 4 | entries), [1 ..] the faces.  The tag tells a list head from whatever
                // fresh memory holds; the binning kernel empties the list after use.  A full list leaves the face its whole image.
                int* list = loose_image + (i / nf) * kLooseList;
                int slot = -1;
                for (int cur = __hip_atomic_load(list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);;) {
                    const int have = (cur >> 4) == loose_stamp ? (cur & 15) : 0;
                    if (have >= kLooseList - 1) break;
                    const int want = (loose_stamp << 4) | (have + 1);
                    const int seen = atomicCAS(list, cur, want);
                    if (seen == cur) { slot = have; break; }
                    cur = seen;
                }
                if (slot >= 0) { list[1 + slot] = (int)(i % nf); loose_box[i] = make_int4(0x7fffffff, -1, 0x7fffffff, -1); }
                else           loose_box[i] = make_int4(0, 0x7ffffff0, 0, 0x7ffffff0);       // every pixel
            }
        }
    }

    // bin record: the cull box (the binning kernel tests boxes only, see there)
    if (in_range) reinterpret_cast<float4*>(boxes)[i] = make_float4(xlo, xhi, ylo, yhi);

    // the record leaves in 16-byte stores (REC is a multiple of 4 floats)
    float r[REC];
#pragma unroll
    for (int k = 0; k < REC; k++) r[k] = 0.f;
    r[kRecBox + 0] = xlo; r[kRecBox + 1] = xhi; r[kRecBox + 2] = ylo; r[kRecBox + 3] = yhi;
#pragma unroll
    for (int k = 0; k < 9; k++) r[kRecInv + k] = g.inv[k];
    // The clipped, renormalised barycentrics are >= 0 and sum to 1 (at least one raw weight is >= 1/3), so the
    // perspective depth 1 / sum(w_k / z_k) lies between the smallest and the largest vertex depth up to rounding:
    // with a 1e-4 margin on both sides the near / far test (:810, :994) can never fire for this face.
    const float zlo = fminf(fminf(f[2], f[5]), f[8]), zhi = fmaxf(fmaxf(f[2], f[5]), f[8]);
    const int depth_safe = (zlo > 0.f && zlo >= near_ * 1.0001f && zhi <= far_ * 0.9999f) ? kBitDepthSafe : 0;
    r[kRecBits] = __int_as_float(g.obt | (g.front << 3) | depth_safe);
    r[kRecWCull + 0] = wcull[0]; r[kRecWCull + 1] = wcull[1]; r[kRecWCull + 2] = wcull[2];
    r[kRecLoose] = __int_as_float(loose ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3;
        float a[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { a[j] = g.sym[3 * k + j] - g.sym[3 * k1 + j]; r[kRecEdge + 3 * k + j] = a[j]; }
        put_rcp(r, kRecRDen + 2 * k, 1. / (double)(a[k] - a[k1]));
        r[kRecXY + 2 * k] = f[3 * k]; r[kRecXY + 2 * k + 1] = f[3 * k + 1];
        put_rcp(r, kRecRZ + 2 * k, 1. / (double)f[3 * k + 2]);
    }
    if (TEXM == kTexSurface1 && in_range) {
        const long nxt = (i + 1 < total_faces) ? i + 1 : i;   // reference reads the next face's texel (:179-182); none after the last
#pragma unroll
        for (int k = 0; k < 3; k++) { r[kRecTex + k] = textures[i * 3 + k]; r[kRecTex + 3 + k] = textures[nxt * 3 + k]; }
    } else if (TEXM == kTexVertex && in_range) {
#pragma unroll
        for (int k = 0; k < 9; k++) r[kRecTex + k] = textures[i * 9 + k];
    }
    // The 64 records of a wavefront are contiguous in HBM (64 * REC floats): they go through LDS so that every store
    // instruction writes 1 KiB of consecutive bytes instead of 64 scattered 16-byte pieces 4 * REC bytes apart (the
    // scattered form made this kernel store-bound: 15.5 us at C2).
    __shared__ __attribute__((aligned(16))) float s_out[kThreads * REC];
    float4* mine4 = reinterpret_cast<float4*>(s_out + (threadIdx.x & 63) * REC);
#pragma unroll
    for (int q = 0; q < REC / 4; q++) mine4[q] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    const long first = (long)blockIdx.x * blockDim.x;                      // first face of this wavefront
    const int nrec = (int)min((long)kThreads, total_faces - first);
    float4* dst4 = reinterpret_cast<float4*>(records + first * REC);
    const float4* src4 = reinterpret_cast<const float4*>(s_out);
#if GENDR_FS_ABLATE == 1        // measurement build: one 16-byte store per lane instead of the whole record
    if ((threadIdx.x & 63) < nrec) dst4[threadIdx.x & 63] = src4[threadIdx.x & 63];
    return;
#endif
    for (int q = threadIdx.x & 63; q < nrec * (REC / 4); q += 64) dst4[q] = src4[q];
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
// pixel / tile geometry
// ---------------------------------------------------------------------------------------------
// (2.*idx + 1. - is) / is of kernel.cu:718-719.  The reference evaluates it in double and rounds to float;
// numerator and denominator are integers below 2^24, so the float division rounds to the same value
// (and div_by() reproduces that float division exactly).
__device__ __forceinline__ float pixel_coord(int idx, int is, double r_is)
{
    return (float)((double)(float)(2 * idx + 1 - is) * r_is);    // r_is = RN(1 / (double)is), computed once on the host; exact in every build (per tile, not per pair)
}

// box = (xlo, xhi, ylo, yhi).  A rectangle of pixel centres misses the box iff every centre fails the
// per-pixel test "x > xhi || x < xlo || y > yhi || y < ylo".
__device__ __forceinline__ bool rect_hits_box(float rx_lo, float rx_hi, float ry_lo, float ry_hi, const float4& box)
{
    return !(rx_lo > box.y || rx_hi < box.x || ry_lo > box.w || ry_hi < box.z);
}

// ---------------------------------------------------------------------------------------------
// wave-tile bookkeeping shared by forward and backward
// ---------------------------------------------------------------------------------------------
struct TileCtx {
    int   b;            // batch item
    int   tile;         // global tile id (b * tiles_per_image + ty * tiles_x + tx)
    int   x0, y0;       // first pixel column / row of the tile (uniform)
    int   xi, row;      // this lane's pixel
    bool  valid;        // pixel inside the image
    float xp, yp;       // pixel centre, kernel.cu:716-719
    long  pix;          // row * is + xi
};

// The render kernels are launched with a quarter of the waves it would take to give every tile of the batch its
// own: wave r of XCD x renders entries r, r + stride, ... of queue x.  In the usual scene (at most a quarter of the
// tiles list a face) that is one tile per wave and no wave is launched in vain.
struct TileWalk { long qbase, qend; int total, live, empties, rank, next, stride, hint_flag; int g8, g4, g2, items; };

// Graded sub-tile split of the render kernels (round 4; round 3 split every tile of a queue alike).  The latency of a launch is
// the latency of one wave on the heaviest tile (ten batches at the headline scene); when a queue lists fewer tiles than the
// chip holds waves for it at once -- small batches: the per-GPU share of a strong-scaling run -- the HEAVY tiles are rendered
// by 8, 4 or 2 waves, each taking 1, 2 or 4 of the tile's 8 pixel rows.  order_tiles_kernel, which sorts the queue records by
// weight class anyway, picks the smallest piece size T (a multiple of 32 pairs, at least 64 = one batch) for which
//     tiles + #(pairs > T) + 2 #(pairs > 2 T) + 4 #(pairs > 4 T)   <=   the waves the chip holds for the queue
// (more work items than resident waves only add a second generation of waves: measured in round 3, slower) and leaves the
// numbers g8, g4, g2 of tiles split 8-, 4- and 2-fold in the control block; the sorted records put those tiles first, so work
// item i of the queue maps to (tile, rows) with three compares.  A wave keeps only the bits of its rows in the coverage masks,
// so its pair list, its batches and its time shrink by the split (the entries are read once per wave, which is cheap; forward
// keeps every pixel's ascending face order, since a pixel belongs to exactly one wave).  Both render kernels use the same
// grades -- the forward kernel's pair hints describe the batches of UNSPLIT tiles, and backward takes them for exactly those.
constexpr int kCtlGrade = 5;          // control[x * kCtlStride + kCtlGrade + 0..2] = g8, g4, g2 of queue x (order_tiles_kernel)
// work item i of the queue -> index of its record in the sorted copy, log2 of the tile's split, the piece
__device__ __forceinline__ int walk_item(const TileWalk& w, int i, int& sl, int& sub)
{
    if (i < 8 * w.g8) { sl = 3; sub = i & 7; return i >> 3; }
    i -= 8 * w.g8;
    if (i < 4 * w.g4) { sl = 2; sub = i & 3; return w.g8 + (i >> 2); }
    i -= 4 * w.g4;
    if (i < 2 * w.g2) { sl = 1; sub = i & 1; return w.g8 + w.g4 + (i >> 1); }
    sl = 0; sub = 0;
    return w.g8 + w.g4 + w.g2 + (i - 2 * w.g2);
}
// pixel lanes (bits) of sub-tile `sub` of 1 << split_log2: whole rows of 8 pixels
__device__ __forceinline__ unsigned long long sub_tile_mask(int split_log2, int sub)
{
    if (split_log2 == 0) return ~0ull;
    const int bits = 64 >> split_log2;
    return ((1ull << bits) - 1ull) << (sub * bits);
}

__device__ __forceinline__ void walk_init(TileWalk& w, const RenderArgs& a, int waves_per_block)
{
    const int xcd = blockIdx.x & 7;                                          // gridDim.x is a multiple of 8
    w.qbase = queue_begin(xcd, a.total_tiles);
    w.qend = queue_begin(xcd + 1, a.total_tiles);
    w.total = __builtin_amdgcn_readfirstlane(a.control[xcd * kCtlStride]);
    // (the render kernels walk the heavy-first copy when there is one: its tiles without any entry sit at the end)
    w.live = a.tile_info != a.tile_info_raw ? __builtin_amdgcn_readfirstlane(a.control[xcd * kCtlStride + kCtlLive]) : w.total;
    w.empties = __builtin_amdgcn_readfirstlane(a.control[(8 + xcd) * kCtlStride]);
    w.hint_flag = __builtin_amdgcn_readfirstlane(a.control[xcd * kCtlStride + kCtlHintFlag]);
    w.stride = (int)(gridDim.x >> 3) * waves_per_block;
    w.rank = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) * waves_per_block + (int)(threadIdx.x >> 6));
    w.next = w.rank;
    w.g8 = w.g4 = w.g2 = 0;
    if (a.tile_info != a.tile_info_raw) {
        const i4v g = *(const GENDR_CONST_AS i4v*)(a.control + xcd * kCtlStride + kCtlGrade - 1);      // (live, g8, g4, g2): one scalar load
        w.g8 = g.y; w.g4 = g.z; w.g2 = g.w;
    }
    w.items = w.live + 7 * w.g8 + 3 * w.g4 + w.g2;
}

// the lane = pixel part of a tile's context
__device__ __forceinline__ void tile_lanes(TileCtx& t, const RenderArgs& a)
{
    const int lane = threadIdx.x & 63;
    t.xi = t.x0 + (lane & 7);
    t.row = t.y0 + (lane >> 3);
    t.valid = t.xi < a.is && t.row < a.is;
    t.xp = pixel_coord(t.xi, a.is, a.r_is);
    t.yp = pixel_coord(a.is - 1 - t.row, a.is, a.r_is);   // yi = is - 1 - row, kernel.cu:716
    t.pix = (long)t.row * a.is + t.xi;
}
// lanes = false: the wave-uniform part only (the coverage kernel, whose lanes are (face, row) slots: tile_lanes() where it meets a loose face)
__device__ __forceinline__ void tile_setup(TileCtx& t, const RenderArgs& a, int tile, bool lanes = true)
{
    t.tile = tile;
    t.b = fast_div(tile, a.div_tpi_m, a.div_tpi_s);
    const int tl = tile - t.b * a.tiles_per_image;
    const int ty = fast_div(tl, a.div_tx_m, a.div_tx_s), tx = tl - ty * a.tiles_x;
    t.x0 = tx * kTile;
    t.y0 = ty * kTile;
    if (lanes) tile_lanes(t, a);
}

// ---------------------------------------------------------------------------------------------
// one (pixel, face) evaluation: kernel.cu:747-786 (forward) == :924-962 (backward)
// ---------------------------------------------------------------------------------------------
struct Pair {
    float w0, w1, w2;        // barycentrics (:39-43)
    float t0, t1, t2;        // t - w of the closest boundary point (:103-105,:157)
    float sign, dx, dy, dis, frag;
    float hint;              // the edge point_to_face() selected, as a float the forward kernel's ballots can test with ONE compare
                             // against an inline constant each (no register for a mask or a literal -- the forward kernel has
                             // none to spare): kHintEdge0 / 1 / 2, or kHintNone when the inside branch selected none of its
                             // candidates (:112 never true).  A pair without a gradient keeps hint = +0.
};
#define kHintEdge0 1.0f
#define kHintEdge1 (-1.0f)
#define kHintEdge2 4.0f
#define kHintNone  __builtin_nanf("")
// bit 0 of the 2-bit code (edge 1, or dead = 3): hint <= 0;  bit 1 (edge 2, or dead): |hint| <> 1 (ordered);  none: unordered
__device__ __forceinline__ bool hint_bit0(float h) { return h <= 0.f; }
__device__ __forceinline__ bool hint_bit1(float h) { const float m = __builtin_fabsf(h); return m < 1.f || m > 1.f; }
__device__ __forceinline__ bool hint_none(float h) { return h != h; }

__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// kernel.cu:76-165 on the face record.  Returns false when the pair must be dropped (NaN barycentrics:
// the reference indexes with v0 = -1 there; DESIGN.md quirk iv).
__device__ __forceinline__ bool point_to_face(Pair& q, const float* r, float xp, float yp)
{
    const float w0 = q.w0, w1 = q.w1, w2 = q.w2;
    const float x0 = r[kRecXY + 0], y0 = r[kRecXY + 1], x1 = r[kRecXY + 2], y1 = r[kRecXY + 3],
                x2 = r[kRecXY + 4], y2 = r[kRecXY + 5];
    if (w0 > 0 && w1 > 0 && w2 > 0 && w0 < 1 && w1 < 1 && w2 < 1) {
        // edge k joins vertex k and k+1: t0[k] = tv, t0[k+1] = 1 - tv, t0[k+2] = 0, then t0 -= w (:91-105)
        const float tva = div_by((w0 * r[kRecEdge + 0] + w1 * r[kRecEdge + 1] + w2 * r[kRecEdge + 2] - r[kRecEdge + 1]), rec_rcp(r, kRecRDen + 0));
        const float a0 = tva - w0, a1 = (1 - tva) - w1, a2 = 0.f - w2;
        const float adx = a0 * x0 + a1 * x1 + a2 * x2, ady = a0 * y0 + a1 * y1 + a2 * y2;
        const float ad = adx * adx + ady * ady;
        const float tvb = div_by((w0 * r[kRecEdge + 3] + w1 * r[kRecEdge + 4] + w2 * r[kRecEdge + 5] - r[kRecEdge + 5]), rec_rcp(r, kRecRDen + 2));
        const float b0 = 0.f - w0, b1 = tvb - w1, b2 = (1 - tvb) - w2;
        const float bdx = b0 * x0 + b1 * x1 + b2 * x2, bdy = b0 * y0 + b1 * y1 + b2 * y2;
        const float bd = bdx * bdx + bdy * bdy;
        const float tvc = div_by((w0 * r[kRecEdge + 6] + w1 * r[kRecEdge + 7] + w2 * r[kRecEdge + 8] - r[kRecEdge + 6]), rec_rcp(r, kRecRDen + 4));
        const float c0 = (1 - tvc) - w0, c1 = 0.f - w1, c2 = tvc - w2;
        const float cdx = c0 * x0 + c1 * x1 + c2 * x2, cdy = c0 * y0 + c1 * y1 + c2 * y2;
        const float cd = cdx * cdx + cdy * cdy;
        // running strict minimum in edge order 0,1,2 starting from 1e8 (:86,:112)
        float best = 100000000.f;
        q.dx = 0.f; q.dy = 0.f; q.t0 = 0.f; q.t1 = 0.f; q.t2 = 0.f; q.hint = kHintNone;
        if (ad < best) { best = ad; q.dx = adx; q.dy = ady; q.t0 = a0; q.t1 = a1; q.t2 = a2; q.hint = kHintEdge0; }
        if (bd < best) { best = bd; q.dx = bdx; q.dy = bdy; q.t0 = b0; q.t1 = b1; q.t2 = b2; q.hint = kHintEdge1; }
        if (cd < best) { best = cd; q.dx = cdx; q.dy = cdy; q.t0 = c0; q.t1 = c1; q.t2 = c2; q.hint = kHintEdge2; }
        q.sign = 1.f;
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
        v0 = m == 2 ? 0 : m + 1;
    }
    const bool e0 = v0 == 0, e1 = v0 == 1;       // selected edge: v0 -> v0+1
    q.hint = e0 ? kHintEdge0 : (e1 ? kHintEdge1 : kHintEdge2);
    // read every candidate into a value first: selecting between array elements directly lets the compiler
    // turn the select into a dynamically indexed load, which would push the record out of SGPRs
    const float E00 = r[kRecEdge + 0], E01 = r[kRecEdge + 1], E02 = r[kRecEdge + 2];
    const float E10 = r[kRecEdge + 3], E11 = r[kRecEdge + 4], E12 = r[kRecEdge + 5];
    const float E20 = r[kRecEdge + 6], E21 = r[kRecEdge + 7], E22 = r[kRecEdge + 8];
    const rcp_t D0 = rec_rcp(r, kRecRDen + 0), D1 = rec_rcp(r, kRecRDen + 2), D2 = rec_rcp(r, kRecRDen + 4);
    const float A0 = e0 ? E00 : (e1 ? E10 : E20);
    const float A1 = e0 ? E01 : (e1 ? E11 : E21);
    const float A2 = e0 ? E02 : (e1 ? E12 : E22);
    const float Av1 = e0 ? A1 : (e1 ? A2 : A0);   // a0[v1]
    const rcp_t rden = e0 ? D0 : (e1 ? D1 : D2);
    const float tv = div_by(w0 * A0 + w1 * A1 + w2 * A2 - Av1, rden);
    const float ta = fminf(fmaxf(tv, 0.f), 1.f);           // min(max(t, 0.), 1.) : the clamp is exact in either precision
    const float tb = fminf(fmaxf(1 - tv, 0.f), 1.f);
    // t[v0] = ta, t[v0+1] = tb, t[v0+2] = clamp(0) = 0 ; then t[k] -= w[k]  (:150-158)
    q.t0 = (e0 ? ta : (e1 ? 0.f : tb)) - w0;
    q.t1 = (e0 ? tb : (e1 ? ta : 0.f)) - w1;
    q.t2 = (e0 ? 0.f : (e1 ? tb : ta)) - w2;
    q.dx = q.t0 * x0 + q.t1 * x1 + q.t2 * x2;
    q.dy = q.t0 * y0 + q.t1 * y1 + q.t2 * y2;
    q.sign = -1.f;
    return true;
}

// The same closest-point evaluation for a pair whose edge is already known (backward: the forward kernel recorded, per
// pair, which edge point_to_face() selected -- the pair hints, see kHintDead): one edge instead of three candidates resp.
// the corner logic, the very expressions of the two branches above on the very operands (inside: t, 1 - t unclamped,
// :99-105; outside: clamped, :150-158), so dx, dy, t and with them the fragment are bit for bit the forward kernel's.
// e0 / e1: the selected edge is 0 / 1 (neither: 2).
__device__ __forceinline__ void point_to_face_edge(Pair& q, const float* r, bool e0, bool e1, bool outside_known = false)
{
    const float w0 = q.w0, w1 = q.w1, w2 = q.w2;
    const float x0 = r[kRecXY + 0], y0 = r[kRecXY + 1], x1 = r[kRecXY + 2], y1 = r[kRecXY + 3],
                x2 = r[kRecXY + 4], y2 = r[kRecXY + 5];
    const bool inside = !outside_known && w0 > 0 && w1 > 0 && w2 > 0 && w0 < 1 && w1 < 1 && w2 < 1;
    const float E00 = r[kRecEdge + 0], E01 = r[kRecEdge + 1], E02 = r[kRecEdge + 2];
    const float E10 = r[kRecEdge + 3], E11 = r[kRecEdge + 4], E12 = r[kRecEdge + 5];
    const float E20 = r[kRecEdge + 6], E21 = r[kRecEdge + 7], E22 = r[kRecEdge + 8];
    const rcp_t D0 = rec_rcp(r, kRecRDen + 0), D1 = rec_rcp(r, kRecRDen + 2), D2 = rec_rcp(r, kRecRDen + 4);
    const float A0 = e0 ? E00 : (e1 ? E10 : E20);
    const float A1 = e0 ? E01 : (e1 ? E11 : E21);
    const float A2 = e0 ? E02 : (e1 ? E12 : E22);
    const float Av1 = e0 ? A1 : (e1 ? A2 : A0);
    const rcp_t rden = e0 ? D0 : (e1 ? D1 : D2);
    const float tv = div_by(w0 * A0 + w1 * A1 + w2 * A2 - Av1, rden);
    const float ta = inside ? tv : fminf(fmaxf(tv, 0.f), 1.f);
    const float tb = inside ? 1 - tv : fminf(fmaxf(1 - tv, 0.f), 1.f);
    q.t0 = (e0 ? ta : (e1 ? 0.f : tb)) - w0;
    q.t1 = (e0 ? tb : (e1 ? ta : 0.f)) - w1;
    q.t2 = (e0 ? 0.f : (e1 ? tb : ta)) - w2;
    q.dx = q.t0 * x0 + q.t1 * x1 + q.t2 * x2;
    q.dy = q.t0 * y0 + q.t1 * y1 + q.t2 * y2;
    q.sign = inside ? 1.f : -1.f;
}

__device__ __forceinline__ bool inside_closed(const Pair& q)
{
    return q.w0 <= 1 && q.w0 >= 0 && q.w1 <= 1 && q.w1 >= 0 && q.w2 <= 1 && q.w2 >= 0;   // :62-64
}

// stage 1: the reference's border test (:747) tightened to the exact cull box, on r[0..16)
__device__ __forceinline__ bool inside_box(const float* r, float xp, float yp)
{
    return !(xp > r[kRecBox + 1] || xp < r[kRecBox + 0] || yp > r[kRecBox + 3] || yp < r[kRecBox + 2]);
}
// exact edge reject (see kRecWCull); false for NaN barycentrics so that those reach the reference's own logic
__device__ __forceinline__ bool beyond_an_edge(const Pair& q, const float* r)
{
    return q.w0 < r[kRecWCull + 0] || q.w1 < r[kRecWCull + 1] || q.w2 < r[kRecWCull + 2];
}
__device__ __forceinline__ void barycentrics(Pair& q, const float* r, float xp, float yp)   // :39-43
{
    q.w0 = r[kRecInv + 0] * xp + r[kRecInv + 1] * yp + r[kRecInv + 2];
    q.w1 = r[kRecInv + 3] * xp + r[kRecInv + 4] * yp + r[kRecInv + 5];
    q.w2 = r[kRecInv + 6] * xp + r[kRecInv + 7] * yp + r[kRecInv + 8];
}

// stage 2 on r[16..34): soft fragment of a pair whose pixel is inside the box.  Returns true if the pair
// contributes (none of the skips at :769, :784 fires).
template <int DIST, int SQ>
__device__ __forceinline__ bool soft_fragment(Pair& q, const float* r, float xp, float yp, const RenderArgs& a, const DistParams& dp,
                                              bool hinted = false, bool e0 = false, bool e1 = false, bool tagged = false)
{
    // hinted: the forward kernel recorded the pair's edge AND that it passed the skip tests (pair hints, backward only);
    // tagged: the coverage kernel proved the edge for every pixel of the entry (region tag) -- the skip tests still apply
    const int dist = DIST >= 0 ? DIST : a.p.dist_func;
    if (dist == kHeaviside) {
        q.sign = 0.f; q.dx = 0.f; q.dy = 0.f; q.dis = 0.f; q.t0 = q.t1 = q.t2 = 0.f; q.hint = kHintEdge0;
        q.frag = inside_closed(q) ? 1.f : 0.f;                                      // :762-764
    } else {
        if (hinted || tagged) point_to_face_edge(q, r, e0, e1, tagged);
        else if (!point_to_face(q, r, xp, yp)) return false;
        float dis = q.dx * q.dx + q.dy * q.dy;                                      // :768
        if (!hinted && q.sign < 0 && dis >= a.thr) return false;                    // :769 (a hinted pair passed it in the forward kernel)
        const bool squared = SQ >= 0 ? (SQ != 0) : (a.p.dist_squared != 0);
        if (!squared) dis = sqrt_rn(dis);                                           // :770-772 (== sqrtf, see sqrt_rn)
        q.dis = dis;
        if constexpr (DIST == kGaussian) q.frag = norm_cdf_tab(div_by(q.sign * dis, dp.rscale), dp.norm_tab);   // Dist<kGaussian>::cdf with the LDS tables
        else if constexpr (DIST >= 0)  q.frag = Dist<(DIST >= 0 ? DIST : 0)>::cdf(q.sign, dis, dp);
        else if constexpr (DIST == -2) q.frag = cdf_light_rt(dist, q.sign, dis, dp);
        else                           q.frag = cdf_rt(dist, q.sign, dis, dp);
    }
    // :784 compares the float fragment with the double literal 1e-6.  (float)1e-6 lies BELOW 1e-6, so the floats that are
    // <= 1e-6 are exactly those <= (float)1e-6: one float compare (NaN: neither form skips the pair)
    static_assert((double)(float)kProbThreshold <= kProbThreshold, "the float compare needs RN(threshold) <= threshold");
    return hinted || !(q.frag <= (float)kProbThreshold);                            // :784
}

// barycentric_clip + depth, kernel.cu:68-72, :807-810
__device__ __forceinline__ float clip_and_depth(const Pair& q, const float* r, float* wc)
{
    wc[0] = fmaxf(fminf(q.w0, 1.f), 0.f);
    wc[1] = fmaxf(fminf(q.w1, 1.f), 0.f);
    wc[2] = fmaxf(fminf(q.w2, 1.f), 0.f);
    float s = wc[0] + wc[1] + wc[2];
    // max(sum, 1e-5) with a double literal, stored as float: (float)1e-5 lies below 1e-5, so s > 1e-5 iff s > (float)1e-5
    static_assert((double)(float)1e-5 < 1e-5, "the float compare needs RN(1e-5) < 1e-5");
    s = (s > (float)1e-5) ? s : (float)1e-5;
    const rcp_t rs = rcp_for_div_by(s);           // three float quotients by one float divisor (s >= 1e-5)
    wc[0] = div_by(wc[0], rs); wc[1] = div_by(wc[1], rs); wc[2] = div_by(wc[2], rs);
    return rcp_rn(div_by(wc[0], rec_rcp(r, kRecRZ + 0)) + div_by(wc[1], rec_rcp(r, kRecRZ + 2)) + div_by(wc[2], rec_rcp(r, kRecRZ + 4)));   // "1. /": one rounding
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
__device__ __forceinline__ void sample_colour(float* c, int& own, const float* wc, const float* r, const RenderArgs& a, long face_lin)
{
    if (TEXM == kTexSurface1) {
        int idx = texel_index(wc, 1, a.p.texel_mode == 1);
        const bool last = face_lin + 1 >= (long)a.B * a.nf;
        own = (idx == 0) ? 0 : -1;
        if (a.p.texel_mode == 1) { idx = 0; own = 0; }
        if (idx != 0 && last) idx = 0;                      // nothing after the last face: own texel, still no gradient
        // idx is 0 (own) or 1 (next face) for R == 1; a negative index cannot occur here
        const bool nxt = idx != 0;
        const float o0 = r[kRecTex + 0], o1 = r[kRecTex + 1], o2 = r[kRecTex + 2];
        const float n0 = r[kRecTex + 3], n1 = r[kRecTex + 4], n2 = r[kRecTex + 5];
        c[0] = nxt ? n0 : o0;
        c[1] = nxt ? n1 : o1;
        c[2] = nxt ? n2 : o2;
    } else if (TEXM == kTexVertex) {
        own = 0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            c[k] = wc[0] * r[kRecTex + k] + wc[1] * r[kRecTex + 3 + k] + wc[2] * r[kRecTex + 6 + k];   // :187-189
    } else {
        const long at = resolve_texel(wc, a, face_lin, own);
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = a.textures[at * 3 + k];
    }
}

// ---------------------------------------------------------------------------------------------
// pair compaction (shared by forward and backward)
//
// Measured on the headline scene: a face touches only ~14 of a tile's 64 pixels, so evaluating "one face per
// loop iteration, lane = pixel" leaves ~78 % of the lanes idle in the expensive stages.  Instead:
//   coverage (cheap, once per forward call, cover_kernel): box test, barycentrics, edge reject for every pixel of the
//            tile against every listed face, lane = (face, two pixel rows), sixteen faces per step -> per tile the entries
//            (face, pixel mask) in ascending face order;
//   walk     the render kernels append the (pixel, face) pairs of the entries -- in ascending (face, pixel) order --
//            to a wavefront-private list in LDS, together with the face's mask;
//   phase B  (lane = pair, dense): every lane fetches ITS pair's face record from L2 and runs the distance,
//            CDF, clip/depth and colour stages; results go back to LDS;
//   phase C  forward : lane = pixel again; each pixel folds the results of its own pairs in list order, i.e. in
//                      ascending face order (t-conorm fold, z-buffer / online softmax) -- the reference's order;
//            backward: nothing sequential is left; one lane per (face, component) sums the partials of the face's
//                      pairs from LDS and issues one hardware fp32 atomic per (tile batch, face, component).
// Everything is wavefront-local: no barriers.
// ---------------------------------------------------------------------------------------------
// A pair in the batch list is one int, (face slot in the batch << 8) | pixel lane; its barycentrics are computed in
// phase B from the gathered record and the pixel centre from the lane (the same expressions on the same operands as
// everywhere else).
struct FaceSeg {           // 8 bytes: one face of a backward batch
    int fn;                // face index inside the batch item
    int span;              // (first pair of the face in the batch) | (its consecutive pairs << 8)
};
constexpr int kFlagContrib = 1;   // passed :769 and :784 -> folds into alpha
constexpr int kFlagDepthOk = 2;   // near <= zp <= far (:810)
constexpr int kFlagRgb     = 4;   // eligible for the RGB aggregation (:816 resp. :825)

// per-lane copy of record floats [4*Q0, 4*Q1) with 16-byte vector loads (phase B: lane = pair)
template <int Q0, int Q1>
__device__ __forceinline__ void gather_record(float* r, const float* __restrict__ rec)
{
    const float4* src = reinterpret_cast<const float4*>(rec);
#pragma unroll
    for (int q = Q0; q < Q1; q++) {
        const float4 v = src[q];
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}
// the distance stage needs floats [20, 42) (+ the flag word at 13 for the obtuse-corner bits);
// depth / colour need [42, REC)
constexpr int kGatherW0 = 1, kGatherW1 = 4;      // floats [4, 16): inv, flag word, wcull -> barycentrics of the pair
constexpr int kGatherA0 = 5, kGatherA1 = 11;     // floats [20, 44)
constexpr int kGatherB0 = 10;                    // floats [40, REC)

// the exact per-pixel tests for one listed face, lane = pixel (used when a tile has no slice of the entry pool);
// returns the ballot of surviving lanes (0 = nothing to do)
template <int REC>
__device__ __forceinline__ unsigned long long collect_pairs(const TileCtx& t, RecPtr rp, Pair& q)
{
    float r[REC];
    load_record<0, kRecStage1>(r, rp);
    bool live = t.valid && inside_box(r, t.xp, t.yp);
    if (!__any(live)) return 0ull;                   // whole wave outside the box
    barycentrics(q, r, t.xp, t.yp);
    live = live && !beyond_an_edge(q, r);
    return __ballot(live);
}

// the cull box of a flagged face from the pixel box loose_faces_kernel left: pixel centres, inclusive (empty: misses everything)
__device__ __forceinline__ float4 loose_box_ndc(const int4& bx, int is, double r_is)
{
    if (bx.x > bx.y || bx.z > bx.w) return make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    if (bx.y >= is || bx.w >= is) return make_float4(-INFINITY, INFINITY, -INFINITY, INFINITY);      // every pixel (a full list, see face_setup_kernel)
    return make_float4(pixel_coord(bx.x, is, r_is), pixel_coord(bx.y, is, r_is), pixel_coord(is - 1 - bx.w, is, r_is), pixel_coord(is - 1 - bx.z, is, r_is));
}

// ---------------------------------------------------------------------------------------------
// faces with a loose cull box (face_setup_kernel): the bounding box of the pixels that can contribute
// ---------------------------------------------------------------------------------------------
// A fixed grid of one-wave workgroups.  The face_setup kernel left, per image, a short list of its flagged faces behind a
// tag word (loose_image, kLooseList ints per image; emptied by the binning kernel after use).  Every wave looks at the lists of all images (lane = image)
// and, for the few images that have one, takes every kLooseWaves-th block of 64 consecutive pixels for every listed face:
// lane = pixel, the face's record arrives by scalar loads, and the pixel is LIVE if it passes the record's (loose) box and is
// inside the face or the closest-point search of the render kernels -- barycentrics() and point_to_face(), the same float
// operations on the same operands -- puts it closer than the cull radius (beyond it the pair fails :769 or :784, which is
// what the radius is defined by; NaN barycentrics drop the pair as everywhere).  Live pixels widen the face's loose_box by
// integer atomics.  The binning and coverage kernels read that box instead of the record's for a flagged face; everything
// downstream still applies the reference's own tests to every pair, so the box only has to contain the pixels that can
// contribute -- it contains exactly the live ones.

template <int REC>
__global__ __launch_bounds__(kThreads) void loose_faces_kernel(const RenderArgs a)
{
    const int lane = threadIdx.x & 63;
    const long P = (long)a.is * a.is;
    const int nblk = (int)((P + 63) >> 6);
    constexpr int G = kLooseWaves;                // the grid (a power of two)
    const int w = (int)blockIdx.x;
    // Pixel blocks per task (the record is loaded once per task): as few as give every wave of the grid about one task -- the
    // waves run almost alone on their SIMDs, so the kernel lasts as long as its chain of dependent loads and one task.  Up to 64
    // images the lists stay in registers (lane = image: the head and the first four faces in one 16-byte load).
    int flagged = 0;
    int4 mine = make_int4(0, 0, 0, 0);
    for (int b0 = 0; b0 < a.B; b0 += 64) {
        const int4 row = b0 + lane < a.B ? *reinterpret_cast<const int4*>(a.loose_image + (long)(b0 + lane) * kLooseList) : make_int4(0, 0, 0, 0);
        if (b0 == 0) mine = row;
        int n = (row.x >> 4) == a.loose_stamp ? (row.x & 15) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
        flagged += __builtin_amdgcn_readfirstlane(n);
    }
    if (flagged == 0) return;
    const int kChunk = (int)min(16L, max(1L, ((long)flagged * nblk + G - 1) / G));
    const int nchunks = (nblk + kChunk - 1) / kChunk;
    int base = 0;                                 // tasks of the flagged images before this one (mod G): task = (image, face, chunk), wave w takes tasks w, w + G, ...
    for (int b0 = 0; b0 < a.B; b0 += 64) {
        const int4 row = b0 == 0 ? mine : (b0 + lane < a.B ? *reinterpret_cast<const int4*>(a.loose_image + (long)(b0 + lane) * kLooseList) : make_int4(0, 0, 0, 0));
        unsigned long long todo = __ballot((row.x >> 4) == a.loose_stamp && (row.x & 15) != 0);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int b = b0 + l;
            const int n = __builtin_amdgcn_readlane(row.x, l) & 15;
            const int f1 = __builtin_amdgcn_readlane(row.y, l), f2 = __builtin_amdgcn_readlane(row.z, l), f3 = __builtin_amdgcn_readlane(row.w, l);
            const int tasks = n * nchunks;                           // (< 2^31: at most 15 faces of at most 2^24 pixel blocks)
            const RecPtr recs = (RecPtr)a.records + (long)b * a.nf * REC;
            for (int t = (w - base) & (G - 1); t < tasks; t += G) {
                const int k = (int)((unsigned)t / (unsigned)nchunks), ch = t - k * nchunks;
                const int f = k == 0 ? f1 : (k == 1 ? f2 : (k == 2 ? f3 : __builtin_amdgcn_readfirstlane(a.loose_image[(long)b * kLooseList + 1 + k])));
                if ((unsigned)f >= (unsigned)a.nf) continue;                 // (a list head that only looked like one: fresh memory)
                float r[kRecStage3];
                load_record<0, kRecStage3>(r, recs + (long)f * REC);
                int c_lo = 0x7fffffff, c_hi = -1, r_lo = 0x7fffffff, r_hi = -1;
                const int blk_end = min(nblk, (ch + 1) * kChunk);
                for (int blk = ch * kChunk; blk < blk_end; blk++) {
                    const unsigned p = (unsigned)blk * 64u + (unsigned)lane;         // (P < 2^31: image sizes are validated)
                    const int row_p = (int)(p / (unsigned)a.is), col = (int)(p - (unsigned)row_p * (unsigned)a.is);
                    const float xp = pixel_coord(col, a.is, a.r_is), yp = pixel_coord(a.is - 1 - row_p, a.is, a.r_is);
                    Pair q;
                    barycentrics(q, r, xp, yp);
                    bool live = false;
                    if ((long)p < P && inside_box(r, xp, yp)) {  // (the record's box: the reference's own border test :747 lies inside it)
                        live = inside_closed(q);                   // (closed: what the heaviside branch of soft_fragment tests)
                        if (!live && point_to_face(q, r, xp, yp)) live = q.sign > 0.f || q.dx * q.dx + q.dy * q.dy < a.cull_r2;
                    }
                    if (live) { c_lo = min(c_lo, col); c_hi = max(c_hi, col); r_lo = min(r_lo, row_p); r_hi = max(r_hi, row_p); }
                }
                if (__any(c_hi >= 0)) {
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        c_lo = min(c_lo, __shfl_xor(c_lo, d)); c_hi = max(c_hi, __shfl_xor(c_hi, d));
                        r_lo = min(r_lo, __shfl_xor(r_lo, d)); r_hi = max(r_hi, __shfl_xor(r_hi, d));
                    }
                    if (lane == 0) {
                        int* bx = reinterpret_cast<int*>(a.loose_box + ((long)b * a.nf + f));
                        atomicMin(bx + 0, c_lo); atomicMax(bx + 1, c_hi); atomicMin(bx + 2, r_lo); atomicMax(bx + 3, r_hi);
                    }
                }
            }
            base = (base + tasks) & (G - 1);
        }
    }
}

#ifndef GENDR_BIN_LOOP_MAX
#define GENDR_BIN_LOOP_MAX 6
#endif
#ifndef GENDR_BIN_THREADS
#define GENDR_BIN_THREADS 512
#endif
// ---------------------------------------------------------------------------------------------
// binning: masks[b][tile][chunk], bit f of chunk c = face 64c+f of image b may touch the 8x8 tile; and the tile
// queues the render kernels walk.
// One workgroup takes a block of 8x8 tiles (a 64x64 pixel super-tile) of one image; its wavefronts share the face
// chunks.  For a chunk, lane = face (coalesced box load) and lane = tile as well: a ballot finds the few faces whose
// box meets the super-tile at all; for each of them the box is broadcast with v_readlane and every tile lane sets
// its bit.  The words go to LDS, from where the mask rows leave in runs of 8 tiles x chunks words (the rows of 8
// horizontally adjacent tiles are contiguous in HBM).  Once all chunks are done the first wavefront knows, per tile,
// whether its row is empty and appends the tile to its queue: listed tiles grow the queue from the front, the others
// -- the background, three quarters of the headline scene -- from the back of the queue's slots (the forward kernel
// writes their pixels in a store-only loop, backward never looks at them).  Two atomics per workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int kBinThreads = GENDR_BIN_THREADS, kBinWaves = kBinThreads / 64;
constexpr int kBinGroup = 32;      // chunks staged in LDS per round
constexpr int kBinLoopMax = GENDR_BIN_LOOP_MAX;   // up to this many candidate faces of a chunk are broadcast one by one

__device__ __forceinline__ float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float4 bcast4(const float4& v, int l) { return make_float4(bcast(v.x, l), bcast(v.y, l), bcast(v.z, l), bcast(v.w, l)); }

__device__ __forceinline__ int wave_exclusive_scan(int v, int& total)
{
    const int lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}

__global__ __launch_bounds__(kBinThreads) __attribute__((amdgpu_waves_per_eu(GENDR_BIN_WAVES, GENDR_BIN_WAVES))) void bin_faces_kernel(const float* __restrict__ boxes, const RenderArgs a, int supers_x, int cull,
                                                                                                                        float4* __restrict__ clear4, long clear_quads)
{
    // the caller's buffer to clear (gendr_params::clear_ptr: the gradients of the coming backward call): one 16-byte store per
    // thread or so, issued before anything else -- this kernel has 400 times the threads of the per-face setup kernel (which
    // did it until round 3 and paid 2.6 us for it: its waves run alone on their SIMDs)
    for (long q = (long)blockIdx.x * kBinThreads + threadIdx.x; q < clear_quads; q += (long)gridDim.x * kBinThreads)
        clear4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the lists of faces with a loose cull box have been used (loose_faces_kernel ran before this launch): empty them for the
    // next call on this workspace
    if (a.loose_flag && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.B; i += kBinThreads) a.loose_image[(long)i * kLooseList] = 0;
    __shared__ unsigned long long s_words[64][kBinGroup + 1];
    __shared__ int s_listed[64];
    GENDR_SPAN_BEGIN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int is = a.is, tiles_x = a.tiles_x, chunks = a.chunks;
    const int per_image = supers_x * supers_x;
    // Workgroups start in index order and the ones under the object live four times as long as the ones over the
    // background (16 against 4 us at C2), so the super-tiles are handed out from the image centre outwards, ring by
    // ring, all images' first ring first: where the object is roughly centred the long workgroups start first instead
    // of somewhere in a 10-us dispatch ramp; where it is not, the order is as good as any other.
    const int nimg = (int)(gridDim.x / per_image);
    const int b = blockIdx.x % nimg;
    int sy, sx;
    {
        const int r = blockIdx.x / nimg;                        // rank of the super-tile, centre first
        int n = 2 - (supers_x & 1), inner = 0;                  // side of the centred square that holds ranks < n * n
        while (n * n <= r) { inner = n * n; n += 2; }
        const int o = (supers_x - n) >> 1, pos = r - inner;     // the ring is the square's border, origin (o, o)
        if (n == 1)               { sy = o; sx = o; }
        else if (pos < n)         { sy = o;         sx = o + pos; }                       // top row
        else if (pos < 2 * n)     { sy = o + n - 1; sx = o + pos - n; }                   // bottom row
        else if (pos < 3 * n - 2) { sy = o + 1 + pos - 2 * n;       sx = o; }             // left column
        else                      { sy = o + 1 + pos - (3 * n - 2); sx = o + n - 1; }     // right column
    }

    // lane t owns tile t of the super-tile: its rectangle and, at the end, its mask words
    const int ty_l = sy * 8 + (lane >> 3), tx_l = sx * 8 + (lane & 7);
    const bool tile_ok = ty_l < tiles_x && tx_l < tiles_x;
    const float rx_lo = pixel_coord(tx_l * 8, is, a.r_is), rx_hi = pixel_coord(min(tx_l * 8 + 7, is - 1), is, a.r_is);
    const float ry_hi = pixel_coord(is - 1 - ty_l * 8, is, a.r_is), ry_lo = pixel_coord(is - 1 - min(ty_l * 8 + 7, is - 1), is, a.r_is);
    const float sx_lo = pixel_coord(sx * 64, is, a.r_is), sx_hi = pixel_coord(min(sx * 64 + 63, is - 1), is, a.r_is);
    const float sy_hi = pixel_coord(is - 1 - sy * 64, is, a.r_is), sy_lo = pixel_coord(is - 1 - min(sy * 64 + 63, is - 1), is, a.r_is);
    const long tile_base = (long)b * a.tiles_per_image;
    int listed_faces = 0;                                                    // first wavefront: faces listed for this lane's tile

    for (int c0 = 0; c0 < chunks; c0 += kBinGroup) {
        const int ng = min(kBinGroup, chunks - c0);
        constexpr int kPerWave = (kBinGroup + kBinWaves - 1) / kBinWaves;
        // this wave's boxes of the group, all requested before the first one is used (round 6: the loads sat behind each other's
        // candidate loops, one round trip to memory per chunk on the critical path of a workgroup that IS the kernel's duration)
        float4 boxv[kPerWave];
#pragma unroll
        for (int u = 0; u < kPerWave; u++) {
            const int ci = wave + u * kBinWaves;
            const int fi = (c0 + ci) * 64 + lane;
            boxv[u] = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);      // misses everything
            if (ci < ng && fi < a.nf) boxv[u] = reinterpret_cast<const float4*>(boxes)[(long)b * a.nf + fi];
        }
#pragma unroll
        for (int u = 0; u < kPerWave; u++) {
            const int ci = wave + u * kBinWaves;
            if (ci >= ng) break;
            const int fi = (c0 + ci) * 64 + lane;
            const bool have = fi < a.nf;
            float4 box = boxv[u];
            if (have && a.loose_flag && a.loose_flag[(long)b * a.nf + fi]) box = loose_box_ndc(a.loose_box[(long)b * a.nf + fi], is, a.r_is);
            unsigned long long mine = 0ull;
            // Box test only (measured in round 2: an exact per-(face, tile) edge test removes a third of the listings but
            // costs this kernel 43 us at C2; the coverage kernel drops those faces for 10 us).
            unsigned long long cand = __ballot(have && (cull ? rect_hits_box(sx_lo, sx_hi, sy_lo, sy_hi, box) : true));
            if (__popcll(cand) <= kBinLoopMax) {
                // a handful: broadcast each box, every tile lane sets its bit
                while (cand) {
                    const int l = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    const float4 fb = bcast4(box, l);
                    if (cull ? rect_hits_box(rx_lo, rx_hi, ry_lo, ry_hi, fb) : true) mine |= 1ull << l;
                }
            } else {
                // many (the super-tiles under the object): the predicate is separable, and the tile columns (rows) a box meets
                // are one run of the eight.  Every face lane turns its box into the two runs -- in pixel-index space, i =
                // (x + 1) is / 2 - 1/2, rounded OUTWARDS by 2^-6 pixel (the float error is below 2^-10 pixel up to 4096^2): the
                // tile masks only have to be a superset, the coverage kernel applies the exact per-pixel box test -- then one
                // ballot per column and per row (16, not one per tile: 64) and every tile lane picks its column's and its row's
                // word with lane-constant select masks.  NaN boxes keep every tile, as the comparisons of rect_hits_box() do.
                unsigned mx = 0xffu, my = 0xffu;
                if (cull) {
                    constexpr float kPixSlack = 0.015625f;                             // 2^-6 pixel
                    const float half_is = 0.5f * (float)is, fis = (float)is;
                    const float px_lo = (box.x + 1.f) * half_is - 0.5f, px_hi = (box.y + 1.f) * half_is - 0.5f;      // pixel columns of the box
                    const float pr_lo = fis - 0.5f - (box.w + 1.f) * half_is, pr_hi = fis - 0.5f - (box.z + 1.f) * half_is;   // pixel rows (row 0 = top = largest y)
                    const float X0 = (float)(sx * 64), Y0 = (float)(sy * 64);
                    const int kx_lo = (int)fmaxf(ceilf((px_lo - X0 - 7.f - kPixSlack) * 0.125f), 0.f), kx_hi = (int)fminf(floorf((px_hi - X0 + kPixSlack) * 0.125f), 7.f);
                    const int ky_lo = (int)fmaxf(ceilf((pr_lo - Y0 - 7.f - kPixSlack) * 0.125f), 0.f), ky_hi = (int)fminf(floorf((pr_hi - Y0 + kPixSlack) * 0.125f), 7.f);
                    mx = kx_hi >= kx_lo && kx_lo <= 7 && kx_hi >= 0 ? ((2u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u) : 0u;
                    my = ky_hi >= ky_lo && ky_lo <= 7 && ky_hi >= 0 ? ((2u << ky_hi) - 1u) & ~((1u << ky_lo) - 1u) : 0u;
                }
                if (!((cand >> lane) & 1ull)) mx = 0u;              // also drops lanes without a face
                unsigned long long colw = 0ull, roww = 0ull;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const unsigned long long cm = __ballot((mx >> k) & 1u), rm = __ballot((my >> k) & 1u);
                    if (__builtin_amdgcn_inverse_ballot_w64(0x0101010101010101ull << k)) colw = cm;     // lanes of tile column k
                    if (__builtin_amdgcn_inverse_ballot_w64(0xffull << (8 * k))) roww = rm;             // lanes of tile row k
                }
                mine = colw & roww;
            }
            s_words[lane][ci] = mine;
        }
        __syncthreads();
        if (wave == 0) {
            for (int ci = 0; ci < ng; ci++) listed_faces += __popcll(s_words[lane][ci]);
            s_listed[lane] = listed_faces != 0;
        }
        // Nobody reads the mask row of a tile that lists no face (such a tile is not queued), so when the whole row is
        // known here -- one group holds all chunks, up to 2048 faces -- the rows of empty tiles are not written at all
        // (the background: 79 % of the tiles of BASELINE config 5, 265 MB of zeros per call at batch 32).
        const bool skip_empty = chunks <= kBinGroup;
        if (skip_empty) __syncthreads();
        // (no entry pool for this option set, see entry_capacity: nobody reads the masks, and they have no buffer)
        for (int idx = threadIdx.x; idx < (a.ent_cap8 > 0 ? 64 * ng : 0); idx += kBinThreads) {   // consecutive threads: consecutive words of a row
            const int tl = idx / ng, ci = idx - tl * ng;
            const int ty = sy * 8 + (tl >> 3), tx = sx * 8 + (tl & 7);
            if (ty < tiles_x && tx < tiles_x && (!skip_empty || s_listed[tl]))
                const_cast<unsigned long long*>(a.masks)[(tile_base + (long)ty * tiles_x + tx) * chunks + c0 + ci] = s_words[tl][ci];
        }
        __syncthreads();
    }
    if (wave != 0) return;

    // tile queues and the tiles' slices of the entry pool.  Usually the 64 tiles belong to one queue; small batches
    // of small images put several into one wave
    const long g = tile_base + (long)ty_l * tiles_x + tx_l;
    const int xq = tile_ok ? queue_of_tile(g, a.total_tiles) : -1;
    if (!tile_ok) listed_faces = 0;
    // A super-tile that lies wholly inside the image and lists nothing anywhere becomes ONE entry of the unlisted
    // queue, -(its first tile) - 1: the forward kernel fills its 64 x 64 pixels with 256-byte row segments instead of
    // 64 tiles' 32-byte ones (image rows of a multiple of four pixels, for 16-byte stores).
    if ((is & 3) == 0 && sx * 64 + 64 <= is && sy * 64 + 64 <= is) {
        const int x0 = __builtin_amdgcn_readlane(xq, 0);
        if (__ballot(listed_faces == 0 && xq == x0) == ~0ull) {
            if (lane == 0) {
                const int base_e = atomicAdd(a.control + (8 + x0) * kCtlStride, 1);
                a.tile_list[queue_begin(x0 + 1, a.total_tiles) - 1 - base_e] = -(int)g - 1;
            }
            GENDR_SPAN_END(1, blockIdx.x);
            return;
        }
    }
    unsigned long long todo = __ballot(tile_ok);
    while (todo) {
        const int x = __builtin_amdgcn_readlane(xq, __builtin_ctzll(todo));
        const unsigned long long mine = __ballot(xq == x);
        todo &= ~mine;
        const unsigned long long listed = __ballot(xq == x && listed_faces != 0);
        const unsigned long long empty = mine & ~listed;
        int need = 0;
        const int before = wave_exclusive_scan(xq == x ? listed_faces : 0, need);
        int base_l = 0, base_e = 0;
        long base_n = 0;
        if (lane == 0) {
            if (listed) base_l = atomicAdd(a.control + x * kCtlStride, __popcll(listed));
            if (empty)  base_e = atomicAdd(a.control + (8 + x) * kCtlStride, __popcll(empty));
            // Entries handed out so far in region x of the pool (ent_cap8 == 0: this option set has no pool at all, see entry_capacity):
            // a 64-bit counter, zeroed per call, that cannot wrap (nothing culled: 2.7e9 listings at 2048^2 x 5120 faces x 64).  A
            // request that runs past the region leaves its tiles -- and every later request's -- without a slice, which the lanes
            // find out by themselves below.  (Until round 6 the counter was read first and only advanced while the request fitted: a
            // second dependent round trip to the memory-side atomic unit at the end of every workgroup.)
            if (need) {
                base_n = a.ent_cap8;
                if (a.ent_cap8 > 0) base_n = (long)atomicAdd(reinterpret_cast<unsigned long long*>(a.control + (16 + x) * kCtlStride), (unsigned long long)need);
            }
        }
        base_l = __builtin_amdgcn_readfirstlane(base_l);
        base_e = __builtin_amdgcn_readfirstlane(base_e);
        base_n = ((long)__builtin_amdgcn_readfirstlane((int)(base_n >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base_n);
        if ((listed >> lane) & 1ull) {
            const long slot = queue_begin(x, a.total_tiles) + base_l + __popcll(listed & lt);
            const long at = base_n + before;                                    // inside region x of the pool
            const int off = (at >= 0 && at + listed_faces <= a.ent_cap8 && (long)x * a.ent_cap8 + at < 0x7fffffffL) ? (int)((long)x * a.ent_cap8 + at) : -1;
            a.tile_list[slot] = (int)g;
            a.tile_info_raw[slot] = make_int4((int)g, off, 0, 0);               // the coverage kernel fills in the entry count
        }
        if ((empty >> lane) & 1ull)  a.tile_list[queue_begin(x + 1, a.total_tiles) - 1 - (base_e + __popcll(empty & lt))] = (int)g;
    }
    GENDR_SPAN_END(1, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// coverage: which pixels of the tile does each listed face reach?  (once per forward call, shared by both passes)
// ---------------------------------------------------------------------------------------------
// One wavefront per listed tile (the same queue walk as the render kernels).  The tile's mask row is first unpacked
// into an ascending face list in LDS.  Then sixteen faces are examined per step: the four lanes of a quad share a face
// slot, lane q of the quad takes pixel rows q and q + 4; the lane gathers its face's first record stage with vector loads
// -- sixteen records in flight per step instead of one scalar-load round trip per face -- and finds, per row, the
// interval of columns that passes the box / edge tests (see the step loop).  The eight row bytes of a face are OR-ed
// together across its quad (two DPP steps); faces that own at least one pixel are appended, in ascending order, to the
// tile's slice of the entry pool (up to sixteen 16-byte stores per step, contiguous).
constexpr int kListCap = 128;      // faces unpacked per round (>= 64: one mask word must fit)

__device__ __forceinline__ unsigned quad_or(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    return v;
}

// TAGS: the region tags of the entries are worked out (RenderArgs::want_tags).  A template parameter since round 6: as a run-time flag the
// compiler had if-converted the second row's tag arithmetic -- 45 vector instructions per step that ran for every call, also where nobody
// reads the tags (BASELINE config 2).
template <int REC, int WAVES = 1, bool TAGS = false>
__global__ __launch_bounds__(kThreads * WAVES) void cover_kernel(const RenderArgs a)
{
    // WAVES == 8 (team calls, gendr_params::team: few tiles, each listing faces by the hundred -- opt_shape.py's 64^2 images with
    // a 4-pixel logistic tail list 200 faces per tile, and one wave per tile walks them in thirteen dependent steps on a chip
    // that holds 1.5 such waves per SIMD): one WORKGROUP per tile.  Every wave unpacks the same kListCap = 8 x 16 faces of a
    // round (only wave 0 stores the list), wave w examines faces [16 w, 16 w + 16) of it, and the survivors are appended in face
    // order through the waves' counts in LDS -- the same entries in the same slots as the one-wave form.
    static_assert(WAVES == 1 || 16 * WAVES == kListCap, "a round of the team form is one step per wave");
    __shared__ int s_flist[kListCap];
    __shared__ int s_cnt[WAVES > 1 ? WAVES : 1];
    __shared__ int s_pairs;
    GENDR_SPAN_BEGIN;
    TileWalk tw;
    walk_init(tw, a, 1);
    const int wave = WAVES > 1 ? (int)(threadIdx.x >> 6) : 0;
    if (WAVES > 1) tw.rank = tw.next = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3));      // one work item per workgroup
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // sixteen faces per step: the four lanes of a quad share a face, lane q of the quad tests pixel rows q and q + 4 of the tile
    const int slot = lane >> 2, prow = lane & 3;
    for (; tw.next < tw.total; tw.next += tw.stride) {
        GENDR_RELOADED_ARGS(a);
        // The queue was appended to as the binning workgroups finished: the super-tiles under the object, with the most
        // faces to examine here, came last.  The waves take the slots from the back so that those start first.
        const int slot_c = tw.total - 1 - tw.next;
        const i4v qi = *(const GENDR_CONST_AS i4v*)(a.tile_info_raw + (tw.qbase + slot_c)); // (tile, first entry) from the binning kernel
        const int tile = qi.x, off = qi.y;
        if (off < 0) continue;                      // no room in the pool: the render kernels test this tile themselves
        TileCtx t;
        tile_setup(t, a, tile, false);
        const float* recs_g = a.records + (long)t.b * a.nf * REC;
        const unsigned long long* mrow = a.masks + (long)tile * a.chunks;
        const int row_a = t.y0 + prow;
        const bool row_ok[2] = {row_a < a.is, row_a + 4 < a.is};
        const float yp_r[2] = {pixel_coord(a.is - 1 - row_a, a.is, a.r_is), pixel_coord(a.is - 5 - row_a, a.is, a.r_is)};
        CoverEnt* out = a.entries + off;
        int nout = 0;
        // pixel centres of the tile's columns: lane l holds column l & 7's (the exact box step below fetches its two columns with
        // ds_bpermute -- a double-precision product each until round 6), xs0 = column 0's
        const float xs_l = pixel_coord(t.x0 + (lane & 7), a.is, a.r_is);
        const float xs0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xs_l)));
        const float pitch = (float)(2. * a.r_is);
        int my_pairs = 0;                          // pixels this lane's (face, row) slots found: summed into the tile's weight

        int word0 = 0, group0 = 0;                 // next 64-word group to load / base word of the loaded one
        unsigned long long wv = 0ull, nz = 0ull;   // this lane's word of the loaded group / its non-zero words still to unpack
        bool done = false;
        while (!done) {
            // ---- unpack up to kListCap faces
            int nlist = 0;
            for (;;) {
                if (!nz) {
                    if (word0 >= a.chunks) { done = true; break; }
                    wv = word0 + lane < a.chunks ? mrow[word0 + lane] : 0ull;
                    nz = __ballot(wv != 0ull);
                    group0 = word0;
                    word0 += 64;
                    continue;
                }
                const int j = __builtin_ctzll(nz);
                const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(wv >> 32), j) << 32)
                                           | (unsigned)__builtin_amdgcn_readlane((int)wv, j);
                const int cnt = __popcll(w);
                if (nlist + cnt > kListCap) break;                       // the word stays in nz for the next round
                nz &= nz - 1;
                if (wave == 0 && lane_in(w)) s_flist[bits_below(w, nlist)] = (group0 + j) * 64 + lane;
                nlist += cnt;
            }
            if (WAVES > 1) { if (threadIdx.x == 0) s_pairs = 0; __syncthreads(); } else __builtin_amdgcn_wave_barrier();
            // ---- sixteen faces per step (round 4: eight, one row per lane -- a wave of this kernel waits for the dependent loads of
            // each step, face list -> record, far longer than it computes: half the steps, and the per-edge terms that do not
            // depend on the row are shared by the lane's two rows)
            CoverEnt my_ent; my_ent.fn = 0; my_ent.npix = 0; my_ent.lo = 0u; my_ent.hi = 0u;      // team form: this wave's entry of the round, stored after the counts are known
            bool my_owns = false;
            unsigned long long my_keep = 0ull;
            for (int i0 = 16 * wave; i0 < nlist; i0 += 16 * WAVES) {
                const bool has = i0 + slot < nlist;
                const int fn = s_flist[has ? i0 + slot : i0];
                float r[kRecStage1];
                gather_record<0, kRecStage1 / 4>(r, recs_g + (long)fn * REC);
                unsigned m8[2] = {0u, 0u};
                unsigned unproven = 0u;                     // region tag: what this lane's rows leave unproven (rows without a pixel: nothing)
                // Inside the record's box the entries only have to be a SUPERSET of the contributing pairs (every listed pair still
                // meets the reference's distance and probability tests, :769 / :784, in the render kernels; the box itself is exact,
                // see below).  Along a pixel row each barycentric is linear in the column c = 0..7,
                // w_k(c) = w_k(0) + c d_k, so the columns that pass all three edge thresholds and the cull box form ONE interval:
                // its ends are three quotients (t_k - w_k(0)) / d_k and the two box columns -- about half the vector instructions
                // of testing the eight pixels one by one (round 2 stepped w_k from pixel to pixel: 30.0 -> 27.2 us at C2; the
                // interval form: see DESIGN.md).  Error budget, all towards MORE pixels: the thresholds t_k are lowered by
                // 2^-19 (|a| + |b| + |c|) -- the model w_k(0) + c d_k and the expression the thresholds were derived for
                // (barycentrics(), three roundings, on the pixel centre pixel_coord() returns) differ by a few roundings of
                // magnitudes below that sum, as before; the quotients (one rounded subtraction, v_rcp_f32, one product: relative
                // error < 2^-21, i.e. < 2^-17 columns wherever the bound lies within [-1, 9]) and the pixel pitch model of the box
                // test (< 2^-13 columns up to 4096^2) are covered by widening the interval by kColSlack = 2^-8 column at either
                // end.  A coefficient that is zero, tiny, infinite or NaN makes its constraint constant along the row (kept unless
                // it provably fails); NaN never drops a pixel (v_max / v_min return the other operand).
                constexpr float kSlack = 1.9073486328125e-06f;                     // 2^-19
                constexpr float kColSlack = 0.00390625f;                           // 2^-8
                // The record's BOX, though, is tested EXACTLY: it contains the reference's own border test (kernel.cu:747), which no
                // later stage repeats for a listed pair -- with a small dist_eps it is the box, not the distribution's tail, that ends a
                // face's reach (round 3's interval form widened the box ends by the column slack like the others and listed pixels whose
                // centre lies up to 2^-8 column outside: found by tools/fuzz_parity.py case 255, a box edge 0.0015 column from a pixel
                // centre).  Rows: the y test below is the per-pixel expression.  Columns (the same for both rows): the widened ends,
                // then one exact step at either end on the pixel centre pixel_coord() returns -- the render kernels' x.
                const float half_is = 0.5f * (float)a.is;                          // 1 / pitch
                int cb_first = max(0, (int)ceilf(fminf(fmaxf((r[kRecBox + 0] - xs0) * half_is, -1.f), 16.f) - kColSlack));
                int cb_last = min(min(7, a.is - 1 - t.x0), (int)floorf(fmaxf(fminf((r[kRecBox + 1] - xs0) * half_is, 9.f), -2.f) + kColSlack));
                // (a column outside 0..7 fetches some other column's centre: the interval is empty then, whatever the step does)
                const float x_first = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(cb_first << 2, __builtin_bit_cast(int, xs_l)));
                const float x_last = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(cb_last << 2, __builtin_bit_cast(int, xs_l)));
                if (x_first < r[kRecBox + 0]) cb_first++;                                         // (NaN ends exclude nothing, as in inside_box())
                if (x_last > r[kRecBox + 1]) cb_last--;
                // what does not depend on the row, once for the lane's two rows (the rows sit in branches of their own: the compiler does
                // not share it by itself -- round 6: 15 vector instructions per step)
                float tk3[3], dk3[3], rdk3[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float ak = r[kRecInv + 3 * k], bk = r[kRecInv + 3 * k + 1], ck = r[kRecInv + 3 * k + 2];
                    // threshold minus the row-independent part of w_k(0) = (ak x0 + ck) + bk y: the row passes where c dk >= tk3 - bk y.
                    // That difference is rounded at the magnitude of the threshold (|wcull_k| <= Rf (|ak| + |bk|), a dozen times the
                    // sum for a cull radius of several image widths), which the 2^-19 sum does not cover: the threshold (<= 0, or
                    // -inf) is lowered by another 2^-21 of itself -- eight roundings of its magnitude.
                    tk3[k] = (r[kRecWCull + k] * 1.000000476837158203125f - kSlack * (fabsf(ak) + fabsf(bk) + fabsf(ck))) - (ak * xs0 + ck);
                    dk3[k] = ak * pitch;
                    rdk3[k] = __builtin_amdgcn_rcpf(dk3[k]);
                }
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const float yp_a = yp_r[rr];
                    if (!(has && row_ok[rr] && !(yp_a > r[kRecBox + 3] || yp_a < r[kRecBox + 2]))) continue;
                    float lo = -1.f, hi = 9.f;                                         // the interval of columns, in column units
                    bool none = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float ak = r[kRecInv + 3 * k], bk = r[kRecInv + 3 * k + 1], ck = r[kRecInv + 3 * k + 2];
                        const float dk = dk3[k];
                        const float u = tk3[k] - bk * yp_a;                            // the row passes where c dk >= u
                        const float q = u * rdk3[k];
                        const bool up = dk > 1e-30f, down = dk < -1e-30f;
                        lo = fmaxf(lo, up ? q : -1.f);
                        hi = fminf(hi, down ? q : 9.f);
                        none = none || (!up && !down && u > 8e-30f);                   // constant along the row (to 7e-30), and failing
                    }
                    const int c_first = max(cb_first, (int)ceilf(fminf(lo, 16.f) - kColSlack));
                    const int c_last = min(cb_last, (int)floorf(fmaxf(hi, -2.f) + kColSlack));
                    if (!none && c_last >= c_first) m8[rr] = ((2u << c_last) - 1u) & ~((1u << c_first) - 1u);
                    // Region tag (see CoverEnt): along the row's pixels c_first .. c_last every barycentric is linear, so its sign
                    // is settled by the two ends -- where both lie beyond a margin of 2^-19 (|a| + |b| + |c|) on the same side (the
                    // model w_k(0) + c d_k and the value barycentrics() computes for the pixel differ by less than half of that,
                    // see kSlack above).  Bits 0..2: w_k <= 0 NOT proven for the row; bits 3..5: w_k > 0 NOT proven.  NaN proves
                    // nothing.
                    if (TAGS && m8[rr]) {
                        const float cf = (float)c_first, cl = (float)c_last;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float ak = r[kRecInv + 3 * k], bk = r[kRecInv + 3 * k + 1], ck = r[kRecInv + 3 * k + 2];
                            const float w = ak * xs0 + bk * yp_a + ck, dk = ak * pitch;
                            const float mk = kSlack * (fabsf(ak) + fabsf(bk) + fabsf(ck));
                            const float wa = w + cf * dk, wb = w + cl * dk;
                            if (!(fmaxf(wa, wb) <= -mk) || !(wa == wa) || !(wb == wb)) unproven |= 1u << k;
                            if (!(fminf(wa, wb) >= mk) || !(wa == wa) || !(wb == wb)) unproven |= 8u << k;
                        }
                    }
                }
                const bool loose_l = has && __float_as_int(r[kRecLoose]) != 0;         // (the four lanes of a quad hold the same record)
                if (loose_l) m8[0] = m8[1] = 0u;
                unsigned v = quad_or(m8[0] << (8 * prow));                            // rows 0-3 of the face, in all four lanes of its quad
                unsigned hi = quad_or(m8[1] << (8 * prow));                           // rows 4-7
                // A face with a LOOSE cull box (face_setup_kernel: seen edge-on, no usable error bound -- its box is the reference's
                // own margin and the binning kernel listed it in every tile of its image) is EVALUATED instead of bounded: lane =
                // pixel of the tile, the record in scalar registers, barycentrics() and point_to_face() -- the render kernels'
                // functions on the same operands -- and the pixel is LIVE if it lies inside the record's box and inside the face or
                // closer to it than the cull radius (beyond the radius the pair fails :769 or :784, which is what the radius is
                // defined by; NaN barycentrics drop the pair as everywhere).  The render kernels still apply the reference's own
                // tests to every listed pair, so the mask only has to contain the pixels that can contribute -- it contains exactly
                // the live ones, and a tile without any drops the face.  (Round 3 ran a launch of its own for this -- every pixel
                // of the image once, leaving a bounding box for the binning kernel -- which cost 5-10 us per call and was therefore
                // switched on from 1024^2 only; here it costs the affected tiles' waves ~1 us per flagged face, spread over the
                // whole chip, and nothing anywhere else.)
                unsigned long long flagged = __ballot(loose_l && prow == 0);
                if (flagged) tile_lanes(t, a);
                while (flagged) {
                    const int l = __builtin_ctzll(flagged);
                    flagged &= flagged - 1;
                    const int fn_l = __builtin_amdgcn_readlane(fn, l);
                    float rl[kRecStage3];
                    load_record<0, kRecStage3>(rl, (RecPtr)recs_g + (long)fn_l * REC);
                    Pair q;
                    barycentrics(q, rl, t.xp, t.yp);
                    bool live = false;
                    if (t.valid && inside_box(rl, t.xp, t.yp)) {
                        live = inside_closed(q);                                       // (closed: what the heaviside branch of soft_fragment tests)
                        if (!live && point_to_face(q, rl, t.xp, t.yp)) live = q.sign > 0.f || q.dx * q.dx + q.dy * q.dy < a.cull_r2;
                    }
                    const unsigned long long lm = __ballot(live);
                    if (lane == l) { v = (unsigned)lm; hi = (unsigned)(lm >> 32); my_pairs += __popcll(lm); }
                }
                my_pairs += __popc(m8[0]) + __popc(m8[1]);
                // the rows' verdicts OR-ed over the face's four lanes (as the row masks above), then the decision tree of
                // kernel.cu:120-142 on the proven signs; a corner region whose obtuse-angle test (:122, :128, :134) would have to be
                // evaluated per pixel gets no tag
                const unsigned up = quad_or(unproven);
                int tag = 0;
                if (TAGS && !loose_l) {
                    const int bits = __float_as_int(r[kRecBits]);
                    const bool n0 = !(up & 1u), n1 = !(up & 2u), n2 = !(up & 4u);          // w_k <= 0 on every pixel
                    const bool p0 = !(up & 8u), p1 = !(up & 16u), p2 = !(up & 32u);       // w_k > 0 on every pixel
                    if ((n0 || p0) && (n1 || p1) && (n2 || p2)) {
                        if (n1 && n2)      tag = (bits & 1) ? 0 : 1;           // v0 = 0
                        else if (n2 && n0) tag = (bits & 2) ? 0 : 2;           // v0 = 1
                        else if (n0 && n1) tag = (bits & 4) ? 0 : 3;           // v0 = 2
                        else if (n0)       tag = 2;                            // v0 = 1
                        else if (n1)       tag = 3;                            // v0 = 2
                        else if (n2)       tag = 1;                            // v0 = 0
                    }
                }
                const bool owns = prow == 0 && (v | hi) != 0u;
                const unsigned long long keep = __ballot(owns);
                CoverEnt e;
                e.fn = fn; e.npix = (__popc(v) + __popc(hi)) | (tag << 8); e.lo = v; e.hi = hi;
                if (WAVES > 1) { my_ent = e; my_owns = owns; my_keep = keep; }
                else {
                    if (owns) out[bits_below(keep, nout)] = e;
                    nout += __popcll(keep);
                }
            }
            if (WAVES > 1) {
                if (lane == 0) s_cnt[wave] = __popcll(my_keep);
                __syncthreads();
                int before = 0, all = 0;
#pragma unroll
                for (int k = 0; k < WAVES; k++) { const int c = s_cnt[k]; all += c; if (k < wave) before += c; }
                if (my_owns) out[bits_below(my_keep, nout + before)] = my_ent;
                nout += all;
                __syncthreads();                       // s_flist and s_cnt are rewritten by the next round
            } else __builtin_amdgcn_wave_barrier();
        }
        // the tile's (pixel, face) pairs = its weight for order_tiles_kernel: summed across the lanes into lane 63 (an
        // inclusive scan inside each row of 16 lanes, then the two DPP row broadcasts: six VALU steps, no LDS)
#define GENDR_DPP_IADD(v, ctrl, rows) ((v) + __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xF, false))
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x111, 0xF);      // row_shr:1
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x112, 0xF);      // row_shr:2
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x114, 0xF);      // row_shr:4
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x118, 0xF);      // row_shr:8
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x142, 0xA);      // row_bcast:15 into rows 1 and 3
        my_pairs = GENDR_DPP_IADD(my_pairs, 0x143, 0xC);      // row_bcast:31 into rows 2 and 3
#undef GENDR_DPP_IADD
        if (WAVES > 1) {
            if (lane == 63) atomicAdd(&s_pairs, my_pairs);
            __syncthreads();
            if (threadIdx.x == 63) a.tile_info_raw[tw.qbase + slot_c] = make_int4(tile, off, nout, s_pairs);
            __syncthreads();                           // s_pairs is cleared by the next tile's first round
        } else if (lane == 63) a.tile_info_raw[tw.qbase + slot_c] = make_int4(tile, off, nout, my_pairs);
    }
    GENDR_SPAN_END(0, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// heavy tiles first
// ---------------------------------------------------------------------------------------------
// Workgroups are dispatched in index order and wave r of a queue takes slot r, so the tiles a launch starts last are
// the ones the binning kernel happened to append last -- and the launch ends when the longest of them does.  Measured on
// the headline scene (tools/wave_trace.py): the last waves of the backward kernel start at 80-88 us and run 40-48 us
// (10-12 batches), so a third of the 129 us passes with the chip emptying.  This kernel (one workgroup per queue) copies
// the queue's records into a second array sorted by descending weight class (the pairs the coverage kernel counted, in
// steps of 32) and the render kernels walk that copy: the heavy tiles start first and the tail is made of one-batch
// tiles (C2: backward kernel 124 -> 107 us, forward 108 -> 92 us; C3 +18 % frames/s).  Launched only for up to
// kOrderTilesMax tiles in all (the host decides; RenderArgs::tile_info then points at the copy): with dozens of tiles per
// wave slot the tail does not matter, and neighbouring tiles running together share their records and output lines in the
// L2 (C5, 2.1 M tiles at batch 32: 3.5 % slower when ordered).
constexpr int kOrderThreads = 1024, kOrderClasses = 32;
constexpr long kOrderTilesMax = 1L << 19;
// weight class of a queue record (tile, first entry, entries, pairs): 0 = no entries at all (a tile with a slice of the pool
// whose list came out empty), else 1 + pairs / 32, capped
// (shift: 5 = steps of 32 pairs; the team kernels' calls, whose live tiles hold thousands of pairs, sort in steps of 512 -- kTeamClassShift)
__device__ __forceinline__ int order_class(const int4& rec, int shift = 5)
{
    if (rec.y >= 0 && rec.z == 0) return 0;
    return min((rec.w >> shift) + 1, kOrderClasses - 1);
}
#ifndef GENDR_TEAM_PIECE
#define GENDR_TEAM_PIECE 8
#endif
constexpr int kTeamClassShift = 9, kTeamPieceClasses = GENDR_TEAM_PIECE;      // team calls: a tile of more than 8 x 512 pairs is cut into parts (backward; measured 4 / 8: 102 / 97 us)

__global__ __launch_bounds__(kOrderThreads) void order_tiles_kernel(const RenderArgs a, int budget, int team)
{
    const int shift = team ? kTeamClassShift : 5;
    __shared__ int s_count[kOrderClasses], s_cursor[kOrderClasses];
    const int x = blockIdx.x;
    const long qbase = queue_begin(x, a.total_tiles);
    const int n = a.control[x * kCtlStride];
    if (threadIdx.x < kOrderClasses) s_count[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kOrderThreads)
        atomicAdd(&s_count[order_class(a.tile_info_raw[qbase + i], shift)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int at = 0;
        for (int c = kOrderClasses - 1; c >= 0; c--) { s_cursor[c] = at; at += s_count[c]; }
        // class 0 = tiles whose coverage list came out EMPTY (every face the binning kernel listed for them was dropped by the
        // exact tests -- the tiles of an image that only a face with a loose cull box reaches): they end up behind all others,
        // and the render kernels treat them as unlisted (kCtlLive = the number of tiles before them)
        const int live = n - s_count[0];
        a.control[x * kCtlStride + kCtlLive] = live;
        // split grades (see TileWalk): class c >= 1 holds the tiles of 32 (c - 1) .. 32 c - 1 pairs; a tile is split 2-, 4-, 8-fold if
        // its class exceeds tc, 2 tc, 4 tc, with the smallest tc >= 2 (pieces of 64 pairs: one batch) whose work items fit `budget`
        int g8 = 0, g4 = 0, g2 = 0;
        if (team) {
            // team calls (gendr_team.h): the backward teams take a heavy tile in 2 or 4 PARTS -- ranges of its batches, there is no
            // order to keep -- of at most kTeamPieceClasses x 512 pairs or so, whatever the number of teams: the parts are dealt out
            // heaviest tile first, so the teams' loads even out; the forward teams ignore the grades (a fold cannot be cut)
            // (2 or 4 parts: the classes end at kOrderClasses - 1 = 31 = tiles of 15 872 pairs and more, below the 4 x kTeamPieceClasses an
            // 8-fold grade would start at -- ADVICE r5: the n8 count of round 5 was dead code)
            int n2 = 0, n4 = 0;
            for (int c = kOrderClasses - 1; c > kTeamPieceClasses; c--) {
                n2 += s_count[c];
                if (c > 2 * kTeamPieceClasses) n4 += s_count[c];
            }
            g8 = 0; g4 = n4; g2 = n2 - n4;
        } else if (live > 0 && live < budget) {
            int above[kOrderClasses + 1];                    // above[c] = tiles of a class > c
            above[kOrderClasses] = 0;
            for (int c = kOrderClasses - 1; c >= 0; c--) above[c] = above[c + 1] + (c + 1 < kOrderClasses ? s_count[c + 1] : 0);
            for (int tc = 2; tc < kOrderClasses; tc++) {
                const int n2 = above[tc], n4 = above[min(2 * tc, kOrderClasses)], n8 = above[min(4 * tc, kOrderClasses)];
                if (live + n2 + 2 * n4 + 4 * n8 <= budget) { g8 = n8; g4 = n4 - n8; g2 = n2 - n4; break; }
            }
        }
        a.control[x * kCtlStride + kCtlGrade] = g8;
        a.control[x * kCtlStride + kCtlGrade + 1] = g4;
        a.control[x * kCtlStride + kCtlGrade + 2] = g2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kOrderThreads) {
        const int4 rec = a.tile_info_raw[qbase + i];
        a.tile_info[qbase + atomicAdd(&s_cursor[order_class(rec, shift)], 1)] = rec;
    }
}

// A tile's (pixel, face) pairs as a list of 4-byte codes (face << 6 | pixel lane), in ascending (face, pixel) order, built
// in a wavefront-private LDS buffer from the tile's coverage entries, and handed to body(first code, pairs) in batches of
// 64 consecutive codes -- lane l of phase B takes code l of the batch.  A batch is just a window of the list: faces are
// split wherever the window ends, every batch but the tile's last is full.
//   * entries of the tile's slice of the pool arrive eight at a time by two scalar loads (face and mask in scalar registers:
//     round 6; pixel mode and the fallback below keep up to 64 entries in lane-indexed registers read back with v_readlane);
//     the wave-uniform append step: the lanes (= pixels) whose bit is set store their code at list position base + (set bits
//     below the lane, v_mbcnt on the scalar mask) -- five vector instructions per entry, no per-batch bookkeeping (round 2
//     built the batches entry by entry with their face tables and split decisions: ~45 instructions per entry, and per tile
//     that was as much as a batch of pair math);
//   * entries are appended until at least kFillCodes are listed; the full batches run -- from ONE call site, so that the
//     caller's phase B is compiled once -- and the remainder (< 64 codes) moves to the front of the buffer;
//   * `pixels` restricts the list to the pixel rows this wave renders (sub-tile split, see TileWalk);
//   * a tile without a slice of the entry pool (off < 0: the pool is exhausted, e.g. a heavy-tailed distribution lists
//     every face in every tile) produces its entries here instead, up to 64 at a time: ALL faces of the image are walked
//     with the face's first record stage in SGPRs (scalar loads) and every lane applies the exact per-pixel tests
//     (collect_pairs) -- same entries, only slower (the reference's own traversal).
constexpr int kChunkBatches = 4;
constexpr int kFillCodes = kChunkBatches * 64, kCodeCap = kFillCodes + 64;

// Dense entries (round 4): an entry whose mask holds ALL 64 pixels of the tile -- the rule wherever a distribution's tail
// reaches tens of pixels (BASELINE config 4: logistic, 35 pixels; the reference's opt_shape.py at 64^2) -- is not turned into
// codes at all: `dense(face)` runs it with lane = pixel, the face's record in scalar registers (one s_load stream instead of
// 64 per-lane gathers; the closest-point search, CDF, depth and colour on scalar operands) and folds the result straight into
// the pixel's registers -- no code list, no per-pixel pair sets, no round trip through LDS.  The codes listed BEFORE it are
// flushed first (their faces come first in every pixel's fold order), as a last partial batch if need be.  Compiled into the
// kernels of long-tailed distributions only (dense_path<DIST>()): the others never meet such entries often enough to pay for
// the registers.
template <int DIST> __host__ __device__ constexpr bool dense_path()
{
#ifndef GENDR_DENSE_GAMMA
#define GENDR_DENSE_GAMMA 0
#endif
    return DIST < 0 || DIST == kLogistic || (GENDR_DENSE_GAMMA && DIST == kGamma);      // the runtime-dispatch kernels, and BASELINE config 4's
}

#ifndef GENDR_PIXEL_MODE_AVG
#define GENDR_PIXEL_MODE_AVG 36
#endif
constexpr int kPixelModeAvg = GENDR_PIXEL_MODE_AVG;
// (tile, first entry, entries, pairs) of an unsplit tile with a slice of the entry pool -> rendered in pixel mode?
__device__ __forceinline__ bool tile_in_pixel_mode(const i4v& ti, int split_log2)
{
    return split_log2 == 0 && ti.y >= 0 && ti.z > 0 && ti.w >= kPixelModeAvg * ti.z;
}

#ifndef GENDR_SCALAR_ENTRIES
#define GENDR_SCALAR_ENTRIES 1
#endif
#ifndef GENDR_SE_GROUP
#define GENDR_SE_GROUP 8          // entries per round of scalar loads (4 or 8: two s_load_dwordx16 in flight; +0.7 % at C2 batch 64, +1 % at batch 1)
#endif
template <int REC, bool DENSE, typename Body, typename DenseBody>
__device__ __forceinline__ void for_each_batch(const RenderArgs& a, const TileCtx& t, int off, int cnt, unsigned long long pixels, bool pixel_mode, int* s_code, Body body, DenseBody dense)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int4* ents = reinterpret_cast<const int4*>(a.entries + max(off, 0));
    // fallback: the next face to examine
    const RecPtr recs = (RecPtr)a.records + (long)t.b * a.nf * REC;
    int next_fn = 0;
    bool exhausted = off >= 0;

    int e0 = 0, n = 0, j = 0;
    int4 e = make_int4(0, 0, 0, 0);
    // PIXEL MODE: a tile whose entries hold kPixelModeAvg pixels and more on average (the queue record knows: pairs / entries)
    // is rendered entry by entry with lane = pixel throughout -- every entry takes the dense path under its own pixel mask, no
    // code list, no batches: at 54 of 64 pixels per entry (BASELINE config 4) a step of ~200 instructions for one face beats
    // 54 / 64 of a batch of ~450, and nothing has to be flushed.
    if (DENSE && pixel_mode) {
        for (; e0 < cnt; e0 += 64) {
            n = min(64, cnt - e0);
            if (lane < n) e = ents[e0 + lane];
            for (j = 0; j < n; j++) {
                const int fn = __builtin_amdgcn_readlane(e.x, j);
                const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(e.w, j) << 32) | (unsigned)__builtin_amdgcn_readlane(e.z, j);
                dense(fn, m, (__builtin_amdgcn_readlane(e.y, j) >> 8) & 3);
            }
        }
        return;
    }
    int npairs = 0;                                  // codes in s_code[0, npairs)
    int dense_fn = -1, dense_tag = 0;                // a full-tile entry waiting behind the codes listed so far (and its region tag)
    bool done = false;
    for (;;) {
        // ---- fill: append entries until kFillCodes are listed, a dense entry turns up or the tile's entries are used up
        while (!done && npairs < kFillCodes) {
#if GENDR_SCALAR_ENTRIES
            if (off >= 0) {
                // the tile's slice of the entry pool, eight entries to a round of two scalar loads: face and mask arrive in scalar registers,
                // no lane-indexed copy to read back (a load may run past the slice: into the next tile's entries, or the
                // workspace block that follows the pool)
                if (e0 >= cnt) { done = true; break; }
                const GENDR_CONST_AS i16v* gp = (const GENDR_CONST_AS i16v*)(a.entries + off + e0);
                i16v gq[GENDR_SE_GROUP / 4];
#pragma unroll
                for (int q = 0; q < GENDR_SE_GROUP / 4; q++) gq[q] = gp[q];
#pragma unroll
                for (int k = 0; k < GENDR_SE_GROUP; k++) {
                    if (k > 0 && (e0 >= cnt || npairs >= kFillCodes)) break;
                    const i16v g = gq[k >> 2];
                    const int fn = g[4 * (k & 3)];
                    const unsigned long long m = (((unsigned long long)(unsigned)g[4 * (k & 3) + 3] << 32) | (unsigned)g[4 * (k & 3) + 2]) & pixels;
                    e0++;
                    if (DENSE && m == ~0ull) { dense_fn = fn; dense_tag = (g[4 * (k & 3) + 1] >> 8) & 3; break; }
                    if (lane_in(m)) s_code[bits_below(m, npairs)] = (fn << 6) | lane;
                    npairs += __popcll(m);
                }
                if (DENSE && dense_fn >= 0) break;
                continue;
            }
#endif
            if (j == n) {
                // next group of up to 64 entries into lane-indexed registers
                j = 0; n = 0;
                if (off >= 0) {
                    n = min(64, cnt - e0);
                    if (lane < n) e = ents[e0 + lane];
                    e0 += n;
                } else {
                    while (n < 64 && !exhausted) {
                        if (next_fn >= a.nf) { exhausted = true; break; }
                        const int fn = next_fn++;
                        Pair q;
                        const unsigned long long m = collect_pairs<REC>(t, recs + (long)fn * REC, q);
                        if (m) {
                            if (lane == n) e = make_int4(fn, 0, (int)(unsigned)m, (int)(unsigned)(m >> 32));
                            n++;
                        }
                    }
                }
                if (n <= 0) { done = true; break; }
            }
            const int fn = __builtin_amdgcn_readlane(e.x, j);
            const unsigned long long m = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(e.w, j) << 32) | (unsigned)__builtin_amdgcn_readlane(e.z, j))
                                         & pixels;              // this wave's rows of the tile, see sub_tile_mask
            j++;
            if (DENSE && m == ~0ull) { dense_fn = fn; dense_tag = (__builtin_amdgcn_readlane(e.y, j - 1) >> 8) & 3; break; }
            if (lane_in(m)) s_code[bits_below(m, npairs)] = (fn << 6) | lane;
            npairs += __popcll(m);
        }
        // ---- drain: the full batches; at the end of the tile, or ahead of a dense entry, the partial one as well
        const bool flush = done || dense_fn >= 0;
        const int nb = flush ? (npairs + 63) >> 6 : npairs >> 6;
        if (nb > 0) {
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < nb; k++) body(k << 6, min(64, npairs - (k << 6)));
        }
        if (DENSE && dense_fn >= 0) {
            __builtin_amdgcn_wave_barrier();
            dense(dense_fn, ~0ull, dense_tag);
            dense_fn = -1;
            npairs = 0;
            continue;
        }
        if (done) break;
        const int rem = npairs - (nb << 6);
        __builtin_amdgcn_wave_barrier();
        const int keep = lane < rem ? s_code[(nb << 6) + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane < rem) s_code[lane] = keep;
        npairs = rem;
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct FwdRes {            // 32 bytes: what phase C needs from a pair
    float frag;            // soft fragment D
    float z;               // zp (hard RGB) or zn (softmax)
    float c0, c1, c2;      // sampled colour
    int   flags;
    int   fn;              // face index (hard RGB keeps the nearest face's, kernel.cu:819)
    int   pad1;
};

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void render_forward_body(const RenderArgs& a)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int WAVES = kThreads / 64;
    __shared__ int s_code[WAVES][kCodeCap];                 // the tile's pair list (face << 6 | pixel), see for_each_batch
    __shared__ float2 s_xy[WAVES][64];                      // pixel centres of the tile, fetched by pair lanes
    __shared__ __attribute__((aligned(16))) FwdRes  s_res[WAVES][64];
    __shared__ unsigned long long s_mask[WAVES][64];        // per pixel: which pairs of the running batch are its own

    const int wave = threadIdx.x >> 6;
    const long P = (long)a.is * a.is;
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);
    constexpr bool kSil = RGB == kRgbNone;       // alpha-only: `rgba` is one plane [B,is,is], nothing else is written
    GENDR_SPAN_BEGIN;

    TileWalk tw;
    walk_init(tw, a, WAVES);

    // Tiles no face is listed for: what the loop below leaves for an untouched pixel (kernel.cu:728-740, :845-861).
    // First, so that these stores (two thirds of the output planes in the headline scene) drain while the wave computes.
#if GENDR_ABLATE != 5
    auto fill_tile = [&](int tile) __attribute__((always_inline)) {
        TileCtx t;
        tile_setup(t, a, tile);
        if (!t.valid) return;
        if constexpr (kSil) { a.rgba[(long)t.b * P + t.pix] = 0.f; return; }
        float* out = a.rgba + (long)t.b * 4 * P + t.pix;
        float* aux = a.aux + (long)t.b * 2 * P + t.pix;
        const bool with_aux = !a.p.skip_unlisted_aux;      // backward never reads the aggrs_info of an unlisted tile
        out[3 * P] = 0.f;
        if (!rgb_soft) {
            if (!a.p.background_from_buffer) {
#pragma unroll
                for (int k = 0; k < 3; k++) out[k * P] = a.p.background[k];
            }
            if (with_aux) { aux[0] = 10000000.f; aux[P] = -1.f; }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (a.p.background_from_buffer) { const float bgk = out[k * P]; out[k * P] = (bgk * a.softmax_sum0) / a.softmax_sum0; }
                else out[k * P] = a.bg_soft[k];
            }
            if (with_aux) { aux[0] = a.softmax_sum0; aux[P] = a.p.aggr_rgb_eps; }
        }
    };
    // 64 x 64 pixels of one plane from `first` (16-byte aligned): lane = (row & 3, 4 pixels), sixteen 1-KiB stores
    auto fill_plane = [&](float* first, float v) __attribute__((always_inline)) {
        const int lane = threadIdx.x & 63;
        float4* p4 = reinterpret_cast<float4*>(first + (long)(lane >> 4) * a.is + (lane & 15) * 4);
        const float4 v4 = make_float4(v, v, v, v);
#pragma unroll
        for (int it = 0; it < 16; it++) p4[(long)it * a.is] = v4;           // 4 rows further = is float4s further
    };
    const bool wide_ok = !a.p.background_from_buffer && ((reinterpret_cast<unsigned long long>(a.rgba) | (kSil ? 0ull : reinterpret_cast<unsigned long long>(a.aux))) & 15ull) == 0ull;
    // (... and the listed tiles whose coverage list came out empty -- behind the live ones in the heavy-first copy: same loop, so
    // that fill_tile is inlined once)
#ifndef GENDR_FILL_DIV
#define GENDR_FILL_DIV 2
#endif
    // which waves fill: the last 1 / GENDR_FILL_DIV of the queue's waves (dispatched last: their stores are spread over the launch
    // instead of saturating the memory system while every wave of the first generation waits behind its share)
    const int fill_stride = (GENDR_FILL_DIV > 1 && tw.stride >= 8 * GENDR_FILL_DIV) ? tw.stride / GENDR_FILL_DIV : tw.stride;
    const int fill_first = tw.stride - fill_stride;
    for (int r = tw.rank >= fill_first ? tw.rank - fill_first : 0x7fffffff; r < tw.empties + (tw.total - tw.live); r += fill_stride) {
        const int tile = __builtin_amdgcn_readfirstlane(r < tw.empties ? a.tile_list[tw.qend - 1 - r] : a.tile_info[tw.qbase + tw.live + (r - tw.empties)].x);
        // a negative entry is an empty super-tile (see bin_faces_kernel): tiles g0 + 8 rows of 8
        const int g0 = tile < 0 ? -tile - 1 : tile;
        if (tile >= 0 || !wide_ok) {
            const int n = tile < 0 ? 64 : 1;
            for (int k = 0; k < n; k++) fill_tile(g0 + (k >> 3) * a.tiles_x + (k & 7));
            continue;
        }
        const int b = fast_div(g0, a.div_tpi_m, a.div_tpi_s);
        const int tl = g0 - b * a.tiles_per_image;
        const int ty = fast_div(tl, a.div_tx_m, a.div_tx_s), tx = tl - ty * a.tiles_x;
        const long at = (long)ty * 8 * a.is + tx * 8;
        if constexpr (kSil) { fill_plane(a.rgba + (long)b * P + at, 0.f); continue; }
        float* out = a.rgba + (long)b * 4 * P + at;
        float* aux = a.aux + (long)b * 2 * P + at;
#pragma unroll
        for (int k = 0; k < 3; k++)
            fill_plane(out + k * P, rgb_soft ? a.bg_soft[k] : a.p.background[k]);
        fill_plane(out + 3 * P, 0.f);
        if (!a.p.skip_unlisted_aux) {
            fill_plane(aux, rgb_soft ? a.softmax_sum0 : 10000000.f);
            fill_plane(aux + P, rgb_soft ? a.p.aggr_rgb_eps : -1.f);
        }
    }
#endif

    // pair hints for backward (PairHints): written for the tiles that are not split among several waves
    const bool hints_q = a.hints != nullptr;
    if (hints_q && tw.rank == 0 && (threadIdx.x & 63) == 0) atomicOr(a.control + (blockIdx.x & 7) * kCtlStride + kCtlHintFlag, 1);
    for (; tw.next < tw.items; tw.next += tw.stride) {
    GENDR_RELOADED_ARGS(a);
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int split_log2, sub;
    const int slot_i = walk_item(tw, tw.next, split_log2, sub);
    const i4v ti = *(const GENDR_CONST_AS i4v*)(a.tile_info + (tw.qbase + slot_i));   // (tile, first entry, entries, pairs): scalar load
    const unsigned long long my_rows = sub_tile_mask(split_log2, sub);
    const bool pixel_mode = dense_path<DIST>() && tile_in_pixel_mode(ti, split_log2);      // (no batches, hence no hints: the same decision in backward)
    PairHints* hint_slot = (hints_q && split_log2 == 0 && ti.y >= 0 && !pixel_mode) ? a.hints + ti.y : nullptr;   // next batch's slot (a tile without a slice of the entry pool has none)
    TileCtx t;
    tile_setup(t, a, ti.x);
    t.valid = t.valid && lane_in(my_rows);        // the pixels this wave renders

    s_xy[wave][lane] = make_float2(t.xp, t.yp);

    // per-pixel state, kernel.cu:728-740
    float bg[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        bg[k] = kSil ? 0.f : ((a.p.background_from_buffer && t.valid) ? a.rgba[((long)t.b * 4 + k) * P + t.pix] : a.p.background[k]);
    float alpha = 0.f;
    float ssum = a.softmax_sum0, smax = a.p.aggr_rgb_eps;
    float col[3];
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = rgb_soft ? bg[k] * ssum : bg[k];
    float depth_min = 10000000.f;
    int face_min = -1;

    const float* recs_g = a.records + (long)t.b * a.nf * REC;

    // the fold of one evaluated pair into its pixel's state (lane = pixel): kernel.cu:791-838
    auto fold = [&](const FwdRes& res) __attribute__((always_inline)) {
        const int fn = res.fn;
        if (!(res.flags & kFlagContrib)) return;
        // alpha, kernel.cu:791-803
        if (alpha_func == kAlphaHard) {
            if ((double)res.frag > 0.5) alpha = 1.f;
        } else if constexpr (ALPHA > 0) {
            alpha = TConorm<(ALPHA > 0 ? ALPHA : 1)>::fold(alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
        } else if constexpr (ALPHA == -2) {
            alpha = tconorm_fold_light_rt(alpha_func, alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
        } else {
            alpha = tconorm_fold_rt(alpha_func, alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
        }
        if constexpr (kSil) return;
        if (!(res.flags & kFlagRgb)) return;
        if (!rgb_soft) {                                                     // :815-822
            if (res.z < depth_min) {
                depth_min = res.z;
                face_min = fn;
                col[0] = res.c0; col[1] = res.c1; col[2] = res.c2;
            }
        } else {                                                             // :824-838
            const float zn = res.z;
            // exp_delta_zp and exp_z of :827-832: one of the two is exp(0) == 1 exactly
            const bool deeper = zn > smax;
            const float e = exp_f(div_by(deeper ? smax - zn : zn - smax, GENDR_R_GAMMA(a)));
            const float edz = deeper ? e : 1.f;
            const float ez = deeper ? 1.f : e;
            if (deeper) smax = zn;
            ssum = edz * ssum + ez * res.frag;
            col[0] = edz * col[0] + ez * res.frag * res.c0;
            col[1] = edz * col[1] + ez * res.frag * res.c1;
            col[2] = edz * col[2] + ez * res.frag * res.c2;
        }
    };

    auto run_batch = [&](int base, int np) __attribute__((always_inline)) {
#if GENDR_ABLATE == 1
        alpha += (float)np; return;
#endif
        // Which pairs of the batch are this pixel's?  Every pair lane sets its bit in its pixel's word: a 64-bit LDS OR
        // (the few pairs of one pixel -- one per face of the batch -- serialise on their word, different pixels do not).
        s_mask[wave][lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        int code = 0;
        if (lane < np) {
            code = s_code[wave][base + lane];
            __hip_atomic_fetch_or(&s_mask[wave][code & 63], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // ---- phase B: one pair per lane
        float hint = 0.f;                                       // what backward may know about the pair (PairHints); +0: no gradient
        if (lane < np) {
            const int fn = code >> 6;
            const long face_lin = (long)t.b * a.nf + fn;
            float r[REC];
            const float* rg = recs_g + (long)fn * REC;
            gather_record<kGatherW0, kGatherW1>(r, rg);
            gather_record<kGatherA0, kGatherA1>(r, rg);
            const float2 pc = s_xy[wave][code & 63];
            const float pxp = pc.x, pyp = pc.y;
            Pair q;
            barycentrics(q, r, pxp, pyp);
            FwdRes res;
            res.flags = 0; res.frag = 0.f; res.z = 0.f; res.c0 = res.c1 = res.c2 = 0.f; res.fn = fn; res.pad1 = 0;
#if GENDR_ABLATE == 6       // phase table (tools/fwd_phases.sh): the gathers and the barycentrics only
            res.frag = q.w0 + q.w1 + q.w2 + r[kRecEdge] + r[kRecXY + 5];
            if (false)
#elif GENDR_ABLATE == 7     // ... and the distance + CDF stage, nothing behind it
            if (soft_fragment<DIST, SQ>(q, r, pxp, pyp, a, dp)) { res.flags = kFlagContrib; res.frag = q.frag; hint = q.hint; }
            if (false)
#endif
            if (kSil) {
                // alpha needs the fragment only: the reference folds it before it looks at the depth (:795-810)
                if (soft_fragment<DIST, SQ>(q, r, pxp, pyp, a, dp)) { res.flags = kFlagContrib; res.frag = q.frag; hint = q.hint; }
            } else if (soft_fragment<DIST, SQ>(q, r, pxp, pyp, a, dp)) {
                hint = q.hint;                  // (backward repeats the depth test :810 / :994 itself: it needs the depth anyway)
                gather_record<kGatherB0, REC / 4>(r, rg);
                res.flags = kFlagContrib;
                res.frag = q.frag;
                float wc[3];
                const float zp = clip_and_depth(q, r, wc);
                if (!(zp < a.p.near_ || zp > a.p.far_)) {                         // :810
                    res.flags |= kFlagDepthOk;
                    const bool front = (__float_as_int(r[kRecBits]) & kBitFront) != 0;
                    const bool eligible = rgb_soft ? (front || a.p.double_side)                         // :825
                                                   : (inside_closed(q) && (a.p.double_side || front));  // :816
                    if (eligible) {
                        res.flags |= kFlagRgb;
                        res.z = rgb_soft ? div_by(a.p.far_ - zp, GENDR_R_ZRANGE(a)) : zp;   // zp_norm (:826) or zp
                        float cc[3]; int own;
                        sample_colour<TEXM>(cc, own, wc, r, a, face_lin);
                        res.c0 = cc[0]; res.c1 = cc[1]; res.c2 = cc[2];
                    }
                }
            }
            s_res[wave][lane] = res;
        }
        if (hint_slot) {
            // The batch's pair hints for backward: one compare per hint word, and the 16 bytes leave through the SCALAR
            // data cache (s_store_dwordx4: the ballots are scalar registers already; a vector store by one lane costs four moves,
            // the address and the exec juggling -- measured +1.3 us per launch at the headline scene).  Written back by the
            // s_dcache_wb at the end of the wave; partial lines shared with other waves are byte-masked (tools/micro/sstore.hip).
            typedef unsigned u4s __attribute__((ext_vector_type(4)));
            const unsigned long long h_lo = __ballot(hint_bit0(hint));
            const unsigned long long h_hi = __ballot(hint_bit1(hint));
            u4s hv; hv.x = (unsigned)h_lo; hv.y = (unsigned)(h_lo >> 32); hv.z = (unsigned)h_hi; hv.w = (unsigned)(h_hi >> 32);
            asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(hv), "s"(hint_slot) : "memory");
            hint_slot++;
            if (__ballot(hint_none(hint)) && lane == 0) atomicOr(a.control + (blockIdx.x & 7) * kCtlStride + kCtlHintFlag, 2);
        }
        __builtin_amdgcn_wave_barrier();
#if GENDR_ABLATE == 2 || GENDR_ABLATE == 6 || GENDR_ABLATE == 7
        alpha += s_res[wave][lane].frag; return;
#endif
        // ---- phase C: every pixel folds its own pairs; their list positions ascend with the face index
        for (unsigned long long todo = s_mask[wave][lane]; todo; todo &= todo - 1) fold(s_res[wave][__builtin_ctzll(todo)]);
        __builtin_amdgcn_wave_barrier();
    };

    // A dense entry (see for_each_batch): face `fn` reaches all 64 pixels of the tile.  lane = pixel, the record in scalar
    // registers, the same pair functions on the same operands as phase B, and the fold in place.
    auto run_dense = [&](int fn, unsigned long long mask, int tag) __attribute__((always_inline)) {
        if constexpr (dense_path<DIST>()) {
            const long face_lin = (long)t.b * a.nf + fn;
            const bool mine = lane_in(mask) && t.valid;
            // the record arrives by scalar loads, stage by stage: a stage's floats are requested when the previous stage has been
            // consumed, so that only one stage occupies scalar registers at a time (the kernel has few to spare: the whole record
            // at once was spilled lane by lane)
            float r[REC];
            RecPtr rs = uniform_rec_ptr(recs_g + (long)fn * REC);
            load_record<4 * kGatherW0, 4 * kGatherW1>(r, rs);
            Pair q;
            barycentrics(q, r, t.xp, t.yp);
            asm volatile("" : "+s"(rs) : "v"(q.w0), "v"(q.w1), "v"(q.w2));
            load_record<4 * kGatherA0, 4 * kGatherA1>(r, rs);
            FwdRes res;
            res.flags = 0; res.frag = 0.f; res.z = 0.f; res.c0 = res.c1 = res.c2 = 0.f; res.fn = fn; res.pad1 = 0;
            bool contributes = false;
            q.frag = 0.f;
            if (mine) contributes = soft_fragment<DIST, SQ>(q, r, t.xp, t.yp, a, dp, false, tag == 1, tag == 2, tag != 0);
            if constexpr (!kSil) {
                asm volatile("" : "+s"(rs) : "v"(q.frag));
                load_record<4 * kGatherB0, REC>(r, rs);
            }
            if (contributes) {
                res.flags = kFlagContrib;
                res.frag = q.frag;
                if constexpr (!kSil) {
                    float wc[3];
                    const float zp = clip_and_depth(q, r, wc);
                    if (!(zp < a.p.near_ || zp > a.p.far_)) {                         // :810
                        res.flags |= kFlagDepthOk;
                        const bool front = (__float_as_int(r[kRecBits]) & kBitFront) != 0;
                        const bool eligible = rgb_soft ? (front || a.p.double_side)                         // :825
                                                       : (inside_closed(q) && (a.p.double_side || front));  // :816
                        if (eligible) {
                            res.flags |= kFlagRgb;
                            res.z = rgb_soft ? div_by(a.p.far_ - zp, GENDR_R_ZRANGE(a)) : zp;   // zp_norm (:826) or zp
                            float cc[3]; int own;
                            sample_colour<TEXM>(cc, own, wc, r, a, face_lin);
                            res.c0 = cc[0]; res.c1 = cc[1]; res.c2 = cc[2];
                        }
                    }
                }
            }
            fold(res);
        }
    };

    for_each_batch<REC, dense_path<DIST>()>(a, t, ti.y, ti.z, my_rows, pixel_mode, s_code[wave], run_batch, run_dense);

    if constexpr (kSil) {
        // alpha plane, and the tile's share of the fused IoU sums (opt_shape.py:20-24: intersect = sum(a t),
        // union = sum(a + t - a t) = sum(t) + sum(a (1 - t)); unlisted tiles have a = 0 and add nothing)
        if (t.valid) a.rgba[(long)t.b * P + t.pix] = alpha;
        if (a.target) {
            const float tv = t.valid ? a.target[(long)t.b * P + t.pix] : 0.f;
            float s1 = t.valid ? alpha * tv : 0.f, s2 = t.valid ? alpha * (1.f - tv) : 0.f;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
            if (lane == 0) {
                if (s1 != 0.f) unsafeAtomicAdd(a.iou_sums + 2 * t.b, s1);
                if (s2 != 0.f) unsafeAtomicAdd(a.iou_sums + 2 * t.b + 1, s2);
            }
        }
    } else if (t.valid) {
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
            for (int k = 0; k < 3; k++) out[k * P] = div_f(col[k], ssum);
            aux[0] = ssum;
            aux[P] = smax;
        }
    }
    __builtin_amdgcn_wave_barrier();
    }   // tile loop
    if (hints_q) asm volatile("s_waitcnt lgkmcnt(0)\n s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");   // the hints' scalar stores reach the L2
    GENDR_SPAN_END(2, blockIdx.x * WAVES + wave);
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) void render_forward_kernel(const RenderArgs a)
{
    render_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

// Same body, register budget capped for GENDR_FWD_WAVES (7) waves per SIMD (72 VGPRs): used for the specialised option
// sets, whose natural allocation sits a few registers above an occupancy step; the handful of spilled dwords costs
// less than the waves (the _w6 / _w5 suffixes date from the first budgets tried).
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_FWD_WAVES))) void render_forward_kernel_w6(const RenderArgs a)
{
    render_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_LIGHT_FWD_WAVES))) void render_forward_kernel_wl(const RenderArgs a)
{
    render_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_FULL_WAVES))) void render_forward_kernel_wf(const RenderArgs a)
{
    render_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_HALPHA_WAVES))) void render_forward_kernel_wa(const RenderArgs a)
{
    render_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

// ---------------------------------------------------------------------------------------------
// backward of one (pixel, face) pair: kernel.cu:965-1052.  Recomputes the pair exactly as the forward kernel does (whatever
// decides which pairs contribute) and forms the partials of the 9 vertex components (gv) and of the face's texture
// (gt: 3 texel components for surface T = 1, 9 vertex-colour components; surface T > 1: the one texel `tex_own` of the
// face's block that receives tex_val[3], or -1).  Returns false when the pair contributes nothing (gv / gt are then
// undefined).  Shared by the tile kernel (lane = pair of a batch) and the per-face deterministic kernel (lane = pixel).
// ---------------------------------------------------------------------------------------------
template <int TEXM> struct GradSlots { static constexpr int n = TEXM == kTexSurface1 ? 12 : (TEXM == kTexVertex ? 18 : 9); };
template <int TEXM> struct GradTex { static constexpr int n = GradSlots<TEXM>::n > 9 ? GradSlots<TEXM>::n - 9 : 1; };

struct PixIn {             // 48 bytes: per-pixel inputs of the backward pass, kernel.cu:916-917, :973, :980, :1013, :1021
    float g[4], out[4], ssum, smax, xp, yp;
};

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM, int PRELOADED = 0>
__device__ __forceinline__ bool backward_pair(const RenderArgs& a, const DistParams& dp, const float* rg, const PixIn& px, int fn, long face_lin,
                                              float (&gv)[9], float (&gt)[GradTex<TEXM>::n], int& tex_own, float (&tex_val)[3],
                                              bool hinted = false, bool e0 = false, bool e1 = false, bool mine = true, bool tagged = false)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NT = GradTex<TEXM>::n;
    constexpr bool kSil = RGB == kRgbNone;
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const int dist = DIST >= 0 ? DIST : a.p.dist_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);
    const bool squared = SQ >= 0 ? (SQ != 0) : (a.p.dist_squared != 0);
    // PRELOADED 1: rg is the caller's register copy of the whole record (one face per wavefront: loaded once, not per pair);
    // PRELOADED 2: rg points at the record of a wave-uniform face (a dense entry, see for_each_batch): scalar loads, stage by
    // stage -- a stage's floats are requested when the previous stage has been consumed, so that only one stage occupies
    // scalar registers at a time (the kernels have few to spare: the whole record at once was spilled lane by lane)
    float r[REC];
    RecPtr rs = nullptr;
    if constexpr (PRELOADED == 2) rs = uniform_rec_ptr(rg);
    if constexpr (PRELOADED == 1) {
#pragma unroll
        for (int k = 0; k < REC; k++) r[k] = rg[k];
    } else if constexpr (PRELOADED == 2) {
        load_record<4 * kGatherW0, 4 * kGatherW1>(r, rs);
    } else {
        gather_record<kGatherW0, kGatherW1>(r, rg);
        gather_record<kGatherA0, kGatherA1>(r, rg);
    }
    const float pxp = px.xp, pyp = px.yp;
    Pair q;
    barycentrics(q, r, pxp, pyp);
    if constexpr (PRELOADED == 2) {
        asm volatile("" : "+s"(rs) : "v"(q.w0), "v"(q.w1), "v"(q.w2));
        load_record<4 * kGatherA0, 4 * kGatherA1>(r, rs);
    }
    // Two stages one after the other, not nested: `live` is narrowed by the first and guards the second, and the
    // partials are defined in the second only (as values that survive a nest of early exits they were
    // re-initialised at every level of it: 58 moves per batch; the empty asm below keeps the compiler from turning
    // the final select back into such a default).
    float C_xy = 0.f, zp = 0.f;
    float wc[3];
    bool live = false;
    q.frag = 0.f;
    if (mine) live = soft_fragment<DIST, SQ>(q, r, pxp, pyp, a, dp, hinted, e0, e1, tagged);     // (`mine`: the lane holds a pair at all -- dense entries under a pixel mask)
    if constexpr (PRELOADED == 2) {
        asm volatile("" : "+s"(rs) : "v"(q.frag));
        load_record<4 * kGatherB0, REC>(r, rs);
    }
    if (live) {
        // alpha only, and this face's depth cannot fail the near / far test (see face_setup_kernel): no depth stage
        const bool need_depth = !(kSil && (__float_as_int(r[kRecBits]) & kBitDepthSafe));
        if constexpr (PRELOADED == 0) { if (need_depth) gather_record<kGatherB0, REC / 4>(r, rg); }
        // alpha partial, kernel.cu:973-987 (hard alpha leaves g[3] unscaled, as the reference does)
        float C_alpha = px.g[3];
        if (alpha_func != kAlphaHard) {
            if constexpr ((ALPHA == kProbabilistic || ALPHA == kEinstein) && !GENDR_EXACT_GRADIENT)
                                            C_alpha *= TConorm<(ALPHA > 0 ? ALPHA : 1)>::grad_fp32(px.out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
            else if constexpr (ALPHA > 0)   C_alpha *= TConorm<(ALPHA > 0 ? ALPHA : 1)>::grad(px.out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
            else if constexpr (ALPHA == -2) C_alpha *= tconorm_grad_light_rt(alpha_func, px.out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
            else                            C_alpha *= tconorm_grad_rt(alpha_func, px.out[3], q.frag, a.p.aggr_alpha_t_conorm_p);
        }
        C_xy += C_alpha;

        if (need_depth) {
            zp = clip_and_depth(q, r, wc);
            live = !(zp < a.p.near_ || zp > a.p.far_);                  // :994 drops the whole pair
        }
    }
    {
        if (live) {
#pragma unroll
            for (int k = 0; k < 9; k++) gv[k] = 0.f;
#pragma unroll
            for (int k = 0; k < NT; k++) gt[k] = 0.f;
            const bool front = (__float_as_int(r[kRecBits]) & kBitFront) != 0;
            if constexpr (kSil) {
                // no colour term: C_xy stays the alpha partial
            } else if (!rgb_soft) {                                     // :997-1004
                if ((float)fn == px.smax) {
                    if constexpr (TEXM == kTexVertex) {
#pragma unroll
                        for (int k = 0; k < 3; k++)
#pragma unroll
                            for (int j = 0; j < 3; j++) gt[3 * j + k] = wc[j] * px.g[k];
                    } else {
                        float cc[3]; int own;
                        sample_colour<TEXM>(cc, own, wc, r, a, face_lin);
                        if (own >= 0) {
                            if constexpr (TEXM == kTexSurface1) {
#pragma unroll
                                for (int k = 0; k < 3; k++) gt[k] = px.g[k];
                            } else {
#pragma unroll
                                for (int k = 0; k < 3; k++)
                                    unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + own) * 3 + k, px.g[k]);
                            }
                        }
                    }
                }
            } else if (front || a.p.double_side) {                      // :1006-1030
                const float zn = div_by(a.p.far_ - zp, GENDR_R_ZRANGE(a));
                const float zs = grad_div(q.frag * exp_f(div_by(zn - px.smax, GENDR_R_GAMMA(a))), px.ssum);   // :1010
                float cc[3]; int own;
                sample_colour<TEXM>(cc, own, wc, r, a, face_lin);
                float C_rgb = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if constexpr (TEXM == kTexVertex) {
#pragma unroll
                        for (int j = 0; j < 3; j++) gt[3 * j + k] = zs * (wc[j] * px.g[k]);
                    } else if constexpr (TEXM == kTexSurface1) {
                        if (own >= 0) gt[k] = zs * px.g[k];
                    } else {
                        if (own >= 0) { tex_own = own; tex_val[k] = zs * px.g[k]; }
                    }
                    C_rgb += px.g[k] * (cc[k] - px.out[k]);             // :1021
                }
                C_rgb *= zs;                                            // :1023
                C_xy += grad_div(C_rgb, q.frag);                        // :1024
#if GENDR_EXACT_GRADIENT
                const float C_z = div_by(div_by(C_rgb, GENDR_R_GAMMA(a)), GENDR_R_NZRANGE(a)) * zp * zp;   // :1026
                gv[2] = div_by(div_by(C_z * wc[0], rec_rcp(r, kRecRZ + 0)), rec_rcp(r, kRecRZ + 0));
                gv[5] = div_by(div_by(C_z * wc[1], rec_rcp(r, kRecRZ + 2)), rec_rcp(r, kRecRZ + 2));
                gv[8] = div_by(div_by(C_z * wc[2], rec_rcp(r, kRecRZ + 4)), rec_rcp(r, kRecRZ + 4));
#else
                // C_rgb / gamma / (near - far) * zp^2 * w_k / z_k^2 with the float reciprocals (:1026-1029)
                const float C_z = C_rgb * ((float)a.r_gamma * (float)a.r_nzrange) * zp * zp;
                const float rz0 = (float)rec_rcp(r, kRecRZ + 0), rz1 = (float)rec_rcp(r, kRecRZ + 2), rz2 = (float)rec_rcp(r, kRecRZ + 4);
                gv[2] = C_z * wc[0] * (rz0 * rz0);
                gv[5] = C_z * wc[1] * (rz1 * rz1);
                gv[8] = C_z * wc[2] * (rz2 * rz2);
#endif
            }

            // distance gradient, kernel.cu:1034-1052.  Heaviside: D' = 0 times uninitialised
            // values in the reference -> defined as exactly 0 here (DESIGN.md quirk i).
            if (dist != kHeaviside) {
                if constexpr (DIST == kLogistic) C_xy *= div_by(q.frag * (1 - q.frag), dp.rscale);   // :378-380: y is the CDF just computed
                else if constexpr (DIST >= 0)  C_xy *= Dist<(DIST >= 0 ? DIST : 0)>::pdf(q.sign, q.dis, dp);
                else if constexpr (DIST == -2) C_xy *= pdf_light_rt(dist, q.sign, q.dis, dp);
                else                           C_xy *= pdf_rt(dist, q.sign, q.dis, dp);
                const float tw[3] = {q.t0 + q.w0, q.t1 + q.w1, q.t2 + q.w2};
                if (squared) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        gv[3 * k + 0] = 2 * q.sign * C_xy * tw[k] * q.dx;
                        gv[3 * k + 1] = 2 * q.sign * C_xy * tw[k] * q.dy;
                    }
                } else {
                    // (double)num / max(sqrt(dx^2+dy^2), 1e-6), rounded to float (:1049).  When the divisor is
                    // the float square root, one double reciprocal serves all six quotients exactly (div_by);
                    // below 1e-6 the divisor is the double literal and the true divisions are kept.
                    const float len = q.dis;   // == sqrtf(dx*dx + dy*dy): the very value soft_fragment() computed (:771)
#if GENDR_EXACT_GRADIENT
                    if ((double)len >= 1e-6) {
                        const rcp_t rlen = rcp_for_div_by(len);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            gv[3 * k + 0] = div_by(q.sign * C_xy * tw[k] * q.dx, rlen);
                            gv[3 * k + 1] = div_by(q.sign * C_xy * tw[k] * q.dy, rlen);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            gv[3 * k + 0] = (float)((double)(q.sign * C_xy * tw[k] * q.dx) / 1e-6);
                            gv[3 * k + 1] = (float)((double)(q.sign * C_xy * tw[k] * q.dy) / 1e-6);
                        }
                    }
#else
                    const float rlen = (len > (float)1e-6) ? grad_rcp(len) : 1e6f;      // len >= 1e-6 (double) iff len > (float)1e-6 < 1e-6
                    const float sx = q.sign * C_xy * (q.dx * rlen), sy = q.sign * C_xy * (q.dy * rlen);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        gv[3 * k + 0] = sx * tw[k];
                        gv[3 * k + 1] = sy * tw[k];
                    }
#endif
                }
            }
        }
    }
    return live;
}

// the per-pixel inputs of the backward pass (zeros for a lane without a pixel)
template <int RGB>
__device__ __forceinline__ PixIn load_pixel_inputs(const RenderArgs& a, int b, long pix, bool valid, float xp, float yp)
{
    constexpr bool kSil = RGB == kRgbNone;       // alpha-only: `rgba` / `grad_rgba` are single planes, see RenderArgs
    const long P = (long)a.is * a.is;
    PixIn pi;
#pragma unroll
    for (int k = 0; k < 4; k++) { pi.g[k] = 0.f; pi.out[k] = 0.f; }
    pi.ssum = 1.f; pi.smax = 0.f; pi.xp = xp; pi.yp = yp;
    if (kSil) {
        if (valid) {
            pi.out[3] = a.rgba[(long)b * P + pix];
            if (a.grad_iou) {
                // d loss / d alpha of the fused IoU sums: g1 * t + g2 * (1 - t)
                const float tv = a.target[(long)b * P + pix];
                pi.g[3] = a.grad_iou[2 * b] * tv + a.grad_iou[2 * b + 1] * (1.f - tv);
            } else {
                pi.g[3] = a.grad_rgba[(long)b * P + pix];
            }
        }
    } else if (valid) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pi.g[k] = a.grad_rgba[((long)b * 4 + k) * P + pix];
            pi.out[k] = a.rgba[((long)b * 4 + k) * P + pix];
        }
        pi.ssum = a.aux[((long)b * 2 + 0) * P + pix];
        pi.smax = a.aux[((long)b * 2 + 1) * P + pix];
    }
    return pi;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void render_backward_body(const RenderArgs& a)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NG = GradSlots<TEXM>::n;       // 9 vertex components, then texture components summed per face in LDS
    constexpr int NT = NG > 9 ? NG - 9 : 1;
    constexpr int WAVES = kThreads / 64;
    __shared__ int s_code[WAVES][kCodeCap];      // the tile's pair list (face << 6 | pixel), see for_each_batch
    __shared__ __attribute__((aligned(16))) PixIn   s_pix[WAVES][64];
    __shared__ __attribute__((aligned(8))) FaceSeg s_seg[WAVES][64];   // the faces of the running batch: (face, first pair, pairs)
    __shared__ float s_val[WAVES][NG * 65];      // per-pair gradient partials, component-major, rows padded to 65

    const int wave = threadIdx.x >> 6;
    const long P = (long)a.is * a.is;
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const int dist = DIST >= 0 ? DIST : a.p.dist_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);
    const bool squared = SQ >= 0 ? (SQ != 0) : (a.p.dist_squared != 0);
    constexpr bool kSil = RGB == kRgbNone;       // alpha-only: `rgba` / `grad_rgba` are single planes, see RenderArgs
#if GENDR_TIMERS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif

#if GENDR_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tr[0] = __builtin_readcyclecounter();
    tr[6] = __builtin_amdgcn_s_memrealtime();
#endif
    TileWalk tw;
    walk_init(tw, a, WAVES);
    GENDR_T(0);                                   // 0: wave start-up (queue lengths)
    GENDR_STAMP(1);
    // the forward kernel's pair hints hold for the tiles of this queue that are rendered unsplit (the same tiles in both kernels:
    // the grades come from order_tiles_kernel) unless it met a pair a hint cannot describe: batch k of such a tile is the same
    // 64 pairs in both
    const bool hinted_q = a.hints != nullptr && tw.hint_flag == 1;
    for (; tw.next < tw.items; tw.next += tw.stride) {
    GENDR_RELOADED_ARGS(a);
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int split_log2, sub;
    const int slot_i = walk_item(tw, tw.next, split_log2, sub);
    const i4v ti = *(const GENDR_CONST_AS i4v*)(a.tile_info + (tw.qbase + slot_i));   // (tile, first entry, entries, pairs): scalar load
    if (ti.y >= 0 && ti.z == 0) continue;                       // an empty coverage list (no heavy-first copy: such tiles are not sorted out)
    const unsigned long long my_rows = sub_tile_mask(split_log2, sub);
    const bool pixel_mode = dense_path<DIST>() && tile_in_pixel_mode(ti, split_log2);
    const PairHints* hint_slot = (hinted_q && split_log2 == 0 && ti.y >= 0 && !pixel_mode) ? a.hints + ti.y : nullptr;   // next batch's hints
    if (hint_slot && !dense_path<DIST>() && ti.w <= 4 * 64) {
        // a tile whose few batches hold no pair with a gradient is done before its pixel inputs are fetched (an image with a
        // face seen edge-on lists that face -- no error bound, 64 dead pairs -- in every one of its tiles).  Not in the kernels
        // with the dense path: a dense entry of the tile runs outside the batches, so the batches' hints say nothing about its
        // pairs and the tile's pair count says nothing about the number of batches (found by tools/fuzz_parity.py in round 4: a
        // tile whose only live pairs belonged to a dense entry was skipped -- the texture gradient of an image-filling face came
        // out 40 % short)
        unsigned long long all_dead = ~0ull;
        for (int k = 0; k < ((ti.w + 63) >> 6); k++) {
            const GENDR_CONST_AS unsigned long long* hp = (const GENDR_CONST_AS unsigned long long*)(hint_slot + k);
            all_dead &= hp[0] & hp[1];
        }
        if (all_dead == ~0ull) continue;
    }
    TileCtx t;
    tile_setup(t, a, ti.x);
    t.valid = t.valid && lane_in(my_rows);        // the pixels whose pairs this wave differentiates
    s_pix[wave][lane] = load_pixel_inputs<RGB>(a, t.b, t.pix, t.valid, t.xp, t.yp);
    GENDR_T(1);                                   // 1: tile record + the pixel's inputs parked in LDS
#if GENDR_TRACE
    if (!tr[2]) GENDR_STAMP(2);
#endif

    const float* recs_g = a.records + (long)t.b * a.nf * REC;

    auto run_batch = [&](int base, int np) __attribute__((always_inline)) {
        GENDR_T(2);                               // 2: entry list + code list since the last batch
#if GENDR_TRACE
        if (!tr[3]) GENDR_STAMP(3);
        tr[5] += 1;
#endif
#if GENDR_ABLATE == 3
        return;
#endif
        // the forward kernel's hints for the batch's pairs (PairHints): which edge, or nothing to do at all
        bool e0 = false, e1 = false, dead = false;
        const bool hinted = hint_slot != nullptr;
        if (hinted) {
            const GENDR_CONST_AS unsigned long long* hp = (const GENDR_CONST_AS unsigned long long*)hint_slot;   // scalar load
            const unsigned long long h_lo = hp[0], h_hi = hp[1];
            hint_slot++;
            // no pair of the batch gets a gradient (lanes past the batch's end read as dead too): nothing to do -- the tiles a
            // face without an error bound is listed in (seen edge-on: every tile of its image) consist of such batches
            if ((h_lo & h_hi) == ~0ull) return;
            e0 = __builtin_amdgcn_inverse_ballot_w64(~h_lo & ~h_hi);
            e1 = __builtin_amdgcn_inverse_ballot_w64(h_lo & ~h_hi);
            dead = __builtin_amdgcn_inverse_ballot_w64(h_lo & h_hi);
        }
        // The batch's codes, and its faces: the pairs of one face are consecutive, so a lane whose face differs from
        // its left neighbour's heads a segment; the ballot of the heads gives every head its slot, first pair and
        // length (a dozen vector instructions per batch for what the entry-by-entry batch builder used to keep).
        const int code = lane < np ? s_code[wave][base + lane] : -64;
        const int fn_l = code >> 6;
        const int fn_left = __builtin_amdgcn_update_dpp(-2, fn_l, 0x138, 0xF, 0xF, false);     // wave_shr:1, lane 0 keeps -2
        const unsigned long long heads = __ballot(lane < np && fn_l != fn_left);
        const int nfaces = __popcll(heads);
        if (lane_in(heads)) {
            const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;
            FaceSeg sg;
            sg.fn = fn_l; sg.span = lane | ((above ? __builtin_ctzll(above) + 1 : np - lane) << 8);
            s_seg[wave][bits_below(heads)] = sg;
        }
        if (lane < np) {
            const PixIn px = s_pix[wave][code & 63];
            const int fn = fn_l;
            const long face_lin = (long)t.b * a.nf + fn;
            float gv[9];                       // d loss / d (x,y,z) of the 3 vertices, kernel.cu:967
            float gt[NT];                      // texture partials
            int tex_own = -1;
            float tex_val[3] = {0.f, 0.f, 0.f};
            bool live = false;
            if (!dead) live = backward_pair<DIST, ALPHA, RGB, SQ, TEXM>(a, dp, recs_g + (long)fn * REC, px, fn, face_lin, gv, gt, tex_own, tex_val, hinted, e0, e1);
            if constexpr (TEXM == kTexSurfaceN) {
                if (live && tex_own >= 0) {
#pragma unroll
                    for (int k = 0; k < 3; k++) unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + tex_own) * 3 + k, tex_val[k]);
                }
            }
            GENDR_T(4);                           // 4: the pair math (incl. the second gather)
            // every pair lane publishes its partials (zeros if the pair dropped out): column = pair index
#pragma unroll
            for (int k = 0; k < 9; k++) asm("" : "+v"(gv[k]));
#pragma unroll
            for (int k = 0; k < NG - 9; k++) asm("" : "+v"(gt[k]));
#pragma unroll
            for (int k = 0; k < 9; k++) s_val[wave][k * 65 + lane] = live ? gv[k] : 0.f;
#pragma unroll
            for (int k = 0; k < NG - 9; k++) s_val[wave][(9 + k) * 65 + lane] = live ? gt[k] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        GENDR_T(5);                               // 5: partials to LDS
#if GENDR_ABLATE == 4
        return;
#endif
        // The pairs of one face are contiguous.  One lane per (face, component) sums its segment in pair order and
        // issues one hardware fp32 atomic: deterministic inside the batch, no LDS atomics.
        for (int e = lane; e < nfaces * NG; e += 64) {
            const int slot = e / NG, k = e - slot * NG;
            const FaceSeg sg = s_seg[wave][slot];
            const int cnt = sg.span >> 8;
            const float* col = &s_val[wave][k * 65 + (sg.span & 255)];
            // four independent partial sums so that the LDS reads of a segment overlap instead of chaining
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) { v0 += col[i]; v1 += col[i + 1]; v2 += col[i + 2]; v3 += col[i + 3]; }
            for (; i < cnt; i++) v0 += col[i];
            const float v = (v0 + v1) + (v2 + v3);
#if GENDR_ABLATE == 8
            if (v == 12345.678f) {
#else
            if (v != 0.f) {
#endif
                const long face_lin = (long)t.b * a.nf + sg.fn;
                if (k < 9) unsafeAtomicAdd(a.grad_faces + face_lin * 9 + k, v);
                else       unsafeAtomicAdd(a.grad_textures + face_lin * (NG - 9) + (k - 9), v);
            }
        }
        __builtin_amdgcn_wave_barrier();
        GENDR_T(6);                               // 6: segment sums + atomics issued
    };

    // A dense entry (see for_each_batch): face `fn` reaches all 64 pixels of the tile.  lane = pixel, the record in scalar
    // registers, backward_pair() as everywhere; the partials of the ONE face are summed by four lanes per component (sixteen
    // consecutive pairs each, combined inside the quad) and leave as one atomic per component.
    auto run_dense = [&](int fn, unsigned long long mask, int tag) __attribute__((always_inline)) {
        if constexpr (dense_path<DIST>()) {
            const long face_lin = (long)t.b * a.nf + fn;
            const bool mine = lane_in(mask) && t.valid;
            const PixIn px = s_pix[wave][lane];
            float gv[9];
            float gt[NT];
            int tex_own = -1;
            float tex_val[3] = {0.f, 0.f, 0.f};
            const bool live = backward_pair<DIST, ALPHA, RGB, SQ, TEXM, 2>(a, dp, recs_g + (long)fn * REC, px, fn, face_lin, gv, gt, tex_own, tex_val, false, tag == 1, tag == 2, mine, tag != 0);
            if constexpr (TEXM == kTexSurfaceN) {
                if (live && tex_own >= 0) {
#pragma unroll
                    for (int k = 0; k < 3; k++) unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + tex_own) * 3 + k, tex_val[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 9; k++) asm("" : "+v"(gv[k]));
#pragma unroll
            for (int k = 0; k < NG - 9; k++) asm("" : "+v"(gt[k]));
#pragma unroll
            for (int k = 0; k < 9; k++) s_val[wave][k * 65 + lane] = live ? gv[k] : 0.f;
#pragma unroll
            for (int k = 0; k < NG - 9; k++) s_val[wave][(9 + k) * 65 + lane] = live ? gt[k] : 0.f;
            __builtin_amdgcn_wave_barrier();
            constexpr int PER = NG <= 16 ? 4 : 2;                // lanes per component (NG = 9, 12: 4; 18: 2)
            constexpr int LEN = 64 / PER;
            const int k = lane / PER, seg = lane % PER;
            float v = 0.f;
            if (k < NG) {
                const float* col = &s_val[wave][k * 65 + seg * LEN];
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
                for (int i = 0; i < LEN; i += 4) { v0 += col[i]; v1 += col[i + 1]; v2 += col[i + 2]; v3 += col[i + 3]; }
                v = (v0 + v1) + (v2 + v3);
            }
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
            if (PER == 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            if (k < NG && seg == 0 && v != 0.f) {
                if (k < 9) unsafeAtomicAdd(a.grad_faces + face_lin * 9 + k, v);
                else       unsafeAtomicAdd(a.grad_textures + face_lin * (NG - 9) + (k - 9), v);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    for_each_batch<REC, dense_path<DIST>()>(a, t, ti.y, ti.z, my_rows, pixel_mode, s_code[wave], run_batch, run_dense);
    __builtin_amdgcn_wave_barrier();
    GENDR_T(7);                                   // 7: tail of the tile (entry walk after the last batch)
    }   // tile loop
#if GENDR_TIMERS
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 8; i++) atomicAdd(reinterpret_cast<unsigned long long*>(a.control + 16 * kCtlStride + 64) + i, tacc[i]);
#endif
#if GENDR_TRACE
    GENDR_STAMP(4);
    tr[7] = __builtin_amdgcn_s_memrealtime();
    {
        const unsigned w = (unsigned)blockIdx.x * WAVES + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && w < (1u << 17))
            for (int i = 0; i < 8; i++) g_wave_trace[w][i] = tr[i];
    }
#endif
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) void render_backward_kernel(const RenderArgs a)
{
    render_backward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

// register budget capped for 5 waves per SIMD (96 VGPRs), see render_forward_kernel_w6
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_BWD_WAVES))) void render_backward_kernel_w5(const RenderArgs a)
{
    render_backward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_LIGHT_BWD_WAVES))) void render_backward_kernel_wl(const RenderArgs a)
{
    render_backward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_FULL_WAVES))) void render_backward_kernel_wf(const RenderArgs a)
{
    render_backward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GENDR_HALPHA_WAVES))) void render_backward_kernel_wa(const RenderArgs a)
{
    render_backward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

// ---------------------------------------------------------------------------------------------
// deterministic backward (gendr_params::deterministic): one wavefront per (image, face), no atomics
// ---------------------------------------------------------------------------------------------
// The tile kernel adds a face's partial sums with hardware fp32 atomics: one per (tile batch, face, component), in the order
// the tiles happen to be dispatched -- like the reference's atomicAdd per (pixel, face) (kernel.cu:1054-1063), the result
// differs in the last bits from run to run (experiments/train_reconstruction.py:582-586 warns about it).  This kernel
// gathers instead: the wavefront of a face walks the pixels of the face's cull box in raster order, 64 at a time
// (lane = pixel), applies the exact per-pixel tests and the same backward_pair() as the tile kernel, keeps one running sum
// per lane and component (a pixel of the box always lands on the same lane, in the same order), and combines the 64 lane sums
// in a fixed butterfly at the end.  Every gradient element has exactly one writer and one summation order: two calls on the
// same inputs return bit-identical gradients, whatever the batch, the dispatch order or the build's tile size.
// It needs nothing from the tile queues (only the face records and cull boxes), and costs what a gather costs: the lanes
// of a 64-pixel step that fall outside the triangle idle (measured at C2: see DESIGN.md).
constexpr int kDetBigSteps = 64;      // a face whose cull box takes more 64-pixel steps than this goes to the band kernel
constexpr int kDetBandRows = 8;       // image rows per work item of the band kernel
constexpr int kDetBigCap = 2048;      // deferred faces the scratch region of the workspace holds
constexpr int kDetSlots = 18;         // floats per partial row (the largest GradSlots)
#ifndef GENDR_DET_THREADS
#define GENDR_DET_THREADS 256
#endif
constexpr int kDetThreads = GENDR_DET_THREADS;     // wavefronts (= faces) per workgroup x 64

struct DetBox { int x0, W, yi0, yi1; bool empty; };

// the pixels whose centres can lie inside the face's cull box: index of the first / last pixel centre inside [lo, hi], in
// double with a thousandth of a pixel of slack (a centre is correctly rounded from an integer: off by < 1e-7 of the image);
// pixel centre of index i: (2 i + 1 - is) / is
__device__ __forceinline__ DetBox det_box(const RenderArgs& a, const float* __restrict__ boxes, long face_lin)
{
    const float4 box = reinterpret_cast<const float4*>(boxes)[face_lin];
    const double is_d = (double)a.is;
    auto first_index = [&](float v) { const double f = ceil(((double)v * is_d + is_d - 1.) * 0.5 - 1e-3); return f != f ? 0 : (int)fmin(fmax(f, 0.), is_d - 1.); };
    auto last_index = [&](float v) { const double f = floor(((double)v * is_d + is_d - 1.) * 0.5 + 1e-3); return f != f ? a.is - 1 : (int)fmin(fmax(f, 0.), is_d - 1.); };
    DetBox d;
    d.empty = box.x > box.y || box.z > box.w;                            // (NaN boxes fall through: whole image)
    d.x0 = first_index(box.x);
    d.W = last_index(box.y) - d.x0 + 1;
    d.yi0 = first_index(box.z);
    d.yi1 = last_index(box.w);
    if (d.W <= 0 || d.yi1 < d.yi0) d.empty = true;
    return d;
}

// Rows [ya, yb] (indices from the bottom of the image, kernel.cu:716) of the face's box: per lane and component the sum of
// the partials of the lane's pixels, in raster order.  lane -> pixel: rows of up to 64 pixels; a narrower box puts 64 / W
// rows into one step (one division per wavefront); a pixel always lands on the same lane and step for given (W, ya).
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void det_rows(const RenderArgs& a, const DistParams& dp, const float (&rec)[record_floats(TEXM)], int b, int fn, long face_lin,
                                         int x0, int W, int ya, int yb, float (&acc)[GradSlots<TEXM>::n])
{
    constexpr int NG = GradSlots<TEXM>::n, NT = GradTex<TEXM>::n;
    const int lane = threadIdx.x & 63;
    const double is_d = (double)a.is;
    const int wp = min(W, 64), rows_per_step = 64 / wp;
    const int ry_l = lane / wp, rx_l = lane - ry_l * wp;
    const bool lane_used = ry_l < rows_per_step;
    for (int yr = ya; yr <= yb; yr += rows_per_step) {
    // A wide box (one row per step) of a thin face is mostly empty: the three edge tests are linear in x, so the stretch of
    // the row that can pass them is an interval -- computed in double with a pixel of slack on either side (every pixel
    // still takes the exact tests below).
    int rx_first = 0, rx_last = W - 1;
    if (wp == 64) {
        const double ypd = (double)pixel_coord(yr, a.is, a.r_is);
        double xa = -1e30, xb = 1e30;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double ea = rec[kRecInv + 3 * k], eb = rec[kRecInv + 3 * k + 1], ec = rec[kRecInv + 3 * k + 2], et = rec[kRecWCull + k];
            if (et > -1e30 && ea == ea && fabs(ea) < 1e30 && ea != 0.) {
                const double rhs = (et - 1e-5 * (fabs(ea) + fabs(eb) + fabs(ec) + fabs(et))) - (eb * ypd + ec);
                const double x = rhs / ea;
                if (x == x) { if (ea > 0.) xa = fmax(xa, x); else xb = fmin(xb, x); }
            }
        }
        const double ia = floor((xa * is_d + is_d - 1.) * 0.5) - 1. - x0, ib = ceil((xb * is_d + is_d - 1.) * 0.5) + 1. - x0;
        rx_first = __builtin_amdgcn_readfirstlane((int)fmin(fmax(ia, 0.), (double)W));
        rx_last = __builtin_amdgcn_readfirstlane((int)fmax(fmin(ib, (double)(W - 1)), -1.));
    }
    for (int rx0 = rx_first; rx0 <= rx_last; rx0 += wp) {
        const int rx = rx0 + rx_l, yi = yr + ry_l;
        const int xi = x0 + rx;
        const float xp = pixel_coord(xi, a.is, a.r_is), yp = pixel_coord(yi, a.is, a.r_is);
        bool live = lane_used && rx < W && yi <= yb && inside_box(rec, xp, yp);
        Pair q0;
        barycentrics(q0, rec, xp, yp);
        live = live && !beyond_an_edge(q0, rec);
        if (!__any(live)) continue;
        const long pix = (long)(a.is - 1 - yi) * a.is + xi;              // row 0 = top: yi = is - 1 - row (kernel.cu:716)
        const PixIn px = load_pixel_inputs<RGB>(a, b, pix, live, xp, yp);
        float gv[9], gt[NT];
        int tex_own = -1;
        float tex_val[3] = {0.f, 0.f, 0.f};
        bool ok = false;
        if (live) ok = backward_pair<DIST, ALPHA, RGB, SQ, TEXM, 1>(a, dp, rec, px, fn, face_lin, gv, gt, tex_own, tex_val);
#pragma unroll
        for (int k = 0; k < 9; k++) asm("" : "+v"(gv[k]));
#pragma unroll
        for (int k = 0; k < NG - 9; k++) asm("" : "+v"(gt[k]));
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k] += ok ? gv[k] : 0.f;
#pragma unroll
        for (int k = 0; k < NG - 9; k++) acc[9 + k] += ok ? gt[k] : 0.f;
        if constexpr (TEXM == kTexSurfaceN) {
            // surface texels, T > 1: a pair feeds one texel of the face's block; texel by texel, a fixed butterfly over the
            // lanes and one writer (the rows of a face are walked in ascending order by exactly one wavefront at a time:
            // the band kernel leaves these faces to the per-face kernel)
            if (!ok) tex_own = -1;
            for (int j = 0; j < a.T; j++) {
                if (!__any(tex_own == j)) continue;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float v = tex_own == j ? tex_val[k] : 0.f;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                    if (lane == 0) a.grad_textures[(face_lin * a.T + j) * 3 + k] += v;
                }
            }
        }
    }
    }
}

// the 64 lane sums of every component, combined in a fixed butterfly; lane k returns component k
template <int NG>
__device__ __forceinline__ float det_combine(const float (&acc)[NG])
{
    const int lane = threadIdx.x & 63;
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < NG; k++) {
        float v = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == k) mine = v;
    }
    return mine;
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void render_backward_faces_body(const RenderArgs& a, const float* __restrict__ boxes)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NG = GradSlots<TEXM>::n;
    const int lane = threadIdx.x & 63;
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};
    // Workgroups are dispatched round-robin over the 8 XCDs: XCD x takes the images x, x + 8, ... one after the other, so
    // that the planes of the image its waves are gathering from stay in that XCD's L2.
    const int xcd = blockIdx.x & 7;
    const long j = (long)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);     // kDetThreads / 64 faces per workgroup
    const int b = xcd + 8 * (int)(j / a.nf), fn = (int)(j % a.nf);
    if (b >= a.B) return;
    const long face_lin = (long)b * a.nf + fn;
    const DetBox d = det_box(a, boxes, face_lin);
    if (d.empty) return;
    // A box that takes many steps -- a sliver whose cull box degenerated to a large part of the image (23 of the 81920 faces
    // of the headline batch, but one wavefront scanning 256^2 pixels alone takes as long as all the others together) -- is
    // left to the band kernel, which cuts it into bands of kDetBandRows rows.
    bool by_bands = false;
    if ((TEXM != kTexSurfaceN || RGB == kRgbNone) && a.det_list) {
        const int wp = min(d.W, 64), rows_per_step = 64 / wp;
        const long steps = (long)((d.yi1 - d.yi0) / rows_per_step + 1) * ((d.W + wp - 1) / wp);
        if (steps > kDetBigSteps) {
            int at = 0;
            if (lane == 0) at = atomicAdd(a.det_count, 1);
            at = __builtin_amdgcn_readfirstlane(at);
            if (at < kDetBigCap) {
                if (lane == 0) a.det_list[at] = (int)face_lin;
                return;
            }
            // The list is full (which faces found a slot depends on the order the waves arrived in): this wave walks the
            // face itself, but band by band and summing the bands in ascending order -- the very sums, in the very order,
            // the band kernel and det_reduce_kernel form -- so the gradient does not depend on who got a slot.
            by_bands = true;
        }
    }
    float rec[REC];
    load_record<0, REC>(rec, (RecPtr)a.records + face_lin * REC);         // the whole record, wave-uniform: scalar loads
    float acc[NG];
    float mine = 0.f;
    if (by_bands) {
        const int nbands = (a.is + kDetBandRows - 1) / kDetBandRows;
        for (int band = 0; band < nbands; band++) {
#pragma unroll
            for (int k = 0; k < NG; k++) acc[k] = 0.f;
            const int ya = max(d.yi0, band * kDetBandRows), yb = min(d.yi1, band * kDetBandRows + kDetBandRows - 1);
            if (ya <= yb) det_rows<DIST, ALPHA, RGB, SQ, TEXM>(a, dp, rec, b, fn, face_lin, d.x0, d.W, ya, yb, acc);
            mine += det_combine<NG>(acc);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NG; k++) acc[k] = 0.f;
        det_rows<DIST, ALPHA, RGB, SQ, TEXM>(a, dp, rec, b, fn, face_lin, d.x0, d.W, d.yi0, d.yi1, acc);
        // lane k adds component k to the gradient -- one read-modify-write round trip for all of them
        mine = det_combine<NG>(acc);
    }
    if (lane < NG) {
        float* dst = lane < 9 ? a.grad_faces + face_lin * 9 + lane : a.grad_textures + face_lin * (NG - 9) + (lane - 9);
        *dst += mine;
    }
}

// deferred faces: one work item = (face, band of kDetBandRows image rows); its component sums go to a row of the scratch
// region, every row has exactly one writer
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void render_backward_bands_body(const RenderArgs& a, const float* __restrict__ boxes)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NG = GradSlots<TEXM>::n;
    const int lane = threadIdx.x & 63;
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};
    const int n = min(__builtin_amdgcn_readfirstlane(*a.det_count), kDetBigCap);
    const int nbands = (a.is + kDetBandRows - 1) / kDetBandRows;
    const long items = (long)n * nbands;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int bi = (int)(item / nbands), band = (int)(item - (long)bi * nbands);
        const long face_lin = a.det_list[bi];
        const int b = (int)(face_lin / a.nf), fn = (int)(face_lin - (long)b * a.nf);
        const DetBox d = det_box(a, boxes, face_lin);
        float acc[NG];
#pragma unroll
        for (int k = 0; k < NG; k++) acc[k] = 0.f;
        const int ya = max(d.yi0, band * kDetBandRows), yb = min(d.yi1, band * kDetBandRows + kDetBandRows - 1);
        if (!d.empty && ya <= yb) {
            float rec[REC];
            load_record<0, REC>(rec, (RecPtr)a.records + face_lin * REC);
            det_rows<DIST, ALPHA, RGB, SQ, TEXM>(a, dp, rec, b, fn, face_lin, d.x0, d.W, ya, yb, acc);
        }
        const float mine = det_combine<NG>(acc);
        if (lane < NG) a.det_partial[(item) * kDetSlots + lane] = mine;
    }
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kDetThreads) void render_backward_faces_kernel(const RenderArgs a, const float* __restrict__ boxes)
{
    render_backward_faces_body<DIST, ALPHA, RGB, SQ, TEXM>(a, boxes);
}
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(kThreads) void render_backward_bands_kernel(const RenderArgs a, const float* __restrict__ boxes)
{
    render_backward_bands_body<DIST, ALPHA, RGB, SQ, TEXM>(a, boxes);
}

// the bands of a deferred face, summed in ascending order (one wavefront per face, lane = component) and added to the gradient
__global__ __launch_bounds__(kThreads) void det_reduce_kernel(const RenderArgs a, int ng)
{
    const int lane = threadIdx.x & 63;
    const int n = min(*a.det_count, kDetBigCap);
    const int bi = blockIdx.x;
    if (bi >= n || lane >= ng) return;
    const int nbands = (a.is + kDetBandRows - 1) / kDetBandRows;
    const long face_lin = a.det_list[bi];
    float v = 0.f;
    for (int band = 0; band < nbands; band++) v += a.det_partial[((long)bi * nbands + band) * kDetSlots + lane];
    float* dst = lane < 9 ? a.grad_faces + face_lin * 9 + lane : a.grad_textures + face_lin * (ng - 9) + (lane - 9);
    *dst += v;
}

// Exhaustive check of sqrt_rn / rcp_rn against the compiler's correctly rounded expansions: every float bit pattern
// in [2^-96, 2^96] (what == 0: sqrt, 1: reciprocal of +x, 2: reciprocal of -x).  out[0] = mismatches, out[1] = tested,
// out[2..] = up to 14 offending bit patterns.
__global__ __launch_bounds__(256) void selftest_kernel(int what, unsigned long long* out)
{
    // what 0..2: sqrt_rn, rcp_rn(+x), rcp_rn(-x) on every float of [2^-96, 2^96];  what 3 / 4: norm_cdf(u) / norm_cdf(-u) against the
    // library's double normcdf rounded to float -- what the reference's kernel compiled for this platform computes -- on every
    // float u of [0, 6] (the polynomial's whole range and what lies beyond it)
    const bool ncdf = what >= 3;
    __shared__ double s_ntab[kNormRows * kNormRow];          // what 3 / 4: the table form (norm_cdf_tab), 5 / 6: the polynomial form (norm_cdf)
    for (int k = threadIdx.x; k < kNormRows * kNormRow; k += 256) s_ntab[k] = (&kNormTab[0][0])[k];
    __syncthreads();
    const unsigned lo = ncdf ? 0u : __float_as_uint(0x1p-96f), hi = ncdf ? __float_as_uint(6.f) : __float_as_uint(0x1p+96f);
    unsigned long long bad = 0, n = 0;
    for (unsigned long long u = (unsigned long long)lo + blockIdx.x * 256ull + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * 256ull) {
        float x = __uint_as_float((unsigned)u);
        if (what == 2 || what == 4 || what == 6) x = -x;
        float got, want;
        if (ncdf)           { got = what <= 4 ? norm_cdf_tab(x, s_ntab) : norm_cdf(x); want = (float)normcdf((double)x); }
        else if (what == 0) { got = sqrt_rn(x); want = sqrtf(x); }
        else                { got = rcp_rn(x); want = 1.f / x; }
        n++;
        // (norm_cdf: values at or below 5e-7 are equal for this purpose -- the pair is skipped at 1e-6, kernel.cu:784, whatever they are)
        if (__float_as_uint(got) != __float_as_uint(want) && !(ncdf && got <= 5e-7f && want <= 5e-7f)) {
            bad++;
            const unsigned long long slot = atomicAdd(out + 15, 1ull);
            if (slot < 12) out[2 + slot] = u;
            // the largest difference in units of the last place
            const long long d = (long long)__float_as_uint(got) - (long long)__float_as_uint(want);
            atomicMax(out + 14, (unsigned long long)(d < 0 ? -d : d));
        }
    }
    atomicAdd(out + 0, bad);
    atomicAdd(out + 1, n);
}

}  // namespace gendr
