// gendr_capi.hip -- extern "C" entry points of libgendr_hip.so (declared in include/gendr_hip.h).
//
// Replaces the launchers forward_render_cuda / backward_render_cuda (kernel.cu:1071-1227) and the scalar
// exports (kernel.cu:1230-1270) of the reference.  Differences by design: launches go to the caller's
// stream (the reference uses the null stream, kernel.cu:1103,1118,1190), invalid options are rejected
// with an error code instead of a device printf + NaN, nothing is allocated here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include "gendr_kernels.h"
#include "gendr_team.h"
#include "gendr_project.h"
#include "gendr_voxel.h"
#include "gendr_texture.h"
#include "gendr_light.h"
#include "compat/gendr_f64.h"

using namespace gendr;

namespace {

int texture_mode(const gendr_params* p, int T)
{
    if (p->texture_type == 1) return kTexVertex;
    return T == 1 ? kTexSurface1 : kTexSurfaceN;
}

typedef void (*render_kernel_t)(const RenderArgs);
typedef void (*faces_kernel_t)(const RenderArgs, const float*);

// Specialised instantiations: the option sets of BASELINE.json's configs and of the reference's experiment
// scripts get their own kernel (only their own CDF / t-conorm branch is compiled in); everything else runs
// the runtime-dispatch kernel of its texture mode, which carries all 18 x 10 branches.
struct KernelKey { int dist, alpha, rgb, sq, texm; };
struct KernelEntry { KernelKey key; render_kernel_t fwd, bwd; faces_kernel_t det, det_bands; };   // det: deterministic backward (runtime-dispatch rows only)

#define GENDR_SPECIALISE(D, A, RGB, SQ, TEXM) \
    { {D, A, RGB, SQ, TEXM}, render_forward_kernel<D, A, RGB, SQ, TEXM>, render_backward_kernel<D, A, RGB, SQ, TEXM>, nullptr, nullptr }

// same, with the register budget capped for 6 (forward) / 5 (backward) waves per SIMD
#define GENDR_SPECIALISE_OCC(D, A, RGB, SQ, TEXM) \
    { {D, A, RGB, SQ, TEXM}, render_forward_kernel_w6<D, A, RGB, SQ, TEXM>, render_backward_kernel_w5<D, A, RGB, SQ, TEXM>, nullptr, nullptr }

// same with explicit register-budget suffixes for the forward / backward kernels (_wl 5/4, _wa 4, _wf 2 waves per SIMD)
#define GENDR_SPECIALISE_K2(D, A, RGB, SQ, TEXM, KF, KB) \
    { {D, A, RGB, SQ, TEXM}, render_forward_kernel_##KF<D, A, RGB, SQ, TEXM>, render_backward_kernel_##KB<D, A, RGB, SQ, TEXM>, nullptr, nullptr }
#define GENDR_SPECIALISE_K(D, A, RGB, SQ, TEXM, KF, KB) GENDR_SPECIALISE_K2(D, A, RGB, SQ, TEXM, KF, KB)

#ifndef C2B
#define C2B w5
#endif
#ifndef C5F
#define C5F wa
#endif
#ifndef C5B
#define C5B wa
#endif
// -DGENDR_DEV_MIN=1 (tools/devbuild.sh): a library with the kernels of ONE regime only -- opt_shape.py's two renderers and the team
// kernels -- that compiles in seconds instead of minutes, for A/B experiments on those kernels.  Never shipped: every other option
// set lands on a kernel of another option set.  -DGENDR_DEV_MIN=2 / 3 / 4: BASELINE config 2's / 3's / 4's kernels only (tools/ab.sh, tools/ab_cfg.sh).
#ifndef GENDR_DEV_MIN
#define GENDR_DEV_MIN 0
#endif
#if GENDR_DEV_MIN == 2
const KernelEntry kSpecialised[] = {
    GENDR_SPECIALISE_K(kUniform,     kProbabilistic, 1, 0, kTexSurface1, w6, C2B),
};
#elif GENDR_DEV_MIN == 3      // BASELINE config 3's kernels only
const KernelEntry kSpecialised[] = {
    GENDR_SPECIALISE_K(kGaussian,    kEinstein,      1, 1, kTexSurface1, w6, wa),
};
#elif GENDR_DEV_MIN == 4      // BASELINE config 4's
const KernelEntry kSpecialised[] = {
    GENDR_SPECIALISE_OCC(kLogistic,  kProbabilistic, 1, 0, kTexSurface1),
};
#elif GENDR_DEV_MIN
const KernelEntry kSpecialised[] = {
    GENDR_SPECIALISE_OCC(kLogistic,  kProbabilistic, 0, 0, kTexSurface1),
    GENDR_SPECIALISE(kHeaviside,     kAlphaHard,     0, 0, kTexSurface1),
};
#else
const KernelEntry kSpecialised[] = {
    GENDR_SPECIALISE_K(kUniform,     kProbabilistic, 1, 0, kTexSurface1, w6, C2B),  // C2 headline; library defaults
    GENDR_SPECIALISE_K(kGaussian,    kEinstein,      1, 1, kTexSurface1, w6, wa),   // C3 (backward: 4 waves, less spill)
    GENDR_SPECIALISE_OCC(kLogistic,  kProbabilistic, 1, 0, kTexSurface1),   // C4
    GENDR_SPECIALISE_K(kGamma,       kYager,         1, 0, kTexVertex, C5F, C5B),   // C5
    GENDR_SPECIALISE_OCC(kUniform,   kProbabilistic, 0, 0, kTexSurface1),   // train_reconstruction.py:181-196,557 (uniform, hard RGB)
    GENDR_SPECIALISE_OCC(kLogistic,  kProbabilistic, 0, 0, kTexSurface1),   // opt_shape.py:134-145 soft renderer (its default --dist-func logistic, hard RGB)
    GENDR_SPECIALISE(kHeaviside,     kAlphaHard,     0, 0, kTexSurface1),   // opt_shape.py:146-157 hard renderer (it passes dist_squared=True: dead for dist_func 0, see pick_kernel)
    // alpha-only (silhouette) kernels, SURVEY f-4: what opt_shape / train_reconstruction actually consume
    GENDR_SPECIALISE_K(kUniform,     kProbabilistic, kRgbNone, 0, kTexSurfaceN, w6, wa),
    GENDR_SPECIALISE_K(kLogistic,    kProbabilistic, kRgbNone, 0, kTexSurfaceN, w6, wa),
    GENDR_SPECIALISE(kHeaviside,     kAlphaHard,     kRgbNone, 0, kTexSurfaceN),
};
#endif

// Team kernels (gendr_team.h; gendr_params::team): one workgroup per tile, for calls of few tiles with thousands of pairs each.
// The option sets whose one-wave kernels have the dense path and that the reference's scripts put into that regime.
struct TeamEntry { KernelKey key; render_kernel_t fwd, bwd; };
#define GENDR_TEAM_ROW(D, A, RGB, SQ, TEXM) \
    { {D, A, RGB, SQ, TEXM}, render_forward_team_kernel<D, A, RGB, SQ, TEXM>, render_backward_team_kernel<D, A, RGB, SQ, TEXM> }
const TeamEntry kTeam[] = {
    GENDR_TEAM_ROW(kLogistic, kProbabilistic, 0, 0, kTexSurface1),          // opt_shape.py:134-145 soft renderer
#if !GENDR_DEV_MIN
    GENDR_TEAM_ROW(kLogistic, kProbabilistic, 1, 0, kTexSurface1),          // BASELINE config 4's option set at small batches
    GENDR_TEAM_ROW(kLogistic, kProbabilistic, kRgbNone, 0, kTexSurfaceN),   // the alpha-only twin
#endif
};
// ... and the runtime-dispatch row of the 13 light distributions x 5 light aggregators (surface texture, T = 1): what opt_shape.py
// renders with any other --dist-func / --aggr-func of those (key: the classes of kGeneric)
const TeamEntry kTeamGeneric = { {-2, -2, -1, -1, kTexSurface1}, render_forward_team_kernel_wl<-2, -2, -1, -1, kTexSurface1>, render_backward_team_kernel<-2, -2, -1, -1, kTexSurface1> };

#if !GENDR_DEV_MIN
// alpha-only runtime-dispatch kernels, by the same four classes as kGeneric
#define GENDR_SIL_ROW(D, A, K) \
    { {D, A, kRgbNone, -1, kTexSurfaceN}, render_forward_kernel_##K<D, A, kRgbNone, -1, kTexSurfaceN>, render_backward_kernel_##K<D, A, kRgbNone, -1, kTexSurfaceN>, \
      render_backward_faces_kernel<D, A, kRgbNone, -1, kTexSurfaceN>, render_backward_bands_kernel<D, A, kRgbNone, -1, kTexSurfaceN> }
const KernelEntry kGenericSil[2][2] = {
    { GENDR_SIL_ROW(-1, -1, wf), GENDR_SIL_ROW(-1, -2, wf) },
    { GENDR_SIL_ROW(-2, -1, wa), GENDR_SIL_ROW(-2, -2, wl) },
};

// Runtime-dispatch kernels, four classes by what has to be compiled in: the 13 "light" distributions or all 18, the 5
// "light" alpha aggregators (max ... hamacher) or all 10.  The heavy branches (gamma's series, the transcendental
// t-conorms) cost registers, so an option set only pays for the heavy half it actually needs.  K = kernel suffix
// (register budget): _wl light x light, _wa light distributions x all aggregators, _wf whenever the heavy
// distributions are compiled in (their register need does not fit more than two waves per SIMD without heavy spills).
#define GENDR_GENERIC_ROW(D, A, TEXM, K) \
    { {D, A, -1, -1, TEXM}, render_forward_kernel_##K<D, A, -1, -1, TEXM>, render_backward_kernel_##K<D, A, -1, -1, TEXM>, \
      render_backward_faces_kernel<D, A, -1, -1, TEXM>, render_backward_bands_kernel<D, A, -1, -1, TEXM> }
#define GENDR_GENERIC_CLASS(D, A, K) \
    { GENDR_GENERIC_ROW(D, A, kTexSurface1, K), GENDR_GENERIC_ROW(D, A, kTexVertex, K), GENDR_GENERIC_ROW(D, A, kTexSurfaceN, K) }

// [distribution class: 0 = all, 1 = light][alpha class: 0 = all, 1 = light][texture mode]
const KernelEntry kGeneric[2][2][3] = {
    { GENDR_GENERIC_CLASS(-1, -1, wf), GENDR_GENERIC_CLASS(-1, -2, wf) },
    { GENDR_GENERIC_CLASS(-2, -1, wa), GENDR_GENERIC_CLASS(-2, -2, wl) },
};
#endif

const KernelEntry& pick_generic(const gendr_params* p, int texm, bool silhouette);

const KernelEntry& pick_kernel(const gendr_params* p, int texm, bool silhouette = false)
{
    const int rgb = silhouette ? kRgbNone : p->aggr_rgb_func;
    // dist_squared does nothing for dist_func 0: the fragment is the inside test (kernel.cu:762-764; soft_fragment()) and there is
    // no distance gradient (:375-376) -- the reference's opt_shape hard renderer passes True (opt_shape.py:149)
    const int sq = (p->dist_squared && p->dist_func != kHeaviside) ? 1 : 0;
    for (const KernelEntry& e : kSpecialised) {
        if (e.key.dist == p->dist_func && e.key.alpha == p->aggr_alpha_func && e.key.rgb == rgb &&
            e.key.sq == sq && e.key.texm == texm)
            return e;
    }
    return pick_generic(p, texm, silhouette);
}

// the runtime-dispatch row of an option set (also the home of the deterministic backward kernels)
const KernelEntry& pick_generic(const gendr_params* p, int texm, bool silhouette)
{
#if GENDR_DEV_MIN
    return kSpecialised[0];
#else
    if (silhouette) return kGenericSil[is_light_dist(p->dist_func) ? 1 : 0][is_light_alpha(p->aggr_alpha_func) ? 1 : 0];
    return kGeneric[is_light_dist(p->dist_func) ? 1 : 0][is_light_alpha(p->aggr_alpha_func) ? 1 : 0][texm];
#endif
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" float gendr_cull_radius(const gendr_params* p);

// gendr_params::team.  Automatic: the call holds at most kTeamMaxTilesShort tiles, or at most kTeamMaxTiles and the cull radius is at
// least kTeamMinRadiusPx pixels -- few tiles, each listing a face for every pixel within that radius.  Measured (tools/teamrule*.sh, one MI355X, forward +
// backward call, one-wave kernels -> team kernels): opt_shape.py's shape (24 views of 64^2, sigma 1e-2: 1 536 tiles, 4.4 pixels, 4 000
// pairs per live tile) 0.406 -> 0.234 ms; 8 views 0.333 -> 0.167; 128 views (8 192 tiles) 1.07 -> 0.86; sigma 3e-2 (pixel-mode tiles)
// 1.75 -> 0.87; 128^2 x 16 0.53 -> 0.39; 256^2 x 8 (BASELINE config 4's regime at a strong-scaling share) 0.76 -> 0.54; at 16 384 tiles
// the two are level, from 32 768 tiles the one-wave kernels win (512^2 x 32: 4.3 against 6.9 ms) -- with thousands of tiles in flight
// every SIMD has its waves and a team's barriers only cost.  A team kernel for BASELINE config 2's option set (244 pairs per tile)
// was measured too: backward -20 % up to 4 096 tiles, forward level, slower from 8 192 -- not built in.  Needs the entry pool (the
// team walks coverage entries) and the exact culling.
constexpr long kTeamMaxTiles = 8192, kTeamMaxTilesShort = 4096;
constexpr float kTeamMinRadiusPx = 2.f;
const TeamEntry* pick_team(const gendr_params* p, int texm, bool silhouette, long total_tiles, long ent_cap8)
{
    if (p->team < 0 || !p->cull || ent_cap8 <= 0 || p->deterministic) return nullptr;
    const int rgb = silhouette ? kRgbNone : p->aggr_rgb_func;
    const int sq = (p->dist_squared && p->dist_func != kHeaviside) ? 1 : 0;
    const TeamEntry* t = nullptr;
    for (const TeamEntry& e : kTeam)
        if (e.key.dist == p->dist_func && e.key.alpha == p->aggr_alpha_func && e.key.rgb == rgb && e.key.sq == sq && e.key.texm == texm) t = &e;
    bool generic = false;
    if (!t && !silhouette && texm == kTexSurface1 && is_light_dist(p->dist_func) && is_light_alpha(p->aggr_alpha_func) && p->dist_func != kHeaviside) {
        t = &kTeamGeneric;
        generic = true;
    }
    if (!t || p->team > 0) return t;
    // an option set with a specialised one-wave kernel keeps it unless it has a specialised team kernel too: the runtime-dispatch team
    // kernel pays ~30 % per pair for its branches (measured at BASELINE config 2's option set, 256^2 x 4: forward 49 -> 65 us)
    if (generic)
        for (const KernelEntry& e : kSpecialised)
            if (e.key.dist == p->dist_func && e.key.alpha == p->aggr_alpha_func && e.key.rgb == rgb && e.key.sq == sq && e.key.texm == texm) return nullptr;
    // up to 4 096 tiles whatever the tail (opt_shape.py sweeps sigma down to 1e-7 at 1 536 tiles: 0.095 -> 0.076 ms at sigma 1e-4 -- light
    // tiles, the gain is the backward call's), up to 8 192 where a tile holds pairs by the thousand
    if (total_tiles <= kTeamMaxTilesShort) return t;
    if (total_tiles > kTeamMaxTiles) return nullptr;
    const float r = gendr_cull_radius(p);
    return (r < 1e18f && r * (float)p->image_size * 0.5f >= kTeamMinRadiusPx) ? t : nullptr;
}

// The coverage kernel's team form (cover_kernel<REC, 8>: one 8-wave workgroup per listed tile, wave w examines faces [16 w, 16 w + 16)
// of every 128 the tile lists; the same entries in the same slots).  It pays where a tile lists faces by the dozen AND the call holds
// too few tiles to fill the chip with one wave each -- independent of whether the option set has team RENDER kernels (the heavy
// distributions at opt_shape.py's shape have none and still list 200 faces per tile).  Measured (tools/tcover_sweep.sh, forward phase,
// one-wave -> team form, logistic with sigma 1e-4 ... 3e-2): 64^2 x 24 0.053 -> 0.048 ... 0.521 -> 0.473 ms (the coverage kernel itself at
// opt_shape.py's shape 30.8 -> 20.3 us, at 70^2 x 5 with sigma 3e-2 89 -> 29 us), 64^2 x 8 0.044 -> 0.038 ... 0.320 -> 0.260; 128^2 x 8 and
// 256^2 x 1 level up to sigma 3e-3, then 0.131 -> 0.126 / 0.130 -> 0.122 and 0.561 -> 0.518 / 0.447 -> 0.407; 256^2 x 4 with short lists
// 0.046 -> 0.049: hence a rule on the faces a tile can expect to list -- nf x ((tile + 2 radii) / image + the width of a face of
// 2 / nf of the image)^2 >= 32, two steps' worth -- for calls of up to kTeamMaxTiles tiles.  gendr_params::team: -1 never, 2 always.
#ifndef GENDR_TEAM_COVER_MAX_TILES
#define GENDR_TEAM_COVER_MAX_TILES 8192
#endif
constexpr float kTeamCoverMinFaces = 32.f;
constexpr long kTeamCoverMaxTiles = GENDR_TEAM_COVER_MAX_TILES;
bool team_cover(const gendr_params* p, int nf, long total_tiles, long ent_cap8)
{
    if (p->team < 0 || !p->cull || ent_cap8 <= 0) return false;
    if (p->team >= 2) return true;
    if (total_tiles > kTeamCoverMaxTiles) return false;
    const float r = gendr_cull_radius(p);
    if (!(r < 1e18f)) return false;
    const float span = ((float)kTile + r * (float)p->image_size) / (float)p->image_size + sqrtf(2.f / (float)std::max(nf, 1));
    return (float)nf * span * span >= kTeamCoverMinFaces;
}

struct Workspace {
    size_t boxes_off, records_off, masks_off, lists_off, tileinfo_off, entries_off, hints_off, sorted_off, loose_off, control_off, det_off, total;
    bool ordered;              // the render kernels walk the heavy-first copy of the queue records (order_tiles_kernel)
    bool loose_lists;          // large images: flagged faces are boxed by loose_faces_kernel ahead of the binning kernel
    bool hints;                // the forward kernel leaves pair hints for the backward kernel (gendr_params::pair_hints)
    int tiles_x, chunks, supers_x, ncontrol;
    long ent_cap8;
};

// Entry pool (CoverEnt, 16 bytes each): every listed (tile, face) gets a slot, so the worst case is tiles * nf.  Sized
// for 32 listings per tile plus, per face, the tiles a face of a few pixels reaches with the option set's cull radius
// (at least 64) -- C2 lists 5.7 per tile and 4.6 per face, C4 (37-pixel radius) 45 per tile and 145 per face, C5 144 per
// face -- doubled for fewer than 8 batch items (the 8 regions of the pool are then bands of an image, and the bands
// under the object hold most of the listings), never more than the worst case.  A tile that finds the pool exhausted is
// still rendered exactly (first entry = -1: the render kernels apply the per-pixel tests themselves), only slower.
long entry_capacity(long B, long tiles, long nf, const gendr_params* p)
{
    const long worst = B * tiles * nf;
    const float r = gendr_cull_radius(p);                        // NDC units; +inf: no radius (or cull switched off)
    // No useful radius (cauchy, reciprocal, levy_rev; any distribution whose tail reaches half the image): every tile lists
    // every face and every entry would be a full pixel mask -- the pool would be the worst case (1.3 GiB at C2, 20 GiB at
    // C4) and add nothing.  Such option sets get NO pool: every listed tile takes the render kernels' own walk over all
    // faces (first entry = -1), which is the traversal they need anyway.
    if (!(r < 1.0f)) return 0;
    double per_face = 64.;
    {
        const double reach = 2. * (double)r * p->image_size / 2. / kTile + 3.;   // tiles across: 2 r in pixels / 8, plus the face
        per_face = fmax(per_face, 2. * reach * reach);
    }
    double want = 32. * (double)B * tiles + per_face * (double)B * nf;
    if (B < 8) want *= 2.;
    if (want > (double)worst) want = (double)worst;
    if (want > (double)0x7fffff00L) want = (double)0x7fffff00L;   // entry offsets are ints
    if (p->pool_entries_max > 0 && want > (double)p->pool_entries_max) want = (double)p->pool_entries_max;   // caller's limit
    return ((long)want + 7) / 8 * 8;
}

// gendr_params::pair_hints.  Automatic: on while the cull radius stays within kHintRadiusPx pixels -- the hints save the
// backward kernel the closest-point search (C2 -8.5 us of 104, C3 -6 %, C5 -7 %) and cost the forward kernel 3-4 % (C2 +3 us);
// with long tails (C4: 37 pixels) the backward kernel is not bound by its vector arithmetic and gains nothing.
constexpr float kHintRadiusPx = 16.f;
bool pair_hints_enabled(const gendr_params* p)
{
    if (!GENDR_PAIR_HINTS || p->pair_hints < 0 || !p->cull) return false;
    if (p->pair_hints > 0) return true;
    const float r = gendr_cull_radius(p);
    return r * (float)p->image_size * 0.5f <= kHintRadiusPx;
}

// workspace layout: [bin records B*nf*16 f32][face records B*nf*REC f32][tile masks B*tiles*chunks u64]
//                   [tile queues B*tiles i32][queue records B*tiles 4 x i32][entry pool][pair hints, one slot per entry slot]
//                   [heavy-first copy of the queue records, up to kOrderTilesMax tiles][control counters]
Workspace workspace_layout(int B, int nf, int T, const gendr_params* p)
{
    Workspace w;
    const int texm = texture_mode(p, T);
    w.tiles_x = (p->image_size + kTile - 1) / kTile;
    w.chunks = (nf + 63) / 64;
    w.supers_x = (w.tiles_x + 7) / 8;
    const size_t tiles = (size_t)B * w.tiles_x * w.tiles_x;
    w.ent_cap8 = entry_capacity(B, (long)w.tiles_x * w.tiles_x, nf, p) / 8;
    w.boxes_off = 0;
    w.records_off = align256((size_t)B * nf * kBinRec * sizeof(float));
    w.masks_off = w.records_off + align256((size_t)B * nf * record_floats(texm) * sizeof(float));
    // tile masks (binning -> coverage): not needed when there is no entry pool -- the coverage kernel then has nothing to do
    w.lists_off = w.masks_off + (w.ent_cap8 > 0 ? align256(tiles * w.chunks * sizeof(unsigned long long)) : 0);
    w.tileinfo_off = w.lists_off + align256(tiles * sizeof(int));
    w.entries_off = w.tileinfo_off + align256(tiles * sizeof(int4));
    // pair hints of the forward kernel for the backward kernel, one 16-byte slot per entry slot (PairHints in gendr_kernels.h)
    w.hints_off = w.entries_off + align256((size_t)w.ent_cap8 * 8 * sizeof(CoverEnt));
    w.hints = pair_hints_enabled(p) && w.ent_cap8 > 0;
    w.sorted_off = w.hints_off + (w.hints ? align256((size_t)w.ent_cap8 * 8 * sizeof(PairHints)) : 0);
    // heavy-first order of the queue records (up to kOrderTilesMax tiles; measured in round 3: skipping it below 8192 tiles,
    // where every tile could start at once, made batch 8 slower -- 123 vs 108 us per step -- since the sub-tile split puts
    // more work items than wave slots into the launch); no pool: no pair counts to order by
    w.ordered = (long)tiles <= kOrderTilesMax && tiles >= 16 && w.ent_cap8 > 0;
    // faces with a loose cull box, large images (loose_faces_kernel): [flag B*nf i32][box B*nf 4 x i32][list per image B x 16 i32]
    w.loose_off = w.sorted_off + (w.ordered ? align256(tiles * sizeof(int4)) : 0);
    // (loose_faces == 2 forces the list path at any image size: it is the default at 1024^2 and more -- BASELINE config 5 -- and
    // the parity suites run at 32 ... 768 pixels: ADVICE r4)
    w.loose_lists = (long)w.tiles_x * w.tiles_x >= GENDR_LOOSE_MIN_TILES || p->loose_faces == 2;
    w.control_off = w.loose_off + (w.loose_lists ? align256((size_t)B * nf * sizeof(int)) + align256((size_t)B * nf * sizeof(int4)) + align256((size_t)B * kLooseList * sizeof(int)) : 0);
    w.ncontrol = kCtlInts;
    // deterministic backward: [count + list of the deferred faces][their band sums]
    w.det_off = w.control_off + align256((size_t)w.ncontrol * sizeof(int));
    const size_t det_bands = (size_t)(p->image_size + kDetBandRows - 1) / kDetBandRows;
    w.total = w.det_off + (p->deterministic ? align256((size_t)(kDetBigCap + 64) * sizeof(int)) + align256((size_t)kDetBigCap * det_bands * kDetSlots * sizeof(float)) : 0);
    return w;
}

// gendr_params::deterministic: one wavefront per (image, face), see render_backward_faces_body; the few faces with a large
// cull box are cut into row bands (second launch) whose sums a third launch adds up in a fixed order.  Uses the scratch region
// at the end of the workspace (the one part of it that gendr_backward writes).
int launch_deterministic_backward(RenderArgs& a, const void* workspace, int B, int nf, int T, const gendr_params* p, bool silhouette, void* stream)
{
    // XCD x renders the images x, x + 8, ...; kDetThreads / 64 faces per workgroup
    const long per_xcd = ((long)((B + 7) / 8) * nf + (kDetThreads / 64) - 1) / (kDetThreads / 64);
    const long blocks = 8L * per_xcd;
    if (blocks > 0x7fffffffL) return GENDR_E_SHAPE;
    const Workspace w = workspace_layout(B, nf, T, p);
    hipStream_t s = (hipStream_t)stream;
    char* base = static_cast<char*>(const_cast<void*>(workspace));
    const float* boxes = reinterpret_cast<const float*>(base + w.boxes_off);
    a.det_count = reinterpret_cast<int*>(base + w.det_off);
    a.det_list = a.det_count + 64;
    a.det_partial = reinterpret_cast<float*>(base + w.det_off + align256((size_t)(kDetBigCap + 64) * sizeof(int)));
    if (hipMemsetAsync(a.det_count, 0, sizeof(int), s) != hipSuccess) return GENDR_E_LAUNCH;
    const int texm = texture_mode(p, T);
    const KernelEntry& k = pick_generic(p, texm, silhouette);
    hipLaunchKernelGGL(k.det, dim3((unsigned)blocks), dim3(kDetThreads), 0, s, a, boxes);
    if (texm != kTexSurfaceN || silhouette) {
        hipLaunchKernelGGL(k.det_bands, dim3(8192), dim3(kThreads), 0, s, a, boxes);
        const int ng = silhouette ? 9 : (texm == kTexSurface1 ? 12 : 18);
        hipLaunchKernelGGL(det_reduce_kernel, dim3(kDetBigCap), dim3(kThreads), 0, s, a, ng);
    }
    return hipGetLastError() == hipSuccess ? GENDR_OK : GENDR_E_LAUNCH;
}

// (m, shift) with n / d == (n * m) >> shift for every 0 <= n < 2^30, d >= 1: m = ceil(2^(30 + l) / d), l = ceil(log2 d)  (m <= 2^31 + 1)
static void div_magic(unsigned d, unsigned& m, int& shift)
{
    int l = 0;
    while ((1ull << l) < d) l++;
    shift = 30 + l;
    m = (unsigned)(((1ull << shift) + d - 1) / d);
}

int fill_args(RenderArgs& a, const void* workspace, const float* textures, int B, int nf, int T, const gendr_params* p)
{
    const int texm = texture_mode(p, T);
    const Workspace w = workspace_layout(B, nf, T, p);
    memset(&a, 0, sizeof(a));
    a.records = reinterpret_cast<const float*>(static_cast<const char*>(workspace) + w.records_off);
    a.masks = reinterpret_cast<const unsigned long long*>(static_cast<const char*>(workspace) + w.masks_off);
    a.tile_list = reinterpret_cast<int*>(static_cast<char*>(const_cast<void*>(workspace)) + w.lists_off);
    a.control = reinterpret_cast<int*>(static_cast<char*>(const_cast<void*>(workspace)) + w.control_off);
    a.tile_info_raw = reinterpret_cast<int4*>(static_cast<char*>(const_cast<void*>(workspace)) + w.tileinfo_off);
    a.tile_info = w.ordered ? reinterpret_cast<int4*>(static_cast<char*>(const_cast<void*>(workspace)) + w.sorted_off) : a.tile_info_raw;
    a.entries = reinterpret_cast<CoverEnt*>(static_cast<char*>(const_cast<void*>(workspace)) + w.entries_off);
    a.ent_cap8 = w.ent_cap8;
    if (w.loose_lists) {
        char* lb = static_cast<char*>(const_cast<void*>(workspace)) + w.loose_off;
        a.loose_flag = reinterpret_cast<const int*>(lb);
        a.loose_box = reinterpret_cast<int4*>(lb + align256((size_t)B * nf * sizeof(int)));
        a.loose_image = reinterpret_cast<int*>(lb + align256((size_t)B * nf * sizeof(int)) + align256((size_t)B * nf * sizeof(int4)));
    }
    {
        const float r = gendr_cull_radius(p);
        // (radius (1 + 2^-10))^2 rounded up: a computed squared distance that reaches it lies beyond the radius
        a.cull_r2 = r < 1e18f ? nextafterf((float)((double)r * (double)r * (1. + 1. / 512.)), INFINITY) : INFINITY;
    }
    a.hints = w.hints ? reinterpret_cast<PairHints*>(static_cast<char*>(const_cast<void*>(workspace)) + w.hints_off) : nullptr;
    a.textures = textures;
    a.B = B; a.nf = nf; a.T = T;
    a.R = (int)sqrt((double)T);                                  // kernel.cu:1098
    a.is = p->image_size;
    a.tiles_x = w.tiles_x;
    a.tiles_per_image = w.tiles_x * w.tiles_x;
    a.total_tiles = a.tiles_per_image * B;
    div_magic((unsigned)a.tiles_per_image, a.div_tpi_m, a.div_tpi_s);
    div_magic((unsigned)a.tiles_x, a.div_tx_m, a.div_tx_s);
    a.total_blocks = (a.total_tiles + (kThreads / 64) - 1) / (kThreads / 64);
    a.chunks = w.chunks;
    a.rec_floats = record_floats(texm);
    a.p = *p;
    a.thr = p->dist_eps * p->dist_scale;                         // float * float, kernel.cu:725
    a.softmax_sum0 = expf(p->aggr_rgb_eps / p->aggr_rgb_gamma);  // kernel.cu:729
    for (int k = 0; k < 3; k++) {
        const volatile float prod = p->background[k] * a.softmax_sum0;          // (float product, then float quotient: the device's two roundings)
        a.bg_soft[k] = prod / a.softmax_sum0;
    }
    a.r_scale = 1. / (double)p->dist_scale;
    a.r_gamma = 1. / (double)p->aggr_rgb_gamma;
    a.r_zrange = 1. / (double)(p->far_ - p->near_);              // float subtraction first, as kernel.cu:826
    a.r_nzrange = 1. / (double)(p->near_ - p->far_);             // kernel.cu:1026
    a.r_is = 1. / (double)p->image_size;
    a.rf_scale = (float)a.r_scale; a.rf_gamma = (float)a.r_gamma; a.rf_zrange = (float)a.r_zrange; a.rf_nzrange = (float)a.r_nzrange;
    if (p->dist_func == kGamma || p->dist_func == kGammaRev) {   // kernel.cu:309,:420-421: the pair-independent factors, in double
        a.gamma_k0 = (float)(1. / tgamma((double)p->dist_shape + 1.));
        a.gamma_pdf_c = pow(1. / (double)p->dist_scale, (double)p->dist_shape) / tgamma((double)p->dist_shape);
    }
    return texm;
}

// Grid of a render kernel / the coverage kernel, in one-wave workgroups: wave r of XCD x takes work items r, r + stride, ... of
// queue x, so the grid only has to hold the work items of the longest queue -- and every wave beyond that costs dispatch time:
// the dispatcher issues 600-1000 waves per microsecond, so the 16 384-wave floor of rounds 1-3 put 16-27 us under every
// launch of a small batch (measured in round 4 at batch 8: render kernels 27 / 30 us whatever the tiles were split into).
// Per queue (an eighth of the tiles):
//   * a quarter of its tiles (large batches: at most a quarter to a third of the tiles list a face -- one tile per wave, no
//     wave launched in vain; measured: 64 > 128 > 256 > 512 threads per workgroup, a quarter > a half),
//   * but one wave per tile up to 2048 of them (small batches are bound by the longest wave, not by the number of waves:
//     batch 16 192 -> 154 us per step in round 1),
//   * and never fewer than the work items the graded sub-tile split may create (`split_items`, see split_budget).
// A multiple of 8 so that every XCD gets the same number of workgroups.
#ifndef GENDR_GRID_DIV
#define GENDR_GRID_DIV 4
#endif
#ifndef GENDR_GRID_ONE
#define GENDR_GRID_ONE 2048
#endif
int render_blocks(int total_blocks, int split_items = 0)
{
    const int per_queue = (total_blocks + 7) / 8;
    int waves = (per_queue + GENDR_GRID_DIV - 1) / GENDR_GRID_DIV;
    waves = std::max(waves, std::min(per_queue, GENDR_GRID_ONE));
    waves = std::max(waves, split_items);
    return 8 * std::max(waves, 1);
}

// Waves of a render kernel the chip holds at once, per tile queue: occupancy (one-wave workgroups per CU) x CUs / 8.
// Queried once per kernel and device.
int resident_per_queue(render_kernel_t k, int threads = kThreads)
{
    // (one slot per render kernel of the dispatch tables and device: ~60 kernels; a thread that fills the cache keeps querying
    // the kernels that did not fit -- ADVICE r4: 16 slots were fewer than a test session's option sets)
    struct Slot { render_kernel_t k; int dev; int v; };
    constexpr int kSlots = 256;
    static thread_local Slot cache[kSlots];
    static thread_local int used = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    for (int i = 0; i < used; i++)
        if (cache[i].k == k && cache[i].dev == dev) return cache[i].v;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), threads, 0) != hipSuccess || per_cu <= 0) per_cu = threads == kThreads ? 16 : 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    (void)hipGetLastError();
    const int v = per_cu * cus / 8;
    if (getenv("GENDR_DEBUG")) fprintf(stderr, "gendr: resident_per_queue per_cu %d cus %d -> %d\n", per_cu, cus, v);
    if (used < kSlots) cache[used++] = Slot{k, dev, v};
    return v;
}

// Work items (tile pieces) per queue the graded sub-tile split may create (order_tiles_kernel): what the chip holds of the
// render kernel with the smaller occupancy at once -- and not one more.  (Rounds 4-5 added a quarter "because the pieces are
// uneven"; the wave time line of the backward kernel at 8 frames -- tools/wave_trace.py -- then showed what the surplus does: work
// items beyond the resident waves start when a slot frees, 14 us into a 32-us launch, and the launch ends with them.  Measured,
// forward + backward call in ms, budget x 1.25 -> x 1: batch 1 0.074 -> 0.069, 2 0.079 -> 0.076, 4 0.083 -> 0.078, 8 0.100 -> 0.096,
// 16 0.116 -> 0.109, from 32 unchanged; x 0.875: batch 8 0.094 but batch 16 0.117; x 1.5: slower everywhere -- tools/ab_batches.sh.)
// Both render kernels walk the same grades (the pair hints depend on it), and both are launched with at least that many waves per queue.
int split_budget(const gendr_params* p, int texm, bool silhouette)
{
    const KernelEntry& k = pick_kernel(p, texm, silhouette);
    const int r = std::min(resident_per_queue(k.fwd), resident_per_queue(k.bwd));
    return r;
}

// Grid of a team kernel: the workgroups the chip holds at once (a team keeps its tile's state in LDS: a second generation of
// workgroups would only queue behind the first), but no more than the tiles of the longest queue; team r of XCD x takes the
// records r, r + stride, ... of queue x -- heaviest first.
int team_blocks(render_kernel_t k, int threads, int total_tiles)
{
    const int per_queue = (total_tiles + 7) / 8;
    return 8 * std::max(1, std::min(per_queue, resident_per_queue(k, threads)));
}

int check_launch()
{
    return hipGetLastError() == hipSuccess ? GENDR_OK : GENDR_E_LAUNCH;
}

}  // namespace

extern "C" {

int gendr_abi_version(void) { return GENDR_ABI_VERSION; }
int gendr_params_size(void) { return (int)sizeof(gendr_params); }

const char* gendr_error_string(int code)
{
    switch (code) {
    case GENDR_OK:              return "ok";
    case GENDR_E_NULL:          return "a required pointer is NULL";
    case GENDR_E_SHAPE:         return "B, nf, T or image_size out of range";
    case GENDR_E_DIST_FUNC:     return "unknown dist_func id (valid: 0..17)";
    case GENDR_E_ALPHA_FUNC:    return "unknown aggr_alpha_func id (valid: 0..9)";
    case GENDR_E_RGB_FUNC:      return "aggr_rgb_func must be 0 (hard) or 1 (softmax)";
    case GENDR_E_TEXTURE_TYPE:  return "texture_type must be 0 (surface) or 1 (vertex, T == 3)";
    case GENDR_E_DIST_PARAM:    return "invalid distribution parameter (dist_scale < 0, dist_eps < 1, or gamma dist_shape < 0)";
    case GENDR_E_TCONORM_PARAM: return "invalid t-conorm parameter p for the chosen aggr_alpha_func";
    case GENDR_E_LAUNCH:        return "kernel launch failed";
    case GENDR_E_WORKSPACE:     return "workspace buffer missing";
    default:                    return "unknown error";
    }
}

unsigned long long gendr_workspace_bytes(int B, int nf, int T, const gendr_params* p)
{
    if (!p || B < 0 || nf < 0 || T < 1 || p->image_size < 1) return 0;
    return (unsigned long long)workspace_layout(B, nf, T, p).total;
}

int gendr_validate(const gendr_params* p, int B, int nf, int T)
{
    if (!p) return GENDR_E_NULL;
    if (B < 0 || nf < 0 || T < 1 || p->image_size < 1 || p->image_size > 32768) return GENDR_E_SHAPE;
    if ((long long)B * ((p->image_size + kTile - 1) / kTile) * ((p->image_size + kTile - 1) / kTile) > 0x3fffffffLL) return GENDR_E_SHAPE;
    if (nf >= (1 << 24)) return GENDR_E_SHAPE;                   // face index is carried in a float (kernel.cu:853,998)
    if (p->dist_func < 0 || p->dist_func >= kNumDist) return GENDR_E_DIST_FUNC;
    if (p->aggr_alpha_func < 0 || p->aggr_alpha_func >= kNumAlpha) return GENDR_E_ALPHA_FUNC;
    if (p->aggr_rgb_func != 0 && p->aggr_rgb_func != 1) return GENDR_E_RGB_FUNC;
    if (p->texture_type != 0 && p->texture_type != 1) return GENDR_E_TEXTURE_TYPE;
    if (p->texture_type == 1 && T != 3) return GENDR_E_TEXTURE_TYPE;
    if (!(p->dist_scale >= 0.f) || !(p->dist_eps >= 1.f)) return GENDR_E_DIST_PARAM;   // functional/renderer.py:96,101
    if ((p->dist_func == kGamma || p->dist_func == kGammaRev) && p->dist_shape < 0.f) return GENDR_E_DIST_PARAM;   // kernel.cu:296
    const float tp = p->aggr_alpha_t_conorm_p;
    switch (p->aggr_alpha_func) {
    case kHamacher:       if (tp < 0.f) return GENDR_E_TCONORM_PARAM; break;                  // kernel.cu:491
    case kFrank:          if (tp <= 0.f || tp == 1.f) return GENDR_E_TCONORM_PARAM; break;    // :501
    case kYager: case kAczelAlsina: case kDombi:
                          if (tp <= 0.f) return GENDR_E_TCONORM_PARAM; break;                 // :512,:522,:534
    case kSchweizerSklar: if (tp >= 0.f) return GENDR_E_TCONORM_PARAM; break;                 // :552
    default: break;
    }
    if (tp != tp) return GENDR_E_TCONORM_PARAM;
    return GENDR_OK;
}

int gendr_uses_team(int B, int nf, int T, const gendr_params* p, int silhouette)
{
    if (silhouette) T = 4;                                       // kSilT: the workspace layout of the alpha-only entry points
    if (gendr_validate(p, B, nf, T) != GENDR_OK || B == 0) return 0;
    const Workspace w = workspace_layout(B, nf, T, p);
    return pick_team(p, texture_mode(p, T), silhouette != 0, (long)B * w.tiles_x * w.tiles_x, w.ent_cap8) ? 1 : 0;
}

int gendr_uses_team_cover(int B, int nf, int T, const gendr_params* p)
{
    if (gendr_validate(p, B, nf, T) != GENDR_OK || B == 0) return 0;
    const Workspace w = workspace_layout(B, nf, T, p);
    return team_cover(p, nf, (long)B * w.tiles_x * w.tiles_x, w.ent_cap8) ? 1 : 0;
}

float gendr_sigmoid_forward(int function_id, float sign, float x, float scale, float dist_shape, float dist_shift)
{
    const DistParams d = make_dist_params(scale, dist_shape, dist_shift);
    return cdf_rt(function_id, sign, x, d);
}
float gendr_sigmoid_backward(int function_id, float sign, float x, float scale, float dist_shape, float dist_shift)
{
    const DistParams d = make_dist_params(scale, dist_shape, dist_shift);
    return pdf_rt(function_id, sign, x, d);
}
float gendr_t_conorm_forward(int t_conorm_id, float a_existing, float b_new, int face_id, float t_conorm_p)
{
    (void)face_id;
    return tconorm_fold_rt(t_conorm_id, a_existing, b_new, t_conorm_p);
}
float gendr_t_conorm_backward(int t_conorm_id, float a_all, float b_current, int number_of_faces, float t_conorm_p)
{
    (void)number_of_faces;
    return tconorm_grad_rt(t_conorm_id, a_all, b_current, t_conorm_p);
}

// Distance d (NDC) such that every outside pixel farther than d from the triangle is skipped by the
// reference itself: either D(-x) <= 1e-6 (kernel.cu:784; searched against half that threshold so that the
// few-ulp difference between host and device libm cannot matter) or d^2 >= dist_eps * tau (kernel.cu:769).
// D(-x) is not exactly monotone in float arithmetic (wigner_semicircle loses ~1e-6 to cancellation right where it
// crosses the threshold), so the bisection's crossing is only a candidate: a sweep over [candidate, search end]
// then moves the radius beyond the LAST argument at which D(-x) still exceeds the limit
// (tests/test_host_logic.py::test_cull_radius_is_an_upper_bound_for_every_distribution scans the result).
// The sweep costs a few thousand CDF evaluations, so the result is cached per thread for the last option set.
static float cull_radius_uncached(const gendr_params* p)
{
    const float thr = p->dist_eps * p->dist_scale;
    float r_eps = sqrtf(thr) * (1.f + 1e-6f) + 1e-30f;
    if (!(r_eps == r_eps)) r_eps = INFINITY;
    if (p->dist_func == kHeaviside) return 0.f;      // outside pixels: check_pixel_inside fails, fragment = 0 (kernel.cu:762-764)
    const DistParams d = make_dist_params(p->dist_scale, p->dist_shape, p->dist_shift);
    const double limit = 0.5 * kProbThreshold;
    // x is the CDF argument: distance, or squared distance with dist_squared
    const float x_hi = p->dist_squared ? 64.f : 8.f;
    const float f_hi = cdf_rt(p->dist_func, -1.f, x_hi, d);
    float r_cdf = INFINITY;
    if (f_hi == f_hi && (double)f_hi <= limit) {
        float lo = 0.f, hi = x_hi;                   // invariant: cdf(-hi) <= limit
        const float f0 = cdf_rt(p->dist_func, -1.f, 0.f, d);
        if (f0 == f0 && (double)f0 <= limit) hi = 0.f;
        for (int it = 0; it < 64 && hi > lo; it++) {
            const float mid = 0.5f * (lo + hi);
            if (mid <= lo || mid >= hi) break;
            const float fm = cdf_rt(p->dist_func, -1.f, mid, d);
            if (fm == fm && (double)fm <= limit) hi = mid; else lo = mid;
        }
        // verification sweep: a fine linear grid over [hi, 2 hi] (where rounding noise competes with the limit) and
        // a geometric grid up to the search end; the radius moves past the last offender
        const int kLin = 2048, kGeo = 2048;
        const float start = hi > 0.f ? hi : 1e-12f;
        float last_bad = -1.f, after_bad = x_hi;      // last offending sample and the first clean sample behind it
        const double ratio = (double)x_hi / (double)start;
        for (int i = 0; i <= kLin + kGeo; i++) {
            const float x = i <= kLin ? start * (1.f + (float)i / kLin)
                                      : (float)((double)start * pow(ratio, (double)(i - kLin) / kGeo));
            if (!(x <= x_hi)) continue;
            const float fx = cdf_rt(p->dist_func, -1.f, x, d);
            if (!(fx == fx && (double)fx <= limit)) { if (x > last_bad) { last_bad = x; after_bad = x_hi; } }
            else if (x > last_bad && x < after_bad) after_bad = x;
        }
        if (last_bad >= 0.f) {
            // the offending region ends between the two samples (a cut like gamma's xs / tau > 15 is a jump): bisect
            float blo = last_bad, bhi = after_bad;
            for (int it = 0; it < 48; it++) {
                const float mid = 0.5f * (blo + bhi);
                if (mid <= blo || mid >= bhi) break;
                const float fm = cdf_rt(p->dist_func, -1.f, mid, d);
                if (fm == fm && (double)fm <= limit) bhi = mid; else blo = mid;
            }
            hi = fmaxf(hi, bhi);
        }
        // wigner_semicircle evaluates tau^2 - x^2 in float: up to ~1e-6 of cancellation noise on D right where it
        // crosses the limit, isolated spikes a sweep cannot see.  Beyond its support end (u < -1) it is exactly 0.
        if (p->dist_func == kWigner) hi = fmaxf(hi, p->dist_scale);
        r_cdf = p->dist_squared ? sqrtf(hi) : hi;
        r_cdf = r_cdf * (1.f + 1e-6f);
    }
    return fminf(r_cdf, r_eps);
}

float gendr_cull_radius(const gendr_params* p)
{
    if (!p || !p->cull) return INFINITY;
    struct Key { int dist, squared; float scale, shape, shift, eps; };
    static thread_local Key last = {-1, 0, 0.f, 0.f, 0.f, 0.f};
    static thread_local float last_r = 0.f;
    const Key k = {p->dist_func, p->dist_squared ? 1 : 0, p->dist_scale, p->dist_shape, p->dist_shift, p->dist_eps};
    if (memcmp(&k, &last, sizeof(Key)) == 0) return last_r;
    const float r = cull_radius_uncached(p);
    last = k;
    last_r = r;
    return r;
}

static int face_setup_impl(const float* faces, const float* textures, void* workspace,
                           int B, int nf, int T, const gendr_params* p, void* stream, bool silhouette);

// ---- alpha-only rendering (SURVEY f-4) ------------------------------------------------------------------------------
// The workspace is laid out for T = 4 surface texels, i.e. texture mode kTexSurfaceN: records without texels; no
// texture pointer is ever dereferenced by the alpha-only kernels.
static const int kSilT = 4;

unsigned long long gendr_silhouette_workspace_bytes(int B, int nf, const gendr_params* p)
{
    if (!p || p->texture_type != 0) return 0;
    return gendr_workspace_bytes(B, nf, kSilT, p);
}

int gendr_silhouette_forward(const float* faces, float* alpha, void* workspace, const float* target, float* iou_sums,
                             int B, int nf, const gendr_params* p, void* stream)
{
    if (p && p->texture_type != 0) return GENDR_E_TEXTURE_TYPE;
    const int v = gendr_validate(p, B, nf, kSilT);
    if (v != GENDR_OK) return v;
    if (!alpha || (target && !iou_sums)) return GENDR_E_NULL;
    if (B == 0) return GENDR_OK;
    if (!workspace) return GENDR_E_WORKSPACE;
    if ((long)B * nf > 0 && !faces) return GENDR_E_NULL;
    const int e = face_setup_impl(faces, faces /* never read in this texture mode */, workspace, B, nf, kSilT, p, stream, true);
    if (e != GENDR_OK) return e;
    if (target && hipMemsetAsync(iou_sums, 0, (size_t)B * 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) return GENDR_E_LAUNCH;
    RenderArgs a;
    const int texm = fill_args(a, workspace, nullptr, B, nf, kSilT, p);
    a.rgba = alpha;
    a.target = target;
    a.iou_sums = iou_sums;
    a.p.background_from_buffer = 0;
    if (const TeamEntry* tk = pick_team(p, texm, true, a.total_tiles, a.ent_cap8)) {
        hipLaunchKernelGGL(tk->fwd, dim3(team_blocks(tk->fwd, 64 * kTeamFwdWaves, a.total_tiles)), dim3(64 * kTeamFwdWaves), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    const KernelEntry& k = pick_kernel(p, texm, true);
    hipLaunchKernelGGL(k.fwd, dim3(render_blocks(a.total_blocks, split_budget(p, texm, true))), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

int gendr_silhouette_backward(const float* alpha, const void* workspace, const float* grad_alpha,
                              const float* target, const float* grad_iou, float* grad_faces,
                              int B, int nf, const gendr_params* p, void* stream)
{
    if (p && p->texture_type != 0) return GENDR_E_TEXTURE_TYPE;
    const int v = gendr_validate(p, B, nf, kSilT);
    if (v != GENDR_OK) return v;
    if (B == 0 || nf == 0) return GENDR_OK;
    if (!alpha || !grad_faces) return GENDR_E_NULL;
    if (!grad_alpha && !(target && grad_iou)) return GENDR_E_NULL;
    if (!workspace) return GENDR_E_WORKSPACE;
    RenderArgs a;
    const int texm = fill_args(a, workspace, nullptr, B, nf, kSilT, p);
    a.rgba = const_cast<float*>(alpha);
    a.grad_rgba = grad_alpha;
    a.target = target;
    a.grad_iou = grad_alpha ? nullptr : grad_iou;
    a.grad_faces = grad_faces;
    a.grad_textures = grad_faces;            // never written: the alpha-only kernels have no texture term
    a.p.background_from_buffer = 0;
    if (p->deterministic) return launch_deterministic_backward(a, workspace, B, nf, kSilT, p, true, stream);
    if (const TeamEntry* tk = pick_team(p, texm, true, a.total_tiles, a.ent_cap8)) {
        hipLaunchKernelGGL(tk->bwd, dim3(team_blocks(tk->bwd, 64 * kTeamBwdWaves, a.total_tiles)), dim3(64 * kTeamBwdWaves), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    const KernelEntry& k = pick_kernel(p, texm, true);
    hipLaunchKernelGGL(k.bwd, dim3(render_blocks(a.total_blocks, split_budget(p, texm, true))), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

// ---- float64 instantiation (kernel.cu:1102,1117,1189 AT_DISPATCH_FLOATING_TYPES) ------------------------------------
static int fill_args_f64(f64::Args& a, const double* faces, const double* textures, void* workspace,
                         int B, int nf, int T, const gendr_params* p)
{
    memset(&a, 0, sizeof(a));
    a.faces = faces; a.textures = textures; a.info = static_cast<const double*>(workspace);
    a.B = B; a.nf = nf; a.T = T; a.R = (int)sqrt((double)T); a.is = p->image_size;
    a.p = *p;
    const float thr = p->dist_eps * p->dist_scale;               // float * float, kernel.cu:725
    a.thr = (double)thr;
    a.sqrt_thr = sqrt((double)thr);                              // sqrt(scalar_t), kernel.cu:747
    a.softmax_sum0 = (double)expf(p->aggr_rgb_eps / p->aggr_rgb_gamma);   // exp of the float arguments, kernel.cu:729
    return GENDR_OK;
}

unsigned long long gendr_workspace_bytes_f64(int B, int nf, int T, const gendr_params* p)
{
    if (!p || B < 0 || nf < 0 || T < 1 || p->image_size < 1) return 0;
    return (unsigned long long)align256((size_t)B * nf * 27 * sizeof(double)) + 256;
}

int gendr_forward_f64(const double* faces, const double* textures, double* rgba, double* aggrs_info,
                      void* workspace, int B, int nf, int T, const gendr_params* p, void* stream)
{
    const int v = gendr_validate(p, B, nf, T);
    if (v != GENDR_OK) return v;
    if (!rgba || !aggrs_info) return GENDR_E_NULL;
    if (B == 0) return GENDR_OK;
    if (!workspace) return GENDR_E_WORKSPACE;
    const long total = (long)B * nf;
    if (total > 0 && (!faces || !textures)) return GENDR_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (total > 0)
        hipLaunchKernelGGL(f64::face_info_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, faces, static_cast<double*>(workspace), total);
    f64::Args a;
    fill_args_f64(a, faces, textures, workspace, B, nf, T, p);
    a.rgba = rgba; a.aux = aggrs_info;
    const long pixels = (long)B * p->image_size * p->image_size;
    hipLaunchKernelGGL(f64::forward_kernel, dim3((unsigned)((pixels + f64::kThreadsD - 1) / f64::kThreadsD)), dim3(f64::kThreadsD), 0, s, a);
    return check_launch();
}

int gendr_backward_f64(const double* faces, const double* textures, const double* rgba, const double* aggrs_info,
                       const void* workspace, const double* grad_rgba, double* grad_faces, double* grad_textures,
                       int B, int nf, int T, const gendr_params* p, void* stream)
{
    const int v = gendr_validate(p, B, nf, T);
    if (v != GENDR_OK) return v;
    if (B == 0 || nf == 0) return GENDR_OK;
    if (!faces || !textures || !rgba || !aggrs_info || !grad_rgba || !grad_faces || !grad_textures) return GENDR_E_NULL;
    if (!workspace) return GENDR_E_WORKSPACE;
    f64::Args a;
    fill_args_f64(a, faces, textures, const_cast<void*>(workspace), B, nf, T, p);
    a.rgba = const_cast<double*>(rgba); a.aux = const_cast<double*>(aggrs_info);
    a.grad_rgba = grad_rgba; a.grad_faces = grad_faces; a.grad_textures = grad_textures;
    a.p.background_from_buffer = 0;
    const long pixels = (long)B * p->image_size * p->image_size;
    hipLaunchKernelGGL(f64::backward_kernel, dim3((unsigned)((pixels + f64::kThreadsD - 1) / f64::kThreadsD)), dim3(f64::kThreadsD), 0,
                       (hipStream_t)stream, a);
    return check_launch();
}

#if GENDR_TRACE
// diagnostic builds only: copies the wave trace of the last backward launch to the host (n_waves x 8 x u64)
int gendr_trace_read(unsigned long long* dst, int n_waves)
{
    if (hipDeviceSynchronize() != hipSuccess) return GENDR_E_LAUNCH;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_trace), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? GENDR_OK : GENDR_E_LAUNCH;
}
int gendr_span_read(unsigned long long* dst, int kernel, int n)
{
    if (hipDeviceSynchronize() != hipSuccess || kernel < 0 || kernel > 2) return GENDR_E_LAUNCH;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_span_trace), (size_t)n * 2 * sizeof(unsigned long long),
                               (size_t)kernel * (1 << 16) * 2 * sizeof(unsigned long long)) == hipSuccess ? GENDR_OK : GENDR_E_LAUNCH;
}
#endif

int gendr_selftest(int what, unsigned long long* report16, void* stream)
{
    if (!report16) return GENDR_E_NULL;
    if (what < 0 || what > 6) return GENDR_E_SHAPE;
    if (hipMemsetAsync(report16, 0, 16 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return GENDR_E_LAUNCH;
    hipLaunchKernelGGL(selftest_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, what, report16);
    return check_launch();
}

int gendr_face_info(const float* faces, float* faces_info, int B, int nf, void* stream)
{
    if (!faces || !faces_info) return GENDR_E_NULL;
    const long total = (long)B * nf;
    if (total == 0) return GENDR_OK;
    // one wavefront per workgroup: 81920 faces are only 1280 wavefronts, spread them over all CUs
    const int blocks = total > 0 ? (int)((total + 63) / 64) : 1;                        // also zeroes the control block
    hipLaunchKernelGGL(face_info_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, faces, faces_info, total);
    return check_launch();
}

// face records + cull boxes, tile masks, tile queues, coverage entries
int gendr_face_setup(const float* faces, const float* textures, void* workspace,
                     int B, int nf, int T, const gendr_params* p, void* stream)
{
    return face_setup_impl(faces, textures, workspace, B, nf, T, p, stream, false);
}

// `silhouette`: the alpha-only render kernels follow (their occupancy sets the budget of the sub-tile split)
static int face_setup_impl(const float* faces, const float* textures, void* workspace,
                           int B, int nf, int T, const gendr_params* p, void* stream, bool silhouette)
{
    const int v = gendr_validate(p, B, nf, T);
    if (v != GENDR_OK) return v;
    const long total = (long)B * nf;
    if (B == 0) return GENDR_OK;
    if (!workspace) return GENDR_E_WORKSPACE;
    if (total > 0 && (!faces || !textures)) return GENDR_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int texm = texture_mode(p, T);
    const Workspace w = workspace_layout(B, nf, T, p);
    float* boxes = reinterpret_cast<float*>(static_cast<char*>(workspace) + w.boxes_off);
    float* recs = reinterpret_cast<float*>(static_cast<char*>(workspace) + w.records_off);
    int* control = reinterpret_cast<int*>(static_cast<char*>(workspace) + w.control_off);
    // one wavefront per workgroup: 81920 faces are only 1280 wavefronts, spread them over all CUs
    const int blocks = total > 0 ? (int)((total + 63) / 64) : 1;                        // also zeroes the control block
    const float sthr = sqrtf(p->dist_eps * p->dist_scale);       // sqrt(threshold), kernel.cu:725,747
    if (p->clear_ptr && ((reinterpret_cast<unsigned long long>(p->clear_ptr) & 15ull) || (p->clear_floats & 3ull))) return GENDR_E_SHAPE;
    float4* clear4 = p->clear_ptr && total > 0 ? static_cast<float4*>(p->clear_ptr) : nullptr;
    const long clear_quads = clear4 ? (long)(p->clear_floats / 4) : 0;
    const float cull_r = gendr_cull_radius(p);
    RenderArgs a;
    fill_args(a, workspace, textures, B, nf, T, p);
    // Faces with a loose cull box: face_setup_kernel flags them in their record and the coverage kernel evaluates them on the
    // pixels of every tile they are listed in (only with a finite cull radius and an entry pool: no pool, no coverage kernel).
    // In images of GENDR_LOOSE_MIN_TILES tiles and more (1024^2) -- where such a face is listed in tens of thousands of tiles
    // -- loose_faces_kernel first evaluates it on every pixel once (one more launch: 10 us when an image has such a face, 5 us
    // when none has) and the binning kernel lists it by the bounding box of its live pixels.  The per-image lists of that
    // path start with a tag word; the binning kernel clears it after use, so a workspace that is used again -- also by the
    // replay of a captured HIP graph, whose kernel arguments never change -- starts from empty lists, and fresh memory holds the
    // tag only by a 2^-27 chance (the kernel then ignores face numbers out of range).
    const int loose_on = (GENDR_LOOSE_FACES && cull_r < INFINITY && total > 0 && p->cull && p->loose_faces >= 0 && w.ent_cap8 > 0) ? 1 : 0;
    if (!loose_on || !w.loose_lists) a.loose_flag = nullptr;
    a.loose_stamp = 0x05a17c3d;                                                  // 27 bits: the list heads hold tag << 4 | entries
    if (texm == kTexSurface1)
        hipLaunchKernelGGL(face_setup_kernel<kTexSurface1>, dim3(blocks), dim3(64), 0, s, faces, textures, boxes, recs, total, sthr, cull_r, control, w.ncontrol, p->near_, p->far_, (float4*)nullptr, 0L,
                           loose_on, const_cast<int*>(a.loose_flag), a.loose_box, a.loose_image, a.loose_stamp, nf);
    else if (texm == kTexVertex)
        hipLaunchKernelGGL(face_setup_kernel<kTexVertex>, dim3(blocks), dim3(64), 0, s, faces, textures, boxes, recs, total, sthr, cull_r, control, w.ncontrol, p->near_, p->far_, (float4*)nullptr, 0L,
                           loose_on, const_cast<int*>(a.loose_flag), a.loose_box, a.loose_image, a.loose_stamp, nf);
    else
        hipLaunchKernelGGL(face_setup_kernel<kTexSurfaceN>, dim3(blocks), dim3(64), 0, s, faces, textures, boxes, recs, total, sthr, cull_r, control, w.ncontrol, p->near_, p->far_, (float4*)nullptr, 0L,
                           loose_on, const_cast<int*>(a.loose_flag), a.loose_box, a.loose_image, a.loose_stamp, nf);
    int e = check_launch();
    if (e != GENDR_OK) return e;
    if (a.loose_flag) {
        if (texm == kTexSurface1)    hipLaunchKernelGGL(loose_faces_kernel<record_floats(kTexSurface1)>, dim3(kLooseWaves), dim3(kThreads), 0, s, a);
        else if (texm == kTexVertex) hipLaunchKernelGGL(loose_faces_kernel<record_floats(kTexVertex)>, dim3(kLooseWaves), dim3(kThreads), 0, s, a);
        else                         hipLaunchKernelGGL(loose_faces_kernel<record_floats(kTexSurfaceN)>, dim3(kLooseWaves), dim3(kThreads), 0, s, a);
        e = check_launch();
        if (e != GENDR_OK) return e;
    }
    // one workgroup per (image, 64x64 super-tile): tile masks and the tile queues
    const long bblocks = (long)B * w.supers_x * w.supers_x;
    if (bblocks > 0x7fffffffL) return GENDR_E_SHAPE;
    hipLaunchKernelGGL(bin_faces_kernel, dim3((unsigned)bblocks), dim3(kBinThreads), 0, s, boxes, a, w.supers_x, p->cull, clear4, clear_quads);   // (also clears the caller's gradient buffer)
    e = check_launch();
    if (e != GENDR_OK) return e;
    // coverage entries of the listed tiles: one wave per queue slot, same walk as the render kernels
#ifndef GENDR_COVER_GRID_MUL
#define GENDR_COVER_GRID_MUL 1
#endif
    if (w.ent_cap8 == 0) return GENDR_OK;                       // no entry pool: every listed tile takes the render kernels' own walk
    {
        // region tags (CoverEnt) pay where entries cover most of a tile -- a cull radius of a tile's width and more -- and only the
        // kernels with the dense path read them (measured at BASELINE config 2, where neither holds: +5 us in the coverage kernel)
        const KernelEntry& k = pick_kernel(p, texm, silhouette);
        const bool dense = k.key.dist < 0 || k.key.dist == kLogistic || (GENDR_DENSE_GAMMA && k.key.dist == kGamma);           // dense_path<DIST>() of gendr_kernels.h
        a.want_tags = (dense && cull_r * (float)p->image_size * 0.5f >= (float)kTile) ? 1 : 0;
    }
    if (team_cover(p, nf, a.total_tiles, a.ent_cap8)) {
        // one 8-wave workgroup per listed tile, at most twice what the chip holds of them per queue
        const int per_queue = (int)((a.total_tiles + 7) / 8);
        const int tblocks = 8 * std::max(1, std::min(per_queue, 256));
#define GENDR_COVER_LAUNCH(W, GRID) do { \
        if (a.want_tags) { \
            if (texm == kTexSurface1)    hipLaunchKernelGGL((cover_kernel<record_floats(kTexSurface1), W, true>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
            else if (texm == kTexVertex) hipLaunchKernelGGL((cover_kernel<record_floats(kTexVertex), W, true>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
            else                         hipLaunchKernelGGL((cover_kernel<record_floats(kTexSurfaceN), W, true>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
        } else { \
            if (texm == kTexSurface1)    hipLaunchKernelGGL((cover_kernel<record_floats(kTexSurface1), W, false>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
            else if (texm == kTexVertex) hipLaunchKernelGGL((cover_kernel<record_floats(kTexVertex), W, false>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
            else                         hipLaunchKernelGGL((cover_kernel<record_floats(kTexSurfaceN), W, false>), dim3(GRID), dim3(kThreads * W), 0, s, a); \
        } } while (0)
        GENDR_COVER_LAUNCH(8, tblocks);
    } else {
        const int cblocks = render_blocks(a.total_blocks) * GENDR_COVER_GRID_MUL;
        GENDR_COVER_LAUNCH(1, cblocks);
    }
    e = check_launch();
    if (e != GENDR_OK) return e;
    // heavy tiles first: the render kernels walk the sorted copy of the queue records
    if (w.ordered) {
        const bool team = pick_team(p, texm, silhouette, a.total_tiles, a.ent_cap8) != nullptr;
        hipLaunchKernelGGL(order_tiles_kernel, dim3(8), dim3(kOrderThreads), 0, s, a, team ? 0 : split_budget(p, texm, silhouette), team ? 1 : 0);
        return check_launch();
    }
    return GENDR_OK;
}

int gendr_forward(const float* faces, const float* textures, float* rgba, float* aggrs_info,
                  void* workspace, int B, int nf, int T, const gendr_params* p, void* stream)
{
    const int v = gendr_validate(p, B, nf, T);
    if (v != GENDR_OK) return v;
    if (!rgba || !aggrs_info) return GENDR_E_NULL;
    if (B == 0) return GENDR_OK;
    if (!workspace) return GENDR_E_WORKSPACE;
    const int e = gendr_face_setup(faces, textures, workspace, B, nf, T, p, stream);
    if (e != GENDR_OK) return e;

    RenderArgs a;
    const int texm = fill_args(a, workspace, textures, B, nf, T, p);
    a.rgba = rgba;
    a.aux = aggrs_info;
    if (const TeamEntry* tk = pick_team(p, texm, false, a.total_tiles, a.ent_cap8)) {
        hipLaunchKernelGGL(tk->fwd, dim3(team_blocks(tk->fwd, 64 * kTeamFwdWaves, a.total_tiles)), dim3(64 * kTeamFwdWaves), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    const KernelEntry& k = pick_kernel(p, texm);
    hipLaunchKernelGGL(k.fwd, dim3(render_blocks(a.total_blocks, split_budget(p, texm, false))), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

int gendr_backward(const float* faces, const float* textures, const float* rgba, const float* aggrs_info,
                   const void* workspace, const float* grad_rgba,
                   float* grad_faces, float* grad_textures,
                   int B, int nf, int T, const gendr_params* p, void* stream)
{
    (void)faces;
    const int v = gendr_validate(p, B, nf, T);
    if (v != GENDR_OK) return v;
    if (B == 0 || nf == 0) return GENDR_OK;
    if (!textures || !rgba || !aggrs_info || !grad_rgba || !grad_faces || !grad_textures) return GENDR_E_NULL;
    if (!workspace) return GENDR_E_WORKSPACE;

    RenderArgs a;
    const int texm = fill_args(a, workspace, textures, B, nf, T, p);
    a.rgba = const_cast<float*>(rgba);
    a.aux = const_cast<float*>(aggrs_info);
    a.grad_rgba = grad_rgba;
    a.grad_faces = grad_faces;
    a.grad_textures = grad_textures;
    a.p.background_from_buffer = 0;
    if (p->deterministic) return launch_deterministic_backward(a, workspace, B, nf, T, p, false, stream);
    if (const TeamEntry* tk = pick_team(p, texm, false, a.total_tiles, a.ent_cap8)) {
        hipLaunchKernelGGL(tk->bwd, dim3(team_blocks(tk->bwd, 64 * kTeamBwdWaves, a.total_tiles)), dim3(64 * kTeamBwdWaves), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    const KernelEntry& k = pick_kernel(p, texm);
    hipLaunchKernelGGL(k.bwd, dim3(render_blocks(a.total_blocks, split_budget(p, texm, false))), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

int gendr_load_textures(const float* image, const float* face_uv, const int* is_update, float* textures,
                        int nf, int texture_res, int image_height, int image_width, void* stream)
{
    if (nf < 0 || texture_res < 1 || image_height < 1 || image_width < 1) return GENDR_E_SHAPE;
    if (nf == 0) return GENDR_OK;
    if (!image || !face_uv || !is_update || !textures) return GENDR_E_NULL;
    const long texels = (long)nf * texture_res * texture_res;
    hipLaunchKernelGGL(load_textures_kernel, dim3((unsigned)((texels + kTexThreads - 1) / kTexThreads)), dim3(kTexThreads), 0,
                       (hipStream_t)stream, image, face_uv, is_update, textures, texels, texture_res, image_height, image_width);
    return check_launch();
}

int gendr_create_texture_image(const float* face_uv, const float* textures, float* image, int nf, int texture_res_in,
                               int image_rows, int image_cols, int tile_width, float eps, void* stream)
{
    if (nf < 0 || texture_res_in < 1 || image_rows < 1 || image_cols < 1 || tile_width < 1 || image_cols % tile_width)
        return GENDR_E_SHAPE;
    if (nf == 0) return GENDR_OK;
    if (!face_uv || !textures || !image) return GENDR_E_NULL;
    const long pixels = (long)image_rows * image_cols;
    hipLaunchKernelGGL(create_texture_image_kernel, dim3((unsigned)((pixels + kTexThreads - 1) / kTexThreads)), dim3(kTexThreads), 0,
                       (hipStream_t)stream, face_uv, textures, image, pixels, nf, texture_res_in, image_cols / tile_width,
                       tile_width, eps);
    return check_launch();
}

size_t gendr_voxelize_workspace_bytes(int B, int voxel_size)
{
    if (B <= 0 || voxel_size <= 64) return 0;                  // up to 64^3 the flood fill lives in LDS
    const size_t W = ((size_t)voxel_size + 63) / 64;
    return (size_t)B * 2 * voxel_size * voxel_size * W * sizeof(u64);
}

int gendr_voxelize(const float* faces, int* voxels, void* workspace, int B, int nf, int voxel_size, void* stream)
{
    if (B < 0 || nf < 0 || voxel_size < 1 || voxel_size > 1024) return GENDR_E_SHAPE;
    if (B == 0) return GENDR_OK;
    if (!voxels || (nf > 0 && !faces)) return GENDR_E_NULL;
    const int vs = voxel_size, W = (vs + 63) / 64;
    const bool in_lds = vs <= 64;
    if (!in_lds && !workspace) return GENDR_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(voxels, 0, (size_t)B * vs * vs * vs * sizeof(int), st) != hipSuccess) return GENDR_E_LAUNCH;
    if (nf > 0) {
        const int work = vs * vs > nf ? vs * vs : nf;
        hipLaunchKernelGGL(voxel_surface_kernel, dim3((work + kVoxSurfaceThreads - 1) / kVoxSurfaceThreads, 4, B),
                           dim3(kVoxSurfaceThreads), 0, st, faces, voxels, nf, vs);
    }
    if (in_lds)
        hipLaunchKernelGGL(voxel_fill_kernel<true>, dim3(B), dim3(kVoxFillThreads), (size_t)2 * vs * vs * W * sizeof(u64), st,
                           voxels, (u64*)nullptr, vs, W);
    else
        hipLaunchKernelGGL(voxel_fill_kernel<false>, dim3(B), dim3(kVoxFillThreads), 0, st, voxels, (u64*)workspace, vs, W);
    return check_launch();
}

int gendr_light_faces(const float* vertices, const int* face_index, const float* textures, float* out,
                      int B, int nv, int nf, int T, int index_batched, const gendr_light_params* lp, void* stream)
{
    if (B < 0 || nv < 0 || nf < 0 || T < 0) return GENDR_E_SHAPE;
    if (!lp) return GENDR_E_NULL;
    if (lp->n_directional < 0 || lp->n_directional > GENDR_MAX_DIRECTIONAL) return GENDR_E_SHAPE;
    if ((long)B * nf * T == 0) return GENDR_OK;
    if (!vertices || !face_index || !textures || !out) return GENDR_E_NULL;
    const long faces = (long)B * nf;
    hipLaunchKernelGGL(light_faces_kernel, dim3((unsigned)((faces + kLightThreads - 1) / kLightThreads)), dim3(kLightThreads), 0,
                       (hipStream_t)stream, vertices, face_index, textures, out, B, nv, nf, T, index_batched, *lp);
    return check_launch();
}

int gendr_light_faces_backward(const float* vertices, const int* face_index, const float* textures, const float* grad_out,
                               float* grad_textures, float* grad_vertices,
                               int B, int nv, int nf, int T, int index_batched, const gendr_light_params* lp, void* stream)
{
    if (B < 0 || nv < 0 || nf < 0 || T < 0) return GENDR_E_SHAPE;
    if (!lp) return GENDR_E_NULL;
    if (lp->n_directional < 0 || lp->n_directional > GENDR_MAX_DIRECTIONAL) return GENDR_E_SHAPE;
    if ((long)B * nf * T == 0) return GENDR_OK;
    if (!vertices || !face_index || !textures || !grad_out) return GENDR_E_NULL;
    const long faces = (long)B * nf;
    hipLaunchKernelGGL(light_faces_backward_kernel, dim3((unsigned)((faces + kLightThreads - 1) / kLightThreads)), dim3(kLightThreads), 0,
                       (hipStream_t)stream, vertices, face_index, textures, grad_out, grad_textures, grad_vertices,
                       B, nv, nf, T, index_batched, *lp);
    return check_launch();
}

int gendr_camera_rotation(const float* eye, const float* target, const float* up, float* camera, int B,
                          int target_is_direction, void* stream)
{
    if (B < 0) return GENDR_E_SHAPE;
    if (B == 0) return GENDR_OK;
    if (!eye || !target || !up || !camera) return GENDR_E_NULL;
    hipLaunchKernelGGL(camera_rotation_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       eye, target, up, camera, B, target_is_direction);
    return check_launch();
}

int gendr_camera_rotation_backward(const float* eye, const float* target, const float* up, const float* grad_camera,
                                   float* grad_eye, float* grad_target, float* grad_up, int B, int target_is_direction,
                                   void* stream)
{
    if (B < 0) return GENDR_E_SHAPE;
    if (B == 0) return GENDR_OK;
    if (!eye || !target || !up || !grad_camera) return GENDR_E_NULL;
    hipLaunchKernelGGL(camera_rotation_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       eye, target, up, grad_camera, grad_eye, grad_target, grad_up, B, target_is_direction);
    return check_launch();
}

int gendr_project_faces(const float* vertices, const int* face_index, const float* camera, float* face_vertices,
                        int B, int nv, int nf, int index_batched, int perspective, float width_or_scale, void* stream)
{
    if (B < 0 || nv < 0 || nf < 0) return GENDR_E_SHAPE;
    if ((long)B * nf == 0) return GENDR_OK;
    if (!vertices || !face_index || !camera || !face_vertices) return GENDR_E_NULL;
    const long total = (long)B * nf * 3;
    const long blocks = (total + kProjThreads - 1) / kProjThreads;
    if (blocks > 0x7fffffffL) return GENDR_E_SHAPE;
    hipLaunchKernelGGL(project_faces_kernel, dim3((unsigned)blocks), dim3(kProjThreads), 0, (hipStream_t)stream,
                       vertices, face_index, camera, face_vertices, B, nv, nf, index_batched, perspective, width_or_scale);
    return check_launch();
}

int gendr_project_faces_backward(const float* vertices, const int* face_index, const float* camera,
                                 const float* grad_face_vertices, float* grad_vertices, float* grad_camera,
                                 int B, int nv, int nf, int index_batched, int perspective, float width_or_scale, void* stream)
{
    if (B < 0 || nv < 0 || nf < 0) return GENDR_E_SHAPE;
    if ((long)B * nf == 0) return GENDR_OK;
    if (!vertices || !face_index || !camera || !grad_face_vertices || !grad_vertices) return GENDR_E_NULL;
    const int blocks_per_item = (int)(((long)nf * 3 + kProjThreads - 1) / kProjThreads);
    const long blocks = (long)blocks_per_item * B;
    if (blocks > 0x7fffffffL) return GENDR_E_SHAPE;
    hipLaunchKernelGGL(project_faces_backward_kernel, dim3((unsigned)blocks), dim3(kProjThreads), 0, (hipStream_t)stream,
                       vertices, face_index, camera, grad_face_vertices, grad_vertices, grad_camera,
                       B, nv, nf, index_batched, perspective, width_or_scale, blocks_per_item);
    return check_launch();
}

}  // extern "C"
