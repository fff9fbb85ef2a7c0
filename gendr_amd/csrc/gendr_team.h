// gendr_team.h -- team kernels: one WORKGROUP per 8x8 pixel tile (gendr_params::team, ABI 7).
//
// The render kernels of gendr_kernels.h give a tile to one wavefront.  That is the right shape for BASELINE config 2 (22 k listed
// tiles of ~250 pairs) and the wrong one for what the reference's experiments/opt_shape.py renders (:134-145, :259-303: 24 views
// at 64^2, logistic, sigma 10^-2, a 4-pixel tail on 2-pixel faces): ~800 live tiles of 4 000 - 11 000 (pixel, face) pairs each.
// There the graded sub-tile split (TileWalk) cuts a tile into eight one-row pieces so that the chip has something to do, and
// every piece walks ALL of the tile's ~200 coverage entries for its 8 pixels, folds its results on 8 lanes, finds faces cut
// into 8-pair segments (backward: three times the per-face sums and atomics), and gets no pair hints (they describe the batches
// of unsplit tiles): measured 169 us forward, 170 us backward for 3.3 M pairs, five times the per-pair cost of config 2.
//
// A team renders the tile UNSPLIT:
//   * the tile's pair list (codes face << 6 | pixel, ascending) is built once, cooperatively, in a ring in shared LDS: every wave
//     loads the same 32 coverage entries, scans their pixel counts, and expands every eighth entry;
//   * batch k of the tile is codes [64 k, 64 k + 64) of that list -- the windows of the unsplit walk (for_each_batch), so the
//     pair hints keep their meaning: forward writes slot k, backward reads slot k;
//   * forward: eight B-waves evaluate eight batches side by side (phase B of render_forward_body: lane = pair, per-lane record
//     gather, distance, CDF, depth, colour) into a double-buffered result array; the ninth wave (lane = pixel) folds a chunk of
//     eight batches per pixel in ascending pair order -- the reference's order, kernel.cu:791-838 -- while the B-waves evaluate the
//     next chunk: one workgroup barrier per chunk;
//   * backward: eight symmetric waves take batches k = w, w + 8, ...: backward_pair(), per-face segment sums, one atomic per (batch,
//     face, component), exactly the unsplit kernel's batch.
// The pair functions are the ones every other path uses (barycentrics(), soft_fragment(), clip_and_depth(), sample_colour(),
// backward_pair()) on the same operands: forward results are bit-identical to the one-wave kernels' and to the all-pairs walk.
// Tiles of nearly full entries (36 pixels and more on average: the one-wave kernels' pixel mode) are rendered by the team with
// lane = pixel: a chunk is eight ENTRIES, one per B-wave, each evaluated on all 64 pixels with the face's record in scalar registers,
// folded entry by entry (backward: the entries dealt out to the eight waves).  A tile without a slice of the entry pool is rendered by
// one wave of the team with the all-faces walk.
//
// Measured (one MI355X, opt_shape.py's shape, 3.3 M pairs): forward kernel 169 -> 88 us, backward kernel 170 -> 88 us; what the stages
// of the design were worth is in DESIGN.md.
#pragma once

#include "gendr_kernels.h"

namespace gendr {

#ifndef GENDR_TEAM_B
#define GENDR_TEAM_B 8
#endif
#ifndef GENDR_TEAM_ABLATE
#define GENDR_TEAM_ABLATE 0                        // diagnostic builds (tools/devbuild.sh): 1 no fold, 2 no pair math, 3 no hints
#endif
#ifndef GENDR_TEAM_FWD_ROWS
#define GENDR_TEAM_FWD_ROWS 0                      // 1: the forward teams take a graded tile in row parts (measured: slower -- the fold chain
#endif                                             //    of a part is as long as the whole tile's); 0: whole tiles
constexpr bool kTeamFwdRows = GENDR_TEAM_FWD_ROWS != 0;
constexpr int kTeamB = GENDR_TEAM_B;               // B-waves of a forward team / waves of a backward team
constexpr int kTeamFwdWaves = kTeamB + 1;          // + the fold wave (wave 0)
constexpr int kTeamBwdWaves = kTeamB;
constexpr int kTeamRing = 4096;                    // codes in the shared ring (16 KB)
constexpr int kTeamRound = 32;                     // coverage entries appended per build round (at most 2048 codes)
static_assert((kTeamRing & (kTeamRing - 1)) == 0 && kTeamRound * 64 * 2 <= kTeamRing, "a round must fit beside a round's remainder");

// exclusive prefix sum over the 64 lanes without LDS: an inclusive scan inside each row of 16 lanes (four DPP row shifts), then the
// two DPP row broadcasts (lane 15 -> row 1 and 3, lane 31 -> rows 2 and 3) -- six vector instructions; __shfl_up steps are six
// ds_bpermute round trips, and both roles of a team scan once per chunk
__device__ __forceinline__ int wave_exclusive_scan_dpp(int v)
{
    int s = v;
#define GENDR_TEAM_DPP_ADD(ctrl, rows) s += __builtin_amdgcn_update_dpp(0, s, (ctrl), (rows), 0xF, false)
    GENDR_TEAM_DPP_ADD(0x111, 0xF);      // row_shr:1
    GENDR_TEAM_DPP_ADD(0x112, 0xF);      // row_shr:2
    GENDR_TEAM_DPP_ADD(0x114, 0xF);      // row_shr:4
    GENDR_TEAM_DPP_ADD(0x118, 0xF);      // row_shr:8
    GENDR_TEAM_DPP_ADD(0x142, 0xA);      // row_bcast:15 into rows 1 and 3
    GENDR_TEAM_DPP_ADD(0x143, 0xC);      // row_bcast:31 into rows 2 and 3
#undef GENDR_TEAM_DPP_ADD
    return s - v;
}

// what the fold wave needs from a pair (FwdRes of gendr_kernels.h in 24 bytes: the team keeps 2 x 8 x 64 of them in LDS)
struct __attribute__((aligned(8))) TeamRes { float frag, z, c0, c1, c2; int fnflags; };   // fnflags = face << 3 | kFlag*

// Build rounds: while another round of kTeamRound entries is sure to fit into the ring, every wave of the team loads the same
// entries (lane = entry), scans their pixel counts, and builder wave w expands entries w, w + NB, ... of the round (the lanes
// whose pixel bit is set store the code at list position base + set bits below the lane, as for_each_batch does).  The state
// (e_next, built) stays identical in all waves.  w < 0: a wave that only follows the state (the fold wave).
template <int NB>
__device__ __forceinline__ void team_build(const int4* __restrict__ ents, int cnt, int& e_next, int& built, int consumed, int* s_ring, int w,
                                           int first_code = 0, int end_code = 0x7fffffff, unsigned long long pixels = ~0ull)
{
    // [first_code, end_code): the codes the caller will read (a PART of the tile's list, backward) -- rounds that end before the range
    // are scanned, not expanded, and nothing is built past its end.
    // `pixels`: the pixel rows of the tile the caller renders (a row part, forward): the entries' masks are cut to them.
    // The entries of the following round are requested before the current one is expanded (a round is one L2 round trip otherwise).
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto more = [&]() { return e_next < cnt && built < end_code && built - consumed + kTeamRound * 64 <= kTeamRing; };
    if (!more()) return;
    int4 e = make_int4(0, 0, 0, 0);
    if (lane < min(kTeamRound, cnt - e_next)) e = ents[e_next + lane];
    do {
        const int n = min(kTeamRound, cnt - e_next);
        int4 e_ahead = make_int4(0, 0, 0, 0);
        if (lane < min(kTeamRound, cnt - e_next - n)) e_ahead = ents[e_next + n + lane];
        const unsigned long long mine = (((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z) & pixels;     // (lanes >= n hold zeros)
        const int np = __popcll(mine);
        const int pos = wave_exclusive_scan_dpp(np);
        const int total = __builtin_amdgcn_readlane(pos + np, 63);
        if (w >= 0 && built + total > first_code) {
            for (int j = w; j < n; j += NB) {
                const int fn = __builtin_amdgcn_readlane(e.x, j);
                const unsigned long long m = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(e.w, j) << 32) | (unsigned)__builtin_amdgcn_readlane(e.z, j)) & pixels;
                const int base = built + __builtin_amdgcn_readlane(pos, j);
                if (lane_in(m)) s_ring[bits_below(m, base) & (kTeamRing - 1)] = (fn << 6) | lane;
            }
        }
        built += total;
        e_next += n;
        e = e_ahead;
    } while (more());
}

// per-pixel state of the forward fold, kernel.cu:728-740
struct FwdPix { float alpha, ssum, smax, c0, c1, c2, depth_min; int face_min; };

// kernel.cu:791-838: one evaluated pair folded into its pixel's state (the `fold` of render_forward_body)
template <int ALPHA, int RGB>
__device__ __forceinline__ void team_fold(FwdPix& px, const TeamRes& res, const RenderArgs& a, int alpha_func, bool rgb_soft)
{
    const int flags = res.fnflags & 7;
    if (!(flags & kFlagContrib)) return;
    if (alpha_func == kAlphaHard) {
        if ((double)res.frag > 0.5) px.alpha = 1.f;
    } else if constexpr (ALPHA > 0) {
        px.alpha = TConorm<(ALPHA > 0 ? ALPHA : 1)>::fold(px.alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
    } else if constexpr (ALPHA == -2) {
        px.alpha = tconorm_fold_light_rt(alpha_func, px.alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
    } else {
        px.alpha = tconorm_fold_rt(alpha_func, px.alpha, res.frag, a.p.aggr_alpha_t_conorm_p);
    }
    if constexpr (RGB == kRgbNone) return;
    if (!(flags & kFlagRgb)) return;
    if (!rgb_soft) {                                                     // :815-822
        if (res.z < px.depth_min) {
            px.depth_min = res.z;
            px.face_min = res.fnflags >> 3;
            px.c0 = res.c0; px.c1 = res.c1; px.c2 = res.c2;
        }
    } else {                                                             // :824-838
        const float zn = res.z;
        const bool deeper = zn > px.smax;
        const float e = exp_f(div_by(deeper ? px.smax - zn : zn - px.smax, GENDR_R_GAMMA(a)));
        const float edz = deeper ? e : 1.f;
        const float ez = deeper ? 1.f : e;
        if (deeper) px.smax = zn;
        px.ssum = edz * px.ssum + ez * res.frag;
        px.c0 = edz * px.c0 + ez * res.frag * res.c0;
        px.c1 = edz * px.c1 + ez * res.frag * res.c1;
        px.c2 = edz * px.c2 + ez * res.frag * res.c2;
    }
}

// what follows the soft fragment of a contributing pair (:807-826): depth, eligibility, colour -> res
template <int RGB, int TEXM>
__device__ __forceinline__ void team_depth_colour(TeamRes& res, int fn, const Pair& q, const float* r, const RenderArgs& a, bool rgb_soft, long face_lin)
{
    int flags = kFlagContrib;
    res.frag = q.frag;
    if constexpr (RGB != kRgbNone) {
        float wc[3];
        const float zp = clip_and_depth(q, r, wc);
        if (!(zp < a.p.near_ || zp > a.p.far_)) {                         // :810
            flags |= kFlagDepthOk;
            const bool front = (__float_as_int(r[kRecBits]) & kBitFront) != 0;
            const bool eligible = rgb_soft ? (front || a.p.double_side)                         // :825
                                           : (inside_closed(q) && (a.p.double_side || front));  // :816
            if (eligible) {
                flags |= kFlagRgb;
                res.z = rgb_soft ? div_by(a.p.far_ - zp, GENDR_R_ZRANGE(a)) : zp;   // zp_norm (:826) or zp
                float cc[3]; int own;
                sample_colour<TEXM>(cc, own, wc, r, a, face_lin);
                res.c0 = cc[0]; res.c1 = cc[1]; res.c2 = cc[2];
            }
        }
    }
    res.fnflags = (fn << 3) | flags;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__device__ __forceinline__ void team_forward_body(const RenderArgs& a)
{
    constexpr int REC = record_floats(TEXM);
    constexpr bool kSil = RGB == kRgbNone;
    __shared__ int s_ring[kTeamRing];
    __shared__ float2 s_xy[64];                                          // pixel centres of the tile, fetched by pair lanes
    __shared__ TeamRes s_res[2][kTeamB * 64];                            // chunk c -> buffer c & 1: the chunk's results, GROUPED BY PIXEL
    __shared__ unsigned long long s_cnt[3][64];                          // chunk c -> buffer c % 3: per pixel, byte w = its pairs in B-wave w's batch
    __shared__ unsigned long long s_bmask[kTeamB][64];                   // private to a B-wave: per pixel, the pair lanes of its running batch
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const bool folder = wave == 0;
    const int wb = wave - 1;                                             // B-wave 0 .. kTeamB - 1
    const long P = (long)a.is * a.is;
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};
    const int alpha_func = ALPHA >= 0 ? ALPHA : a.p.aggr_alpha_func;
    const bool rgb_soft = RGB >= 0 ? (RGB == 1) : (a.p.aggr_rgb_func == 1);

    static_assert(kTeamB <= 8, "one byte per B-wave in a 64-bit count word");
    for (int i = threadIdx.x; i < kTeamB * 64; i += 64 * kTeamFwdWaves) (&s_bmask[0][0])[i] = 0ull;
    if (threadIdx.x < 3 * 64) (&s_cnt[0][0])[threadIdx.x] = 0ull;

    TileWalk tw;
    walk_init(tw, a, 1);
    tw.rank = (int)(blockIdx.x >> 3);                                    // rank / stride in WORKGROUPS of queue blockIdx.x & 7
    tw.stride = (int)(gridDim.x >> 3);
#ifndef GENDR_TEAM_PRIO
#define GENDR_TEAM_PRIO 3
#endif
    // The fold wave is the team's critical path -- one dependent chain per pixel, a pair per step -- and shares its SIMD with six
    // B-waves that are there for throughput: it issues first whenever it can (measured: see DESIGN.md).
    if (folder && GENDR_TEAM_PRIO) __builtin_amdgcn_s_setprio(GENDR_TEAM_PRIO);

    // ---- tiles no face is listed for (kernel.cu:728-740, :845-861), shared by all waves of all teams: as render_forward_body
    {
        auto fill_tile = [&](int tile) __attribute__((always_inline)) {
            TileCtx t;
            tile_setup(t, a, tile);
            if (!t.valid) return;
            if constexpr (kSil) { a.rgba[(long)t.b * P + t.pix] = 0.f; return; }
            float* out = a.rgba + (long)t.b * 4 * P + t.pix;
            float* aux = a.aux + (long)t.b * 2 * P + t.pix;
            const bool with_aux = !a.p.skip_unlisted_aux;
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
        auto fill_plane = [&](float* first, float v) __attribute__((always_inline)) {
            float4* p4 = reinterpret_cast<float4*>(first + (long)(lane >> 4) * a.is + (lane & 15) * 4);
            const float4 v4 = make_float4(v, v, v, v);
#pragma unroll
            for (int it = 0; it < 16; it++) p4[(long)it * a.is] = v4;
        };
        const bool wide_ok = !a.p.background_from_buffer && ((reinterpret_cast<unsigned long long>(a.rgba) | (kSil ? 0ull : reinterpret_cast<unsigned long long>(a.aux))) & 15ull) == 0ull;
        for (int r = tw.rank * kTeamFwdWaves + wave; r < tw.empties + (tw.total - tw.live); r += tw.stride * kTeamFwdWaves) {
            const int tile = __builtin_amdgcn_readfirstlane(r < tw.empties ? a.tile_list[tw.qend - 1 - r] : a.tile_info[tw.qbase + tw.live + (r - tw.empties)].x);
            const int g0 = tile < 0 ? -tile - 1 : tile;                  // a negative entry is an empty super-tile (bin_faces_kernel)
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
    }

    // pair hints for backward (PairHints): every team tile is rendered unsplit, slot k = batch k
    const bool hints_q = a.hints != nullptr;
    if (hints_q && tw.rank == 0 && threadIdx.x == 0) atomicOr(a.control + (blockIdx.x & 7) * kCtlStride + kCtlHintFlag, 1);
    __syncthreads();                                                     // the cleared masks

    int chunk_ctr = 0;                                                   // chunks so far: the parity selects the result buffer
    // work items: (tile, rows) -- order_tiles_kernel graded the heavy tiles into 8, 4 or 2 parts (TileWalk; team calls: by weight alone).
    // A fold cannot be cut, a tile can: a forward part is 1, 2 or 4 of the tile's pixel rows (sub_tile_mask), rendered by a team of
    // its own -- the part's B-chain shrinks with its pairs, its fold runs on fewer lanes but is bound by latency, not by lanes.
    for (int item = tw.rank; item < (kTeamFwdRows ? tw.items : tw.live); item += tw.stride) {
        GENDR_RELOADED_ARGS(a);
        int part_log2 = 0, part = 0;
        const int slot_i = kTeamFwdRows ? walk_item(tw, item, part_log2, part) : item;
        const i4v ti = *(const GENDR_CONST_AS i4v*)(a.tile_info + (tw.qbase + slot_i));   // (tile, first entry, entries, pairs)
        const unsigned long long my_rows = sub_tile_mask(part_log2, part);
        TileCtx t;
        tile_setup(t, a, ti.x);
        const float* recs_g = a.records + (long)t.b * a.nf * REC;
        const bool solo = ti.y < 0;                                      // one wave, lane = pixel: the tile has no slice of the entry pool
        const bool dense_tile = !solo && tile_in_pixel_mode(ti, 0);      // nearly full entries: lane = pixel, an entry per B-wave
        t.valid = t.valid && lane_in(my_rows);                 // the pixels this team renders

        FwdPix px;
        float bg[3];
        if (folder) {
#pragma unroll
            for (int k = 0; k < 3; k++)
                bg[k] = kSil ? 0.f : ((a.p.background_from_buffer && t.valid) ? a.rgba[((long)t.b * 4 + k) * P + t.pix] : a.p.background[k]);
            px.alpha = 0.f;
            px.ssum = a.softmax_sum0; px.smax = a.p.aggr_rgb_eps;
            px.c0 = rgb_soft ? bg[0] * px.ssum : bg[0];
            px.c1 = rgb_soft ? bg[1] * px.ssum : bg[1];
            px.c2 = rgb_soft ? bg[2] * px.ssum : bg[2];
            px.depth_min = 10000000.f;
            px.face_min = -1;
        }

        // lane = pixel, the face's record in scalar registers (run_dense of render_forward_body): the result of (face, this lane's pixel)
        auto dense_eval = [&](int fn, unsigned long long mask, int tag) __attribute__((always_inline)) -> TeamRes {
            const long face_lin = (long)t.b * a.nf + fn;
            const bool mine = lane_in(mask) && t.valid;
            float r[REC];
            RecPtr rs = uniform_rec_ptr(recs_g + (long)fn * REC);
            load_record<4 * kGatherW0, 4 * kGatherW1>(r, rs);
            Pair q;
            barycentrics(q, r, t.xp, t.yp);
            asm volatile("" : "+s"(rs) : "v"(q.w0), "v"(q.w1), "v"(q.w2));
            load_record<4 * kGatherA0, 4 * kGatherA1>(r, rs);
            TeamRes res;
            res.frag = 0.f; res.z = 0.f; res.c0 = res.c1 = res.c2 = 0.f; res.fnflags = fn << 3;
            bool contributes = false;
            q.frag = 0.f;
            if (mine) contributes = soft_fragment<DIST, SQ>(q, r, t.xp, t.yp, a, dp, false, tag == 1, tag == 2, tag != 0);
            if constexpr (!kSil) {
                asm volatile("" : "+s"(rs) : "v"(q.frag));
                load_record<4 * kGatherB0, REC>(r, rs);
            }
            if (contributes) team_depth_colour<RGB, TEXM>(res, fn, q, r, a, rgb_soft, face_lin);
            return res;
        };
        if (solo) {
            if (!folder) continue;
            // pool exhausted: the exact per-pixel tests for every face of the image (collect_pairs), as for_each_batch's fallback,
            // evaluated and folded in place by the fold wave alone
            const RecPtr recs = (RecPtr)a.records + (long)t.b * a.nf * REC;
            for (int fn = 0; fn < a.nf; fn++) {
                Pair q;
                const unsigned long long m = collect_pairs<REC>(t, recs + (long)fn * REC, q);
                if (m) team_fold<ALPHA, RGB>(px, dense_eval(fn, m, 0), a, alpha_func, rgb_soft);
            }
        } else if (dense_tile) {
            // PIXEL MODE for a team (for_each_batch: entries of kPixelModeAvg pixels and more on average -- no pair list pays): a chunk
            // is eight ENTRIES, B-wave w evaluates entry c + w on all 64 pixels (lane = pixel, the record in scalar registers) into
            // slots [64 w, 64 w + 64) of the chunk's result buffer, and the fold wave folds slots lane, 64 + lane, ... in order while the
            // B-waves evaluate the next eight entries.
            const GENDR_CONST_AS i4v* ents = (const GENDR_CONST_AS i4v*)(a.entries + ti.y);
            const int cnt = ti.z;
            for (int c = 0; c < cnt; c += kTeamB, chunk_ctr++) {
                const int buf = chunk_ctr & 1;
                if (!folder && c + wb < cnt) {
                    const i4v e = ents[c + wb];                          // (face, npix | tag << 8, mask lo, mask hi): scalar load
                    const unsigned long long m = ((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z;
                    s_res[buf][wb * 64 + lane] = dense_eval(e.x, m & my_rows, (e.y >> 8) & 3);
                }
                __syncthreads();
                if (folder) {
                    const int n = min(kTeamB, cnt - c);
                    for (int j0 = 0; j0 < n; j0 += 4) {
                        TeamRes r4[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            r4[j].frag = 0.f; r4[j].z = 0.f; r4[j].c0 = r4[j].c1 = r4[j].c2 = 0.f; r4[j].fnflags = 0;
                            if (j0 + j < n) r4[j] = s_res[buf][(j0 + j) * 64 + lane];
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) team_fold<ALPHA, RGB>(px, r4[j], a, alpha_func, rgb_soft);
                    }
                }
            }
        } else {
            if (wave == 1) s_xy[lane] = make_float2(t.xp, t.yp);
            const int4* ents = reinterpret_cast<const int4*>(a.entries + ti.y);
            const int cnt = ti.z;
            PairHints* hint_base = (hints_q && part_log2 == 0) ? a.hints + ti.y : nullptr;   // (the hints describe the batches of UNCUT tiles)
            int e_next = 0, built = 0, done = 0;                         // entries appended, codes appended, batches evaluated
            // Phase C needs every pixel's pairs of a chunk in list order.  The results of a chunk are therefore stored GROUPED BY PIXEL:
            // result slot = (pairs of the chunk on lower pixels) + (the pixel's pairs in lower batches of the chunk) + (the pixel's pairs on
            // lower lanes of the same batch).  The B-waves announce a chunk's pairs one chunk ahead -- B-wave w adds 1 to byte w of its
            // pairs' pixels' count words -- so that when a chunk is evaluated its count words are complete: every wave turns them into the
            // pixels' first slots by the same scan, and the fold wave walks `total` consecutive slots per pixel: a load, the fold, a
            // counter per step (the first version walked per-batch pair masks: 75 instructions a step, 150 us of a 240-us launch).
            auto announce = [&](int k, int cbuf) __attribute__((always_inline)) {       // B-waves: batch k's pairs into count buffer cbuf
                const int np = min(64, built - (k << 6));
                if (lane < np) {
                    const int code = s_ring[((k << 6) + lane) & (kTeamRing - 1)];
                    __hip_atomic_fetch_add(&s_cnt[cbuf][code & 63], 1ull << (8 * wb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            auto pixel_slots = [&](int cbuf, int& total) __attribute__((always_inline)) -> int {   // lane = pixel: its first slot, its pairs
                const unsigned long long cw = s_cnt[cbuf][lane];
                total = (int)(__builtin_amdgcn_sad_u8((unsigned)cw, 0u, 0u) + __builtin_amdgcn_sad_u8((unsigned)(cw >> 32), 0u, 0u));
                return wave_exclusive_scan_dpp(total);
            };
            for (;;) {
                team_build<kTeamB>(ents, cnt, e_next, built, done * 64, s_ring, wb, 0, 0x7fffffff, my_rows);
                const bool last = e_next >= cnt;
                const int nb = last ? (built - done * 64 + 63) >> 6 : (built - done * 64) >> 6;
                __syncthreads();                                         // the ring (and s_xy) are written
                if (nb > 0) {
                    if (!folder && wb < nb) announce(done + wb, chunk_ctr % 3);
                    __syncthreads();                                     // the first chunk's counts
                }
                for (int c = 0; c < (GENDR_TEAM_ABLATE == 5 ? 0 : nb); c += kTeamB, chunk_ctr++) {
                    const int buf = chunk_ctr & 1, cbuf = chunk_ctr % 3;
                    const int k = done + c + wb;                         // this B-wave's batch of the chunk
                    if (!folder && c + wb < nb) {
                        // ---- phase B: one pair per lane (run_batch of render_forward_body)
                        int total_l;
                        const int first_l = pixel_slots(cbuf, total_l);  // (of pixel `lane`; the pair's pixel fetches its own below)
                        const int np = min(64, built - (k << 6));
                        float hint = 0.f;                                // +0: no gradient
                        const int code = lane < np ? s_ring[((k << 6) + lane) & (kTeamRing - 1)] : 0;
                        const int pixel = code & 63;
                        const int first_p = __shfl(first_l, pixel);     // (all lanes take part: a permute reads nothing from an idle lane)
                        if (lane < np) {
                            // the pair's rank among the batch's pairs of its pixel: the lanes below it in the pixel's private mask
                            __hip_atomic_fetch_or(&s_bmask[wb][pixel], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            __builtin_amdgcn_wave_barrier();
                            const unsigned long long same = s_bmask[wb][pixel];
                            const unsigned long long cw = s_cnt[cbuf][pixel];
                            __builtin_amdgcn_wave_barrier();
                            s_bmask[wb][pixel] = 0ull;
                            // ... plus the pixel's pairs in the batches of B-waves 0 .. wb - 1 (bytes below wb), plus the pixel's first slot
                            const unsigned long long below = wb == 0 ? 0ull : cw & ((1ull << (8 * wb)) - 1ull);
                            const int slot = first_p
                                           + (int)(__builtin_amdgcn_sad_u8((unsigned)below, 0u, 0u) + __builtin_amdgcn_sad_u8((unsigned)(below >> 32), 0u, 0u))
                                           + __popcll(same & ((1ull << lane) - 1ull));
                            const int fn = code >> 6;
                            const long face_lin = (long)t.b * a.nf + fn;
                            float r[REC];
                            const float* rg = recs_g + (long)fn * REC;
                            TeamRes res;
                            res.frag = 0.f; res.z = 0.f; res.c0 = res.c1 = res.c2 = 0.f; res.fnflags = fn << 3;
#if GENDR_TEAM_ABLATE == 4
                            Pair q; q.hint = 0.f; q.frag = 0.f;
                            if (false) {
#else
                            gather_record<kGatherW0, kGatherW1>(r, rg);
                            gather_record<kGatherA0, kGatherA1>(r, rg);
                            const float2 pc = s_xy[pixel];
                            Pair q;
                            barycentrics(q, r, pc.x, pc.y);
#if GENDR_TEAM_ABLATE == 2
                            if (false) {
#else
                            if (soft_fragment<DIST, SQ>(q, r, pc.x, pc.y, a, dp)) {
#endif
#endif
                                hint = q.hint;
                                if constexpr (!kSil) gather_record<kGatherB0, REC / 4>(r, rg);
                                team_depth_colour<RGB, TEXM>(res, fn, q, r, a, rgb_soft, face_lin);
                            }
                            s_res[buf][slot] = res;
                        }
                        if (hint_base && GENDR_TEAM_ABLATE != 3) {
                            // the batch's pair hints leave through the scalar data cache (see render_forward_body)
                            typedef unsigned u4s __attribute__((ext_vector_type(4)));
                            const unsigned long long h_lo = __ballot(hint_bit0(hint));
                            const unsigned long long h_hi = __ballot(hint_bit1(hint));
                            u4s hv; hv.x = (unsigned)h_lo; hv.y = (unsigned)(h_lo >> 32); hv.z = (unsigned)h_hi; hv.w = (unsigned)(h_hi >> 32);
                            PairHints* slot = hint_base + k;
                            asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(hv), "s"(slot) : "memory");
                            if (__ballot(hint_none(hint)) && lane == 0) atomicOr(a.control + (blockIdx.x & 7) * kCtlStride + kCtlHintFlag, 2);
                        }
                        // the next chunk's pairs (of this phase: a later phase's codes are not listed yet) are announced now
                        if (c + kTeamB + wb < nb) announce(k + kTeamB, (chunk_ctr + 1) % 3);
                    }
                    __syncthreads();                                     // chunk c is evaluated; the B-waves go on to chunk c + 1
                    if (folder && GENDR_TEAM_ABLATE != 6) {
                        // ---- phase C: every pixel folds its pairs of the chunk: consecutive slots, ascending list order = ascending
                        // face order; the next result is requested before the current one is folded
                        int total;
                        const int first = pixel_slots(cbuf, total);
                        s_cnt[cbuf][lane] = 0ull;                        // (used again by chunk c + 3: announced after the next barrier)
                        // four results per step: their loads are in flight together, so a step pays one LDS latency for four folds
                        for (int i0 = 0; __any(i0 < total); i0 += 4) {
                            TeamRes r4[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                r4[j].frag = 0.f; r4[j].z = 0.f; r4[j].c0 = r4[j].c1 = r4[j].c2 = 0.f; r4[j].fnflags = 0;   // (no flag: folds nothing)
                                if (i0 + j < total) r4[j] = s_res[buf][first + i0 + j];
                            }
#if GENDR_TEAM_ABLATE != 1
#pragma unroll
                            for (int j = 0; j < 4; j++) team_fold<ALPHA, RGB>(px, r4[j], a, alpha_func, rgb_soft);
#endif
                        }
                    }
                }
                done += nb;
                if (last) break;
            }
        }

        if (!folder) continue;
        // ---- epilogue, kernel.cu:845-861 (the fold wave: lane = pixel)
        if constexpr (kSil) {
            if (t.valid) a.rgba[(long)t.b * P + t.pix] = px.alpha;
            if (a.target) {
                const float tv = t.valid ? a.target[(long)t.b * P + t.pix] : 0.f;
                float s1 = t.valid ? px.alpha * tv : 0.f, s2 = t.valid ? px.alpha * (1.f - tv) : 0.f;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
                if (lane == 0) {
                    if (s1 != 0.f) unsafeAtomicAdd(a.iou_sums + 2 * t.b, s1);
                    if (s2 != 0.f) unsafeAtomicAdd(a.iou_sums + 2 * t.b + 1, s2);
                }
            }
        } else if (t.valid) {
            float* out = a.rgba + (long)t.b * 4 * P + t.pix;
            float* aux = a.aux + (long)t.b * 2 * P + t.pix;
            out[3 * P] = px.alpha;
            if (!rgb_soft) {
                out[0] = (px.face_min != -1) ? px.c0 : bg[0];
                out[P] = (px.face_min != -1) ? px.c1 : bg[1];
                out[2 * P] = (px.face_min != -1) ? px.c2 : bg[2];
                aux[0] = px.depth_min;
                aux[P] = (float)px.face_min;
            } else {
                out[0] = div_f(px.c0, px.ssum);
                out[P] = div_f(px.c1, px.ssum);
                out[2 * P] = div_f(px.c2, px.ssum);
                aux[0] = px.ssum;
                aux[P] = px.smax;
            }
        }
    }
    if (hints_q && !folder) asm volatile("s_waitcnt lgkmcnt(0)\n s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");   // the hints' scalar stores reach the L2
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// register budgets: the specialised option sets run three teams per CU (seven waves per SIMD: 72 registers, as the one-wave kernel _w6);
// the runtime-dispatch kernel of the light distributions x light aggregators two (five waves: 96 registers, as _wl)
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(64 * kTeamFwdWaves) __attribute__((amdgpu_waves_per_eu(GENDR_FWD_WAVES)))
void render_forward_team_kernel(const RenderArgs a)
{
    team_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}
template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(64 * kTeamFwdWaves) __attribute__((amdgpu_waves_per_eu(GENDR_LIGHT_FWD_WAVES)))
void render_forward_team_kernel_wl(const RenderArgs a)
{
    team_forward_body<DIST, ALPHA, RGB, SQ, TEXM>(a);
}

template <int DIST, int ALPHA, int RGB, int SQ, int TEXM>
__global__ __launch_bounds__(64 * kTeamBwdWaves) __attribute__((amdgpu_waves_per_eu(GENDR_BWD_WAVES)))
void render_backward_team_kernel(const RenderArgs a)
{
    constexpr int REC = record_floats(TEXM);
    constexpr int NG = GradSlots<TEXM>::n;
    constexpr int NT = NG > 9 ? NG - 9 : 1;
    __shared__ int s_ring[kTeamRing];
    __shared__ __attribute__((aligned(16))) PixIn s_pix[64];            // the tile's per-pixel inputs, fetched by pair lanes
    __shared__ __attribute__((aligned(8))) FaceSeg s_seg[kTeamBwdWaves][64];
    __shared__ float s_val[kTeamBwdWaves][NG * 65];
    __shared__ rcp_t s_gamma[(DIST == kGamma || DIST == kGammaRev || DIST == -1) ? kGammaSteps : 1];
    __shared__ double s_ntab[(DIST == kGaussian && GENDR_NORMTAB_LDS) ? kNormRows * kNormRow : 1];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const DistParams dp = {a.p.dist_scale, a.p.dist_shape, a.p.dist_shift, GENDR_R_SCALE(a), a.gamma_k0, a.gamma_pdf_c, gamma_table<DIST>(s_gamma, a), norm_table<DIST>(s_ntab)};

    TileWalk tw;
    walk_init(tw, a, 1);
    tw.rank = (int)(blockIdx.x >> 3);
    tw.stride = (int)(gridDim.x >> 3);
    const bool hinted_q = a.hints != nullptr && tw.hint_flag == 1;

    // work items: (tile, part) -- order_tiles_kernel graded the heavy tiles into 8, 4 or 2 parts (TileWalk; team calls: by weight alone)
    for (int item = tw.rank; item < tw.items; item += tw.stride) {
        GENDR_RELOADED_ARGS(a);
        int part_log2, part;
        const int slot_i = walk_item(tw, item, part_log2, part);
        const i4v ti = *(const GENDR_CONST_AS i4v*)(a.tile_info + (tw.qbase + slot_i));
        if (ti.y >= 0 && ti.z == 0) continue;                           // an empty coverage list (no heavy-first copy)
        const bool solo = ti.y < 0;                                     // no slice of the entry pool: wave 0 alone walks all faces
        const bool dense_tile = !solo && tile_in_pixel_mode(ti, 0);     // nearly full entries: lane = pixel, the entries dealt out to the waves
        if (solo && (wave != 0 || part != 0)) continue;                 // (a solo tile is not cut: its first part is all of it)
        TileCtx t;
        tile_setup(t, a, ti.x);
        const float* recs_g = a.records + (long)t.b * a.nf * REC;
        if (wave == 0) s_pix[lane] = load_pixel_inputs<RGB>(a, t.b, t.pix, t.valid, t.xp, t.yp);

        if (solo || dense_tile) {
            // lane = pixel, the ONE face's partials summed by four lanes per component (run_dense of render_backward_body)
            if (dense_tile) __syncthreads(); else __builtin_amdgcn_wave_barrier();       // s_pix is written
            const PixIn px = s_pix[lane];
            auto dense = [&](int fn, unsigned long long mask, int tag) __attribute__((always_inline)) {
                const long face_lin = (long)t.b * a.nf + fn;
                const bool mine = lane_in(mask) && t.valid;
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
                constexpr int PER = NG <= 16 ? 4 : 2;
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
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
                if (PER == 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
                if (k < NG && seg == 0 && v != 0.f) {
                    if (k < 9) unsafeAtomicAdd(a.grad_faces + face_lin * 9 + k, v);
                    else       unsafeAtomicAdd(a.grad_textures + face_lin * (NG - 9) + (k - 9), v);
                }
                __builtin_amdgcn_wave_barrier();
            };
            if (dense_tile) {
                // PIXEL MODE for a team: backward keeps no order, so this part's share of the tile's entries is dealt out to the eight
                // waves, each evaluating its entries on all 64 pixels
                const GENDR_CONST_AS i4v* ents = (const GENDR_CONST_AS i4v*)(a.entries + ti.y);
                const int e_lo = (int)(((long)part * ti.z) >> part_log2), e_hi = (int)(((long)(part + 1) * ti.z) >> part_log2);
                for (int j = e_lo + wave; j < e_hi; j += kTeamBwdWaves) {
                    const i4v e = ents[j];
                    dense(e.x, ((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z, (e.y >> 8) & 3);
                }
                __syncthreads();                                         // every wave has read s_pix
                continue;
            }
            {
                const RecPtr recs = (RecPtr)a.records + (long)t.b * a.nf * REC;
                for (int fn = 0; fn < a.nf; fn++) {
                    Pair q;
                    const unsigned long long m = collect_pairs<REC>(t, recs + (long)fn * REC, q);
                    if (m) dense(fn, m, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }

        const int4* ents = reinterpret_cast<const int4*>(a.entries + ti.y);
        const int cnt = ti.z;
        const PairHints* hint_base = (hinted_q && (part_log2 == 0 || !kTeamFwdRows)) ? a.hints + ti.y : nullptr;   // (row parts of forward: no hints)
        // this part's batches [k_lo, k_hi) of the tile's list (the last part runs to the list's end, whatever the record's pair count says)
        const int nb_tile = (ti.w + 63) >> 6;
        const int k_lo = (int)(((long)part * nb_tile) >> part_log2);
        const int k_hi = part + 1 == (1 << part_log2) ? 0x3fffff : (int)(((long)(part + 1) * nb_tile) >> part_log2);
        int e_next = 0, built = 0, done = k_lo;
        for (;;) {
            team_build<kTeamBwdWaves>(ents, cnt, e_next, built, done * 64, s_ring, wave, k_lo * 64, k_hi * 64);
            const bool last = e_next >= cnt || built >= k_hi * 64;
            const int avail = min(last ? (built + 63) >> 6 : built >> 6, k_hi);       // batches [done, avail) are listed in full
            const int nb = max(avail - done, 0);
            __syncthreads();                                             // the ring (and s_pix) are written
            for (int k = done + wave; k < done + nb; k += kTeamBwdWaves) {
                // ---- one batch, as run_batch of render_backward_body
                const int np = min(64, built - (k << 6));
                bool e0 = false, e1 = false, dead = false;
                const bool hinted = hint_base != nullptr;
                if (hinted) {
                    const GENDR_CONST_AS unsigned long long* hp = (const GENDR_CONST_AS unsigned long long*)(hint_base + k);   // scalar load
                    const unsigned long long h_lo = hp[0], h_hi = hp[1];
                    if ((h_lo & h_hi) == ~0ull) continue;               // no pair of the batch gets a gradient
                    e0 = __builtin_amdgcn_inverse_ballot_w64(~h_lo & ~h_hi);
                    e1 = __builtin_amdgcn_inverse_ballot_w64(h_lo & ~h_hi);
                    dead = __builtin_amdgcn_inverse_ballot_w64(h_lo & h_hi);
                }
                const int code = lane < np ? s_ring[((k << 6) + lane) & (kTeamRing - 1)] : -64;
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
                    const PixIn px = s_pix[code & 63];
                    const int fn = fn_l;
                    const long face_lin = (long)t.b * a.nf + fn;
                    float gv[9];
                    float gt[NT];
                    int tex_own = -1;
                    float tex_val[3] = {0.f, 0.f, 0.f};
                    bool live = false;
                    if (!dead) live = backward_pair<DIST, ALPHA, RGB, SQ, TEXM>(a, dp, recs_g + (long)fn * REC, px, fn, face_lin, gv, gt, tex_own, tex_val, hinted, e0, e1);
                    if constexpr (TEXM == kTexSurfaceN) {
                        if (live && tex_own >= 0) {
#pragma unroll
                            for (int c = 0; c < 3; c++) unsafeAtomicAdd(a.grad_textures + (face_lin * a.T + tex_own) * 3 + c, tex_val[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 9; c++) asm("" : "+v"(gv[c]));
#pragma unroll
                    for (int c = 0; c < NG - 9; c++) asm("" : "+v"(gt[c]));
#pragma unroll
                    for (int c = 0; c < 9; c++) s_val[wave][c * 65 + lane] = live ? gv[c] : 0.f;
#pragma unroll
                    for (int c = 0; c < NG - 9; c++) s_val[wave][(9 + c) * 65 + lane] = live ? gt[c] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
                for (int e = lane; e < nfaces * NG; e += 64) {
                    const int slot = e / NG, c = e - slot * NG;
                    const FaceSeg sg = s_seg[wave][slot];
                    const int n = sg.span >> 8;
                    const float* col = &s_val[wave][c * 65 + (sg.span & 255)];
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                    int i = 0;
                    for (; i + 4 <= n; i += 4) { v0 += col[i]; v1 += col[i + 1]; v2 += col[i + 2]; v3 += col[i + 3]; }
                    for (; i < n; i++) v0 += col[i];
                    const float v = (v0 + v1) + (v2 + v3);
                    if (v != 0.f) {
                        const long face_lin = (long)t.b * a.nf + sg.fn;
                        if (c < 9) unsafeAtomicAdd(a.grad_faces + face_lin * 9 + c, v);
                        else       unsafeAtomicAdd(a.grad_textures + face_lin * (NG - 9) + (c - 9), v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            done += nb;
            __syncthreads();                                             // every wave has read its codes (and, after the last phase, s_pix)
            if (last) break;
        }
    }
}

}  // namespace gendr
