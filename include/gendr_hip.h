/*
 * gendr_hip.h -- C ABI of libgendr_hip.so, the MI355X (gfx950) generalized soft
 * rasterizer.  This is the drop-in boundary for the one hot path of
 * Felix-Petersen/gendr: what the reference binds through pybind11 in
 * gendr/cuda/generalized_renderer_cuda.cpp:230-237 (module
 * `gendr.cuda.generalized_renderer`).  Plain pointers and sizes, no torch
 * types.  All device pointers are fp32, contiguous, and owned by the caller;
 * nothing is allocated inside, every call is asynchronous on `stream`
 * (a hipStream_t passed as void*), there is no global state.
 *
 * Return value of every int function: 0 on success, a negative GENDR_E_* code
 * otherwise (never print-and-continue as kernel.cu:1111-1113 does).
 */
#ifndef GENDR_HIP_H
#define GENDR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GENDR_ABI_VERSION 7

enum {
    GENDR_OK              = 0,
    GENDR_E_NULL          = -1,  /* a required pointer is NULL */
    GENDR_E_SHAPE         = -2,  /* B, nf, T or image_size out of range */
    GENDR_E_DIST_FUNC     = -3,  /* dist_func not in 0..17 (kernel.cu:361) */
    GENDR_E_ALPHA_FUNC    = -4,  /* aggr_alpha_func not in 0..9 (kernel.cu:561) */
    GENDR_E_RGB_FUNC      = -5,  /* aggr_rgb_func not in {0,1} */
    GENDR_E_TEXTURE_TYPE  = -6,  /* texture_type not in {0,1}, or T inconsistent with it */
    GENDR_E_DIST_PARAM    = -7,  /* dist_scale < 0, dist_eps < 1 (functional/renderer.py:96,101), gamma shape < 0 (kernel.cu:296) */
    GENDR_E_TCONORM_PARAM = -8,  /* invalid t-conorm p (kernel.cu:491,501,512,522,534,552) */
    GENDR_E_LAUNCH        = -9,  /* hipGetLastError() != hipSuccess after a launch */
    GENDR_E_WORKSPACE     = -10  /* workspace buffer missing */
};

/* Scalar options of forward_render / backward_render, in the reference's order
 * (generalized_renderer_cuda.cpp:80-96, kernel.cu:1077-1092).  `near`/`far`
 * carry a trailing underscore only because <windows.h>-style macros exist. */
typedef struct gendr_params {
    int   image_size;              /* internal size (already doubled for anti-aliasing, gendr/renderer.py:68) */
    int   dist_func;               /* 0..17 */
    float dist_scale;              /* tau */
    int   dist_squared;
    float dist_shape;
    float dist_shift;
    float dist_eps;
    int   aggr_alpha_func;         /* 0..9 */
    float aggr_alpha_t_conorm_p;
    int   aggr_rgb_func;           /* 0 hard, 1 softmax */
    float aggr_rgb_eps;
    float aggr_rgb_gamma;
    float near_;
    float far_;
    int   double_side;
    int   texture_type;            /* 0 surface (T = R*R texels), 1 vertex (T = 3) */
    /* ---- additions of this ABI (no reference counterpart) ---- */
    float background[3];           /* used when background_from_buffer == 0 */
    int   background_from_buffer;  /* 1: rgba planes 0..2 arrive pre-filled with the background,
                                      as functional/renderer.py:144-151 hands them to forward_render */
    int   texel_mode;              /* 0: reference-faithful surface texel index (kernel.cu:179-184 may
                                         index the following face's texels); 1: clamp to the face's own block */
    int   cull;                    /* 1: exact tile culling (default), 0: visit every (pixel, face) pair */
    void* clear_ptr;               /* optional, gendr_forward / gendr_face_setup only: a float buffer the setup stage (its   */
    unsigned long long clear_floats;   /* binning kernel) zero-fills on the way (16-byte aligned, a multiple of 4 floats) -- the
                                      gradient buffers of the coming gendr_backward call, which saves that call's
                                      caller a fill launch.  NULL / 0: nothing is cleared.  Honoured by the float32
                                      entry points (gendr_forward, gendr_silhouette_forward, gendr_face_setup) only. */
    /* ---- ABI 5 ---- */
    int   deterministic;           /* 1: gendr_backward / gendr_silhouette_backward sum every face's gradient in a fixed
                                      order (one wavefront per face, no atomics): two calls on the same inputs return
                                      bit-identical gradients.  The reference's atomicAdd order is not fixed
                                      (kernel.cu:1054-1063; experiments/train_reconstruction.py:582-586 warns about it).
                                      Slower; 0 (default): hardware fp32 atomics, one per (tile batch, face, component).
                                      Float32 entry points only: gendr_backward_f64 (the reference's double instantiation,
                                      one fp64 atomic per pair and component like kernel.cu:1054-1063) IGNORES this field. */
    int   skip_unlisted_aux;       /* 1: gendr_forward does not write `aggrs_info` for the 8x8 tiles no face reaches (their
                                      RGBA is still written).  gendr_backward never reads those pixels' aggrs_info, so the
                                      autograd path sets it; a caller that hands aggrs_info out (forward_render of
                                      generalized_renderer_cuda.cpp:74-150) leaves it 0. */
    unsigned long long pool_entries_max;   /* 0: automatic.  Otherwise an upper limit on the entries of the coverage pool
                                      (16 bytes each) inside the workspace: tiles that find the pool exhausted are
                                      rendered exactly all the same by the slower all-faces walk.  gendr_workspace_bytes
                                      honours it. */
    /* ---- ABI 6 ---- */
    int   pair_hints;              /* Pair hints: gendr_forward records, two bits per evaluated (pixel, face) pair, which edge
                                      of the face the closest-point search selected (kernel.cu:76-165) and whether the pair
                                      passed the skip tests (:769, :784); gendr_backward then evaluates that one edge instead
                                      of repeating the search -- same float operations for that edge, bit-identical values.
                                      0 (default): on when the option set's cull radius is at most 16 pixels (with long
                                      tails the backward call is not bound by its arithmetic and the forward call's extra
                                      work does not pay: measured at BASELINE config 4); 1: on; -1: off.  The hints live in
                                      the workspace (16 bytes per coverage-pool entry; gendr_workspace_bytes honours the
                                      setting) and are tied to the gendr_forward call that filled it: gendr_face_setup
                                      alone leaves none and gendr_backward then repeats the search. */
    int   loose_faces;             /* Faces whose cull box is loose -- the error bound of the exact culling dwarfs the cull radius: the
                                      determinant is clamped or tiny, the face is seen edge-on -- would be listed, with every pixel,
                                      in every tile of their image.  With this on, the binning stage of gendr_forward /
                                      gendr_face_setup evaluates such a face on the pixels of its image (the render kernels' own pair
                                      functions) and lists it only in the tiles that hold a pixel which can contribute at all.
                                      Results are the same either way.  0 (default) and 1: on (since round 4 it is part of the coverage
                                      kernel and costs nothing where no face is flagged; in images of 1024^2 and more a launch of its
                                      own first narrows the boxes of such faces, per-image lists in the workspace); -1: off;
                                      2: on, with that launch and its lists at EVERY image size (what 1024^2 and more take by default:
                                      lets small test images exercise it). */
    /* ---- ABI 7 ---- */
    int   team;                    /* Team kernels: where few tiles each hold thousands of (pixel, face) pairs -- small images, few
                                      views, a distribution whose tail spans pixels: the reference's experiments/opt_shape.py
                                      renders 24 views at 64^2 with a 4-pixel logistic tail -- one workgroup of nine (forward) or
                                      eight (backward) wavefronts renders a tile: the tile's pair list is built once, in shared
                                      LDS, its batches of 64 pairs are evaluated by eight wavefronts side by side, and (forward)
                                      a ninth folds their results per pixel in the reference's order.  Results do not depend on
                                      it (forward bit for bit).  0 (default): on for the option sets that have a team kernel
                                      (logistic / probabilistic, surface texture with T = 1, and its alpha-only twin) when the
                                      call holds at most 4096 tiles (8192 with a cull radius of 2 pixels and more); 1: on wherever a
                                      team kernel exists; -1: off.  gendr_forward and gendr_backward must be called with the
                                      same setting (the pair hints depend on it); ignored with `deterministic`.
                                      The COVERAGE kernel has a team form of its own (one 8-wave workgroup per listed tile, the same
                                      entries in the same slots), chosen for calls of up to 8192 tiles whose tiles can expect to
                                      list 32 faces and more, whatever the option set: -1 switches it off as well, 2 = 1 with that
                                      form at every shape (tests). */
} gendr_params;

/* Bytes of the caller-owned workspace that gendr_face_setup / gendr_forward fill and gendr_backward
 * reads: per-face bin records and face records (this build's replacement for `faces_info`), the
 * per-tile face masks, tile queues and queue records of the exact culling, the pool of per-tile
 * coverage entries (face, pixel mask) both render kernels walk, and (up to 2^19 tiles) the copy of the
 * queue records sorted heaviest tile first.  Depends on the option set (the pool grows with the cull
 * radius).  0 on invalid arguments. */
unsigned long long gendr_workspace_bytes(int B, int nf, int T, const gendr_params* p);

/* Validates the option set exactly as the reference's asserts / device checks do. */
int gendr_validate(const gendr_params* p, int B, int nf, int T);

/* 1 if gendr_forward / gendr_backward (silhouette != 0: gendr_silhouette_forward / _backward) render this call with the team
 * kernels (gendr_params::team), else 0.  A host-side decision on the shapes and the option set only; for tests and reports. */
int gendr_uses_team(int B, int nf, int T, const gendr_params* p, int silhouette);
/* 1 if the coverage kernel of this call (gendr_face_setup, gendr_forward, gendr_silhouette_forward) runs in its team form. */
int gendr_uses_team_cover(int B, int nf, int T, const gendr_params* p);

/* Per-face preprocessing into this build's record layout (replaces forward_render_inv_cuda_kernel,
 * kernel.cu:620-676, launched at :1100-1109) followed by the tile binning of the exact culling.
 * gendr_forward() runs it itself; it is exported for callers that hold only `faces` when they reach
 * backward (the pybind-shaped backward_render).
 *   workspace [gendr_workspace_bytes()] out, 256-byte aligned */
int gendr_face_setup(const float* faces, const float* textures, void* workspace,
                     int B, int nf, int T, const gendr_params* p, void* stream);

/* replaces forward_render (generalized_renderer_cuda.cpp:74-127 -> kernel.cu:1071-1152).
 *   faces        [B,nf,9]    in   (x,y,z per vertex, NDC)
 *   textures     [B,nf,T,3]  in
 *   rgba         [B,4,is,is] out  (in/out when background_from_buffer)   = `soft_colors`
 *   aggrs_info   [B,2,is,is] out  (softmax_sum, softmax_max) or (depth_min, face_index_min)
 *   workspace    [gendr_workspace_bytes()] out, kept by the caller for backward */
int gendr_forward(const float* faces, const float* textures, float* rgba, float* aggrs_info,
                  void* workspace, int B, int nf, int T, const gendr_params* p, void* stream);

/* replaces backward_render (generalized_renderer_cuda.cpp:130-192 -> kernel.cu:1155-1227).
 *   grad_faces [B,nf,9] and grad_textures [B,nf,T,3] must be zero-filled by the
 *   caller (functional/renderer.py:191-196); gradients are accumulated into them.
 *   workspace: as left by gendr_forward / gendr_face_setup (read only). */
int gendr_backward(const float* faces, const float* textures, const float* rgba, const float* aggrs_info,
                   const void* workspace, const float* grad_rgba,
                   float* grad_faces, float* grad_textures,
                   int B, int nf, int T, const gendr_params* p, void* stream);

/* ---- alpha-only rendering with the silhouette loss fused into the epilogue -------------------------
 * The experiment scripts consume only the alpha channel (opt_shape.py:257,296-303; train_reconstruction.py:41-46).
 * These two calls render / differentiate that channel alone: no RGB or aggrs_info planes are written or read (4 B per
 * pixel out instead of 24, 8 B per pixel in instead of 40), no depth / colour / softmax work per pair, no textures.
 *   alpha      [B,is,is]  out / in   == channel 3 of gendr_forward's rgba, bit for bit
 *   target     [B,is,is]  in, optional: target silhouettes; then iou_sums [B,2] receives per view
 *              sum(alpha * target) and sum(alpha * (1 - target)), from which intersect = sums[0] and
 *              union = sum(target) + sums[1] of iou_loss (opt_shape.py:20-24, train_reconstruction.py:30-36)
 *   backward:  either grad_alpha [B,is,is] (d loss / d alpha), or target + grad_iou [B,2] (d loss / d iou_sums): then
 *              the per-pixel gradient grad_iou[b,0] * t + grad_iou[b,1] * (1 - t) is formed on the fly.
 *   grad_faces [B,nf,9] zero-filled by the caller; gradients are accumulated.  texture_type must be 0. */
unsigned long long gendr_silhouette_workspace_bytes(int B, int nf, const gendr_params* p);
int gendr_silhouette_forward(const float* faces, float* alpha, void* workspace, const float* target, float* iou_sums,
                             int B, int nf, const gendr_params* p, void* stream);
int gendr_silhouette_backward(const float* alpha, const void* workspace, const float* grad_alpha,
                              const float* target, const float* grad_iou, float* grad_faces,
                              int B, int nf, const gendr_params* p, void* stream);

/* float64 instantiation of the two render calls: the reference dispatches its kernels over AT_DISPATCH_FLOATING_TYPES
 * (kernel.cu:1102,1117,1189), so float64 tensors are computed in double.  Same arguments as gendr_forward /
 * gendr_backward with double buffers; the workspace (gendr_workspace_bytes_f64) holds faces_info [B,nf,27] in double.
 * A correctness path (one lane per pixel, the reference's own skip tests, fp64 atomics), not the tuned one. */
unsigned long long gendr_workspace_bytes_f64(int B, int nf, int T, const gendr_params* p);
int gendr_forward_f64(const double* faces, const double* textures, double* rgba, double* aggrs_info,
                      void* workspace, int B, int nf, int T, const gendr_params* p, void* stream);
int gendr_backward_f64(const double* faces, const double* textures, const double* rgba, const double* aggrs_info,
                       const void* workspace, const double* grad_rgba, double* grad_faces, double* grad_textures,
                       int B, int nf, int T, const gendr_params* p, void* stream);

/* The reference's per-face preprocessing in its own layout, faces_info [B,nf,27] =
 * inv[9], sym[9], obt[3], 0[6] (kernel.cu:620-676, functional/renderer.py:139), for
 * callers of the pybind-shaped forward_render that inspect faces_info. */
int gendr_face_info(const float* faces, float* faces_info, int B, int nf, void* stream);

/* replace sigmoid_forward/backward, t_conorm_forward/backward
 * (generalized_renderer_cuda.cpp:195-227,233-236; kernel.cu:1230-1270): host-callable scalars. */
float gendr_sigmoid_forward(int function_id, float sign, float x, float scale, float dist_shape, float dist_shift);
float gendr_sigmoid_backward(int function_id, float sign, float x, float scale, float dist_shape, float dist_shift);
float gendr_t_conorm_forward(int t_conorm_id, float a_existing, float b_new, int face_id, float t_conorm_p);
float gendr_t_conorm_backward(int t_conorm_id, float a_all, float b_current, int number_of_faces, float t_conorm_p);

/* Distance (in NDC units) beyond which an outside pixel provably contributes nothing
 * (D <= 1e-6, kernel.cu:784, or d^2 >= dist_eps*tau, kernel.cu:769); +inf if no such
 * distance exists for the option set.  Used by the face-setup kernel. */
float gendr_cull_radius(const gendr_params* p);

/* ---- the step right before the hot path, fused ------------------------------------
 * Replaces the tensor passes of look_at (gendr/functional/look_at.py:59-67: subtract eye, rotate),
 * perspective / orthogonal (gendr/transform.py:14-47) and the face gather
 * (gendr/functional/face_vertices.py:24-27) by one kernel each way.
 *   vertices      [B,nv,3]           in
 *   face_index    [B or 1,nf,3] i32  in   (index_batched = 1 if it has a batch axis of size B)
 *   camera        [B,12]             in   rotation rows x_axis, y_axis, z_axis (look_at.py:52-59), then eye
 *   face_vertices [B,nf,9]           out
 *   width_or_scale: tan(viewing_angle) if perspective (transform.py:21-23), else the orthogonal scale.
 * Backward accumulates into zero-filled grad_vertices [B,nv,3] and (optional, may be NULL) grad_camera [B,12]. */
/* ---- per-face texel blocks <-> texture atlas ---------------------------------------
 * gendr_load_textures replaces load_textures (gendr/cuda/load_textures_cuda.cpp, load_textures_cuda_kernel.cu:14-104):
 *   image [H,W,3] (already flipped vertically by the caller, functional/load_obj.py:104), face_uv [nf,3,2] in [0,1],
 *   is_update [nf] i32, textures [nf,R*R,3] in/out (faces with is_update == 0 keep their texels).
 * gendr_create_texture_image replaces create_texture_image (create_texture_image_cuda_kernel.cu:16-112):
 *   face_uv [nf,3,2] in atlas pixels, textures [nf,R_in*R_in,3], image [rows,cols,3] in/out (pixels of tiles
 *   without a face keep their value), cols = tile_width * texture_res_out, eps as in functional/save_obj.py:31. */
int gendr_load_textures(const float* image, const float* face_uv, const int* is_update, float* textures,
                        int nf, int texture_res, int image_height, int image_width, void* stream);
int gendr_create_texture_image(const float* face_uv, const float* textures, float* image, int nf, int texture_res_in,
                               int image_rows, int image_cols, int tile_width, float eps, void* stream);

/* ---- mesh -> occupancy grid --------------------------------------------------------
 * Replaces voxelize_sub1 (x3) / sub2 / sub3 / sub4 and the host loop around sub4
 * (gendr/cuda/voxelization_cuda.cpp:20-89, voxelization_cuda_kernel.cu:36-194, functional/voxelization.py:11-62).
 *   faces  [B,nf,9]          in   face vertices already in voxel units (voxelization.py:51-52: faces *= size)
 *   voxels [B,vs,vs,vs] i32  out  1 = surface or enclosed, 0 = reachable from the grid boundary (1 - visible)
 *   workspace: gendr_voxelize_workspace_bytes(B, vs) bytes (0 for vs <= 64; may then be NULL). */
size_t gendr_voxelize_workspace_bytes(int B, int voxel_size);
int gendr_voxelize(const float* faces, int* voxels, void* workspace, int B, int nf, int voxel_size, void* stream);

/* Lighting of surface textures (gendr/lighting.py:48-71 with functional/lighting.py:11-48 and Mesh.surface_normals,
 * gendr/mesh.py:109-117): out[b,f,t,:] = textures[b,f,t,:] * (ambient_intensity * ambient_color
 *     + sum_i intensity[i] * (color[i] * relu(<n_f, direction[i]>))),  n_f = normalize(cross(v2 - v1, v0 - v1), eps 1e-6).
 *   vertices [B,nv,3], face_index [B or 1,nf,3] i32, textures / out [B,nf,T,3].
 * Backward: grad_textures [B,nf,T,3] is written, grad_vertices [B,nv,3] (zero-filled) accumulated; either may be NULL. */
#define GENDR_MAX_DIRECTIONAL 4
typedef struct {
    float ambient_intensity;
    float ambient_color[3];
    int   n_directional;                                  /* 0..GENDR_MAX_DIRECTIONAL */
    float intensity[GENDR_MAX_DIRECTIONAL];
    float color[GENDR_MAX_DIRECTIONAL][3];
    float direction[GENDR_MAX_DIRECTIONAL][3];
} gendr_light_params;
int gendr_light_faces(const float* vertices, const int* face_index, const float* textures, float* out,
                      int B, int nv, int nf, int T, int index_batched, const gendr_light_params* lp, void* stream);
int gendr_light_faces_backward(const float* vertices, const int* face_index, const float* textures, const float* grad_out,
                               float* grad_textures, float* grad_vertices,
                               int B, int nv, int nf, int T, int index_batched, const gendr_light_params* lp, void* stream);

/* camera [B,12] from eye / target / up [B,3] each (look_at.py:52-59: z = normalize(at - eye), x = normalize(up x z),
 * y = normalize(z x x), F.normalize eps 1e-5; look.py: z = normalize(direction) when target_is_direction), and its
 * hand-derived backward (grad_eye / grad_target / grad_up may each be NULL). */
int gendr_camera_rotation(const float* eye, const float* target, const float* up, float* camera, int B,
                          int target_is_direction, void* stream);
int gendr_camera_rotation_backward(const float* eye, const float* target, const float* up, const float* grad_camera,
                                   float* grad_eye, float* grad_target, float* grad_up, int B, int target_is_direction,
                                   void* stream);
int gendr_project_faces(const float* vertices, const int* face_index, const float* camera, float* face_vertices,
                        int B, int nv, int nf, int index_batched, int perspective, float width_or_scale, void* stream);
int gendr_project_faces_backward(const float* vertices, const int* face_index, const float* camera,
                                 const float* grad_face_vertices, float* grad_vertices, float* grad_camera,
                                 int B, int nv, int nf, int index_batched, int perspective, float width_or_scale, void* stream);

/* Self-test of the short correctly-rounded forms the pair math uses for sqrtf(x) and 1.f / x (gendr_math.h: sqrt_rn,
 * rcp_rn): compares them with the compiler's IEEE expansions for EVERY float bit pattern in [2^-96, 2^96].
 *   what: 0 sqrt, 1 reciprocal of +x, 2 reciprocal of -x; 3 / 4: the normal CDF of the gaussian distribution at +u / -u in the
 *   TABLE form the kernels specialised for the gaussian evaluate (gendr_math.h: norm_cdf_tab -- e^(-u^2/2) g(u) in double from
 *   tables in LDS; round 6), 5 / 6: the same in the polynomial form of the runtime-dispatch kernels (norm_cdf: a degree-30
 *   polynomial) -- each against the library's normcdf(double) rounded to float, what the reference's kernel.cu:293 computes when
 *   compiled for this platform, for every float u in [0, 6].
 *   report16 (device, 16 x u64): [0] mismatches, [1] values tested, [2..13] offending bit patterns, [14] largest difference
 *   in units of the last place. */
int gendr_selftest(int what, unsigned long long* report16, void* stream);

const char* gendr_error_string(int code);
int gendr_abi_version(void);
int gendr_params_size(void);   /* sizeof(gendr_params) as compiled, for binding sanity checks */

#ifdef __cplusplus
}
#endif
#endif
