/*
 * gendr_oracle.h -- CPU oracle for the generalized soft rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gendr_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY PINNED (since round 3) to outputs of the reference's own kernels: the
 * reference (Felix-Petersen/gendr) ships no tests, golden vectors or fixtures
 * for this path and its implementation is CUDA, but the DEVICE half of
 * gendr/cuda/generalized_renderer_cuda_kernel.cu compiles for gfx950 as it
 * stands (oracle/build_ref.py: PyTorch-ROCm's translator for the two CUDA
 * includes, clang --cuda-device-only; no reference file edited, nothing
 * supplied in place of anything -> oracle/_ref/*.co).  oracle/ref_gpu.py
 * launches those kernels on the GPU box and tests/test_gpu_reference_pin.py
 * holds this restatement to their outputs: float64 to 1e-9 on the whole option
 * matrix and at the BASELINE configurations (face preprocessing bit for bit),
 * float32 under the element-wise rule of tests/criteria.py.  What the pin
 * cannot cover: nvcc's own contraction choices and CUDA's libm -- the pin build
 * uses -ffp-contract=off and ROCm's device libm.  On the
 * CPU side the restatement is additionally held by closed-form CDF/pdf checks
 * against scipy, finite differences of the fp64 build and a second independent
 * restatement in PyTorch (oracle/torch_ref.py).
 *
 * Two instantiations exist, mirroring AT_DISPATCH_FLOATING_TYPES
 * (kernel.cu:1102,1117,1189): *_f32 follows the float instantiation including
 * its float<->double promotions expression by expression (double literals such
 * as `1.`, `0.5`, `1e-6`, M_PI promote the sub-expression they appear in);
 * *_f64 is the double instantiation.
 */
#ifndef GENDR_ORACLE_H
#define GENDR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the scalar argument list of forward_render / backward_render
 * (generalized_renderer_cuda.cpp:74-96,130-155).  Scalars are `float` exactly
 * as in the reference launcher signature (kernel.cu:1077-1092), also for the
 * double instantiation. */
typedef struct {
    int   image_size;
    int   dist_func;              /* 0..17, kernel.cu:217-239 */
    float dist_scale;
    int   dist_squared;
    float dist_shape;
    float dist_shift;
    float dist_eps;
    int   aggr_alpha_func;        /* 0..9, kernel.cu:462-470 */
    float aggr_alpha_t_conorm_p;
    int   aggr_rgb_func;          /* 0 hard, 1 softmax */
    float aggr_rgb_eps;
    float aggr_rgb_gamma;
    float near_;
    float far_;
    int   double_side;
    int   texture_type;           /* 0 surface, 1 vertex */
    /* Decisions where the reference is undefined (DESIGN.md "reference quirks"):
     * texel_mode 0 = reference-faithful surface texel index (may run into the
     *                following face's texels, kernel.cu:179-184); reads past the
     *                end of the whole tensor are clamped to the face's own
     *                last texel instead of reading the heap;
     *            1 = clamp the texel index to the face's own R x R block. */
    int   texel_mode;
    int   num_threads;            /* OpenMP threads; <=0 -> runtime default */
    /* Sensitivity analysis only (tests/criteria.py): the contribution threshold of kernel.cu:13,:784 is
     * multiplied by this factor; <= 0 means 1 (0.000001 * 1.0 is exact, so the default is the reference's
     * comparison bit for bit).  Elements whose value changes when the threshold moves by a few percent are
     * decided by last-ulp libm differences and are reported separately by the parity tests. */
    float prob_threshold_scale;
} gendr_oracle_opts;

/* scalar functions, float instantiation (kernel.cu:1230-1270) */
float gendr_oracle_sigmoid_forward(int id, float sign, float x, float scale, float shape, float shift);
float gendr_oracle_sigmoid_backward(int id, float sign, float x, float scale, float shape, float shift);
float gendr_oracle_t_conorm_forward(int id, float a_existing, float b_new, int face_id, float p);
float gendr_oracle_t_conorm_backward(int id, float a_all, float b_current, int nf, float p);
/* double instantiation of the same templates */
double gendr_oracle_sigmoid_forward_f64(int id, double sign, double x, double scale, double shape, double shift);
double gendr_oracle_sigmoid_backward_f64(int id, double sign, double x, double scale, double shape, double shift);
double gendr_oracle_t_conorm_forward_f64(int id, double a_existing, double b_new, int face_id, double p);
double gendr_oracle_t_conorm_backward_f64(int id, double a_all, double b_current, int nf, double p);

/* faces [B,nf,9] -> faces_info [B,nf,27] (kernel.cu:620-676).  faces_info must
 * be zero-filled by the caller (functional/renderer.py:136). */
void gendr_oracle_face_info_f32(const float* faces, float* faces_info, int B, int nf);
void gendr_oracle_face_info_f64(const double* faces, double* faces_info, int B, int nf);

/* forward (kernel.cu:680-862).  soft_colors [B,4,is,is] must arrive pre-filled
 * with the background colour in planes 0..2 (functional/renderer.py:144-151);
 * aggrs_info [B,2,is,is] is written. */
void gendr_oracle_forward_f32(const float* faces, const float* textures, const float* faces_info,
                              float* aggrs_info, float* soft_colors,
                              int B, int nf, int T, const gendr_oracle_opts* o);
void gendr_oracle_forward_f64(const double* faces, const double* textures, const double* faces_info,
                              double* aggrs_info, double* soft_colors,
                              int B, int nf, int T, const gendr_oracle_opts* o);

/* backward (kernel.cu:866-1065).  grad_faces [B,nf,9], grad_textures
 * [B,nf,T,3] are overwritten.  Per-pair contributions are computed in the
 * instantiation's type exactly as the reference does, but summed in double
 * (the reference's atomicAdd order is unspecified).  abs_faces / abs_textures
 * (may be NULL) receive sum |contribution| per element, for
 * conditioning-aware tolerances. */
void gendr_oracle_backward_f32(const float* faces, const float* textures, const float* soft_colors,
                               const float* faces_info, const float* aggrs_info,
                               float* grad_faces, float* grad_textures, const float* grad_soft_colors,
                               float* abs_faces, float* abs_textures,
                               int B, int nf, int T, const gendr_oracle_opts* o);
void gendr_oracle_backward_f64(const double* faces, const double* textures, const double* soft_colors,
                               const double* faces_info, const double* aggrs_info,
                               double* grad_faces, double* grad_textures, const double* grad_soft_colors,
                               double* abs_faces, double* abs_textures,
                               int B, int nf, int T, const gendr_oracle_opts* o);

/* number of (pixel, face) pairs that survive all three skip tests in the
 * forward loop (kernel.cu:747,769,784) -- used to size the culling report. */
long long gendr_oracle_count_pairs_f32(const float* faces, const float* faces_info,
                                       int B, int nf, const gendr_oracle_opts* o);

int gendr_oracle_max_threads(void);
/* sensitivity analysis only: +1 / -1 moves every single-precision libm result of the float instantiation one ulp up /
 * down, +-2 ... +-7 up or down by (six different bits of) a hash of the
 * result's bits (neighbouring pairs move against each other), 0 (the default) leaves it alone (gendr_oracle.c).  Process-wide; not to be changed while a call is running. */
void gendr_oracle_set_libm_jitter(int j);

#ifdef __cplusplus
}
#endif
#endif
