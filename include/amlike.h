/* amlike.h -- C ABI of the MI355X acoustic-likelihood engine (libamlike.so).
 *
 * Plain C: opaque handles, raw pointers + sizes + leading dimensions (in
 * floats, what kaldi::MatrixBase::Stride() returns), int status codes.  No
 * exception and no C++/torch type crosses this boundary.  The reference keeps
 * the same shape of boundary under its own C++ matrix classes: extern "C"
 * launchers taking raw device pointers and a {rows, cols, stride} struct
 * (/root/reference/src/cudamatrix/cu-kernels-ansi.h:33-168,
 *  /root/reference/src/cudamatrix/cu-matrixdim.h:47-88).
 *
 * All reference paths below are relative to /root/reference/src.
 */
#ifndef AMLIKE_H_
#define AMLIKE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  The C++ shim layer rethrows them as std::runtime_error with
 * the reference's message (base/kaldi-error.cc:173-185). */
#define AMLIKE_OK 0
#define AMLIKE_EINVAL 1     /* bad argument / handle */
#define AMLIKE_EDIM 2       /* "Dim mismatch: data dim = .. vs. model dim = .."  gmm/decodable-am-diag-gmm.cc:49-52 */
#define AMLIKE_ENONFINITE 3 /* "Invalid answer (overflow or invalid variances/features?)" gmm/diag-gmm.cc:523-524 */
#define AMLIKE_EHIP 4       /* a HIP runtime call failed (CU_SAFE_CALL analogue, cudamatrix/cu-common.h:38-45) */
#define AMLIKE_EOOM 5       /* device allocation failed (cudamatrix/cu-device.cc:640-652) */
#define AMLIKE_ENODEV 6     /* no usable GPU (cudamatrix/cu-device.cc:99-138) */
#define AMLIKE_EBUSY 7      /* both submission slots of the handle are in flight: wait for one first */

typedef struct amlike_gmm amlike_gmm_t;   /* packed AmDiagGmm resident in HBM */
typedef struct amlike_nnet amlike_nnet_t; /* nnet2 forward pipeline resident in HBM */

/* ---- device / process ---------------------------------------------------- */

/* Replaces CuDevice::SelectGpuId (cudamatrix/cu-device.cc:64-138): makes `device` the
 * device on which the calling thread creates handles from now on (and the process's default
 * for threads that never call it).  May be called for several devices, from one thread or
 * from one thread per device: a handle stays on the device it was created on, and every
 * call that takes a handle runs on the handle's device whatever thread makes it.
 * Returns AMLIKE_ENODEV when `device` is not a visible gfx950 device. */
int amlike_init(int device);

/* The device a handle lives on. */
int32_t amlike_gmm_device(const amlike_gmm_t *gmm);
int32_t amlike_nnet_device(const amlike_nnet_t *nnet);

/* Last error text of the calling thread (KALDI_ERR message analogue). */
const char *amlike_last_error(void);

/* Number of visible devices (0 when there is none). */
int amlike_device_count(void);

/* Page-locked host memory for the buffers given to the *_submit calls below (and, if the
 * caller likes, to the synchronous host-pointer calls): copies to and from such memory run at
 * the full rate of the link and overlap with kernels.  The reference copies from ordinary
 * memory (CuMatrix::CopyToMat, cudamatrix/cu-matrix.cc:387-412); this pair has no counterpart
 * there. */
int amlike_host_alloc(size_t bytes, void **ptr);
int amlike_host_free(void *ptr);

/* ---- diagonal-GMM acoustic model ----------------------------------------- */

/* Replaces the in-memory form of AmDiagGmm (gmm/am-diag-gmm.h:36-111: a vector
 * of separately allocated DiagGmm, gmm/diag-gmm.h:231-237).  The caller passes
 * the model flattened in pdf order:
 *   gconsts[G], means_invvars[G x dim], inv_vars[G x dim]  (row-major, dense),
 *   pdf_offsets[num_pdfs + 1]: pdf j owns Gaussians [pdf_offsets[j], pdf_offsets[j+1]).
 * gconsts are the values DiagGmm::ComputeGconsts produces (gmm/diag-gmm.cc:114-152).
 * Host arrays are only read during the call; the handle owns a re-packed device copy.
 * Any dim: the fused kernels cover dim <= 64, larger ones go through the per-Gaussian score
 * product and a log-sum-exp pass (same answers, lower rate). */
int amlike_gmm_create(const float *gconsts, const float *means_invvars,
                      const float *inv_vars, const int32_t *pdf_offsets,
                      int32_t num_pdfs, int32_t num_gauss, int32_t dim,
                      amlike_gmm_t **out);
int amlike_gmm_destroy(amlike_gmm_t *gmm);

/* Replaces DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-152) for num_gauss
 * Gaussians given flat (host arithmetic, same precisions as the reference):
 * gc = log w - 0.5 D log 2pi + sum_d (0.5 log iv - 0.5 miv^2 / iv); +inf -> -inf and
 * counted in *num_bad (may be NULL); NaN -> AMLIKE_ENONFINITE
 * ("not a number in gconst computation"). */
int amlike_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                               const float *inv_vars, int32_t num_gauss, int32_t dim,
                               float *gconsts, int32_t *num_bad);
int32_t amlike_gmm_num_pdfs(const amlike_gmm_t *gmm);
int32_t amlike_gmm_num_gauss(const amlike_gmm_t *gmm);
int32_t amlike_gmm_dim(const amlike_gmm_t *gmm);

/* Replaces the frame x pdf loop of gmm-compute-likes
 * (gmmbin/gmm-compute-likes.cc:71-77 -> AmDiagGmm::LogLikelihood,
 * gmm/am-diag-gmm.h:108-111 -> DiagGmm::LogLikelihood, gmm/diag-gmm.cc:517-526)
 * and the per-(frame,pdf) body of DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased
 * (gmm/decodable-am-diag-gmm.cc:28-72), computed densely:
 *   out[t*ldo + j] = log sum_m exp(gconst_m + means_invvars_m . x_t - 0.5 inv_vars_m . x_t^2)
 * feats: [T x dim] with row stride ld; out: [T x num_pdfs] with row stride ldo.
 * log_sum_exp_prune <= 0: plain LogSumExp(); > 0: terms below max - prune are
 * dropped (matrix/kaldi-vector.cc:766-784).
 * Host-pointer form: copies in, computes, copies out, synchronises.
 * Returns AMLIKE_ENONFINITE if any output is NaN/Inf (the output is still written). */
int amlike_gmm_score(amlike_gmm_t *gmm, const float *feats, int32_t T, int32_t ld,
                     int32_t feat_dim, float log_sum_exp_prune, float *out, int32_t ldo);

/* The same work, not waited for: the copy in, the kernels and the copy out are enqueued and the
 * call returns a ticket; amlike_gmm_score_wait(ticket) blocks until `out` is complete and
 * returns what amlike_gmm_score would have.  A handle takes two submissions at a time
 * (AMLIKE_EBUSY beyond that); they complete in submission order, and the copy out of one
 * overlaps the kernels of the next.  This is what lets a decoder work on utterance k while
 * utterance k+1 is scored (the reference computes likelihoods inside the decoder's loop,
 * gmm/decodable-am-diag-gmm.cc:28-72, so it has no such call).  feats and out must stay valid
 * until the wait returns; page-locked buffers (amlike_host_alloc) make the call truly
 * asynchronous, ordinary memory works but may block in the copies.
 * At most 2^30 output elements (T x num_pdfs) per submission. */
int amlike_gmm_score_submit(amlike_gmm_t *gmm, const float *feats, int32_t T, int32_t ld,
                            int32_t feat_dim, float log_sum_exp_prune, float *out, int32_t ldo,
                            int32_t *ticket);
int amlike_gmm_score_wait(amlike_gmm_t *gmm, int32_t ticket);

/* Device-pointer form: feats/out are device memory, the work is enqueued on
 * `stream` (a hipStream_t, or NULL for the default stream) and NOT waited for.
 * The NaN/Inf flag of the launches made on `stream` is read with amlike_gmm_poll(gmm, stream).
 * A handle keeps one set of working memory (operand images of the features, repair list,
 * partial sums, control and flag words) per stream it is called with, for the 8 streams used
 * last: calls on different streams of one handle may be made from different threads and run on
 * the GPU at the same time; calls on one stream run in order.  (A ninth stream takes over the
 * working memory of the stream unused for longest, after a device-wide synchronisation.) */
int amlike_gmm_score_dev(amlike_gmm_t *gmm, const float *d_feats, int32_t T, int32_t ld,
                         int32_t feat_dim, float log_sum_exp_prune, float *d_out,
                         int32_t ldo, void *stream);

/* amlike_gmm_score_dev for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  Same arguments and results as amlike_gmm_score_dev without pruning
 * (log_sum_exp_prune <= 0), but the scratch memory of the launch (operand images of the
 * features, the repair list, partial sums of split pdfs) comes from an array built into the
 * library, amlike_gmm_noalloc_scratch_bytes() (8 MiB) per GPU, instead of the handle's
 * growable buffers: the call allocates nothing, and returns AMLIKE_EINVAL when T frames of this
 * model need more than that (the swbd tri4 shape: up to about 6 000 frames per call), for models
 * the fused kernels do not cover (dimension above 64), and before amlike_init.  Every call on a
 * GPU uses that GPU's one array: calls that overlap in time on one GPU, from any handle or
 * stream, race; issue them on one stream or order them yourself.  Its NaN/Inf report is kept
 * per handle, not per stream: amlike_gmm_poll(gmm, any stream) returns it. */
int amlike_gmm_score_dev_noalloc(amlike_gmm_t *gmm, const float *d_feats, int32_t T, int32_t ld,
                                 int32_t feat_dim, float *d_out, int32_t ldo, void *stream);
size_t amlike_gmm_noalloc_scratch_bytes(void);

/* Waits for `stream` and returns AMLIKE_OK or AMLIKE_ENONFINITE for the launches made on
 * that stream (and the handle's amlike_gmm_score_dev_noalloc calls) since the previous poll. */
int amlike_gmm_poll(amlike_gmm_t *gmm, void *stream);

/* Replaces DiagGmm::LogLikelihoods(Matrix) (gmm/diag-gmm.cc:546-562) and, row by
 * row, DiagGmm::LogLikelihoods(Vector) (:528-543): per-Gaussian scores
 *   out[t*ldo + g] = gconst_g + means_invvars_g . x_t - 0.5 inv_vars_g . x_t^2
 * for Gaussians g in [pdf_offsets[pdf_begin], pdf_offsets[pdf_end]) of the packed
 * model (a single DiagGmm is the case num_pdfs == 1).  Host pointers. */
int amlike_gmm_gauss_loglikes(amlike_gmm_t *gmm, int32_t pdf_begin, int32_t pdf_end,
                              const float *feats, int32_t T, int32_t ld, int32_t feat_dim,
                              float *out, int32_t ldo);
int amlike_gmm_gauss_loglikes_dev(amlike_gmm_t *gmm, int32_t pdf_begin, int32_t pdf_end,
                                  const float *d_feats, int32_t T, int32_t ld,
                                  int32_t feat_dim, float *d_out, int32_t ldo, void *stream);

/* Replaces DiagGmm::GaussianSelection(Matrix) (gmm/diag-gmm.cc:801-871) for Gaussian mixture
 * `pdf` of the packed model (a single UBM is the case num_pdfs == 1, pdf == 0): for every frame
 * the num_gselect best Gaussians by per-Gaussian log-likelihood, best first, ties towards
 * the larger index; out_idx is [T x num_gselect] int32 (indices within the mixture, -1 where
 * num_gselect exceeds its size); *tot_loglike (may be NULL) = sum over frames of the log-sum
 * of the selected scores, the function's return value in the reference.  Host pointers. */
int amlike_gmm_gselect(amlike_gmm_t *gmm, int32_t pdf, const float *feats, int32_t T, int32_t ld,
                       int32_t feat_dim, int32_t num_gselect, int32_t *out_idx,
                       double *tot_loglike);

/* Name and grid of the kernels the last score launch used (for profiling output). */
const char *amlike_gmm_kernel_info(const amlike_gmm_t *gmm);

/* Profiling hook (CuDevice::AccuProfile analogue, cudamatrix/cu-device.h:84-92): while enabled,
 * every score launch records a HIP event before and after its scoring kernel, on the stream
 * the kernel is launched on (at most 256 launches between two reads).
 * amlike_gmm_kernel_time waits for them and returns their mean duration in milliseconds and
 * their number, then starts over. */
int amlike_gmm_profile_kernel(amlike_gmm_t *gmm, int32_t enable);
int amlike_gmm_kernel_time(amlike_gmm_t *gmm, float *mean_ms, int32_t *launches);

/* Timing hook: runs `iters` back-to-back launches of the scoring kernel on
 * `stream`, bracketed by hipEvents on that stream, and returns the mean kernel
 * time in milliseconds in *ms_per_launch (CuDevice::AccuProfile analogue,
 * cudamatrix/cu-device.h:84-92). */
int amlike_gmm_time_score_dev(amlike_gmm_t *gmm, const float *d_feats, int32_t T, int32_t ld,
                              int32_t feat_dim, float *d_out, int32_t ldo, void *stream,
                              int32_t iters, float *ms_per_launch);

/* ---- feature front end of the GMM systems -------------------------------- */

typedef struct amlike_xform amlike_xform_t;

/* Replaces `splice-feats --left-context=L --right-context=R | transform-feats final.mat`
 * (egs/wsj/s5/steps/decode.sh:84-105; SpliceFrames, feat/feature-functions.cc:391-412: frame t
 * becomes frames t-L .. t+R side by side, edge frames repeated; featbin/transform-feats.cc:92-108:
 * y = M x, or y = M [x; 1] when M has one column more than x has entries).
 * mat is [rows x cols] row-major dense with cols = dim (L+R+1) or that plus one; mat == NULL:
 * splicing only.  At most 128 rows (or spliced columns when mat is NULL). */
int amlike_xform_create(int32_t dim, int32_t left, int32_t right, const float *mat, int32_t rows,
                        int32_t cols, amlike_xform_t **out);
int amlike_xform_destroy(amlike_xform_t *xf);
int32_t amlike_xform_input_dim(const amlike_xform_t *xf);
int32_t amlike_xform_output_dim(const amlike_xform_t *xf);
int32_t amlike_xform_device(const amlike_xform_t *xf);

/* Applies it to one utterance (host pointers, synchronous), followed -- when spk_mat is not
 * NULL -- by a second transform-feats pass with that [rows2 x cols2] matrix, the per-speaker
 * fMLLR step `transform-feats --utt2spk=ark:utt2spk ark:trans.JOB` of the same scripts.
 * out is [T x (rows2, or the first stage's output dimension)] with row stride ldo. */
int amlike_xform_apply(amlike_xform_t *xf, const float *feats, int32_t T, int32_t ld,
                       const float *spk_mat, int32_t rows2, int32_t cols2, float *out, int32_t ldo);

/* Device-pointer, asynchronous form over a batch of utterances laid back to back
 * (utt_rows[u] frames each; edges are repeated per utterance).  d_spk_mats holds matrices of
 * [rows2 x cols2] back to back in device memory and utt_mat[u] (host array, or NULL for
 * "matrix 0 for all") picks utterance u's, -1 meaning none. */
int amlike_xform_apply_dev(amlike_xform_t *xf, const float *d_in, int32_t ld,
                           const int32_t *utt_rows, int32_t num_utts, const float *d_spk_mats,
                           int32_t rows2, int32_t cols2, const int32_t *utt_mat, float *d_out,
                           int32_t ldo, void *stream);

/* amlike_xform_apply followed by amlike_gmm_score without the trip through host memory: the
 * un-spliced features go up, the log-likelihoods come down (host pointers, synchronous). */
int amlike_gmm_score_raw(amlike_gmm_t *gmm, amlike_xform_t *xf, const float *raw_feats, int32_t T,
                         int32_t ld, const float *spk_mat, int32_t rows2, int32_t cols2,
                         float log_sum_exp_prune, float *out, int32_t ldo);

/* ---- nnet2 forward -------------------------------------------------------- */

/* Component kinds of the nnet2 forward path
 * (nnet2/nnet-component.cc: Splice :2411-2471, FixedAffine :3028-3034,
 * Affine / AffineComponentPreconditioned(Online) :1123-1131, Pnorm :530-535,
 * Normalize :578-592, Softmax :905-918, SumGroup :2352-2357, the elementwise
 * Sigmoid :639, Tanh :673, RectifiedLinear :782, SoftHinge :810-816, Power :731-737,
 * Scale :844-850, FixedScale :3103-3109, FixedBias :3171-3177, and Maxout :443-454,
 * Permute :2262-2270, Dct :2802-2830, FixedLinear :2948-2953, SpliceMax :2623-2654). */
enum amlike_component_kind {
  AMLIKE_SPLICE = 1,
  AMLIKE_AFFINE = 2, /* also FixedAffine, and FixedLinear / a BlockAffine written out densely with
                        bias == NULL or the full bias: out = in * W^T + b, W is [out_dim x in_dim] */
  AMLIKE_PNORM = 3,
  AMLIKE_NORMALIZE = 4,
  AMLIKE_SOFTMAX = 5,
  AMLIKE_SUMGROUP = 6,
  AMLIKE_SIGMOID = 7,
  AMLIKE_TANH = 8,
  AMLIKE_RELU = 9,
  AMLIKE_SOFTHINGE = 10,   /* x > 10 ? x : log1p(exp(x))  (matrix/kaldi-matrix.cc:2434-2448) */
  AMLIKE_POWER = 11,       /* |x|^p  (matrix/kaldi-vector.cc:496-525, include_sign = false) */
  AMLIKE_SCALE = 12,       /* p * x */
  AMLIKE_FIXED_SCALE = 13, /* x[c] * bias[c]: `bias` holds the <Scales> vector */
  AMLIKE_FIXED_BIAS = 14,  /* x[c] + bias[c] */
  AMLIKE_MAXOUT = 15,      /* max over consecutive groups of input_dim / output_dim columns, not below -1e20 */
  AMLIKE_PERMUTE = 16,     /* out[sizes[i]] = in[i]: `sizes` holds <Reorder>, a permutation of 0..dim-1 */
  AMLIKE_DCT = 17,         /* per block of const_dim columns a DCT keeping the first output_dim /
                              (input_dim / const_dim) coefficients; left_context != 0: <Reorder> true */
  AMLIKE_SPLICE_MAX = 18   /* per column the largest value over frames t - left_context .. t + right_context
                              (SpliceMaxComponent :2623-2654); counts towards the net's context like SPLICE */
};

typedef struct amlike_component {
  int32_t kind;          /* enum amlike_component_kind */
  int32_t input_dim;
  int32_t output_dim;
  int32_t left_context;  /* SPLICE, SPLICE_MAX; DCT: the reorder flag */
  int32_t right_context; /* SPLICE, SPLICE_MAX */
  int32_t const_dim;     /* SPLICE: const_component_dim_; DCT: dct_dim */
  float p;               /* PNORM, POWER: the exponent; SCALE: the factor */
  const float *linear;   /* AFFINE: [output_dim x input_dim] row-major, dense */
  const float *bias;     /* AFFINE: [output_dim] or NULL (no offset); FIXED_SCALE, FIXED_BIAS: [input_dim] */
  const int32_t *sizes;  /* SUMGROUP: [output_dim] group sizes; PERMUTE: [input_dim] */
} amlike_component_t;

/* Replaces Nnet (nnet2/nnet-nnet.h:62-) + AmNnet::Priors (nnet2/am-nnet.h) as a
 * device-resident forward pipeline.  priors may be NULL (then num_priors = 0). */
int amlike_nnet_create(const amlike_component_t *components, int32_t num_components,
                       const float *priors, int32_t num_priors, amlike_nnet_t **out);
int amlike_nnet_destroy(amlike_nnet_t *nnet);
int32_t amlike_nnet_input_dim(const amlike_nnet_t *nnet);
int32_t amlike_nnet_output_dim(const amlike_nnet_t *nnet);
int32_t amlike_nnet_left_context(const amlike_nnet_t *nnet);  /* Nnet::LeftContext nnet2/nnet-nnet.cc:39-45 */
int32_t amlike_nnet_right_context(const amlike_nnet_t *nnet); /* Nnet::RightContext :47-53 */

/* Output post-processing of amlike_nnet_compute. */
#define AMLIKE_NNET_POSTERIORS 0 /* NnetComputation output as is (nnet2/nnet-compute.cc:158-165) */
#define AMLIKE_NNET_LOG 1        /* + ApplyFloor(1e-20), ApplyLog (nnet2bin/nnet-am-compute.cc:93-96) */
#define AMLIKE_NNET_LOGLIKES 2   /* + minus log prior, times prob_scale (nnet2/decodable-am-nnet.h:59-69) */

/* Replaces NnetComputation(nnet, feats, pad_input, &out) (nnet2/nnet-compute.h:44-47,
 * nnet2/nnet-compute.cc:62-106,158-165).  feats [T x input_dim] stride ld;
 * out [T_out x output_dim] stride ldo with T_out = T if pad_input else
 * T - left_context - right_context.  Host pointers, synchronous. */
int amlike_nnet_compute(amlike_nnet_t *nnet, const float *feats, int32_t T, int32_t ld,
                        int32_t pad_input, int32_t mode, float prob_scale, float *out,
                        int32_t ldo);
/* Submit / wait form of amlike_nnet_compute, with the rules of amlike_gmm_score_submit. */
int amlike_nnet_compute_submit(amlike_nnet_t *nnet, const float *feats, int32_t T, int32_t ld,
                               int32_t pad_input, int32_t mode, float prob_scale, float *out,
                               int32_t ldo, int32_t *ticket);
int amlike_nnet_compute_wait(amlike_nnet_t *nnet, int32_t ticket);
/* Device-pointer, asynchronous form. */
int amlike_nnet_compute_dev(amlike_nnet_t *nnet, const float *d_feats, int32_t T, int32_t ld,
                            int32_t pad_input, int32_t mode, float prob_scale, float *d_out,
                            int32_t ldo, void *stream);

/* The same for a batch of utterances laid out back to back in d_feats
 * (utt_rows[u] input frames each; outputs back to back in d_out): every kernel of the
 * pipeline runs once over all frames, while the edge padding and the frame splicing
 * respect the utterance boundaries exactly as NnetComputer does per utterance
 * (nnet2/nnet-compute.cc:74-86).  *total_out_rows (may be NULL) = rows written. */
int amlike_nnet_compute_batch_dev(amlike_nnet_t *nnet, const float *d_feats, int32_t ld,
                                  const int32_t *utt_rows, int32_t num_utts, int32_t pad_input,
                                  int32_t mode, float prob_scale, float *d_out, int32_t ldo,
                                  void *stream, int64_t *total_out_rows);

/* Per-kernel timing of the last amlike_nnet_compute* call pattern: runs `iters`
 * forward passes and returns mean milliseconds per pass. */
int amlike_nnet_time_compute_dev(amlike_nnet_t *nnet, const float *d_feats, int32_t T,
                                 int32_t ld, int32_t pad_input, int32_t mode, float prob_scale,
                                 float *d_out, int32_t ldo, void *stream, int32_t iters,
                                 float *ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* AMLIKE_H_ */
