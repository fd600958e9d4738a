// The front end of the GMM systems on the device: frame splicing followed by one or two
// affine transforms, i.e. `splice-feats | transform-feats final.mat | transform-feats
// --utt2spk=... ark:trans` of the reference's decoding scripts
// (/root/reference/egs/wsj/s5/steps/decode.sh:84-105), so that the narrow un-spliced
// features are what crosses PCIe.  Reference arithmetic:
//   SpliceFrames            /root/reference/src/feat/feature-functions.cc:391-412
//   transform-feats (body)  /root/reference/src/featbin/transform-feats.cc:92-108
#include <algorithm>
#include <mutex>
#include <vector>

#include "amlike_common.h"

namespace amlike {

constexpr int kXformFrames = 64;   // frames per block
constexpr int kXformMaxRows = 128; // rows of either matrix

// One block = 64 consecutive frames of the batch.  Thread (f = tid & 63, q = tid >> 6)
// computes outputs j = q, q + 4, ... of frame f, so j is uniform over a wave and the matrix
// element is one broadcast load.  Stage one reads the spliced row straight from the input
// (rows clamped to the frame's own utterance); its result goes through LDS to stage two.
__global__ __launch_bounds__(256) void feat_splice_affine_kernel(
    const float *__restrict__ in, int ld, int dim, int left, int nsplice,
    const int32_t *__restrict__ utt_off,  // [num_utts + 1] first frame of each utterance
    int num_utts, const float *__restrict__ m1, int rows1, int cols1,
    const float *__restrict__ m2s, int rows2, int cols2, const int32_t *__restrict__ utt_mat,
    float *__restrict__ out, int ldo, int total) {
  __shared__ float y1[kXformFrames][kXformMaxRows + 1];
  const int f = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int g = blockIdx.x * kXformFrames + f;
  const bool live = g < total;
  int u = 0;
  if (live) {  // the utterance frame g belongs to
    int lo = 0, hi = num_utts;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (utt_off[mid] <= g) lo = mid; else hi = mid;
    }
    u = lo;
  }
  const int u0 = live ? utt_off[u] : 0, u1 = live ? utt_off[u + 1] : 1;
  const int sdim = nsplice * dim;
  const int n1 = m1 ? rows1 : sdim;  // width after stage one
  if (live) {
    if (m1) {
      for (int j = q; j < rows1; j += 4) {
        const float *w = m1 + (size_t)j * cols1;
        float acc = 0.0f;
        for (int s = 0; s < nsplice; ++s) {
          const int t2 = min(max(g + s - left, u0), u1 - 1);
          const float *x = in + (size_t)t2 * ld;
          for (int d = 0; d < dim; ++d) acc = fmaf(x[d], w[s * dim + d], acc);
        }
        if (cols1 == sdim + 1) acc += w[sdim];
        y1[f][j] = acc;
      }
    } else {
      for (int k = q; k < sdim; k += 4) {
        const int s = k / dim, d = k - s * dim;
        const int t2 = min(max(g + s - left, u0), u1 - 1);
        y1[f][k] = in[(size_t)t2 * ld + d];
      }
    }
  }
  __syncthreads();
  if (!live) return;
  const int mi = (m2s && utt_mat) ? utt_mat[u] : (m2s ? 0 : -1);
  if (mi < 0) {
    for (int j = q; j < n1; j += 4) out[(size_t)g * ldo + j] = y1[f][j];
    return;
  }
  const float *m2 = m2s + (size_t)mi * rows2 * cols2;
  for (int j = q; j < rows2; j += 4) {
    const float *w = m2 + (size_t)j * cols2;
    float acc = 0.0f;
    for (int k = 0; k < n1; ++k) acc = fmaf(y1[f][k], w[k], acc);
    if (cols2 == n1 + 1) acc += w[n1];
    out[(size_t)g * ldo + j] = acc;
  }
}

}  // namespace amlike

using namespace amlike;

struct amlike_xform {
  int device = 0;
  int dim = 0, left = 0, right = 0, rows = 0, cols = 0;
  float *d_mat = nullptr;
  hipStream_t stream = nullptr;
  DeviceBuffer ws_off, ws_in, ws_out, ws_spk;
  std::vector<int32_t> h_off;  // offsets followed by the per-utterance matrix index
  PinnedUploadRing uploads;    // h_off on its way to ws_off
  std::mutex mu;
};

extern "C" {

int amlike_xform_create(int32_t dim, int32_t left, int32_t right, const float *mat, int32_t rows,
                        int32_t cols, amlike_xform_t **out) {
  if (int brc = BindThread()) return brc;
  const int sdim = dim * (left + right + 1);
  if (!out || dim <= 0 || left < 0 || right < 0 || (mat && (rows <= 0 || rows > kXformMaxRows)) ||
      (!mat && sdim > kXformMaxRows)) {
    SetError("amlike_xform_create: invalid argument (at most %d output rows)", kXformMaxRows);
    return AMLIKE_EINVAL;
  }
  if (mat && cols != sdim && cols != sdim + 1) {
    SetError("Transform matrix has bad dimension %dx%d versus feat dim %d", rows, cols, sdim);
    return AMLIKE_EDIM;
  }
  if (CurrentDevice() < 0) {
    int rc = amlike_init(0);
    if (rc != AMLIKE_OK) return rc;
  }
  amlike_xform *x = new amlike_xform;
  x->device = CurrentDevice();
  x->dim = dim;
  x->left = left;
  x->right = right;
  x->rows = mat ? rows : 0;
  x->cols = mat ? cols : 0;
  hipError_t e = hipSuccess;
  if (mat) {
    e = hipMalloc(&x->d_mat, (size_t)rows * cols * sizeof(float));
    if (e == hipSuccess)
      e = hipMemcpy(x->d_mat, mat, (size_t)rows * cols * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    SetError("amlike_xform_create: %s", hipGetErrorString(e));
    amlike_xform_destroy(x);
    return AMLIKE_EHIP;
  }
  *out = x;
  return AMLIKE_OK;
}

int amlike_xform_destroy(amlike_xform_t *x) {
  if (int brc = BindHandle(x)) return brc;
  if (!x) return AMLIKE_OK;
  (void)hipFree(x->d_mat);
  x->ws_off.Free();
  x->ws_in.Free();
  x->ws_out.Free();
  x->ws_spk.Free();
  if (x->stream) (void)hipStreamDestroy(x->stream);
  delete x;
  return AMLIKE_OK;
}

int32_t amlike_xform_input_dim(const amlike_xform_t *x) { return x ? x->dim : 0; }
int32_t amlike_xform_device(const amlike_xform_t *x) { return x ? x->device : -1; }
int32_t amlike_xform_output_dim(const amlike_xform_t *x) {
  if (!x) return 0;
  return x->d_mat ? x->rows : x->dim * (x->left + x->right + 1);
}

int amlike_xform_apply_dev(amlike_xform_t *x, const float *d_in, int32_t ld,
                           const int32_t *utt_rows, int32_t num_utts, const float *d_spk_mats,
                           int32_t rows2, int32_t cols2, const int32_t *utt_mat, float *d_out,
                           int32_t ldo, void *stream) {
  if (int brc = BindHandle(x)) return brc;
  if (!x || !d_in || !d_out || !utt_rows || num_utts <= 0 || ld < x->dim) {
    SetError("amlike_xform_apply: invalid argument");
    return AMLIKE_EINVAL;
  }
  const int n1 = amlike_xform_output_dim(x);
  if (d_spk_mats) {
    if (rows2 <= 0 || rows2 > kXformMaxRows) {
      SetError("amlike_xform_apply: at most %d output rows", kXformMaxRows);
      return AMLIKE_EINVAL;
    }
    if (cols2 != n1 && cols2 != n1 + 1) {
      SetError("Transform matrix has bad dimension %dx%d versus feat dim %d", rows2, cols2, n1);
      return AMLIKE_EDIM;
    }
  }
  // Without a per-utterance choice every utterance takes matrix 0; with one, utterances
  // marked -1 keep the stage-one features, which only makes sense at equal widths.
  const int out_dim = d_spk_mats ? rows2 : n1;
  if (d_spk_mats && utt_mat && rows2 != n1) {
    for (int u = 0; u < num_utts; ++u)
      if (utt_mat[u] < 0) {
        SetError("amlike_xform_apply: utterance %d has no second transform but it changes the dimension", u);
        return AMLIKE_EDIM;
      }
  }
  if (ldo < out_dim) {
    SetError("amlike_xform_apply: ldo %d below the output dimension %d", ldo, out_dim);
    return AMLIKE_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<int32_t> &h = x->h_off;
  h.assign((size_t)2 * num_utts + 1, 0);
  int64_t total = 0;
  for (int u = 0; u < num_utts; ++u) {
    if (utt_rows[u] <= 0) {
      SetError("SpliceFrames: empty input");
      return AMLIKE_EINVAL;
    }
    h[u] = (int32_t)total;
    total += utt_rows[u];
    h[num_utts + 1 + u] = (d_spk_mats && utt_mat) ? utt_mat[u] : 0;
  }
  if (total > ((int64_t)1 << 30)) {
    SetError("amlike_xform_apply: batch too large");
    return AMLIKE_EINVAL;
  }
  h[num_utts] = (int32_t)total;
  int rc = x->ws_off.Reserve(h.size() * sizeof(int32_t));
  if (rc != AMLIKE_OK) return rc;
  rc = x->uploads.Upload(x->ws_off.ptr, h.data(), h.size() * sizeof(int32_t), st);
  if (rc != AMLIKE_OK) return rc;
  const int32_t *d_off = static_cast<const int32_t *>(x->ws_off.ptr);
  const int blocks = (int)((total + kXformFrames - 1) / kXformFrames);
  hipLaunchKernelGGL(feat_splice_affine_kernel, dim3(blocks), dim3(256), 0, st, d_in, ld, x->dim,
                     x->left, x->left + x->right + 1, d_off, num_utts, x->d_mat, x->rows, x->cols,
                     d_spk_mats, rows2, cols2,
                     (d_spk_mats && utt_mat) ? d_off + num_utts + 1 : nullptr, d_out, ldo, (int)total);
  AMLIKE_HIP(hipGetLastError());
  return AMLIKE_OK;
}

int amlike_xform_apply(amlike_xform_t *x, const float *feats, int32_t T, int32_t ld,
                       const float *spk_mat, int32_t rows2, int32_t cols2, float *out, int32_t ldo) {
  if (int brc = BindHandle(x)) return brc;
  if (!x || !feats || !out || T <= 0 || ld < x->dim) {
    SetError(T <= 0 ? "SpliceFrames: empty input" : "amlike_xform_apply: invalid argument");
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(x->mu);
  const int out_dim = spk_mat ? rows2 : amlike_xform_output_dim(x);
  if (ldo < out_dim) {
    SetError("amlike_xform_apply: ldo %d below the output dimension %d", ldo, out_dim);
    return AMLIKE_EINVAL;
  }
  int rc = x->ws_in.Reserve((size_t)T * ld * sizeof(float));
  if (rc == AMLIKE_OK) rc = x->ws_out.Reserve((size_t)T * out_dim * sizeof(float));
  if (rc == AMLIKE_OK && spk_mat) rc = x->ws_spk.Reserve((size_t)std::max(1, rows2 * cols2) * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  AMLIKE_HIP(hipMemcpyAsync(x->ws_in.ptr, feats, ((size_t)(T - 1) * ld + x->dim) * sizeof(float),
                            hipMemcpyHostToDevice, x->stream));
  if (spk_mat && rows2 > 0 && cols2 > 0)
    AMLIKE_HIP(hipMemcpyAsync(x->ws_spk.ptr, spk_mat, (size_t)rows2 * cols2 * sizeof(float),
                              hipMemcpyHostToDevice, x->stream));
  rc = amlike_xform_apply_dev(x, static_cast<const float *>(x->ws_in.ptr), ld, &T, 1,
                              spk_mat ? static_cast<const float *>(x->ws_spk.ptr) : nullptr, rows2,
                              cols2, nullptr, static_cast<float *>(x->ws_out.ptr), out_dim, x->stream);
  if (rc != AMLIKE_OK) return rc;
  AMLIKE_HIP(hipMemcpy2DAsync(out, (size_t)ldo * sizeof(float), x->ws_out.ptr,
                              (size_t)out_dim * sizeof(float), (size_t)out_dim * sizeof(float), T,
                              hipMemcpyDeviceToHost, x->stream));
  AMLIKE_HIP(hipStreamSynchronize(x->stream));
  return AMLIKE_OK;
}

}  // extern "C"
