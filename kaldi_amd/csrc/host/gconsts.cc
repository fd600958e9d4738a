// DiagGmm::ComputeGconsts for the C ABI (host arithmetic, done once at model load).
// Follows /root/reference/src/gmm/diag-gmm.cc:114-152: a float accumulator that is
// widened to double for every "+=", log() in double (log(weight) + offset is summed
// in double before the first narrowing), +inf mapped to -inf, NaN reported as an error.
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "../../../include/amlike.h"
#include "../amlike_common.h"

extern "C" int amlike_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                                          const float *inv_vars, int32_t num_gauss, int32_t dim,
                                          float *gconsts, int32_t *num_bad) {
  if (!weights || !means_invvars || !inv_vars || !gconsts || num_gauss <= 0 || dim <= 0) {
    amlike::SetError("amlike_gmm_compute_gconsts: invalid argument");
    return AMLIKE_EINVAL;
  }
  const double kLog2Pi = 1.8378770664093454835606594728112;
  const float offset = static_cast<float>(-0.5 * kLog2Pi * dim);
  int bad = 0;
  for (int32_t g = 0; g < num_gauss; ++g) {
    float gc = static_cast<float>(std::log(static_cast<double>(weights[g])) + offset);
    const float *miv = means_invvars + static_cast<size_t>(g) * dim;
    const float *iv = inv_vars + static_cast<size_t>(g) * dim;
    for (int32_t d = 0; d < dim; ++d)
      gc += 0.5 * std::log(static_cast<double>(iv[d])) - 0.5 * miv[d] * miv[d] / iv[d];
    if (std::isnan(gc)) {
      amlike::SetError("At component %d, not a number in gconst computation", g);
      return AMLIKE_ENONFINITE;
    }
    if (std::isinf(gc)) {
      ++bad;
      if (gc > 0) gc = -gc;
    }
    gconsts[g] = gc;
  }
  if (num_bad) *num_bad = bad;
  return AMLIKE_OK;
}
