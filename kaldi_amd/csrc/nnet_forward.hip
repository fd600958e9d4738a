// placeholder, replaced by the nnet2 forward implementation
#include "amlike_common.h"
extern "C" {
int amlike_nnet_create(const amlike_component_t *, int32_t, const float *, int32_t, amlike_nnet_t **) { amlike::SetError("nnet: not built"); return AMLIKE_EINVAL; }
int amlike_nnet_destroy(amlike_nnet_t *) { return AMLIKE_OK; }
int32_t amlike_nnet_input_dim(const amlike_nnet_t *) { return 0; }
int32_t amlike_nnet_output_dim(const amlike_nnet_t *) { return 0; }
int32_t amlike_nnet_left_context(const amlike_nnet_t *) { return 0; }
int32_t amlike_nnet_right_context(const amlike_nnet_t *) { return 0; }
int amlike_nnet_compute(amlike_nnet_t *, const float *, int32_t, int32_t, int32_t, int32_t, float, float *, int32_t) { return AMLIKE_EINVAL; }
int amlike_nnet_compute_dev(amlike_nnet_t *, const float *, int32_t, int32_t, int32_t, int32_t, float, float *, int32_t, void *) { return AMLIKE_EINVAL; }
int amlike_nnet_time_compute_dev(amlike_nnet_t *, const float *, int32_t, int32_t, int32_t, int32_t, float, float *, int32_t, void *, int32_t, float *) { return AMLIKE_EINVAL; }
}
