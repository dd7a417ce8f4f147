// Entry points declared in include/nadavca_hip.h whose kernels are not written yet.
// They fail loudly (no CPU fallback); each is removed from here when its kernel lands.
#include "nvk_internal.h"

#ifndef NVK_HAVE_CONSENSUS
extern "C" int nvk_consensus_accumulate_dev(nvk_ctx *, int64_t, int64_t, int, const double *, const int32_t *, const int64_t *, const int64_t *, const int32_t *, const int32_t *, double, int64_t, double *, int64_t *) {
  nvk_set_error("consensus kernels not built into this library");
  return NVK_ERR_UNSUPPORTED;
}
extern "C" int nvk_posterior_dev(nvk_ctx *, int64_t, int, int, double, const double *, const int32_t *, double *) {
  nvk_set_error("posterior kernel not built into this library");
  return NVK_ERR_UNSUPPORTED;
}
#endif
