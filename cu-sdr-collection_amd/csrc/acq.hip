// acq.hip — placeholder until the acquisition kernels land (replaced in the next commit).
#include "gc_internal.h"
extern "C" int gc_acquire_coarse(gc_context*, const gc_acq_params*, int, const int8_t*, gc_acq_result*) {
  gc_set_error("gc_acquire_coarse: not implemented yet");
  return GC_E_UNSUPPORTED;
}
extern "C" int gc_acquire_fine_l1ca(gc_context*, const gc_acq_params*, const int8_t*, int, double, double*) {
  gc_set_error("gc_acquire_fine_l1ca: not implemented yet");
  return GC_E_UNSUPPORTED;
}
