#include "nd4hip_internal.h"
int nd4_getrf(nd4hip_handle*, int64_t, int64_t, const double*, double*, int32_t*) {
  nd4_set_error("nd4_getrf: not implemented yet"); return ND4HIP_ERR_ARG;
}
