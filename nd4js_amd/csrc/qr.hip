#include "nd4hip_internal.h"
int nd4_geqrf_q(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*) {
  nd4_set_error("nd4_geqrf_q: not implemented yet"); return ND4HIP_ERR_ARG;
}
