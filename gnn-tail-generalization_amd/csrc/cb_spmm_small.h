// Narrow-feature (d <= 16) variant of the CSR sum-aggregation (cb_spmm_small.hip); cb_spmm.hip dispatches to it.
#pragma once
#include "cb_common.h"

namespace cb {

// d <= 16 (wider rows are faster on one wavefront per gathered row: measurements in cb_spmm_small.hip)
bool spmm_small_eligible(int64_t d, bool al16);
int launch_spmm_small(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, int64_t d, const float* row_scale,
                      const float* bias, int relu, float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks,
                      const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial, int64_t ld_p, bool al16, hipStream_t st);

}  // namespace cb
