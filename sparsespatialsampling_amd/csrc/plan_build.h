// Device-side plan construction (plan_build.hip), used by s3_interp_plan_create (interp_plan.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace s3 {

// device tables of a plan; every non-null pointer is a hipMalloc allocation owned by the caller (also after an error)
struct PlanTables {
    int32_t *perm = nullptr;             // [nc] processing position -> cell id
    int32_t *tile_cell_begin = nullptr;  // [n_tiles+1]
    int32_t *tile_row_begin = nullptr;   // [n_tiles+1]
    int32_t *rows = nullptr;             // [total_rows] distinct source rows, tile after tile
    uint16_t *loc = nullptr;             // [nc*k] per tile: [m][cell in tile] -> position in the tile's row list
    int64_t n_tiles = 0, total_rows = 0;
};

// returns S3_OK or a negative S3_E* code (message in s3_last_error)
int build_plan_tables(const int32_t *d_idx, int64_t nc, int k, int64_t n_src, const double *d_centers, int dim, int tc,
                      int ucap, hipStream_t st, PlanTables *out);

}  // namespace s3
