/*
 * s3hip.h -- C ABI of libs3hip.so: the MI355X (gfx950) implementation of the S^3 hot path.
 *
 * The reference (JanisGeise/sparseSpatialSampling) is pure Python and has no FFI; its boundary for this path is the
 * set of Python call sites listed per function below (file:line relative to the reference checkout).  A maintainer
 * binds this library with ctypes.  Conventions:
 *
 *   - every pointer named d_* is a DEVICE pointer (HBM of the current HIP device); h_* are host pointers;
 *   - arrays are dense row-major; `double` = IEEE f64; indices are int32 on the device (N < 2^31);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on that stream unless
 *     stated otherwise; nothing allocates inside a call except s3_knn_create / s3_malloc;
 *   - return value 0 = success, negative = error (S3_E*); s3_last_error() returns the message of the last failure on
 *     the calling thread;
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with S3_ENODEV.
 *
 * Cell conventions (reference s_cube.py:188-194, 399-445): a cell is (centre[dim] f64, level i32); children and
 * nodes are enumerated in the reference's direction-table order; child centre = centre + dir*(0.25*width)/2^level,
 * node = centre + dir*(0.5*width)/2^level.
 */
#ifndef S3HIP_H
#define S3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3_OK 0
#define S3_EINVAL (-1)   /* bad argument (shape, k, dim, null pointer) */
#define S3_ENODEV (-2)   /* no usable HIP device */
#define S3_EHIP (-3)     /* HIP runtime error, see s3_last_error() */
#define S3_ENOMEM (-4)

#define S3_DTYPE_F32 0
#define S3_DTYPE_F64 1

#define S3_MAX_K 64

/* what s3_abi_version() of a library built from this header returns; the bindings refuse a library that reports another
 * number (a stale build) with the command that rebuilds it */
#define S3_ABI_VERSION 6

typedef struct s3_knn s3_knn; /* opaque: grid-sorted copy of the original point cloud, resident in HBM */
typedef void *s3_stream;

/* ---- runtime / plumbing ------------------------------------------------------------------------------------ */
const char *s3_last_error(void);
int s3_abi_version(void);
/* stop and join the library's own host threads (transfer lanes); returns their number.  Call once at process exit, before the
 * HIP runtime goes away (the Python bindings register it with atexit); safe at any time -- later calls start new lanes. */
int s3_shutdown(void);
/* debugging aid: install a SIGABRT handler that prints the native frames of the aborting thread to stderr, then aborts as usual */
int s3_debug_abort_backtrace(void);
int s3_device_count(int *h_count);
int s3_set_device(int device);
int s3_malloc(void **d_ptr, size_t bytes);
int s3_free(void *d_ptr);
/* host <-> device copies.  PAGE-LOCKED host memory (hipHostMalloc / s3_host_register): one asynchronous copy on `stream`.  PAGEABLE
 * host memory is never handed to the runtime's copy engine (which would pin the caller's pages on the fly): it goes through the
 * library's page-locked lanes (s3_upload_rows / s3_download) and the call returns when the copy is complete. */
int s3_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, s3_stream stream);
int s3_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, s3_stream stream);
int s3_stream_synchronize(s3_stream stream);
/* device -> pageable host array through persistent pinned buffers drained by several host threads (a plain copy into
 * pageable memory runs at 12-15 GB/s); returns when h_dst is complete */
int s3_download(void *h_dst, const void *d_src, size_t bytes, s3_stream stream);
/* page-lock host memory that was not allocated through HIP (a POSIX shared-memory mapping shared by the ranks of a node) and
 * map it into the device's address space: *d_ptr is what kernels write through.  Undo with s3_host_unregister. */
int s3_host_register(void *h_ptr, size_t bytes, void **d_ptr);
int s3_host_unregister(void *h_ptr);
/* upload target of a snapshot batch (the .to(device) of a host tensor handed to ExportData.export, export.py:128-167):
 * pageable host rows [n_rows][row_bytes] -> device rows with pitch dst_pitch_bytes, staged through persistent pinned
 * buffers filled by several host threads; asynchronous on `stream` (the host data may be reused on return).  The
 * bytes between row_bytes and dst_pitch_bytes of every row but the last are padding and may be overwritten. */
int s3_upload_rows(const void *h_src, int64_t n_rows, int64_t row_bytes, void *d_dst, int64_t dst_pitch_bytes,
                   s3_stream stream);
/* the same for a selection of the host rows: row h_rows[i] of h_src becomes device row i.  A generated grid references
 * only the source points near its cell centres (k neighbours each): uploading just those rows cuts the PCIe volume by
 * the sparsity of the grid. */
int s3_upload_rows_indexed(const void *h_src, const int32_t *h_rows /*[n_sel]*/, int64_t n_sel, int64_t row_bytes,
                           void *d_dst, int64_t dst_pitch_bytes, s3_stream stream);
/* the same for a PIECE of every (selected) row: source row r (h_rows[i], or i when h_rows is NULL) starts at h_src +
 * r * src_row_stride_bytes + src_offset_bytes and contributes n_segments segments of segment_bytes, segment_stride_bytes
 * apart; device row i holds them back to back.  This is how ExportData pipelines one export() call (reference
 * export.py:128-167 hands over the whole batch [N, n_comp, T]): the snapshots [t0, t1) of a field are n_comp segments of
 * (t1 - t0) values, T values apart, so piece j + 1 crosses PCIe while piece j is interpolated and piece j - 1 comes back. */
int s3_upload_row_pieces(const void *h_src, const int32_t *h_rows /*[n_rows] or NULL*/, int64_t n_rows,
                         int64_t src_row_stride_bytes, int64_t src_offset_bytes, int n_segments, int64_t segment_bytes,
                         int64_t segment_stride_bytes, void *d_dst, int64_t dst_pitch_bytes, s3_stream stream);

/* ---- KNN index over the original CFD points -------------------------------------------------------------------
 * Replaces KNeighborsRegressor(...).fit(vertices, target)           s_cube.py:161-163
 *      and NearestNeighbors(...).fit(coordinates)                   export.py:120,423
 * Builds a two-level bucket index on the device (bounding box, counting sort into a uniform grid; buckets holding more
 * than 8x the target occupancy get their own sub-lattice, which keeps strongly graded clouds fast).  Synchronous
 * (returns after the build finished).  `target_occupancy` <= 0 selects the default (3 points per bucket in 2-D, 8 in
 * 3-D). */
int s3_knn_create(const double *d_pts /*[n,dim]*/, int64_t n, int dim, double target_occupancy, s3_stream stream,
                  s3_knn **out);
void s3_knn_destroy(s3_knn *knn);
int s3_knn_info(const s3_knn *knn, int64_t *h_n_buckets, int64_t *h_n_refined);
/* attach the regression target y[n] (the S^3 metric, original point order); permuted into grid order */
int s3_knn_set_values(s3_knn *knn, const double *d_y /*[n]*/, s3_stream stream);

/* exact k nearest neighbours, ascending in (distance, original index); dist = sqrt(sum_j (q_j-p_j)^2).
 * Replaces NearestNeighbors.kneighbors(centers)                      export.py:425,438 */
int s3_knn_query(const s3_knn *knn, const double *d_q /*[nq,dim]*/, int64_t nq, int k, int32_t *d_idx /*[nq,k]*/,
                 double *d_dist /*[nq,k]*/, s3_stream stream);

/* inverse-distance KNN regression with scikit-learn's exact-hit rule and numpy's summation order.
 * Replaces self._knn.predict(...)                                    s_cube.py:224,328,372 */
int s3_idw_predict(const s3_knn *knn, const double *d_q /*[nq,dim]*/, int64_t nq, int k, double *d_yhat /*[nq]*/,
                   s3_stream stream);

/* ---- refine kernels ---------------------------------------------------------------------------------------- */
/* a3: children of `n_par` parents.  Child c of parent p_i gets cell id new_index + i*2^dim + c; its centre/level
 * are written into the cell arrays.  Replaces _compute_cell_centers(to_refine, keep_parent=False)
 *                                                                    s_cube.py:875,399-445 */
int s3_make_children(double *d_center /*[cap,dim]*/, int32_t *d_level /*[cap]*/, const int32_t *d_parents /*[n_par]*/,
                     int64_t n_par, int64_t new_index, int dim, double width, s3_stream stream);

/* a4: metric at the centre of each of n cells (ids first..first+n-1) and at its 2^dim candidate child centres, then
 * gain = level_factor[level] * sum_j |m0 - mj| / gain0.  d_level_factor[64] holds the reference's Python expression
 * 1/2^d*(width/2^level)^d evaluated on the host.  d_scratch: n*(2^dim+1) doubles.
 * Replaces SamplingTree._update_gain + _update_gain                  s_cube.py:207-241,1840-1859 */
int s3_child_gain(const s3_knn *knn, int k, const double *d_center, const int32_t *d_level, int64_t first, int64_t n,
                  int dim, double width, const double *d_level_factor, double gain0, double *d_metric /*[cap]*/,
                  double *d_gain /*[cap]*/, double *d_scratch, s3_stream stream);
/* the same with the predictions of a cell's 2^dim candidate child centres kept in d_child_metric[cap][2^dim]: the centre of
 * a new cell is candidate child c of its parent, a point the parent's call predicted already (same coordinates, same
 * deterministic search -- the reference predicts it a second time, s_cube.py:221-233, and obtains the same number), so with
 * d_parents given only the 2^dim child points of a new cell are searched (8 of 9 queries in 3-D) and its centre's value is
 * d_child_metric[parent][c].  The n cells first..first+n-1 are children number parents_offset.. of the ORDERED parents
 * d_parents[] (2^dim consecutive cells per parent, as s3_make_children creates them).  d_parents == NULL: all 2^dim + 1
 * points are searched (cells whose parent has no entry: the root).  Either way the children's values of the n cells are
 * stored in d_child_metric.  d_scratch: n*(2^dim+1) doubles + 2*(2 + n*2^dim) int32.  With d_parents every cell is given to
 * one wavefront (shared candidate box of the 2^dim child points; selection by histogram instead of sorted insertion:
 * csrc/knn.hip); the queries that scheme cannot answer are listed behind the doubles and handed on: to a grouped search
 * (2^dim queries per wavefront, each its own small box: cells too large for one box), then to one wavefront per query
 * (next to a body, at the edge of the cloud, outside it), and what is still left (refined buckets, ties in distance, k > 48)
 * to the per-lane search.  Same bits whichever kernel answers. */
int s3_child_gain_reuse(const s3_knn *knn, int k, const double *d_center, const int32_t *d_level, int64_t first, int64_t n,
                        int dim, double width, const double *d_level_factor, double gain0, double *d_metric /*[cap]*/,
                        double *d_gain /*[cap]*/, double *d_scratch, const int32_t *d_parents /*or NULL*/,
                        int64_t parents_offset, double *d_child_metric /*[cap][2^dim]*/, s3_stream stream);

/* a12: geometry predicates on the nodes of the listed cells (d_cells == NULL: ids first..first+n-1).  Each call ORs
 * its verdict into d_invalid[n] (zero it first): GeometryObject._apply_mask policy, geometry_base.py:40-76.
 *   box:      lo <= x <= hi in every dimension                       cube_geometry.py:50-74
 *   sphere:   ||x - pos|| <= radius                                  sphere_geometry.py:47-72
 *   cylinder: 0 <= proj <= norm && normal distance <= local radius   cylinder_geometry.py:126-157
 *   polygon:  strictly inside (boundary excluded)                    coordinates_2d.py:54-75
 *   triangle: edge cross products not of mixed sign (outline inside) triangle_geometry.py:80-103
 *   prism:    0 <= proj <= norm && in-plane point inside triangle    prism_geometry.py:90-118
 *   tetrahedra (1 = tetrahedron, 2 = the two halves of a pyramid, union): no inward face normal sees the point
 *             behind its face                                        tetrahedron_geometry.py:121-140,
 *                                                                    pyramid_geometry.py:156-170 */
int s3_mask_box(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n, int dim,
                double width, const double *h_lo, const double *h_hi, int refine_mode, int keep_inside,
                uint8_t *d_invalid, s3_stream stream);
int s3_mask_sphere(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                   int dim, double width, const double *h_pos, double radius, int refine_mode, int keep_inside,
                   uint8_t *d_invalid, s3_stream stream);
int s3_mask_cylinder(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                     double width, const double *h_p0, const double *h_axis, double norm, double r0, double r1,
                     int is_cone, int refine_mode, int keep_inside, uint8_t *d_invalid, s3_stream stream);
int s3_mask_polygon(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                    double width, const double *d_poly /*[nv,2] device*/, int nv, int refine_mode, int keep_inside,
                    uint8_t *d_invalid, s3_stream stream);
int s3_mask_triangle(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                     double width, const double *h_points /*[3][2]*/, int refine_mode, int keep_inside,
                     uint8_t *d_invalid, s3_stream stream);
int s3_mask_prism(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                  double width, const double *h_origin /*[3] first point of the first triangle*/,
                  const double *h_axis /*[3] extrusion*/, double norm, const int32_t *h_dims /*[2] in-plane axes*/,
                  const double *h_triangle /*[3][2] first triangle in those axes*/, int refine_mode, int keep_inside,
                  uint8_t *d_invalid, s3_stream stream);
int s3_mask_tetrahedra(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                       double width, const double *h_positions /*[n_tets][4][3]*/,
                       const double *h_normals /*[n_tets][3][4] inward, column p belongs to point p*/, int n_tets,
                       int refine_mode, int keep_inside, uint8_t *d_invalid, s3_stream stream);

/* bookkeeping of one refine batch on the device-resident cell arrays: parents stop being leaves, valid children
 * become leaves, invalid children get gain 0 (s_cube.py:721-723, 250-251) */
int s3_commit_batch(uint8_t *d_leaf /*[cap]*/, double *d_gain /*[cap]*/, const int32_t *d_parents, int64_t n_par,
                    int64_t first, int64_t n_new, const uint8_t *d_invalid /*[n_new] or NULL*/, s3_stream stream);

/* a6 local part: sum over leaf cells of metric^2 (the captured-metric numerator, s_cube.py:327-335), restricted to
 * cell ids in [begin, end) so that ranks can split the array; deterministic two-stage reduction.  d_out: 1 double.
 * d_scratch: 1024 doubles. */
int s3_sumsq_leaf(const double *d_metric, const uint8_t *d_leaf, int64_t begin, int64_t end, double *d_out,
                  double *d_scratch, s3_stream stream);

/* a8: the n_top leaf cells with the largest (gain, -id), ordered like heapq.nlargest (s_cube.py:601-602).
 * Radix select on the device, final ordering of the n_top survivors on the host: writes the ordered ids to the HOST
 * array h_out[n_top]; h_count receives min(n_top, #leaves).  Synchronous.  d_scratch: s3_topn_scratch_bytes(). */
size_t s3_topn_scratch_bytes(int64_t n_cells, int64_t n_top);
int s3_topn_leaf(const double *d_gain, const uint8_t *d_leaf, int64_t n_cells, int64_t n_top, int32_t *h_out,
                 int64_t *h_count, void *d_scratch, s3_stream stream);

/* ---- export kernels ---------------------------------------------------------------------------------------- */
/* a16: w = 1/clamp(dist, 1e-12), rows normalised to sum 1 (torch's summation order).  export.py:428-429 */
int s3_idw_weights(const double *d_dist /*[nc,k]*/, int64_t nc, int k, double *d_w /*[nc,k]*/, s3_stream stream);

/* a17/a18: out[c, l] = sum_m w[c,m] * data[idx[c,m], l], l over the contiguous (n_comp*T) axis, f64 accumulate,
 * f64 output.  Replaces interpolate_data                             export.py:446-468, and :215 with row_len=1 */
int s3_interp(const double *d_w /*[nc,k]*/, const int32_t *d_idx /*[nc,k]*/, int64_t nc, int k, const void *d_data,
              int dtype, int64_t n_src, int64_t row_len, double *d_out /*[nc,row_len]*/, s3_stream stream);

/* a19 hand-over: [nc][n_comp][T] -> [T][nc][n_comp], the snapshot-major image of an interpolated batch, so that the HDF5
 * sink (one dataset per snapshot, export.py:283-299) gets contiguous snapshots instead of slicing out[:, :, i] on the host. */
int s3_snapshot_major(const double *d_in, int64_t nc, int n_comp, int64_t n_snapshots, double *d_out, s3_stream stream);
/* the same for a SHARD of the cells (several ranks): input cell c becomes row d_rows[c] of an output with n_out
 * rows, [T][n_out][n_comp] -- every rank writes its rows of the one batch buffer the ranks share (host memory registered with
 * s3_host_register: the values cross this rank's own PCIe link, nothing is sent to the rank that writes the file). */
int s3_snapshot_major_rows(const double *d_in, int64_t nc, int n_comp, int64_t n_snapshots, const int32_t *d_rows /*[nc]*/,
                           int64_t n_out, double *d_out, s3_stream stream);

/* Metric upstream of S^3 (what the reference's example scripts compute with torch before the grid is generated:
 * metric = pt.std(field, dim=1), examples/s3_for_OAT15_airfoil.py:91): temporal mean and standard deviation of every row
 * of a snapshot matrix [n_rows][row_len] (row pitch in_stride elements, 0 = row_len), f64 accumulation, one pass over
 * the data.  ddof = 1: torch's default (unbiased), 0: population.  Either output may be NULL. */
int s3_row_moments(const void *d_data, int dtype, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                   double *d_mean /*[n_rows] or NULL*/, double *d_std /*[n_rows] or NULL*/, s3_stream stream);
/* the same of |x|: mean_t sum_c |U_c| of a vector field [N, n_comp, T] (examples/s3_for_cylinder2D_Re100.py:55) is n_comp times
 * the mean over the cell's [n_comp * T] row */
int s3_row_abs_moments(const void *d_data, int dtype, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                   double *d_mean /*[n_rows] or NULL*/, double *d_std /*[n_rows] or NULL*/, s3_stream stream);

/* Planned form of a17 for a static neighbour table (the table ExportData caches at export.py:431-432 and reuses for
 * every snapshot batch and field): the plan de-duplicates the source rows of spatially adjacent cells once (built on the
 * device; Hilbert order of d_centers[nc,dim] when given), the kernel then stages each distinct row once per tile in LDS.
 * Same results as s3_interp.  Source rows must start on 16-byte boundaries and be readable up to the next multiple of
 * 16 bytes (in_stride % 4 == 0 for f32 / % 2 == 0 for f64, in_stride >= row_len rounded up to that multiple); the row
 * length itself is arbitrary (25-snapshot batches of examples/s3_for_cylinder3D_Re3900.py:28-69: pitch 28 or 32).
 * Rows of up to 64 bytes take a short-row kernel (several workgroups per CU, no chunk pipeline); rows of up to 1 KiB on
 * plans with the reference's neighbour counts (k = 8 | 26, export.py:84-85) the persistent kernel (two workgroups per CU
 * walking all tiles, the next tile's rows / weights / ids in flight while the current one is accumulated). */
typedef struct s3_interp_plan s3_interp_plan;
int s3_interp_plan_create(const int32_t *d_idx /*[nc,k]*/, int64_t nc, int k, int64_t n_src,
                          const double *d_centers /*[nc,dim] or NULL*/, int dim, int tile_cells /*0 (=64), 64 or 128*/,
                          s3_stream stream, s3_interp_plan **out);
void s3_interp_plan_destroy(s3_interp_plan *plan);
int s3_interp_plan_info(const s3_interp_plan *plan, int64_t *h_n_tiles, int64_t *h_total_rows);
/* leaf-cell shards for `world` ranks: the plan's tiles (cells in Hilbert order: a run of tiles is a compact
 * blob of the grid) are cut into `world` consecutive runs of nearly equal cost = bytes the kernel moves per snapshot
 * (4 per staged source row, halo included, + 8 per output row + the weights' share).  d_order[nc] (may be NULL) receives
 * the plan's processing order (position -> cell id); rank r owns the cells d_order[h_cuts[r] .. h_cuts[r+1]).
 * h_cuts: world + 1 entries on the host. */
int s3_interp_plan_partition(const s3_interp_plan *plan, int world, int32_t *d_order, int64_t *h_cuts, s3_stream stream);
/* the same cost, cumulated along the plan's processing order and sampled at n_samples + 1 equally spaced cell positions
 * i * nc / n_samples (h_out[0] = 0, h_out[n_samples] = total; linear inside a tile): what a rank that holds the plan of ONE
 * stretch of the Hilbert curve publishes so that all ranks can cut the curve into stretches of equal cost without anybody
 * building the table of all cells (parallel.LeafShards) */
int s3_interp_plan_cost_profile(const s3_interp_plan *plan, int n_samples, double *h_out /*[n_samples + 1]*/, s3_stream stream);
/* in_stride: elements between consecutive source rows of d_data (>= row_len; 0 = row_len).  Rows padded to a multiple
 * of 128 bytes keep every staged segment on one cache line. */
/* the weights of the table, [nc,k] in the caller's cell order, are kept inside the plan in tile order (one contiguous
 * stream per tile): set them once per KNN cache, then pass d_w = NULL to s3_interp_planned; a non-NULL d_w re-attaches
 * the weights before the launch (one extra pass over the table) */
int s3_interp_plan_set_weights(s3_interp_plan *plan, const double *d_w /*[nc,k]*/, s3_stream stream);
int s3_interp_planned(s3_interp_plan *plan, const double *d_w /*[nc,k] or NULL*/, const void *d_data, int dtype,
                      int64_t row_len, int64_t in_stride, double *d_out /*[nc,row_len]*/, s3_stream stream);
/* A batch that already lives in HBM as the reference hands it over -- data[N, n_comp * T], every row of the CFD mesh,
 * export.py:446-468 -- is read WHERE IT LIES: no pass that first gathers the referenced rows into a pitched copy.
 *   s3_interp_plan_set_source_ids  d_ids[n_src]: the row of the full table behind each source row the plan was built on
 *                                  (the `used` list of s3_compact_rows); once per KNN cache
 *   s3_interp_planned_src          d_table[n_table_rows][in_stride]: the full batch; same results as s3_interp_planned on
 *                                  the gathered copy
 * Rows that are not 16-byte aligned (25 fp32 snapshots: 100-byte rows) are read with element alignment by the persistent
 * kernel (plans with k = 8 | 26, rows of at least 16 bytes); nothing beyond the end of a row is touched.  The same
 * relaxed alignment holds for s3_interp_planned on such plans. */
int s3_interp_plan_set_source_ids(s3_interp_plan *plan, const int32_t *d_ids /*[n_src]*/, int64_t n_table_rows, s3_stream stream);
int s3_interp_planned_src(s3_interp_plan *plan, const void *d_table, int dtype, int64_t n_table_rows, int64_t row_len,
                          int64_t in_stride, double *d_out /*[nc,row_len]*/, s3_stream stream);

/* ---- yardsticks of the measurement (bench.py's roofline line; no counterpart in the reference, not on any product path) ----
 * s3_yard_stream      a hand-written streaming kernel over d_src: every lane reads `reads` 16-byte vectors (coalesced) and
 *                     writes `writes` vectors derived from them; (reads, writes) = (1, 1) float4 copy, (7, 2) the read / write
 *                     mix of the headline launch (78 % / 22 %), (4, 0) reads only.  Short-lived workgroups, one contiguous block each (the
 *                     fastest form measured, csrc/yardstick.hip).
 *                     *h_bytes_read / *h_bytes_written = what the launch moved.
 * s3_yard_plan_loads  the loads of a tile plan and nothing else, at the headline kernel's occupancy and load schedule, on the
 *                     table s3_interp_planned_src reads (n_table_rows > 0) or s3_interp_planned's compacted one (0):
 *                     variant 0 one 128-byte line of a row per visit (the shift kernel's schedule), variant 1 two consecutive
 *                     lines (256 contiguous bytes) per visit with the same bytes and loads in flight.  *h_staged_bytes =
 *                     rows staged over all tiles x lines x 128. */
int s3_yard_stream(const void *d_src, void *d_dst, int64_t src_bytes, int64_t dst_bytes, int reads, int writes,
                   int nontemporal, s3_stream stream, int64_t *h_bytes_read, int64_t *h_bytes_written);
int s3_yard_plan_loads(s3_interp_plan *plan, const void *d_table, int64_t n_table_rows, int64_t row_bytes, int64_t stride_bytes,
                       int variant, s3_stream stream, int64_t *h_staged_bytes);
/* re-read the S3_* environment switches of the planned launches (they are parsed once, at the first launch; A/B tools that
 * flip them inside one process call this after every change) */
int s3_debug_reload_env(void);


/* ---- device-side bookkeeping of the KNN cache (replaces torch.unique / fancy indexing on the a16 path) ----------
 * A generated grid that is sparser than the CFD mesh references only part of the source rows (export.py:403-444 keeps
 * the full table; here only the referenced rows are uploaded per batch):
 *   s3_mark_rows     d_flag[idx[i]] = 1 for every entry of a neighbour table (d_flag zeroed by the caller, int32[n_src])
 *   s3_compact_rows  d_flag -> remap in place (position among the marked rows, ascending row id; -1 = unused),
 *                    d_used[0..n_used) = the marked row ids ascending; *h_n_used = their number
 *   s3_remap_indices idx[i] = d_remap[idx[i]] in place
 *   s3_gather_rows   d_dst row i (pitch dst_pitch_bytes) = d_src row ids[i] (ids NULL: row i), row_bytes % 4 == 0:
 *                    the device-side form of the indexed upload for batches that already live in HBM, and the re-pitch
 *                    of dense ragged rows for s3_interp_planned */
int s3_mark_rows(const int32_t *d_idx, int64_t n, int64_t n_src, int32_t *d_flag, s3_stream stream);
int s3_compact_rows(int32_t *d_flag_remap /*[n_src] in: 0/1, out: remap*/, int64_t n_src, int32_t *d_used /*[n_src]*/,
                    int64_t *h_n_used, s3_stream stream);
int s3_remap_indices(int32_t *d_idx, int64_t n, const int32_t *d_remap, int64_t n_src, s3_stream stream);
/* Hilbert-curve order of n points [n,dim] (dim 2 | 3): d_perm[position] = point.  The export path keeps the referenced
 * source rows in this order in HBM, so that the rows a tile of neighbouring cells gathers lie close together (fewer
 * DRAM pages / address translations per tile than with the CFD mesh's arbitrary numbering). */
int s3_spatial_order(const double *d_points, int64_t n, int dim, int32_t *d_perm, s3_stream stream);
/* The two device-wide primitives behind the planner and the device topology (csrc/scan_sort.h, hand-written: three-launch
 * exclusive scan; stable LSD radix sort with 8-bit digits), exported for the tests.  s3_exclusive_scan: d_out[i] = sum of
 * d_in[0 .. i-1], int32 (elem_bytes 4) or int64 (8), in place allowed.  s3_sort_pairs: (key, value) pairs ascending by the low
 * `bits` bits of the keys, stable, in place.  Both return when the work is done.  No counterpart in the reference (torch.unique /
 * numpy fancy indexing do this work there, export.py:403-444). */
int s3_exclusive_scan(const void *d_in, void *d_out, int64_t n, int elem_bytes, s3_stream stream);
int s3_sort_pairs(uint64_t *d_keys, int32_t *d_vals, int64_t n, int bits, s3_stream stream);
/* d_remap[d_ids[i]] = i for the n distinct row ids, every other entry of d_remap[n_src] = -1 */
int s3_positions_of(const int32_t *d_ids, int64_t n, int32_t *d_remap, int64_t n_src, s3_stream stream);
int s3_gather_rows(const void *d_src, int64_t n_src_rows, int64_t row_bytes, int64_t src_pitch_bytes,
                   const int32_t *d_ids /*[n] or NULL*/, int64_t n, void *d_dst, int64_t dst_pitch_bytes, s3_stream stream);


/* ---- multi-GPU (one process per GPU, RCCL over xGMI) -----------------------------------------------------
 * The interpolation shards over the generated cells with no collective.  The refine loop has one exchange per batch:
 * every rank evaluates s3_child_gain for its 1/W slice of the new cells (the reference farms the same work out to a
 * process pool, s_cube.py:207-241) and ONE grouped in-place all-gather returns metric and gain to everybody
 * (s3_comm_allgather_inplace); the captured metric (s_cube.py:317-336) is reduced as per-block partial sums
 * (s3_sumsq_blocks, every rank a share of the blocks), gathered the same way and added in block order on every rank
 * (s3_sum_ordered), so that the result is bit-identical for any number of ranks.
 * Bootstrap: rank 0 calls s3_comm_unique_id and hands the 128 bytes to the other ranks by any host channel. */
typedef struct s3_comm s3_comm;
#define S3_COMM_ID_BYTES 128
/* 1 when the RCCL library can be loaded in this process (creates nothing, never blocks): ranks exchange this BEFORE anybody
 * enters s3_comm_init, whose ncclCommInitRank is a collective without a timeout */
int s3_comm_available(void);
int s3_comm_unique_id(void *h_id_out, size_t bytes /* >= S3_COMM_ID_BYTES */);
int s3_comm_init(const void *h_id, size_t bytes, int rank, int world, s3_comm **out);   /* the current device is used */
void s3_comm_destroy(s3_comm *comm);
int s3_comm_rank(const s3_comm *comm, int *rank, int *world);
int s3_comm_allgather_inplace(s3_comm *comm, void *const *d_arrays, const size_t *bytes_per_rank, int n_arrays,
                              s3_stream stream);
/* every rank's block to ONE rank (the one that writes the export's file): rank r sends bytes_per_rank[r] bytes from d_send,
 * `root` receives the blocks in rank order into d_recv (d_recv may be NULL on the other ranks).  Point-to-point sends in one
 * group: every byte crosses xGMI once, where an all-gather would deliver all blocks to every rank. */
int s3_comm_gather_to_root(s3_comm *comm, const void *d_send, void *d_recv, const size_t *bytes_per_rank /*[world]*/, int root,
                           s3_stream stream);
int s3_comm_allreduce_f64(s3_comm *comm, double *d_buf, int64_t n, int op /*0 sum, 1 max*/, s3_stream stream);
/* captured-metric numerator in a form that does not depend on how the work is split: partial[b] = sum of metric^2 over the
 * leaf cells of the 1024-cell block b (fixed reduction tree inside a block), for blocks [block_begin, block_end);
 * s3_sum_ordered adds n values in index order (fixed tree) into d_out[0] */
#define S3_SUMSQ_BLOCK 1024
int s3_sumsq_blocks(const double *d_metric, const uint8_t *d_leaf, int64_t n_cells, int64_t block_begin, int64_t block_end,
                    double *d_partial /*[n_blocks total], entries [block_begin, block_end) written*/, s3_stream stream);
int s3_sum_ordered(const double *d_values, int64_t n, double *d_out, s3_stream stream);


/* ---- weighted SVD downstream of S^3 (utils.py:302-346, data.py:240-247) -------------------------
 * G[t,t] = sum_n w[n] (x[n,:] - mean[n]) (x[n,:] - mean[n])^T over the rows of the interpolated snapshot matrix x
 * [n_rows, t] (f64, row pitch in_stride), accumulated with v_mfma_f64_16x16x4_f64; centring and weighting are fused into
 * the operand staging.  d_mean: temporal mean per row (s3_row_moments), d_weight: cell area / volume per row.  The
 * eigen-decomposition of the small G is a library call (sparsespatialsampling_amd/svd.py); the mode GEMM: s3_centered_gemm. */
size_t s3_weighted_gram_scratch_bytes(int64_t n_rows, int64_t t);
int s3_weighted_gram(const double *d_x, int64_t n_rows, int64_t t, int64_t in_stride, const double *d_mean,
                     const double *d_weight, double *d_gram /*[t,t]*/, void *d_scratch, s3_stream stream);

/* the tall GEMMs of the same SVD, on the same matrix cores (rows of L may be pitched: the interpolated matrix as it stands):
 *   d_e == NULL:  C = (L - lmean 1^T) B                    modes U = (X - mean) V S^-1 (utils.py:302-346 takes U from the SVD),
 *                                                           coefficients A = (X - mean) V of the directions already found
 *   d_e != NULL:  C = (E - emean 1^T) - (L - lmean 1^T) B   residual (X - mean) - A V^T of a deflation level, in row chunks
 * L [m][k] row pitch l_stride, B [k][n] contiguous, E [m][n] row pitch e_stride, C [m][n] contiguous; means may be NULL. */
int s3_centered_gemm(const double *d_l, int64_t m, int64_t k, int64_t l_stride, const double *d_lmean, const double *d_b,
                     int64_t n, const double *d_e, int64_t e_stride, const double *d_emean, double *d_c, s3_stream stream);

/* Symmetric eigenproblem of the T x T Gram matrix (the step between s3_weighted_gram and the modes; the reference gets the whole
 * decomposition from flowtorch.analysis.SVD, utils.py:302-346).  The one LIBRARY call of the SVD path: rocSOLVER's dsyevd, looked up
 * with dlopen at the first call (s3_sym_eig_available: 1 when it loads); the scaling to a unit diagonal maximum and back is done
 * around it.  d_g [t][t] symmetric (read only), d_lam [t] eigenvalues ASCENDING, d_vec [t][t] row-major with eigenvector j in ROW j,
 * d_scratch s3_sym_eig_scratch_bytes(t) bytes.  Complete on return. */
int s3_sym_eig_available(void);
size_t s3_sym_eig_scratch_bytes(int64_t t);
int s3_sym_eig(const double *d_g, int64_t t, double *d_lam, double *d_vec, void *d_scratch, s3_stream stream);

/* ---- device-resident topology of the sampling tree (a9-a11, a15) ------------------------------------------
 * Neighbour links, shared-node numbering, invalid-cell bookkeeping and the final renumbering of s_cube.py:904-1536,
 * 721-728, 734-772, 1695-1736 on tables that live in HBM.  Every update takes the ORDERED id list the host decided on
 * (the iteration order of the reference's sets) as a host array, copies it, and runs asynchronously on the engine's own
 * stream; the result is the one of the reference's sequential procedure (stale links included).  s3_topo_sync waits.
 * The host engine libs3topo.so (csrc/topology.cpp) implements the same operations and serves the 2:1-balance mode. */
typedef struct s3_topo s3_topo;
int s3_topo_create(int dim, double width, const double *h_root_center /*[dim]*/, s3_topo **out);
void s3_topo_destroy(s3_topo *topo);
/* children of the listed parents in list order (s_cube.py:879-895 / 531-544); relink != 0: all their links once more after
 * the batch (uniform levels, s_cube.py:547-549).  *h_first (may be NULL) = id of the first new cell */
int s3_topo_refine(s3_topo *topo, const int64_t *h_parents, int64_t n, int relink, int64_t *h_first);
/* cell.parent.children = _assign_neighbors(cell.parent, ...) for every listed cell, in list order (s_cube.py:609, 834) */
int s3_topo_relink_parent_of(s3_topo *topo, const int64_t *h_cells, int64_t n);
/* children = [] and removal from the neighbours' rows, in list order (s_cube.py:721-728) */
int s3_topo_mark_invalid(s3_topo *topo, const int64_t *h_cells, int64_t n);
/* wait for the submitted updates; *h_error: 0 ok, 1 a listed parent was not a leaf or listed twice, 2 internal (a node
 * reference chain of a batch did not resolve) */
int s3_topo_sync(s3_topo *topo, int64_t *h_n_cells, int64_t *h_n_nodes, int *h_error);
/* device pointer of a table: 0 level i32, 1 parent i32, 2 first_child i32 (-1 leaf, -2 invalid), 3 nb i32 [n][8|26],
 * 4 node_idx i64 [n][2^d], 5 center f64 [n][d], 6 nodes f64 [n_nodes][d]; valid until the next update */
int s3_topo_table(s3_topo *topo, int which, const void **d_ptr);
/* renumbering (s_cube.py:734-772): leaves, nodes that keep a slot; then the grid into the caller's DEVICE arrays:
 * faces [n_leaf][2^d] (int32 when as32, else int64; leaves in ascending cell id), nodes [n_unique][d] */
int s3_topo_finalize(s3_topo *topo, int64_t *h_n_leaf, int64_t *h_n_unique_nodes);
int s3_topo_export_grid(s3_topo *topo, void *d_faces, int as32, double *d_nodes);
/* centres [n][d] / levels [n] (int64) of the listed cells into DEVICE arrays */
int s3_topo_gather_cells(s3_topo *topo, const int64_t *h_ids, int64_t n, double *d_centers, int64_t *d_levels);

#ifdef __cplusplus
}
#endif
#endif /* S3HIP_H */
