/*
 * s3h5.h -- C ABI of libs3h5.so: the HDF5 sink / source of the S^3 export path, on the HDF5 C library (no h5py).
 *
 * Replaces, for the hot path's output, the h5py calls of the reference's writer and loader
 * (sparseSpatialSampling/data.py:361-430 Datawriter.write_data -> one create_dataset per call;
 *  data.py:22-300 Dataloader -> File.get(...)[()]; export.py:283-299 one dataset per write time and field) with
 *  - synchronous dataset writes / reads / listings, and
 *  - an asynchronous batch writer: the datasets of a whole snapshot batch (`data/<t>/<field>_center`, same on-disk layout)
 *    are queued with one call and written by a background thread while the caller interpolates and downloads the next
 *    batch.
 * Plain pointers and sizes; thread-compatible per file handle; every call into HDF5 is serialised inside the library.
 * Error convention: 0 = ok, negative = failure (message: s3h5_last_error()); S3H5_EEXIST = the dataset exists already.
 */
#ifndef S3H5_H
#define S3H5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3H5_OK 0
#define S3H5_EINVAL (-1)
#define S3H5_EIO (-2)
#define S3H5_ENOENT (-3)
#define S3H5_EEXIST (-17)

/* element types of the caller's arrays */
#define S3H5_F32 0
#define S3H5_F64 1
#define S3H5_I32 2
#define S3H5_I64 3
#define S3H5_U8 4
#define S3H5_MAX_DIMS 8

typedef struct s3h5_file s3h5_file;

const char *s3h5_last_error(void);
int s3h5_version(unsigned *major, unsigned *minor, unsigned *release);          /* of the HDF5 library in use */

/* mode "w" (create / truncate), "a" (read-write, create if missing), "r" (read only) -- h5py.File modes of data.py:330 */
int s3h5_open(const char *path, const char *mode, s3h5_file **out);
int s3h5_close(s3h5_file *f);                                                  /* drains the queue first */

/* one dataset `path` ("grid/faces", "constant/levels", "data/0.1/p_center"): intermediate groups are created,
 * ndim = 0 writes a scalar (constant/size_initial_cell, export.py:262).  S3H5_EEXIST if it is there already. */
int s3h5_write(s3h5_file *f, const char *path, int dtype, int ndim, const int64_t *dims, const void *data);

/* the datasets of one snapshot batch: for i < n_snapshots the dataset `data/<times[i]>/<name>` of shape dims[0..ndim)
 * is written from h_base + i * stride_bytes (snapshot-major host buffer).  Asynchronous: the call
 * returns once the batch is queued; h_base must stay valid until s3h5_flush / s3h5_close / the next s3h5_wait_buffer on it.
 * Datasets that exist already are skipped and counted (reference data.py:404-407 logs and skips).  Datasets of a megabyte
 * or more are created by the library (contiguous, allocated at creation) and their values written by several threads
 * straight into the file at the offsets the library reports. */
int s3h5_write_snapshots_async(s3h5_file *f, const char *group /* "data" */, const char *const *times, int64_t n_snapshots,
                               const char *name, int dtype, int ndim, const int64_t *dims, const void *h_base,
                               int64_t stride_bytes);
/* the same for a batch whose values are still on their way into h_base (ExportData queues the device-to-host copy of an
 * interpolated batch and hands the buffer over at once, so that the copy overlaps the upload of the next batch): the writer
 * starts on the batch once *h_ready >= ready_value -- a 4-byte word in host memory that the producer writes behind the copy,
 * in the copy's stream.  h_ready == NULL: the values are there already. */
int s3h5_write_snapshots_async_when(s3h5_file *f, const char *group, const char *const *times, int64_t n_snapshots,
                                    const char *name, int dtype, int ndim, const int64_t *dims, const void *h_base,
                                    int64_t stride_bytes, const int32_t *h_ready, int32_t ready_value);
/* wait until every queued write has been carried out; *n_skipped (optional) = datasets skipped because they existed */
int s3h5_flush(s3h5_file *f, int64_t *n_skipped);
/* wait until no queued write reads from [h_base, h_base + bytes) any more (before the buffer is overwritten) */
int s3h5_wait_buffer(s3h5_file *f, const void *h_base, size_t bytes);

/* queries */
int s3h5_exists(s3h5_file *f, const char *path);                               /* 1 / 0 / negative error */
int s3h5_shape(s3h5_file *f, const char *path, int *dtype /* S3H5_* or -1 */, int *ndim, int64_t *dims /*[S3H5_MAX_DIMS]*/);
int s3h5_read(s3h5_file *f, const char *path, int dtype, void *out, int64_t n_elements);  /* converted to `dtype` */
/* names of the members of a group in name order, '\n'-separated into buf; *needed = bytes required (call twice) */
int s3h5_list(s3h5_file *f, const char *group, char *buf, size_t buf_bytes, size_t *needed, int64_t *n_members);

#ifdef __cplusplus
}
#endif
#endif /* S3H5_H */
