/* A torch-free host of libs3hip.so in plain C: what a maintainer binding the library from another language would call.
 * KNN index -> neighbour table -> weights -> interpolation (direct and planned) on a small seeded case, checked against a
 * scalar loop in this file (same operation order: f64 FMA in neighbour order).  Built and run by tests/test_abi.py on the
 * GPU box:  gcc -std=c11 -O1 -I include tests/native/c_host.c -o c_host -L sparsespatialsampling_amd -ls3hip -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "s3hip.h"

#define CHECK(call)                                                             \
    do {                                                                        \
        int rc_ = (call);                                                       \
        if (rc_ != 0) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, s3_last_error());     \
            return 1;                                                           \
        }                                                                       \
    } while (0)

static double lcg(uint64_t *s) {
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(*s >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
    /* "nosvd": without the weighted-SVD chain (its one library call loads rocSOLVER, a 0.9-GB file that a fresh machine pages in for minutes) */
    const int with_svd = !(argc > 1 && strcmp(argv[1], "nosvd") == 0);
    enum { N = 5000, NC = 700, K = 26, DIM = 3, T = 40 };
    int n_dev = 0;
    CHECK(s3_device_count(&n_dev));
    if (n_dev < 1) { fprintf(stderr, "no device\n"); return 2; }
    CHECK(s3_set_device(0));
    if (s3_abi_version() < 1) return 3;

    uint64_t seed = 42;
    double *pts = malloc(sizeof(double) * N * DIM), *q = malloc(sizeof(double) * NC * DIM);
    float *data = malloc(sizeof(float) * N * T);
    for (int i = 0; i < N * DIM; ++i) pts[i] = lcg(&seed);
    for (int i = 0; i < NC * DIM; ++i) q[i] = lcg(&seed);
    for (int i = 0; i < N * T; ++i) data[i] = (float)(lcg(&seed) - 0.5);

    void *d_pts, *d_q, *d_idx, *d_dist, *d_w, *d_data, *d_out, *d_out2;
    CHECK(s3_malloc(&d_pts, sizeof(double) * N * DIM));
    CHECK(s3_malloc(&d_q, sizeof(double) * NC * DIM));
    CHECK(s3_malloc(&d_idx, sizeof(int32_t) * NC * K));
    CHECK(s3_malloc(&d_dist, sizeof(double) * NC * K));
    CHECK(s3_malloc(&d_w, sizeof(double) * NC * K));
    CHECK(s3_malloc(&d_data, sizeof(float) * N * T));
    CHECK(s3_malloc(&d_out, sizeof(double) * NC * T));
    CHECK(s3_malloc(&d_out2, sizeof(double) * NC * T));
    fprintf(stderr, "c_host: uploading\n");
    CHECK(s3_memcpy_h2d(d_pts, pts, sizeof(double) * N * DIM, NULL));
    CHECK(s3_memcpy_h2d(d_q, q, sizeof(double) * NC * DIM, NULL));
    CHECK(s3_memcpy_h2d(d_data, data, sizeof(float) * N * T, NULL));

    fprintf(stderr, "c_host: kernels\n");
    s3_knn *knn = NULL;
    CHECK(s3_knn_create(d_pts, N, DIM, 0.0, NULL, &knn));
    CHECK(s3_knn_query(knn, d_q, NC, K, d_idx, d_dist, NULL));
    CHECK(s3_idw_weights(d_dist, NC, K, d_w, NULL));
    CHECK(s3_interp(d_w, d_idx, NC, K, d_data, S3_DTYPE_F32, N, T, d_out, NULL));
    s3_interp_plan *plan = NULL;
    CHECK(s3_interp_plan_create(d_idx, NC, K, N, d_q, DIM, 0, NULL, &plan));
    CHECK(s3_interp_planned(plan, d_w, d_data, S3_DTYPE_F32, T, T, d_out2, NULL));
    CHECK(s3_stream_synchronize(NULL));

    int32_t *idx = malloc(sizeof(int32_t) * NC * K);
    double *dist = malloc(sizeof(double) * NC * K), *w = malloc(sizeof(double) * NC * K);
    double *out = malloc(sizeof(double) * NC * T), *out2 = malloc(sizeof(double) * NC * T);
    fprintf(stderr, "c_host: downloading\n");
    CHECK(s3_memcpy_d2h(idx, d_idx, sizeof(int32_t) * NC * K, NULL));
    CHECK(s3_memcpy_d2h(dist, d_dist, sizeof(double) * NC * K, NULL));
    CHECK(s3_memcpy_d2h(w, d_w, sizeof(double) * NC * K, NULL));
    CHECK(s3_memcpy_d2h(out, d_out, sizeof(double) * NC * T, NULL));
    CHECK(s3_memcpy_d2h(out2, d_out2, sizeof(double) * NC * T, NULL));
    CHECK(s3_stream_synchronize(NULL));

    long bad = 0;
    for (int c = 0; c < NC; ++c) {
        /* nearest neighbour by brute force must be the table's first entry, distances ascending */
        int best = 0;
        double bd = 1e300;
        for (int i = 0; i < N; ++i) {
            double s = 0;
            for (int j = 0; j < DIM; ++j) { double t = q[c * DIM + j] - pts[i * DIM + j]; s += t * t; }
            if (s < bd) { bd = s; best = i; }
        }
        bad += idx[c * K] != best || dist[c * K] != sqrt(bd);
        for (int m = 1; m < K; ++m) bad += dist[c * K + m] < dist[c * K + m - 1];
        double sw = 0;
        for (int m = 0; m < K; ++m) sw += w[c * K + m];
        bad += fabs(sw - 1.0) > 1e-14;
        for (int t = 0; t < T; ++t) {
            double acc = 0.0;
            for (int m = 0; m < K; ++m) acc = fma(w[c * K + m], (double)data[(size_t)idx[c * K + m] * T + t], acc);
            bad += out[c * T + t] != acc || out2[c * T + t] != acc;
        }
    }
    /* downstream of the interpolation, still without torch (utils.py:302-346): weighted SVD of the interpolated matrix
     * out [NC][T] by the method of snapshots -- row means, Gram matrix on the f64 matrix cores, the symmetric eigenproblem (the one
     * library call, rocSOLVER behind s3_sym_eig), modes U = (X - mean) V S^-1 -- checked on the host: G v = lambda v, V orthonormal,
     * U^T A U = I for the leading modes */
    long bad_svd = 0;
    fprintf(stderr, "c_host: svd chain\n");
    if (!with_svd) {
        printf("c_host: SVD chain not asked for\n");
    } else if (s3_sym_eig_available()) {
        enum { R = 8 };
        void *d_mean, *d_area, *d_gram, *d_lam, *d_vec, *d_scr, *d_escr, *d_b, *d_u;
        double *area = malloc(sizeof(double) * NC);
        for (int c = 0; c < NC; ++c) area[c] = 0.5 + lcg(&seed);
        CHECK(s3_malloc(&d_mean, sizeof(double) * NC));
        CHECK(s3_malloc(&d_area, sizeof(double) * NC));
        CHECK(s3_malloc(&d_gram, sizeof(double) * T * T));
        CHECK(s3_malloc(&d_lam, sizeof(double) * T));
        CHECK(s3_malloc(&d_vec, sizeof(double) * T * T));
        CHECK(s3_malloc(&d_scr, s3_weighted_gram_scratch_bytes(NC, T)));
        CHECK(s3_malloc(&d_escr, s3_sym_eig_scratch_bytes(T)));
        CHECK(s3_malloc(&d_b, sizeof(double) * T * R));
        CHECK(s3_malloc(&d_u, sizeof(double) * NC * R));
        fprintf(stderr, "c_host:   area up\n");
        CHECK(s3_memcpy_h2d(d_area, area, sizeof(double) * NC, NULL));
        fprintf(stderr, "c_host:   moments\n");
        CHECK(s3_row_moments(d_out, S3_DTYPE_F64, NC, T, T, 1, d_mean, NULL, NULL));
        fprintf(stderr, "c_host:   gram\n");
        CHECK(s3_weighted_gram(d_out, NC, T, T, d_mean, d_area, d_gram, d_scr, NULL));
        CHECK(s3_stream_synchronize(NULL));
        fprintf(stderr, "c_host:   eig\n");
        CHECK(s3_sym_eig(d_gram, T, d_lam, d_vec, d_escr, NULL));
        fprintf(stderr, "c_host:   eig done, downloading\n");
        double *gram = malloc(sizeof(double) * T * T), *lam = malloc(sizeof(double) * T), *vec = malloc(sizeof(double) * T * T);
        double *mean = malloc(sizeof(double) * NC), *b = malloc(sizeof(double) * T * R), *u = malloc(sizeof(double) * NC * R);
        CHECK(s3_memcpy_d2h(gram, d_gram, sizeof(double) * T * T, NULL));
        CHECK(s3_memcpy_d2h(lam, d_lam, sizeof(double) * T, NULL));
        CHECK(s3_memcpy_d2h(vec, d_vec, sizeof(double) * T * T, NULL));
        CHECK(s3_memcpy_d2h(mean, d_mean, sizeof(double) * NC, NULL));
        /* the Gram matrix itself against a plain host sum */
        for (int i = 0; i < T; i += 7)
            for (int j = 0; j < T; j += 5) {
                double g = 0.0;
                for (int c = 0; c < NC; ++c) g += area[c] * (out[c * T + i] - mean[c]) * (out[c * T + j] - mean[c]);
                bad_svd += fabs(g - gram[i * T + j]) > 1e-11 * fabs(gram[0]) + 1e-300;
            }
        const double top = lam[T - 1];
        for (int j = 0; j < T; ++j) {                 /* eigenvector j = ROW j of vec, eigenvalues ascending */
            bad_svd += j > 0 && lam[j] < lam[j - 1];
            double res = 0.0;
            for (int i = 0; i < T; ++i) {
                double gv = 0.0;
                for (int m = 0; m < T; ++m) gv += gram[i * T + m] * vec[j * T + m];
                res = fmax(res, fabs(gv - lam[j] * vec[j * T + i]));
            }
            bad_svd += res > 1e-12 * top;
            for (int l = 0; l <= j; ++l) {
                double dot = 0.0;
                for (int m = 0; m < T; ++m) dot += vec[j * T + m] * vec[l * T + m];
                bad_svd += fabs(dot - (l == j ? 1.0 : 0.0)) > 1e-12;
            }
        }
        /* the R leading modes: B[:, r] = v_r / s_r (descending), U = (X - mean) B; U^T diag(area) U = I */
        for (int r = 0; r < R; ++r)
            for (int m = 0; m < T; ++m) b[m * R + r] = vec[(T - 1 - r) * T + m] / sqrt(lam[T - 1 - r]);
        fprintf(stderr, "c_host:   modes\n");
        CHECK(s3_memcpy_h2d(d_b, b, sizeof(double) * T * R, NULL));
        CHECK(s3_centered_gemm(d_out, NC, T, T, d_mean, d_b, R, NULL, 0, NULL, d_u, NULL));
        CHECK(s3_memcpy_d2h(u, d_u, sizeof(double) * NC * R, NULL));
        for (int r = 0; r < R; ++r)
            for (int l = 0; l <= r; ++l) {
                double dot = 0.0;
                for (int c = 0; c < NC; ++c) dot += area[c] * u[c * R + r] * u[c * R + l];
                bad_svd += fabs(dot - (l == r ? 1.0 : 0.0)) > 1e-9;
            }
        void *more[] = {d_mean, d_area, d_gram, d_lam, d_vec, d_scr, d_escr, d_b, d_u};
        for (unsigned i = 0; i < sizeof(more) / sizeof(more[0]); ++i) CHECK(s3_free(more[i]));
        printf("c_host: Gram -> s3_sym_eig -> modes through the C ABI: s_max %.6g, s_min %.3g, mismatches %ld\n", sqrt(top), sqrt(fabs(lam[0])), bad_svd);
    } else {
        printf("c_host: rocSOLVER not loadable in this process: SVD chain skipped\n");
    }
    bad += bad_svd;
    int64_t n_tiles = 0, n_rows = 0;
    CHECK(s3_interp_plan_info(plan, &n_tiles, &n_rows));
    s3_interp_plan_destroy(plan);
    s3_knn_destroy(knn);
    /* error path: bad arguments are refused on the host, with a message */
    int rc = s3_interp(d_w, d_idx, NC, 65, d_data, S3_DTYPE_F32, N, T, d_out, NULL);
    bad += rc != S3_EINVAL || s3_last_error()[0] == '\0';
    void *bufs[] = {d_pts, d_q, d_idx, d_dist, d_w, d_data, d_out, d_out2};
    for (unsigned i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) CHECK(s3_free(bufs[i]));
    printf("c_host: %d cells x %d snapshots, %lld tiles / %lld staged rows, mismatches %ld\n", NC, T, (long long)n_tiles,
           (long long)n_rows, bad);
    fflush(stdout);
    fprintf(stderr, "c_host: shutting the library's transfer threads down\n");
    s3_shutdown();          /* the staged copies of pageable memory started the library's lane threads: join them before exit */
    fprintf(stderr, "c_host: done\n");
    return bad ? 4 : 0;
}
