/*
 * TEST INFRASTRUCTURE ONLY.  OpenMP variants of two functions of the CPU oracle, used by
 * bench.py's cpu_baseline leg to time the same arithmetic on all host cores, and by tests/ to
 * check that they return exactly what the one-thread functions return.
 *
 * vo_schur_mt restates vo_schur (ba_oracle.c) and vo_two_way_matching_mt restates
 * vo_two_way_matching (match_oracle.c); the reference functions they follow are cited there
 * (src/core/reprojection_error.h:22-51, src/core/optimizer.cc:60-145,
 * src/core/superpoint_frontend.cc:163-211).  Every sum is taken in the order the one-thread
 * function takes it (observations in index order within a camera and within a landmark), so the
 * results are bit-identical; only the loops over independent landmarks / camera rows / matrix rows
 * are spread over threads.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void vo_evaluate(int Nc, int Np, int64_t No, const double *K4, const double *t, const double *rpy,
                 const double *pts, const int32_t *cam, const int32_t *pt, const double *uv,
                 double huber_a, double *cost, double *res, double *Jc, double *Jp);
float vo_desc_distance(const float *a, const float *b);

int vo_max_threads(void) { return omp_get_max_threads(); }

static void inv3_sym(const double *V, double *Vi)
{
    double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
    double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    double det = a * c00 + b * c01 + c * c02;
    double id = 1.0 / det;
    Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
    Vi[3] = Vi[1];    Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
    Vi[6] = Vi[2];    Vi[7] = Vi[5]; Vi[8] = (a * d - b * b) * id;
}

/* Counting sort of observation indices by key (stable: ascending observation index per key). */
static void group_by(const int32_t *key, int64_t No, int n_keys, int64_t *ptr, int64_t *ord)
{
    memset(ptr, 0, sizeof(int64_t) * ((size_t)n_keys + 1));
    for (int64_t o = 0; o < No; o++) ptr[key[o] + 1]++;
    for (int k = 0; k < n_keys; k++) ptr[k + 1] += ptr[k];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n_keys + 1));
    memcpy(fill, ptr, sizeof(int64_t) * ((size_t)n_keys + 1));
    for (int64_t o = 0; o < No; o++) ord[fill[key[o]]++] = o;
    free(fill);
}

/* vo_schur with points free, on `threads` OpenMP threads.  Outputs as vo_schur's S, rhs, Vinv,
 * gp (any may be NULL). */
void vo_schur_mt(int threads, int Nc, int Np, int64_t No, const double *K4, const double *t,
                 const double *rpy, const double *pts, const int32_t *cam, const int32_t *pt,
                 const double *uv, double huber_a, const double *dc, const double *dp, double *S,
                 double *rhs, double *Vinv, double *gp)
{
    if (threads < 1) threads = 1;
    int n = 6 * Nc;
    size_t no = (size_t)(No > 0 ? No : 1);
    double *res = (double *)malloc(sizeof(double) * 2 * no);
    double *Jc = (double *)malloc(sizeof(double) * 12 * no);
    double *Jp = (double *)malloc(sizeof(double) * 6 * no);
    double *W = (double *)malloc(sizeof(double) * 18 * no);      /* Jc^T Jp per observation */
    double *Z = (double *)malloc(sizeof(double) * 18 * no);      /* W Vinv per observation */
    double *gpl = (double *)malloc(sizeof(double) * 3 * ((size_t)Np + 1));
    double *Vil = (double *)calloc(9 * ((size_t)Np + 1), sizeof(double));
    int64_t *lptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)Np + 1));
    int64_t *lord = (int64_t *)malloc(sizeof(int64_t) * no);
    int64_t *cptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)Nc + 1));
    int64_t *cord = (int64_t *)malloc(sizeof(int64_t) * no);
    const int64_t chunk = 4096;
    int64_t n_chunks = (No + chunk - 1) / chunk;
#pragma omp parallel num_threads(threads)
    {
#pragma omp sections
        {
#pragma omp section
            group_by(pt, No, Np, lptr, lord);
#pragma omp section
            group_by(cam, No, Nc, cptr, cord);
        }
#pragma omp for schedule(dynamic, 4)
        for (int64_t c = 0; c < n_chunks; c++) {
            int64_t b = c * chunk, m = (No - b < chunk) ? No - b : chunk;
            vo_evaluate(Nc, Np, m, K4, t, rpy, pts, cam + b, pt + b, uv + 2 * b,
                        c == 0 ? huber_a : 0.0, NULL, res + 2 * b, Jc + 12 * b, Jp + 6 * b);
        }
        /* per landmark: V, gp, (V + D)^-1, and W, Z of its observations */
#pragma omp for schedule(dynamic, 256)
        for (int j = 0; j < Np; j++) {
            double V[9] = { 0 }, g[3] = { 0 };
            for (int64_t x = lptr[j]; x < lptr[j + 1]; x++) {
                const double *jp = Jp + 6 * lord[x], *r = res + 2 * lord[x];
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) V[3 * a + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
                    g[a] += jp[a] * r[0] + jp[3 + a] * r[1];
                }
            }
            memcpy(gpl + 3 * (size_t)j, g, sizeof(g));
            if (lptr[j + 1] == lptr[j]) continue;
            if (dp) { V[0] += dp[3 * j]; V[4] += dp[3 * j + 1]; V[8] += dp[3 * j + 2]; }
            double *Vi = Vil + 9 * (size_t)j;
            inv3_sym(V, Vi);
            for (int64_t x = lptr[j]; x < lptr[j + 1]; x++) {
                int64_t o = lord[x];
                const double *jc = Jc + 12 * o, *jp = Jp + 6 * o;
                double *w = W + 18 * o, *z = Z + 18 * o;
                for (int p = 0; p < 6; p++)
                    for (int q = 0; q < 3; q++) w[3 * p + q] = jc[p] * jp[q] + jc[6 + p] * jp[3 + q];
                for (int p = 0; p < 6; p++)
                    for (int q = 0; q < 3; q++)
                        z[3 * p + q] = w[3 * p] * Vi[q] + w[3 * p + 1] * Vi[3 + q] + w[3 * p + 2] * Vi[6 + q];
            }
        }
        /* per camera: its six rows of S and of rhs have one writer */
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < Nc; c++) {
            double U[36] = { 0 }, g[6] = { 0 };
            for (int64_t x = cptr[c]; x < cptr[c + 1]; x++) {
                const double *jc = Jc + 12 * cord[x], *r = res + 2 * cord[x];
                for (int a = 0; a < 6; a++) {
                    for (int b = 0; b < 6; b++) U[6 * a + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
                    g[a] += jc[a] * r[0] + jc[6 + a] * r[1];
                }
            }
            double rh[6];
            for (int a = 0; a < 6; a++) rh[a] = -g[a];
            if (S) {
                memset(S + (size_t)(6 * c) * n, 0, sizeof(double) * 6 * (size_t)n);
                for (int a = 0; a < 6; a++) {
                    for (int b = 0; b < 6; b++) S[(size_t)(6 * c + a) * n + 6 * c + b] = U[6 * a + b];
                    if (dc) S[(size_t)(6 * c + a) * n + 6 * c + a] += dc[6 * c + a];
                }
            }
            for (int64_t x = cptr[c]; x < cptr[c + 1]; x++) {
                int64_t o = cord[x];
                int j = pt[o];
                const double *z = Z + 18 * o, *gj = gpl + 3 * (size_t)j;
                for (int p = 0; p < 6; p++)
                    rh[p] += z[3 * p] * gj[0] + z[3 * p + 1] * gj[1] + z[3 * p + 2] * gj[2];
                if (!S) continue;
                for (int64_t y = lptr[j]; y < lptr[j + 1]; y++) {
                    int64_t o2 = lord[y];
                    const double *w = W + 18 * o2;
                    double *s = S + (size_t)(6 * c) * n + 6 * cam[o2];
                    for (int p = 0; p < 6; p++)
                        for (int q = 0; q < 6; q++)
                            s[(size_t)p * n + q] -= z[3 * p] * w[3 * q] + z[3 * p + 1] * w[3 * q + 1] +
                                                    z[3 * p + 2] * w[3 * q + 2];
                }
            }
            if (rhs) memcpy(rhs + 6 * c, rh, sizeof(rh));
        }
    }
    if (gp) memcpy(gp, gpl, sizeof(double) * 3 * (size_t)Np);
    if (Vinv) memcpy(Vinv, Vil, sizeof(double) * 9 * (size_t)Np);
    free(res); free(Jc); free(Jp); free(W); free(Z); free(gpl); free(Vil);
    free(lptr); free(lord); free(cptr); free(cord);
}

/* vo_two_way_matching on `threads` OpenMP threads: rows of the distance matrix, the row minima
 * and the column minima (in blocks of 64 columns) are independent. */
int vo_two_way_matching_mt(int threads, const float *d0, int n0, const float *d1, int n1,
                           float nn_thresh, float ratio, int32_t *out_i, int32_t *out_j, float *out_d)
{
    if (n0 <= 0 || n1 <= 0) return 0;
    if (threads < 1) threads = 1;
    float *dmat = (float *)malloc(sizeof(float) * (size_t)n0 * n1);
    int32_t *idx0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n0);
    int32_t *idx1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n1);
    float *second = (float *)malloc(sizeof(float) * (size_t)n0);
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < n0; i++) {
            float *row = dmat + (size_t)i * n1;
            for (int j = 0; j < n1; j++)
                row[j] = vo_desc_distance(d0 + (size_t)i * 256, d1 + (size_t)j * 256);
            int best = 0; float bv = row[0], sv = INFINITY;
            for (int j = 1; j < n1; j++) {
                float v = row[j];
                if (v < bv) { sv = bv; bv = v; best = j; }
                else if (v < sv) sv = v;
            }
            idx0[i] = best; second[i] = sv;
        }
#pragma omp for schedule(dynamic, 1)
        for (int jb = 0; jb < n1; jb += 64) {
            int je = jb + 64 < n1 ? jb + 64 : n1;
            float bv[64]; int32_t bi[64];
            for (int j = jb; j < je; j++) { bv[j - jb] = dmat[j]; bi[j - jb] = 0; }
            for (int i = 1; i < n0; i++) {
                const float *row = dmat + (size_t)i * n1;
                for (int j = jb; j < je; j++)
                    if (row[j] < bv[j - jb]) { bv[j - jb] = row[j]; bi[j - jb] = i; }
            }
            for (int j = jb; j < je; j++) idx1[j] = bi[j - jb];
        }
    }
    int n = 0;
    for (int i = 0; i < n0; i++) {
        int j = idx0[i];
        float d = dmat[(size_t)i * n1 + j];
        if (idx1[j] != i || !(d < nn_thresh)) continue;
        if (ratio < 1.0f && !(d < ratio * second[i])) continue;
        out_i[n] = i; out_j[n] = j; out_d[n] = d; n++;
    }
    free(dmat); free(idx0); free(idx1); free(second);
    return n;
}
