/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement of the descriptor matchers of scomup/visual_slam
 * (reference paths are relative to the reference tree).
 *
 * PARITY UNPINNED: the reference holds no tests or golden vectors for these functions and needs
 * OpenCV (absent here) to build.  cv::Mat::dot is restated as its documented contract: a double
 * accumulation in index order, narrowed to float at the call site
 * (src/core/superpoint_frontend.cc:181); cv::minMaxLoc as "first minimum in scan order".
 * Pinned by numpy brute-force fixtures under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VO_DESC_DIM 256

/* score(): superpoint_frontend.cc:181-183, tracking.cc:24-30, frame.cc:61-63. */
static float desc_distance(const float *a, const float *b)
{
    double acc = 0.0;
    for (int k = 0; k < VO_DESC_DIM; k++) acc += (double)a[k] * (double)b[k];
    float score = (float)acc;
    score = (score < -1.0f) ? -1.0f : (1.0f < score) ? 1.0f : score;
    return sqrtf(2.0f - 2.0f * score);
}

float vo_desc_distance(const float *a, const float *b) { return desc_distance(a, b); }

/* Dense distance matrix; xy0/xy1 NULL = twoWayMatching (superpoint_frontend.cc:171-186), non-NULL
 * = getStereoMatching's geometric mask: entries with |yl-yr| > 20 or xl < xr keep the initial
 * value 1.0 (superpoint_frontend.cc:117,126-129). */
void vo_distance_matrix(const float *d0, int n0, const float *d1, int n1, const int32_t *xy0,
                        const int32_t *xy1, float *dmat)
{
    for (int i = 0; i < n0; i++)
        for (int j = 0; j < n1; j++) {
            float v = 1.0f;
            int masked = 0;
            if (xy0) {
                if (abs(xy0[2 * i + 1] - xy1[2 * j + 1]) > 20) masked = 1;
                if (xy0[2 * i] < xy1[2 * j]) masked = 1;
            }
            if (!masked) v = desc_distance(d0 + (size_t)i * VO_DESC_DIM, d1 + (size_t)j * VO_DESC_DIM);
            dmat[(size_t)i * n1 + j] = v;
        }
}

/* Row argmin, column argmin (first minimum), mutual check and threshold
 * (superpoint_frontend.cc:188-209).  ratio < 1 additionally requires d_best < ratio * d_second
 * over the row (extension, not in the reference; ratio >= 1 disables it).
 * Outputs are ordered by i; returns the number of matches.  out_* sized n0. */
int vo_mutual_nn(const float *dmat, int n0, int n1, float nn_thresh, float ratio, int32_t *out_i,
                 int32_t *out_j, float *out_d)
{
    if (n0 <= 0 || n1 <= 0) return 0;
    int32_t *idx0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n0);
    int32_t *idx1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n1);
    float *second = (float *)malloc(sizeof(float) * (size_t)n0);
    for (int i = 0; i < n0; i++) {
        int best = 0; float bv = dmat[(size_t)i * n1], sv = INFINITY;
        for (int j = 1; j < n1; j++) {
            float v = dmat[(size_t)i * n1 + j];
            if (v < bv) { sv = bv; bv = v; best = j; }
            else if (v < sv) sv = v;
        }
        idx0[i] = best; second[i] = sv;
    }
    for (int j = 0; j < n1; j++) {
        int best = 0; float bv = dmat[j];
        for (int i = 1; i < n0; i++) {
            float v = dmat[(size_t)i * n1 + j];
            if (v < bv) { bv = v; best = i; }
        }
        idx1[j] = best;
    }
    int n = 0;
    for (int i = 0; i < n0; i++) {
        int j = idx0[i];
        float d = dmat[(size_t)i * n1 + j];
        if (idx1[j] != i || !(d < nn_thresh)) continue;
        if (ratio < 1.0f && !(d < ratio * second[i])) continue;
        out_i[n] = i; out_j[n] = j; out_d[n] = d; n++;
    }
    free(idx0); free(idx1); free(second);
    return n;
}

int vo_two_way_matching(const float *d0, int n0, const float *d1, int n1, float nn_thresh,
                        float ratio, int32_t *out_i, int32_t *out_j, float *out_d)
{
    if (n0 <= 0 || n1 <= 0) return 0;
    float *dmat = (float *)malloc(sizeof(float) * (size_t)n0 * n1);
    vo_distance_matrix(d0, n0, d1, n1, NULL, NULL, dmat);
    int n = vo_mutual_nn(dmat, n0, n1, nn_thresh, ratio, out_i, out_j, out_d);
    free(dmat);
    return n;
}

int vo_stereo_matching(const float *d0, const int32_t *xy0, int n0, const float *d1,
                       const int32_t *xy1, int n1, float nn_thresh, float ratio, int32_t *out_i,
                       int32_t *out_j, float *out_d)
{
    if (n0 <= 0 || n1 <= 0) return 0;
    float *dmat = (float *)malloc(sizeof(float) * (size_t)n0 * n1);
    vo_distance_matrix(d0, n0, d1, n1, xy0, xy1, dmat);
    int n = vo_mutual_nn(dmat, n0, n1, nn_thresh, ratio, out_i, out_j, out_d);
    free(dmat);
    return n;
}

/* Frame::computePoint3d's one-way masked nearest neighbour (src/core/frame.cc:47-71): for each
 * left keypoint the first strict minimum over unmasked right keypoints, or -1. */
void vo_stereo_one_way(const float *d0, const int32_t *xy0, int n0, const float *d1,
                       const int32_t *xy1, int n1, int32_t *minloc, float *mind)
{
    for (int i = 0; i < n0; i++) {
        int loc = -1; double mn = 999;
        for (int j = 0; j < n1; j++) {
            if (abs(xy0[2 * i + 1] - xy1[2 * j + 1]) > 20) continue;
            if (xy0[2 * i] < xy1[2 * j]) continue;
            float s = desc_distance(d0 + (size_t)i * VO_DESC_DIM, d1 + (size_t)j * VO_DESC_DIM);
            if (mn > s) { mn = s; loc = j; }
        }
        minloc[i] = loc; mind[i] = (float)mn;
    }
}

/* Tracking::searchProjection's greedy claim (src/core/tracking.cc:462-495) given the integer
 * reprojections: tracked point i scans keypoints j in order, skipping claimed ones and those
 * farther than `radius` pixels, keeps the first strict minimum of the descriptor distance and
 * claims it when that distance is < accept (0.8 in the reference).  reproj_xy: n_tp x 2 ints
 * (cv::Point truncation already applied); width2/height2 = 2*cx, 2*cy bounds (tracking.cc:467-470).
 * claimed (n_keys, in/out): non-zero = keypoint already owns a tracked point.  tracks: n_tp. */
void vo_search_projection(const int32_t *reproj_xy, const float *tp_desc, int n_tp,
                          const int32_t *key_xy, const float *key_desc, int n_keys, float width2,
                          float height2, float radius, double accept, uint8_t *claimed,
                          int32_t *tracks)
{
    for (int i = 0; i < n_tp; i++) {
        tracks[i] = -1;
        int rx = reproj_xy[2 * i], ry = reproj_xy[2 * i + 1];
        if ((float)rx > width2 || rx < 0) continue;
        if ((float)ry > height2 || ry < 0) continue;
        float best = 999; int loc = -1;
        for (int j = 0; j < n_keys; j++) {
            if (claimed[j]) continue;
            int dx = key_xy[2 * j] - rx, dy = key_xy[2 * j + 1] - ry;
            double dist = sqrt((double)(dx * dx + dy * dy));
            if (dist > (double)radius) continue;
            float s = desc_distance(tp_desc + (size_t)i * VO_DESC_DIM, key_desc + (size_t)j * VO_DESC_DIM);
            if (best > s) { best = s; loc = j; }
        }
        if (loc != -1 && (double)best < accept) { tracks[i] = loc; claimed[loc] = 1; }
    }
}
