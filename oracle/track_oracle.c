/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement of the per-frame tracking helpers that sit beside
 * the matcher and the optimiser in scomup/visual_slam (reference paths are relative to the
 * reference tree): Tracking::reprojection, Tracking::removeOutlier, Frame::computePoint3d.
 *
 * PARITY UNPINNED: the reference holds no tests or golden vectors for these functions and needs
 * Eigen / OpenCV (absent here) to build.  The float arithmetic is restated from Eigen's
 * documented evaluation of fixed-size expressions (version unpinned):
 *   - quaternion * vector:  uv = q.vec x v;  uv += uv;  v + q.w * uv + q.vec x uv
 *   - a 3x3 * 3x1 product sums each row as c0 + (c1 + c2)
 *   - Matrix3f::inverse() = cofactors times 1/det, det = c0 + (c1 + c2) over column 0
 * every operation rounded to float on its own (built with -ffp-contract=off), and from the C++
 * conversions at the call sites (cv::Point(float, float) truncates toward zero).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define VO_DESC_DIM 256

float vo_desc_distance(const float *a, const float *b);

/* pose layout: t[3], then quaternion w, x, y, z (transform::Rigid3f, rigid_transform.h:101-103) */

static void cross3f(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen::Quaternionf * Vector3f */
static void quat_rotate_f(const float *q /* w x y z */, const float *v, float *o)
{
    float uv[3], c[3];
    cross3f(q + 1, v, uv);
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    cross3f(q + 1, uv, c);
    for (int k = 0; k < 3; k++) o[k] = (v[k] + q[0] * uv[k]) + c[k];
}

/* Rigid3f * Vector3f (rigid_transform.h:114-119) */
static void rigid_apply_f(const float *pose, const float *p, float *o)
{
    float r[3];
    quat_rotate_f(pose + 3, p, r);
    for (int k = 0; k < 3; k++) o[k] = r[k] + pose[k];
}

/* Rigid3f::inverse() (rigid_transform.h:75-79) */
static void rigid_inverse_f(const float *pose, float *inv)
{
    inv[3] = pose[3]; inv[4] = -pose[4]; inv[5] = -pose[5]; inv[6] = -pose[6];
    float r[3];
    quat_rotate_f(inv + 3, pose, r);
    for (int k = 0; k < 3; k++) inv[k] = -r[k];
}

void vo_rigid_apply_f(const float *pose, const float *p, float *o) { rigid_apply_f(pose, p, o); }
void vo_rigid_inverse_f(const float *pose, float *inv) { rigid_inverse_f(pose, inv); }

/* Tracking::reprojection (tracking.cc:70-78): K4 = fx, fy, cx, cy; K * (x, y, 1) with the zero
 * entries of K contributing exact zeros; division by the homogeneous 1; truncation to int. */
static void reproject_f(const float *pose, const float *K4, const float *Pw, int32_t *xy)
{
    float Pc[3];
    rigid_apply_f(pose, Pw, Pc);
    const float lx = Pc[0] / Pc[1], ly = -Pc[2] / Pc[1];
    const float rz = 0.0f * lx + (0.0f * ly + 1.0f * 1.0f);
    const float rx = (K4[0] * lx + (0.0f * ly + K4[2] * 1.0f)) / rz;
    const float ry = (0.0f * lx + (K4[1] * ly + K4[3] * 1.0f)) / rz;
    /* cv::Point(float, float): float -> int conversion; out-of-range and NaN are undefined in
     * C++, here they saturate / map to INT32_MIN so that the bounds test rejects them */
    xy[0] = (rx >= -2147483648.0f && rx < 2147483648.0f) ? (int32_t)rx : INT32_MIN;
    xy[1] = (ry >= -2147483648.0f && ry < 2147483648.0f) ? (int32_t)ry : INT32_MIN;
}

void vo_reproject_points(const float *pose, const float *K4, const float *Pw, int n,
                         int32_t *xy)
{
    for (int i = 0; i < n; i++) reproject_f(pose, K4, Pw + 3 * (size_t)i, xy + 2 * (size_t)i);
}

/* Tracking::removeOutlier (tracking.cc:346-377) for one frame: tracks[i] = -1 where the key and
 * the integer reprojection are more than max_dist pixels apart.  Returns the number removed. */
int vo_remove_outliers(const float *pose, const float *K4, const float *Pw, int n_tp,
                       const int32_t *key_xy, double max_dist, int32_t *tracks)
{
    int removed = 0;
    for (int i = 0; i < n_tp; i++) {
        const int key_id = tracks[i];
        if (key_id == -1) continue;
        int32_t re[2];
        reproject_f(pose, K4, Pw + 3 * (size_t)i, re);
        /* the reference subtracts and multiplies ints (overflow undefined); widened here */
        const double dx = (double)key_xy[2 * (size_t)key_id] - (double)re[0];
        const double dy = (double)key_xy[2 * (size_t)key_id + 1] - (double)re[1];
        const double dist = sqrt(dx * dx + dy * dy);
        if (dist > max_dist) { tracks[i] = -1; removed++; }
    }
    return removed;
}

/* Frame::setFrameParam's inv_K_ (frame.cc:97-102): Eigen's 3x3 inverse of
 * [[fx 0 cx][0 fy cy][0 0 1]], entries that are products of zeros left as 0. */
void vo_inv_k3(const float *K4, float *invK /* 9, row-major */)
{
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    /* cofactor<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1), i1 = (i+1)%3, i2 = (i+2)%3, same for j */
    const float c00 = fy * 1.0f - cy * 0.0f;
    const float c10 = 0.0f * cx - 1.0f * 0.0f;
    const float c20 = 0.0f * cy - cx * fy;
    const float det = c00 * fx + (c10 * 0.0f + c20 * 0.0f);
    const float invdet = 1.0f / det;
    invK[0] = c00 * invdet;                               /* row 0 = column-0 cofactors / det */
    invK[1] = c10 * invdet;
    invK[2] = c20 * invdet;
    invK[3] = (cy * 0.0f - 0.0f * 1.0f) * invdet;         /* cofactor<0,1> */
    invK[4] = (1.0f * fx - 0.0f * cx) * invdet;           /* cofactor<1,1> */
    invK[5] = (cx * 0.0f - fx * cy) * invdet;             /* cofactor<2,1> */
    invK[6] = (0.0f * 0.0f - fy * 0.0f) * invdet;         /* cofactor<0,2> */
    invK[7] = (0.0f * 0.0f - 0.0f * fx) * invdet;         /* cofactor<1,2> */
    invK[8] = (fx * fy - 0.0f * 0.0f) * invdet;           /* cofactor<2,2> */
}

/* Frame::computePoint3d (frame.cc:45-86) for every left keypoint: one-way nearest right
 * keypoint among those with |yl - yr| <= 20 and xl >= xr (first strict minimum), disparity > 1,
 * depth = bf / disparity, back-projection through inv_K and the inverse pose.
 * good[i] = 1 and Pw[3i..] set when a point was made; minloc[i] = chosen right keypoint or -1. */
void vo_compute_points3d(const float *d0, const int32_t *xy0, int n0, const float *d1,
                         const int32_t *xy1, int n1, const float *K4, float bf,
                         const float *pose, uint8_t *good, float *Pw, int32_t *minloc)
{
    float invK[9], inv_pose[7];
    vo_inv_k3(K4, invK);
    rigid_inverse_f(pose, inv_pose);
    for (int i = 0; i < n0; i++) {
        int loc = -1; double mn = 999;
        const int xl = xy0[2 * (size_t)i], yl = xy0[2 * (size_t)i + 1];
        for (int j = 0; j < n1; j++) {
            const int xr = xy1[2 * (size_t)j], yr = xy1[2 * (size_t)j + 1];
            if (abs(yl - yr) > 20) continue;
            if (xl < xr) continue;
            const float s = vo_desc_distance(d0 + (size_t)i * VO_DESC_DIM, d1 + (size_t)j * VO_DESC_DIM);
            if (mn > s) { mn = s; loc = j; }
        }
        minloc[i] = loc; good[i] = 0;
        Pw[3 * (size_t)i] = Pw[3 * (size_t)i + 1] = Pw[3 * (size_t)i + 2] = 0.0f;
        if (loc == -1) continue;
        const float disparity = (float)(xl - xy1[2 * (size_t)loc]);
        if (disparity <= 1) continue;
        const float depth = bf / disparity;
        const float ip[3] = { (float)xl, (float)yl, 1.0f };
        float pc[3];
        for (int r = 0; r < 3; r++)
            pc[r] = invK[3 * r] * ip[0] + (invK[3 * r + 1] * ip[1] + invK[3 * r + 2] * ip[2]);
        float v[3] = { pc[0] / pc[2], 1.0f, -pc[1] / pc[2] };
        for (int k = 0; k < 3; k++) v[k] = v[k] * depth;
        rigid_apply_f(inv_pose, v, Pw + 3 * (size_t)i);
        good[i] = 1;
    }
}
