/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement of the per-frame tracking helpers that sit beside
 * the matcher and the optimiser in scomup/visual_slam (reference paths are relative to the
 * reference tree): Tracking::reprojection, Tracking::removeOutlier, Frame::computePoint3d,
 * Tracking::updateTrackedPoints.
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

/* ---- Tracking::updateTrackedPoints (tracking.cc:378-451): two-view linear triangulation --------
 * Float construction restated operation by operation; Eigen details (version unpinned):
 *   - Quaternionf::normalized() = coeffs / sqrt(squaredNorm); the 4-term sum is taken as
 *     (x^2 + y^2) + (z^2 + w^2) (Eigen's packet reduction order is not pinned by the reference)
 *   - toRotationMatrix(): tx = 2x, ty = 2y, tz = 2z; twx = tx w ...; R00 = 1 - (tyy + tzz) ...
 *   - Quaternionf::inverse() = conjugate coefficients / squaredNorm
 *   - 3-vector dot and squaredNorm sum left to right
 * The null vector of the 4 x 4 system is what Eigen::JacobiSVD<Matrix4f>::matrixV().col(3) holds up
 * to sign and float round-off.  Here it is the eigenvector of the smallest eigenvalue of A^T A by
 * cyclic Jacobi rotations in double (the float SVD of the reference agrees to the conditioning of
 * the system times float epsilon). */
static void quat_to_rot_f(const float *q /* w x y z */, float *R /* 9 row-major */)
{
    const float n = sqrtf((q[1] * q[1] + q[2] * q[2]) + (q[3] * q[3] + q[0] * q[0]));
    const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

static void quat_inverse_f(const float *q, float *o)
{
    const float n2 = (q[1] * q[1] + q[2] * q[2]) + (q[3] * q[3] + q[0] * q[0]);
    o[0] = q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = -q[3] / n2;
}

/* unit eigenvector of the smallest eigenvalue of the symmetric 4 x 4 matrix B (destroyed) */
static void smallest_eigvec4(double B[4][4], double *v)
{
    double V[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < 4; p++) {
            diag += B[p][p] * B[p][p];
            for (int q = p + 1; q < 4; q++) off += B[p][q] * B[p][q];
        }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                if (B[p][q] == 0.0) continue;
                const double theta = (B[q][q] - B[p][p]) / (2.0 * B[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {
                    const double bkp = B[k][p], bkq = B[k][q];
                    B[k][p] = c * bkp - s * bkq; B[k][q] = s * bkp + c * bkq;
                }
                for (int k = 0; k < 4; k++) {
                    const double bpk = B[p][k], bqk = B[q][k];
                    B[p][k] = c * bpk - s * bqk; B[q][k] = s * bpk + c * bqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int p = 1; p < 4; p++) if (B[p][p] < B[m][m]) m = p;
    for (int k = 0; k < 4; k++) v[k] = V[k][m];
}

/* The 4 x 4 system of one track (float), or 0 when the parallax gate rejects it. */
static int dlt_system_f(const float *pose1, const float *pose2, const float *K4, const int32_t *key1,
                        const int32_t *key2, float A[4][4])
{
    float M1[3][4], M2[3][4], R[9];
    quat_to_rot_f(pose1 + 3, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M1[r][c] = R[3 * r + c]; M1[r][3] = pose1[r]; }
    quat_to_rot_f(pose2 + 3, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M2[r][c] = R[3 * r + c]; M2[r][3] = pose2[r]; }
    const float invfx = 1.0f / K4[0], invfy = 1.0f / K4[1];
    const float xn1[3] = { ((float)key1[0] - K4[2]) * invfx, 1.0f, -((float)key1[1] - K4[3]) * invfy };
    const float xn2[3] = { ((float)key2[0] - K4[2]) * invfx, 1.0f, -((float)key2[1] - K4[3]) * invfy };
    float qi[4], ray1[3], ray2[3];
    quat_inverse_f(pose1 + 3, qi); quat_rotate_f(qi, xn1, ray1);
    quat_inverse_f(pose2 + 3, qi); quat_rotate_f(qi, xn2, ray2);
    const float dot = (ray1[0] * ray2[0] + ray1[1] * ray2[1]) + ray1[2] * ray2[2];
    const float n1 = sqrtf((ray1[0] * ray1[0] + ray1[1] * ray1[1]) + ray1[2] * ray1[2]);
    const float n2 = sqrtf((ray2[0] * ray2[0] + ray2[1] * ray2[1]) + ray2[2] * ray2[2]);
    const float cosp = dot / (n1 * n2);
    if (cosp < 0 || (double)cosp > 0.9998) return 0;
    for (int c = 0; c < 4; c++) {
        A[0][c] = xn1[0] * M1[1][c] - M1[0][c];
        A[1][c] = xn1[2] * M1[1][c] - M1[2][c];
        A[2][c] = xn2[0] * M2[1][c] - M2[0][c];
        A[3][c] = xn2[2] * M2[1][c] - M2[2][c];
    }
    return 1;
}

void vo_dlt_system(const float *pose1, const float *pose2, const float *K4, const int32_t *key1,
                   const int32_t *key2, float *A16, int32_t *ok)
{
    float A[4][4] = { { 0 } };
    *ok = dlt_system_f(pose1, pose2, K4, key1, key2, A);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) A16[4 * r + c] = A[r][c];
}

/* Pw[3i..] is overwritten and updated[i] = 1 where the reference assigns tracked_points_[i]->Pw.
 * pose1/keys1/tracks1 = oldest frame of the window, pose2/keys2/tracks2 = newest.  Returns the
 * number of points updated. */
int vo_triangulate_tracks(const float *pose1, const float *pose2, const float *K4,
                          const int32_t *keys1, const int32_t *tracks1, const int32_t *keys2,
                          const int32_t *tracks2, int n_tp, float *Pw, uint8_t *updated)
{
    int cont = 0;
    for (int i = 0; i < n_tp; i++) {
        updated[i] = 0;
        const int id1 = tracks1[i], id2 = tracks2[i];
        if (id1 == -1 || id2 == -1) continue;
        float A[4][4];
        if (!dlt_system_f(pose1, pose2, K4, keys1 + 2 * (size_t)id1, keys2 + 2 * (size_t)id2, A)) continue;
        double B[4][4], v[4];
        for (int p = 0; p < 4; p++)
            for (int q = 0; q < 4; q++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += (double)A[k][p] * (double)A[k][q];
                B[p][q] = s;
            }
        smallest_eigvec4(B, v);
        const float x3[4] = { (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
        if (x3[3] == 0) continue;
        const float P[3] = { x3[0] / x3[3], x3[1] / x3[3], x3[2] / x3[3] };
        float Pc[3];
        rigid_apply_f(pose1, P, Pc);
        if ((double)Pc[1] <= 0.2) continue;
        rigid_apply_f(pose2, P, Pc);
        if ((double)Pc[1] <= 0.2) continue;
        Pw[3 * (size_t)i] = P[0]; Pw[3 * (size_t)i + 1] = P[1]; Pw[3 * (size_t)i + 2] = P[2];
        updated[i] = 1;
        cont++;
    }
    return cont;
}
