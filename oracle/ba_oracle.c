/*
 * TEST INFRASTRUCTURE ONLY.  CPU fp64 restatement of the bundle-adjustment arithmetic of
 * scomup/visual_slam (reference paths below are relative to the reference tree).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and its numeric core
 * (ceres::AutoDiffCostFunction, ceres::Solve) lives in an un-vendored, un-pinned Ceres that cannot
 * be built here.  This file restates the reference's own residual functor exactly and restates the
 * published Ceres trust-region/Levenberg-Marquardt algorithm; it is pinned only by independent
 * re-derivations (torch autograd Jacobians, numpy dense normal equations) committed as fixtures
 * under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * Conventions restated from the reference:
 *   - pose = translation(3) + (roll,pitch,yaw)(3); q = qz(yaw)*qy(pitch)*qx(roll)
 *     (src/core/reprojection_error.h:32-35); Pc = q*Pw + t (src/transform/rigid_transform.h:114-119)
 *   - camera looks along +Y: ip = (Pc.x/Pc.y, -Pc.z/Pc.y, 1) (reprojection_error.h:43)
 *   - residual = K*ip / (K*ip).z - uv, unweighted (reprojection_error.h:44-48)
 *   - edges: for i in [0, Np-1) for each frame f: tracks[f][i] != -1 (src/core/optimizer.cc:102-143)
 *   - HuberLoss(100) on the first residual block only (optimizer.cc:69,128)
 *   - solver: LM trust region, 100 iterations, Ceres defaults (src/core/tracking.cc:101-110)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double w, x, y, z; } vo_quat;

static vo_quat qmul(vo_quat a, vo_quat b)
{
    vo_quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

/* Eigen::AngleAxis(a, axis) -> quaternion (cos a/2, sin a/2 * axis); product yaw*pitch*roll
 * (reprojection_error.h:32-35). */
static vo_quat rpy_to_quat(const double *rpy)
{
    vo_quat qx = { cos(rpy[0] * 0.5), sin(rpy[0] * 0.5), 0, 0 };
    vo_quat qy = { cos(rpy[1] * 0.5), 0, sin(rpy[1] * 0.5), 0 };
    vo_quat qz = { cos(rpy[2] * 0.5), 0, 0, sin(rpy[2] * 0.5) };
    return qmul(qmul(qz, qy), qx);
}

/* Eigen quaternion * vector: uv = 2*(q.vec x v); v + q.w*uv + q.vec x uv. */
static void qrot(vo_quat q, const double *v, double *o)
{
    double ux = 2.0 * (q.y * v[2] - q.z * v[1]);
    double uy = 2.0 * (q.z * v[0] - q.x * v[2]);
    double uz = 2.0 * (q.x * v[1] - q.y * v[0]);
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}

/* K4 = {fx, fy, cx, cy}: the reference's K is identity with these four entries
 * (src/core/tracking.cc:58-62), stored as float and widened (reprojection_error.h:44). */
void vo_residual(const double *point, const double *uv, const double *t, const double *rpy,
                 const double *K4, double *res)
{
    vo_quat q = rpy_to_quat(rpy);
    double pc[3];
    qrot(q, point, pc);
    pc[0] += t[0]; pc[1] += t[1]; pc[2] += t[2];
    double ipx = pc[0] / pc[1];
    double ipy = -pc[2] / pc[1];
    double rx = K4[0] * ipx + K4[2];
    double ry = K4[1] * ipy + K4[3];
    double rz = 1.0;
    res[0] = rx / rz - uv[0];
    res[1] = ry / rz - uv[1];
}

/* R = Rz(yaw)*Ry(pitch)*Rx(roll), row-major. */
static void rpy_to_R(const double *rpy, double *R)
{
    double sr = sin(rpy[0]), cr = cos(rpy[0]);
    double sp = sin(rpy[1]), cp = cos(rpy[1]);
    double sy = sin(rpy[2]), cy = cos(rpy[2]);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

/* Analytic Jacobians of vo_residual (what ceres::AutoDiffCostFunction<...,2,3,2,3,3> yields for
 * the point, translation and rpy blocks, reprojection_error.h:55).  All 2x3 row-major. */
void vo_jacobian(const double *point, const double *t, const double *rpy, const double *K4,
                 double *Jp, double *Jt, double *Jr)
{
    double R[9];
    rpy_to_R(rpy, R);
    double q[3], pc[3];
    for (int i = 0; i < 3; i++) {
        q[i] = R[3 * i] * point[0] + R[3 * i + 1] * point[1] + R[3 * i + 2] * point[2];
        pc[i] = q[i] + t[i];
    }
    double iy = 1.0 / pc[1];
    /* d res / d Pc */
    double A[6] = { K4[0] * iy, -K4[0] * pc[0] * iy * iy, 0.0,
                    0.0, K4[1] * pc[2] * iy * iy, -K4[1] * iy };
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) {
            Jp[3 * r + c] = A[3 * r] * R[c] + A[3 * r + 1] * R[3 + c] + A[3 * r + 2] * R[6 + c];
            Jt[3 * r + c] = A[3 * r + c];
        }
    /* dPc/droll = R*(x^ x Pw); dPc/dpitch = (Rz*y^) x q; dPc/dyaw = z^ x q */
    double sy = sin(rpy[2]), cy = cos(rpy[2]);
    double m[3][3];
    double xp[3] = { 0.0, -point[2], point[1] };
    for (int i = 0; i < 3; i++)
        m[0][i] = R[3 * i] * xp[0] + R[3 * i + 1] * xp[1] + R[3 * i + 2] * xp[2];
    double ay[3] = { -sy, cy, 0.0 };
    m[1][0] = ay[1] * q[2] - ay[2] * q[1];
    m[1][1] = ay[2] * q[0] - ay[0] * q[2];
    m[1][2] = ay[0] * q[1] - ay[1] * q[0];
    m[2][0] = -q[1]; m[2][1] = q[0]; m[2][2] = 0.0;
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++)
            Jr[3 * r + k] = A[3 * r] * m[k][0] + A[3 * r + 1] * m[k][1] + A[3 * r + 2] * m[k][2];
}

/* Observation set of Optimizer::addReprojectionEdges (optimizer.cc:60-145).
 * tracks: n_frames x n_points row-major (entry = keypoint index in that frame or -1);
 * keys_ptr: n_frames+1 offsets into keys_xy (pairs of int pixel coords, optimizer.cc:30-36).
 * Writes cam/pt/uv when non-NULL; returns the number of observations, or -1 when n_points == 0
 * (the reference underflows size_t there: undefined behaviour, rejected here). */
int64_t vo_build_edges(int n_frames, int n_points, const int32_t *tracks, const int64_t *keys_ptr,
                       const int32_t *keys_xy, int32_t *cam, int32_t *pt, double *uv)
{
    if (n_points <= 0) return -1;
    if (n_frames < 2) return 0;                       /* optimizer.cc:62-63 */
    int64_t n = 0;
    for (int i = 0; i < n_points - 1; i++)            /* optimizer.cc:102: last point skipped */
        for (int f = 0; f < n_frames; f++) {          /* optimizer.cc:119 */
            int k = tracks[(int64_t)f * n_points + i];
            if (k == -1) continue;                    /* optimizer.cc:124 */
            if (cam) {
                cam[n] = f; pt[n] = i;
                uv[2 * n] = (double)keys_xy[2 * (keys_ptr[f] + k)];
                uv[2 * n + 1] = (double)keys_xy[2 * (keys_ptr[f] + k) + 1];
            }
            n++;
        }
    return n;
}

/* Ceres HuberLoss(a): rho(s), rho'(s). */
static void huber(double a, double s, double *rho, double *rho1)
{
    double b = a * a;
    if (s > b) { double r = sqrt(s); *rho = 2.0 * a * r - b; *rho1 = a / r; }
    else { *rho = s; *rho1 = 1.0; }
}

/* Residuals, Jacobians and cost 0.5*sum rho(|r|^2).  huber_a > 0 attaches HuberLoss(huber_a) to
 * observation 0 only (optimizer.cc:128: loss.release() is non-null once).  Because rho'' <= 0 for
 * Huber, Ceres' corrector scales that block's residual and Jacobian rows by sqrt(rho').
 * res: No x 2; Jc: No x 12 (2x6 row-major, columns t0 t1 t2 roll pitch yaw); Jp: No x 6 (2x3). */
void vo_evaluate(int Nc, int Np, int64_t No, const double *K4, const double *t, const double *rpy,
                 const double *pts, const int32_t *cam, const int32_t *pt, const double *uv,
                 double huber_a, double *cost, double *res, double *Jc, double *Jp)
{
    (void)Nc; (void)Np;
    double c = 0.0;
    for (int64_t o = 0; o < No; o++) {
        const double *P = pts + 3 * (int64_t)pt[o];
        const double *tt = t + 3 * (int64_t)cam[o], *rr = rpy + 3 * (int64_t)cam[o];
        double r[2], jp[6], jt[6], jr[6];
        vo_residual(P, uv + 2 * o, tt, rr, K4, r);
        double s = r[0] * r[0] + r[1] * r[1], rho = s, rho1 = 1.0;
        if (o == 0 && huber_a > 0.0) huber(huber_a, s, &rho, &rho1);
        double w = sqrt(rho1);
        c += 0.5 * rho;
        if (res) { res[2 * o] = w * r[0]; res[2 * o + 1] = w * r[1]; }
        if (Jc || Jp) {
            vo_jacobian(P, tt, rr, K4, jp, jt, jr);
            if (Jc)
                for (int rw = 0; rw < 2; rw++)
                    for (int k = 0; k < 3; k++) {
                        Jc[12 * o + 6 * rw + k] = w * jt[3 * rw + k];
                        Jc[12 * o + 6 * rw + 3 + k] = w * jr[3 * rw + k];
                    }
            if (Jp) for (int k = 0; k < 6; k++) Jp[6 * o + k] = w * jp[k];
        }
    }
    if (cost) *cost = c;
}

/* In-place lower Cholesky of a dense symmetric n x n (row-major, lower triangle used), then solve
 * A x = b in place.  Returns 0 or -1 (not positive definite). */
static int chol_solve(double *A, double *b, int n)
{
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            const double *ai = A + (size_t)i * n, *aj = A + (size_t)j * n;
            for (int k = 0; k < j; k++) s -= ai[k] * aj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 0;
}

int vo_chol_solve(double *A, double *b, int n) { return chol_solve(A, b, n); }

static void inv3_sym(const double *V, double *Vi)
{
    /* V row-major 3x3 symmetric */
    double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
    double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    double det = a * c00 + b * c01 + c * c02;
    double id = 1.0 / det;
    Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
    Vi[3] = Vi[1];    Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
    Vi[6] = Vi[2];    Vi[7] = Vi[5]; Vi[8] = (a * d - b * b) * id;
}

/* Camera-block Hessian/gradient and per-landmark Schur elimination:
 *   U_c = sum Jc^T Jc, gc_c = sum Jc^T r, V_j = sum Jp^T Jp, gp_j = sum Jp^T r, W_cj = Jc^T Jp
 *   S = (U + diag(dc)) - sum_j W_j (V_j + diag(dp_j))^-1 W_j^T
 *   rhs = -gc + sum_j W_j (V_j + diag(dp_j))^-1 gp_j
 * dc (6*Nc) / dp (3*Np): LM damping added to the diagonals, or NULL for none.
 * fix_points != 0: points constant (motion-only BA): S = U + dc, rhs = -gc.
 * Outputs (any may be NULL): S dense (6Nc)^2 row-major, rhs 6Nc, U Nc x 36, gc 6Nc,
 * Vinv Np x 9 (inverse of the damped V), gp 3Np, V Np x 9 (undamped). */
void vo_schur(int Nc, int Np, int64_t No, const double *K4, const double *t, const double *rpy,
              const double *pts, const int32_t *cam, const int32_t *pt, const double *uv,
              double huber_a, int fix_points, const double *dc, const double *dp,
              double *S, double *rhs, double *U, double *gc, double *Vinv, double *gp, double *V)
{
    int n = 6 * Nc;
    double *res = (double *)malloc(sizeof(double) * 2 * (size_t)No);
    double *Jc = (double *)malloc(sizeof(double) * 12 * (size_t)No);
    double *Jp = (double *)malloc(sizeof(double) * 6 * (size_t)No);
    vo_evaluate(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, NULL, res, Jc, Jp);
    double *Ul = (double *)calloc((size_t)Nc * 36, sizeof(double));
    double *gcl = (double *)calloc((size_t)n, sizeof(double));
    double *Vl = (double *)calloc((size_t)Np * 9, sizeof(double));
    double *gpl = (double *)calloc((size_t)Np * 3, sizeof(double));
    double *Vil = (double *)calloc((size_t)Np * 9, sizeof(double));
    for (int64_t o = 0; o < No; o++) {
        int c = cam[o], j = pt[o];
        const double *jc = Jc + 12 * o, *jp = Jp + 6 * o, *r = res + 2 * o;
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++)
                Ul[36 * (size_t)c + 6 * a + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
            gcl[6 * c + a] += jc[a] * r[0] + jc[6 + a] * r[1];
        }
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++)
                Vl[9 * (size_t)j + 3 * a + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
            gpl[3 * j + a] += jp[a] * r[0] + jp[3 + a] * r[1];
        }
    }
    if (S) {
        memset(S, 0, sizeof(double) * (size_t)n * n);
        for (int c = 0; c < Nc; c++)
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++)
                    S[(size_t)(6 * c + a) * n + 6 * c + b] = Ul[36 * (size_t)c + 6 * a + b];
        if (dc) for (int i = 0; i < n; i++) S[(size_t)i * n + i] += dc[i];
    }
    if (rhs) for (int i = 0; i < n; i++) rhs[i] = -gcl[i];
    if (!fix_points) {
        /* landmark CSR */
        int64_t *ptr = (int64_t *)calloc((size_t)Np + 1, sizeof(int64_t));
        for (int64_t o = 0; o < No; o++) ptr[pt[o] + 1]++;
        for (int j = 0; j < Np; j++) ptr[j + 1] += ptr[j];
        int64_t *ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(No > 0 ? No : 1));
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * ((size_t)Np + 1));
        memcpy(fill, ptr, sizeof(int64_t) * ((size_t)Np + 1));
        for (int64_t o = 0; o < No; o++) ord[fill[pt[o]]++] = o;
        for (int j = 0; j < Np; j++) {
            int64_t b = ptr[j], e = ptr[j + 1];
            if (e == b) continue;
            double Vd[9];
            memcpy(Vd, Vl + 9 * (size_t)j, sizeof(Vd));
            if (dp) { Vd[0] += dp[3 * j]; Vd[4] += dp[3 * j + 1]; Vd[8] += dp[3 * j + 2]; }
            double *Vi = Vil + 9 * (size_t)j;
            inv3_sym(Vd, Vi);
            if (!S && !rhs) continue;
            int k = (int)(e - b);
            double *W = (double *)malloc(sizeof(double) * 18 * (size_t)k);   /* 6x3 each */
            double *Z = (double *)malloc(sizeof(double) * 18 * (size_t)k);   /* W * Vinv */
            for (int a = 0; a < k; a++) {
                const double *jc = Jc + 12 * ord[b + a], *jp = Jp + 6 * ord[b + a];
                for (int p = 0; p < 6; p++)
                    for (int q2 = 0; q2 < 3; q2++)
                        W[18 * a + 3 * p + q2] = jc[p] * jp[q2] + jc[6 + p] * jp[3 + q2];
                for (int p = 0; p < 6; p++)
                    for (int q2 = 0; q2 < 3; q2++)
                        Z[18 * a + 3 * p + q2] = W[18 * a + 3 * p] * Vi[q2] +
                                                 W[18 * a + 3 * p + 1] * Vi[3 + q2] +
                                                 W[18 * a + 3 * p + 2] * Vi[6 + q2];
            }
            for (int a = 0; a < k; a++) {
                int ca = cam[ord[b + a]];
                if (rhs)
                    for (int p = 0; p < 6; p++)
                        rhs[6 * ca + p] += Z[18 * a + 3 * p] * gpl[3 * j] +
                                           Z[18 * a + 3 * p + 1] * gpl[3 * j + 1] +
                                           Z[18 * a + 3 * p + 2] * gpl[3 * j + 2];
                if (S)
                    for (int a2 = 0; a2 < k; a2++) {
                        int cb = cam[ord[b + a2]];
                        for (int p = 0; p < 6; p++)
                            for (int q2 = 0; q2 < 6; q2++)
                                S[(size_t)(6 * ca + p) * n + 6 * cb + q2] -=
                                    Z[18 * a + 3 * p] * W[18 * a2 + 3 * q2] +
                                    Z[18 * a + 3 * p + 1] * W[18 * a2 + 3 * q2 + 1] +
                                    Z[18 * a + 3 * p + 2] * W[18 * a2 + 3 * q2 + 2];
                    }
            }
            free(W); free(Z);
        }
        free(ptr); free(ord); free(fill);
    }
    if (U) memcpy(U, Ul, sizeof(double) * (size_t)Nc * 36);
    if (gc) memcpy(gc, gcl, sizeof(double) * (size_t)n);
    if (V) memcpy(V, Vl, sizeof(double) * (size_t)Np * 9);
    if (gp) memcpy(gp, gpl, sizeof(double) * (size_t)Np * 3);
    if (Vinv) memcpy(Vinv, Vil, sizeof(double) * (size_t)Np * 9);
    free(res); free(Jc); free(Jp); free(Ul); free(gcl); free(Vl); free(gpl); free(Vil);
}

/* Dense full normal equations H = J^T J (+ diag(damp)), g = J^T r over the free parameters,
 * ordered [camera 0 (t,rpy) ... camera Nc-1 | point 0 ... point Np-1]; points omitted when
 * fix_points.  This is the system SPARSE_NORMAL_CHOLESKY factors (tracking.cc:103); small sizes
 * only (used to check the Schur route).  H: n x n row-major, g: n. */
void vo_normal_dense(int Nc, int Np, int64_t No, const double *K4, const double *t,
                     const double *rpy, const double *pts, const int32_t *cam, const int32_t *pt,
                     const double *uv, double huber_a, int fix_points, double *H, double *g)
{
    int n = 6 * Nc + (fix_points ? 0 : 3 * Np);
    double *res = (double *)malloc(sizeof(double) * 2 * (size_t)No);
    double *Jc = (double *)malloc(sizeof(double) * 12 * (size_t)No);
    double *Jp = (double *)malloc(sizeof(double) * 6 * (size_t)No);
    vo_evaluate(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, NULL, res, Jc, Jp);
    memset(H, 0, sizeof(double) * (size_t)n * n);
    memset(g, 0, sizeof(double) * (size_t)n);
    for (int64_t o = 0; o < No; o++) {
        int idx[9]; double row[2][9];
        int m = fix_points ? 6 : 9;
        for (int a = 0; a < 6; a++) { idx[a] = 6 * cam[o] + a; row[0][a] = Jc[12 * o + a]; row[1][a] = Jc[12 * o + 6 + a]; }
        for (int a = 0; a < 3; a++) { idx[6 + a] = 6 * Nc + 3 * pt[o] + a; row[0][6 + a] = Jp[6 * o + a]; row[1][6 + a] = Jp[6 * o + 3 + a]; }
        for (int a = 0; a < m; a++) {
            g[idx[a]] += row[0][a] * res[2 * o] + row[1][a] * res[2 * o + 1];
            for (int b = 0; b < m; b++)
                H[(size_t)idx[a] * n + idx[b]] += row[0][a] * row[0][b] + row[1][a] * row[1][b];
        }
    }
    free(res); free(Jc); free(Jp);
}

/* ---- Levenberg-Marquardt, restating the documented Ceres trust-region minimizer -------------
 * Jacobi scaling s_i = 1/(1+sqrt(H_ii)) fixed at the first iterate; LM diagonal
 * D_i^2 = clamp(s_i^2 H_ii, min_lm_diagonal, max_lm_diagonal)/radius in scaled coordinates, i.e.
 * damping lambda_i = D_i^2 / s_i^2 added to H_ii in unscaled coordinates; step via per-landmark
 * Schur elimination (algebraically the same step as the reference's SPARSE_NORMAL_CHOLESKY). */

static void lm_damping(int n, const double *hdiag, const double *scale, double radius,
                       double min_d, double max_d, double *lam)
{
    for (int i = 0; i < n; i++) {
        double s2 = scale[i] * scale[i];
        double d = s2 * hdiag[i];
        if (d < min_d) d = min_d;
        if (d > max_d) d = max_d;
        lam[i] = d / radius / s2;
    }
}

/* One LM step at the current linearisation point.
 * scale_c (6Nc) / scale_p (3Np): Jacobi scaling; when scale_c[0] <= 0 on entry they are computed
 * from the current Jacobian and written back.  Outputs dcam (6Nc), dpt (3Np; zeros when
 * fix_points) and the model cost change  -(J d)^T (r + J d / 2).  Returns 0, or -1 when the
 * reduced system is not positive definite. */
int vo_lm_step(int Nc, int Np, int64_t No, const double *K4, const double *t, const double *rpy,
               const double *pts, const int32_t *cam, const int32_t *pt, const double *uv,
               double huber_a, int fix_points, double radius, double min_lm_diag,
               double max_lm_diag, double *scale_c, double *scale_p, double *dcam, double *dpt,
               double *model_cost_change)
{
    int n = 6 * Nc;
    double *U = (double *)malloc(sizeof(double) * 36 * (size_t)Nc);
    double *V = (double *)malloc(sizeof(double) * 9 * (size_t)(Np > 0 ? Np : 1));
    vo_schur(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, 1, NULL, NULL,
             NULL, NULL, U, NULL, NULL, NULL, V);
    double *hc = (double *)malloc(sizeof(double) * (size_t)n);
    double *hp = (double *)malloc(sizeof(double) * 3 * (size_t)(Np > 0 ? Np : 1));
    for (int c = 0; c < Nc; c++) for (int a = 0; a < 6; a++) hc[6 * c + a] = U[36 * (size_t)c + 7 * a];
    for (int j = 0; j < Np; j++) for (int a = 0; a < 3; a++) hp[3 * j + a] = V[9 * (size_t)j + 4 * a];
    if (scale_c[0] <= 0.0) {
        for (int i = 0; i < n; i++) scale_c[i] = 1.0 / (1.0 + sqrt(hc[i]));
        for (int i = 0; i < 3 * Np; i++) scale_p[i] = 1.0 / (1.0 + sqrt(hp[i]));
    }
    double *lc = (double *)malloc(sizeof(double) * (size_t)n);
    double *lp = (double *)malloc(sizeof(double) * 3 * (size_t)(Np > 0 ? Np : 1));
    lm_damping(n, hc, scale_c, radius, min_lm_diag, max_lm_diag, lc);
    lm_damping(3 * Np, hp, scale_p, radius, min_lm_diag, max_lm_diag, lp);
    double *S = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *gp = (double *)malloc(sizeof(double) * 3 * (size_t)(Np > 0 ? Np : 1));
    double *Vinv = (double *)malloc(sizeof(double) * 9 * (size_t)(Np > 0 ? Np : 1));
    vo_schur(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, fix_points, lc, lp,
             S, dcam, NULL, NULL, Vinv, gp, NULL);
    int rc = chol_solve(S, dcam, n);
    double *res = (double *)malloc(sizeof(double) * 2 * (size_t)No);
    double *Jc = (double *)malloc(sizeof(double) * 12 * (size_t)No);
    double *Jp = (double *)malloc(sizeof(double) * 6 * (size_t)No);
    vo_evaluate(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, NULL, res, Jc, Jp);
    memset(dpt, 0, sizeof(double) * 3 * (size_t)Np);
    if (rc == 0 && !fix_points) {
        /* dp_j = -Vinv_j (gp_j + sum_c W_cj^T dc_c) */
        double *acc = (double *)calloc(3 * (size_t)Np, sizeof(double));
        for (int64_t o = 0; o < No; o++) {
            const double *jc = Jc + 12 * o, *jp = Jp + 6 * o, *dc = dcam + 6 * cam[o];
            double a0 = 0, a1 = 0;
            for (int k = 0; k < 6; k++) { a0 += jc[k] * dc[k]; a1 += jc[6 + k] * dc[k]; }
            for (int k = 0; k < 3; k++) acc[3 * pt[o] + k] += jp[k] * a0 + jp[3 + k] * a1;
        }
        int *seen = (int *)calloc((size_t)Np, sizeof(int));
        for (int64_t o = 0; o < No; o++) seen[pt[o]] = 1;
        for (int j = 0; j < Np; j++) {
            if (!seen[j]) continue;
            double b[3];
            for (int k = 0; k < 3; k++) b[k] = gp[3 * j + k] + acc[3 * j + k];
            for (int k = 0; k < 3; k++)
                dpt[3 * j + k] = -(Vinv[9 * (size_t)j + 3 * k] * b[0] +
                                   Vinv[9 * (size_t)j + 3 * k + 1] * b[1] +
                                   Vinv[9 * (size_t)j + 3 * k + 2] * b[2]);
        }
        free(acc); free(seen);
    }
    double mcc = 0.0;
    if (rc == 0)
        for (int64_t o = 0; o < No; o++) {
            const double *jc = Jc + 12 * o, *jp = Jp + 6 * o, *dc = dcam + 6 * cam[o];
            const double *dq = dpt + 3 * pt[o];
            for (int rw = 0; rw < 2; rw++) {
                double m = 0;
                for (int k = 0; k < 6; k++) m += jc[6 * rw + k] * dc[k];
                for (int k = 0; k < 3; k++) m += jp[3 * rw + k] * dq[k];
                mcc -= m * (res[2 * o + rw] + 0.5 * m);
            }
        }
    if (model_cost_change) *model_cost_change = mcc;
    free(U); free(V); free(hc); free(hp); free(lc); free(lp); free(S); free(gp); free(Vinv);
    free(res); free(Jc); free(Jp);
    return rc;
}

/* opts: [0] max_num_iterations [1] initial_trust_region_radius [2] max_radius [3] min_radius
 *       [4] min_relative_decrease [5] min_lm_diagonal [6] max_lm_diagonal [7] function_tolerance
 *       [8] gradient_tolerance [9] parameter_tolerance [10] max_consecutive_invalid_steps
 * summary: [0] iterations run [1] initial cost [2] final cost [3] termination (0 convergence,
 *       1 no convergence, 2 failure) [4] successful steps [5] final radius
 * t, rpy, pts are updated in place (pts untouched when fix_points). */
int vo_solve(int Nc, int Np, int64_t No, const double *K4, double *t, double *rpy, double *pts,
             const int32_t *cam, const int32_t *pt, const double *uv, double huber_a,
             int fix_points, const double *opts, double *summary)
{
    int max_it = (int)opts[0];
    double radius = opts[1], max_radius = opts[2], min_radius = opts[3];
    double min_rel = opts[4], min_d = opts[5], max_d = opts[6];
    double ftol = opts[7], gtol = opts[8], ptol = opts[9];
    int max_invalid = (int)opts[10];
    int n = 6 * Nc, np3 = 3 * Np;
    double *sc = (double *)calloc((size_t)n, sizeof(double));
    double *sp = (double *)calloc((size_t)(np3 > 0 ? np3 : 1), sizeof(double));
    double *dc = (double *)malloc(sizeof(double) * (size_t)n);
    double *dp = (double *)malloc(sizeof(double) * (size_t)(np3 > 0 ? np3 : 1));
    double *t2 = (double *)malloc(sizeof(double) * 3 * (size_t)Nc);
    double *r2 = (double *)malloc(sizeof(double) * 3 * (size_t)Nc);
    double *p2 = (double *)malloc(sizeof(double) * (size_t)(np3 > 0 ? np3 : 1));
    double *gc = (double *)malloc(sizeof(double) * (size_t)n);
    double *gp = (double *)malloc(sizeof(double) * (size_t)(np3 > 0 ? np3 : 1));
    int *seen = (int *)calloc((size_t)(Np > 0 ? Np : 1), sizeof(int));
    int *seenc = (int *)calloc((size_t)Nc, sizeof(int));
    for (int64_t o = 0; o < No; o++) { seen[pt[o]] = 1; seenc[cam[o]] = 1; }
    double cost, cost0, decrease_factor = 2.0;
    int it = 0, term = 1, nsucc = 0, invalid = 0;
    vo_evaluate(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, &cost, NULL, NULL, NULL);
    cost0 = cost;
    if (No == 0) { term = 0; goto done; }
    /* gradient max-norm at the start */
    {
        vo_schur(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, 1, NULL, NULL,
                 NULL, NULL, NULL, gc, NULL, gp, NULL);
        double gmax = 0;
        for (int i = 0; i < n; i++) if (fabs(gc[i]) > gmax) gmax = fabs(gc[i]);
        if (!fix_points) for (int i = 0; i < np3; i++) if (fabs(gp[i]) > gmax) gmax = fabs(gp[i]);
        if (gmax <= gtol) { term = 0; goto done; }
    }
    while (1) {
        if (it >= max_it) { term = 1; break; }
        if (radius < min_radius) { term = 0; break; }
        it++;
        double mcc;
        int rc = vo_lm_step(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, fix_points, radius,
                            min_d, max_d, sc, sp, dc, dp, &mcc);
        if (rc != 0 || !(mcc > 0.0)) {               /* invalid step */
            if (++invalid >= max_invalid) { term = 2; break; }   /* Ceres HandleInvalidStep: fails at the max-th consecutive one */
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        double xn = 0, sn = 0;
        for (int c = 0; c < Nc; c++) {
            if (!seenc[c]) { for (int k = 0; k < 3; k++) { t2[3 * c + k] = t[3 * c + k]; r2[3 * c + k] = rpy[3 * c + k]; } continue; }
            for (int k = 0; k < 3; k++) {
                t2[3 * c + k] = t[3 * c + k] + dc[6 * c + k];
                r2[3 * c + k] = rpy[3 * c + k] + dc[6 * c + 3 + k];
                xn += t[3 * c + k] * t[3 * c + k] + rpy[3 * c + k] * rpy[3 * c + k];
                sn += dc[6 * c + k] * dc[6 * c + k] + dc[6 * c + 3 + k] * dc[6 * c + 3 + k];
            }
        }
        for (int j = 0; j < Np; j++)
            for (int k = 0; k < 3; k++) {
                p2[3 * j + k] = pts[3 * j + k];
                if (!fix_points && seen[j]) {
                    p2[3 * j + k] += dp[3 * j + k];
                    xn += pts[3 * j + k] * pts[3 * j + k];
                    sn += dp[3 * j + k] * dp[3 * j + k];
                }
            }
        xn = sqrt(xn); sn = sqrt(sn);
        double cost_new;
        vo_evaluate(Nc, Np, No, K4, t2, r2, p2, cam, pt, uv, huber_a, &cost_new, NULL, NULL, NULL);
        if (sn <= ptol * (xn + ptol)) { term = 0; break; }          /* parameter tolerance */
        double change = cost - cost_new;
        if (fabs(change) <= ftol * cost) { term = 0; break; }       /* function tolerance */
        double rho = change / mcc;
        if (rho > min_rel) {
            memcpy(t, t2, sizeof(double) * 3 * (size_t)Nc);
            memcpy(rpy, r2, sizeof(double) * 3 * (size_t)Nc);
            if (!fix_points) memcpy(pts, p2, sizeof(double) * (size_t)np3);
            cost = cost_new; nsucc++;
            double f = 1.0 - pow(2.0 * rho - 1.0, 3.0);
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = radius / f;
            if (radius > max_radius) radius = max_radius;
            decrease_factor = 2.0;
            vo_schur(Nc, Np, No, K4, t, rpy, pts, cam, pt, uv, huber_a, 1, NULL, NULL,
                     NULL, NULL, NULL, gc, NULL, gp, NULL);
            double gmax = 0;
            for (int i = 0; i < n; i++) if (fabs(gc[i]) > gmax) gmax = fabs(gc[i]);
            if (!fix_points) for (int i = 0; i < np3; i++) if (fabs(gp[i]) > gmax) gmax = fabs(gp[i]);
            if (gmax <= gtol) { term = 0; break; }
        } else {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
        }
    }
done:
    if (summary) {
        summary[0] = it; summary[1] = cost0; summary[2] = cost; summary[3] = term;
        summary[4] = nsucc; summary[5] = radius;
    }
    free(sc); free(sp); free(dc); free(dp); free(t2); free(r2); free(p2); free(gc); free(gp);
    free(seen); free(seenc);
    return term;
}

/* Pose codec of Optimizer::FromPose / getPose (optimizer.cc:38-59).
 * quat_to_rpy: Eigen toRotationMatrix().eulerAngles(0,1,2): angles (a,b,c) with
 * R = Rx(a)*Ry(b)*Rz(c), a in [0,pi].  q = (w,x,y,z). */
void vo_quat_to_euler012(const double *q, double *e)
{
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double R[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                    2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y) };
    /* Eigen EulerAngles.h with i=0,j=1,k=2 (odd = 0) */
    double r0 = atan2(R[5], R[8]);                   /* atan2(coeff(j,k), coeff(k,k)) */
    double c2 = sqrt(R[0] * R[0] + R[1] * R[1]);     /* norm of (coeff(i,i), coeff(i,j)) */
    double r1, r2;
    if (r0 > 0.0) {
        if (r0 > 0.0) r0 -= M_PI; else r0 += M_PI;
        r1 = atan2(-R[2], -c2);
    } else {
        r1 = atan2(-R[2], c2);
    }
    double s1 = sin(r0), c1 = cos(r0);
    r2 = atan2(s1 * R[6] - c1 * R[3], c1 * R[4] - s1 * R[7]);
    e[0] = -r0; e[1] = -r1; e[2] = -r2;
}

/* getPose: q = qz(r2)*qy(r1)*qx(r0) (optimizer.cc:54-57); out (w,x,y,z). */
void vo_rpy_to_quat(const double *rpy, double *q)
{
    vo_quat r = rpy_to_quat(rpy);
    q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
}
