// Per-frame tracking helpers beside the matcher (reference interfaces: include/vslam_hip.h):
// Tracking::reprojection / searchProjection / removeOutlier (src/core/tracking.cc:70-78,
// 453-496, 346-377, 378-451 updateTrackedPoints) and Frame::computePoint3d (src/core/frame.cc:45-86).
//
// The float geometry replays the reference's Eigen expressions operation by operation (every
// product, sum and quotient rounded to float on its own: contraction into FMAs is switched off
// for this file), descriptor distances use the exact fp64-accumulated form of match_exact.h, so
// integer pixel coordinates, chosen keypoints and claimed tracks are bit-identical to a scalar
// replay of the reference loops.
#include <algorithm>
#include <climits>
#include <new>
#include <vector>

#include "common.h"
#include "match_exact.h"

#pragma clang fp contract(off)

using namespace vs;

namespace {

constexpr int kCandCap = 32;   // stored candidates per tracked point (more -> exact slow path)

struct Pose {
    float t[3];
    float q[4];   // w x y z
};
struct Intr4 {
    float fx, fy, cx, cy;
};

__device__ __forceinline__ void cross3f(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// Eigen::Quaternionf * Vector3f:  uv = q.vec x v;  uv += uv;  v + q.w * uv + q.vec x uv
__device__ __forceinline__ void quat_rotate_f(const float *q, const float *v, float *o)
{
    float uv[3], c[3];
    cross3f(q + 1, v, uv);
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    cross3f(q + 1, uv, c);
    for (int k = 0; k < 3; k++) o[k] = (v[k] + q[0] * uv[k]) + c[k];
}

// Rigid3f * Vector3f (src/transform/rigid_transform.h:114-119)
__device__ __forceinline__ void rigid_apply_f(const Pose &T, const float *p, float *o)
{
    float r[3];
    quat_rotate_f(T.q, p, r);
    for (int k = 0; k < 3; k++) o[k] = r[k] + T.t[k];
}

// Rigid3f::inverse() (src/transform/rigid_transform.h:75-79)
__device__ __forceinline__ Pose rigid_inverse_f(const Pose &T)
{
    Pose inv;
    inv.q[0] = T.q[0]; inv.q[1] = -T.q[1]; inv.q[2] = -T.q[2]; inv.q[3] = -T.q[3];
    float r[3];
    quat_rotate_f(inv.q, T.t, r);
    for (int k = 0; k < 3; k++) inv.t[k] = -r[k];
    return inv;
}

// Tracking::reprojection (tracking.cc:70-78).  K * (x, y, 1) sums each row as c0 + (c1 + c2)
// with the zero entries of K contributing exact zeros; cv::Point(float, float) truncates.
__device__ __forceinline__ int2 reproject_f(const Pose &T, const Intr4 &K, const float *Pw)
{
    float Pc[3];
    rigid_apply_f(T, Pw, Pc);
    const float lx = Pc[0] / Pc[1], ly = -Pc[2] / Pc[1];
    const float rz = 0.0f * lx + (0.0f * ly + 1.0f * 1.0f);
    const float rx = (K.fx * lx + (0.0f * ly + K.cx * 1.0f)) / rz;
    const float ry = (0.0f * lx + (K.fy * ly + K.cy * 1.0f)) / rz;
    int2 o;
    o.x = (rx >= -2147483648.0f && rx < 2147483648.0f) ? (int)rx : INT_MIN;
    o.y = (ry >= -2147483648.0f && ry < 2147483648.0f) ? (int)ry : INT_MIN;
    return o;
}

__global__ void __launch_bounds__(256)
reproject_kernel(Pose T, Intr4 K, const float *__restrict__ Pw, int n, int2 *__restrict__ xy)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p[3] = { Pw[3 * (int64_t)i], Pw[3 * (int64_t)i + 1], Pw[3 * (int64_t)i + 2] };
    xy[i] = reproject_f(T, K, p);
}

// Tracking::removeOutlier for one frame (tracking.cc:357-372).
__global__ void __launch_bounds__(256)
remove_outliers_kernel(Pose T, Intr4 K, const float *__restrict__ Pw, int n_tp,
                       const int2 *__restrict__ key_xy, int n_keys, double max_dist,
                       int *__restrict__ tracks, int *__restrict__ removed)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tp) return;
    const int key_id = tracks[i];
    if (key_id < 0 || key_id >= n_keys) return;
    const float p[3] = { Pw[3 * (int64_t)i], Pw[3 * (int64_t)i + 1], Pw[3 * (int64_t)i + 2] };
    const int2 re = reproject_f(T, K, p);
    const int2 k = key_xy[key_id];
    const double dx = (double)k.x - (double)re.x, dy = (double)k.y - (double)re.y;
    if (sqrt(dx * dx + dy * dy) > max_dist) {
        tracks[i] = -1;
        atomicAdd(removed, 1);
    }
}

// One wave per query descriptor: scans the candidate keypoints in index order, keeps those that
// pass the geometric gate and scores them exactly, 64 at a time (one candidate per lane).
//   WINDOW mode (searchProjection, tracking.cc:473-489): gate = distance to the query's integer
//     reprojection <= radius; candidates with distance < accept are appended to the query's list
//     as packed (distance bits, index) keys.
//   STEREO mode (computePoint3d, frame.cc:53-68): gate = |yl - yr| <= 20 and xl >= xr; the wave
//     minimum of the packed keys is the first strict minimum in scan order.
enum { kWindow = 0, kStereo = 1 };

template <int MODE>
__global__ void __launch_bounds__(64)
gated_nn_kernel(const float *__restrict__ q_desc, const int2 *__restrict__ q_xy, int n_q,
                const float *__restrict__ k_desc, const int2 *__restrict__ k_xy, int n_k,
                float bound_x, float bound_y, float radius, double accept,
                const int *__restrict__ skip, unsigned long long *__restrict__ cand, int *__restrict__ n_cand,
                unsigned char *__restrict__ overflow, int *__restrict__ minloc)
{
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tA[kDescDim];
    __shared__ const float *pA[1];
    __shared__ const float *pB[64];
    __shared__ int queue[128];
    const int lane = threadIdx.x;
    const int q = blockIdx.x;
    if (q >= n_q) return;
    const int2 me = q_xy[q];
    if (MODE == kWindow) {
        // reprojection outside the image (tracking.cc:467-470): no track
        const bool outside = (float)me.x > bound_x || me.x < 0 || (float)me.y > bound_y || me.y < 0;
        if (outside) {
            if (lane == 0) { n_cand[q] = 0; overflow[q] = 0; }
            return;
        }
    }
    int queued = 0, stored = 0;
    bool over = false;
    unsigned long long best = ~0ull;

    auto flush = [&](int count) {   // scores queue[0 .. count), count <= 64
        const int c = lane < count ? queue[lane] : 0;
        __syncthreads();
        if (lane == 0) pA[0] = q_desc + (int64_t)q * kDescDim;
        pB[lane] = k_desc + (int64_t)c * kDescDim;
        __syncthreads();
        const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
        const float d = distance_from_dot(acc);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)c;
        if (MODE == kStereo) {
            if (lane < count && key < best) best = key;
        } else {
            const bool keep = lane < count && (double)d < accept;
            const unsigned long long m = __ballot(keep);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                if (stored + rank < kCandCap) cand[(int64_t)q * kCandCap + stored + rank] = key;
                else over = true;
            }
            stored += __popcll(m);
        }
    };

    for (int j0 = 0; j0 < n_k; j0 += 64) {
        const int j = j0 + lane;
        bool gate = false;
        if (j < n_k) {
            const int2 k = k_xy[j];
            if (MODE == kWindow) {
                const double dx = (double)k.x - (double)me.x, dy = (double)k.y - (double)me.y;
                // keypoints that already own a tracked point never compete (tracking.cc:475-476)
                gate = !skip[j] && !(sqrt(dx * dx + dy * dy) > (double)radius);
            } else {
                int dy = me.y - k.y;
                dy = dy < 0 ? -dy : dy;
                gate = !(dy > 20) && !(me.x < k.x);
            }
        }
        const unsigned long long m = __ballot(gate);
        if (gate) queue[queued + __popcll(m & ((1ull << lane) - 1ull))] = j;
        queued += __popcll(m);
        if (queued >= 64) {
            __syncthreads();
            flush(64);
            __syncthreads();
            const int rest = queued - 64;
            const int moved = lane < rest ? queue[64 + lane] : 0;
            __syncthreads();
            if (lane < rest) queue[lane] = moved;
            queued = rest;
        }
    }
    __syncthreads();
    if (queued > 0) flush(queued);

    if (MODE == kStereo) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) minloc[q] = best == ~0ull ? -1 : (int)(best & 0xffffffffu);
    } else {
        const bool any_over = __ballot(over) != 0ull;
        if (lane == 0) {
            n_cand[q] = stored < kCandCap ? stored : kCandCap;
            overflow[q] = any_over ? 1 : 0;
        }
    }
}

__device__ __forceinline__ int load_shared_int(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_shared_int(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The greedy claim of searchProjection (tracking.cc:462-495): tracked points in index order each
// take their best still-unclaimed keypoint.  Resolved in parallel rounds by ONE workgroup: in a
// round every unresolved point marks all its unclaimed candidates with atomicMin(owner, i); a
// point whose best unclaimed candidate carries its own index cannot lose that keypoint to an
// earlier point, and every keypoint it prefers is already taken by a final decision, so its
// choice equals the sequential one.  The smallest unresolved index always qualifies, so every
// round makes progress.  A point whose candidate list overflowed waits until it is the smallest
// unresolved index and is then decided by an exact scan of all keypoints; later points wait for it.
// claimed / owner / done are int arrays in global memory accessed with agent-scope atomics.
__global__ void __launch_bounds__(1024)
greedy_resolve_kernel(int n_tp, int n_keys, const unsigned long long *__restrict__ cand,
                      const int *__restrict__ n_cand, const unsigned char *__restrict__ overflow,
                      const float *__restrict__ tp_desc, const int2 *__restrict__ reproj,
                      const float *__restrict__ key_desc, const int2 *__restrict__ key_xy,
                      float radius, double accept, int *claimed, int *owner, int *done,
                      int *__restrict__ tracks)
{
    __shared__ int n_unres, ovf_min, unres_min;
    __shared__ unsigned long long slow_best;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_tp; i += 1024) {
        tracks[i] = -1;
        store_shared_int(&done[i], (n_cand[i] == 0 && !overflow[i]) ? 1 : 0);
    }
    for (;;) {
        __syncthreads();
        if (tid == 0) { n_unres = 0; ovf_min = INT_MAX; unres_min = INT_MAX; slow_best = ~0ull; }
        for (int j = tid; j < n_keys; j += 1024) store_shared_int(&owner[j], INT_MAX);
        __syncthreads();
        for (int i = tid; i < n_tp; i += 1024) {
            if (load_shared_int(&done[i])) continue;
            atomicAdd(&n_unres, 1);
            atomicMin(&unres_min, i);
            if (overflow[i]) { atomicMin(&ovf_min, i); continue; }
            for (int c = 0; c < n_cand[i]; c++) {
                const int j = (int)(cand[(int64_t)i * kCandCap + c] & 0xffffffffu);
                if (!load_shared_int(&claimed[j])) atomicMin(&owner[j], i);
            }
        }
        __syncthreads();
        if (n_unres == 0) break;
        const int ovf = ovf_min, first = unres_min;
        for (int i = tid; i < n_tp; i += 1024) {
            if (i >= ovf || load_shared_int(&done[i])) continue;
            unsigned long long best = ~0ull;
            for (int c = 0; c < n_cand[i]; c++) {
                const unsigned long long key = cand[(int64_t)i * kCandCap + c];
                if (!load_shared_int(&claimed[(int)(key & 0xffffffffu)]) && key < best) best = key;
            }
            if (best == ~0ull) { store_shared_int(&done[i], 1); continue; }
            const int j = (int)(best & 0xffffffffu);
            if (load_shared_int(&owner[j]) == i) {
                store_shared_int(&claimed[j], 1);
                tracks[i] = j;
                store_shared_int(&done[i], 1);
            }
        }
        __syncthreads();
        if (ovf == first) {
            // exact scan of every keypoint for the overflowed point (all earlier points are final)
            const int2 me = reproj[ovf];
            unsigned long long best = ~0ull;
            for (int j = tid; j < n_keys; j += 1024) {
                if (load_shared_int(&claimed[j])) continue;
                const int2 k = key_xy[j];
                const double dx = (double)k.x - (double)me.x, dy = (double)k.y - (double)me.y;
                if (sqrt(dx * dx + dy * dy) > (double)radius) continue;
                const float d = exact_distance(tp_desc + (int64_t)ovf * kDescDim, key_desc + (int64_t)j * kDescDim);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
                if (key < best) best = key;
            }
            if (best != ~0ull) atomicMin(&slow_best, best);
            __syncthreads();
            if (tid == 0) {
                const unsigned long long b = slow_best;
                if (b != ~0ull && (double)__uint_as_float((unsigned)(b >> 32)) < accept) {
                    const int j = (int)(b & 0xffffffffu);
                    store_shared_int(&claimed[j], 1);
                    tracks[ovf] = j;
                }
                store_shared_int(&done[ovf], 1);
            }
        }
    }
}

// Frame::setFrameParam's inv_K_ (frame.cc:97-102): Eigen's cofactor inverse of
// [[fx 0 cx][0 fy cy][0 0 1]] (cofactor<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1)).
__device__ __forceinline__ void inv_k3(const Intr4 &K, float *invK)
{
    const float fx = K.fx, fy = K.fy, cx = K.cx, cy = K.cy;
    const float c00 = fy * 1.0f - cy * 0.0f;
    const float c10 = 0.0f * cx - 1.0f * 0.0f;
    const float c20 = 0.0f * cy - cx * fy;
    const float det = c00 * fx + (c10 * 0.0f + c20 * 0.0f);
    const float invdet = 1.0f / det;
    invK[0] = c00 * invdet;
    invK[1] = c10 * invdet;
    invK[2] = c20 * invdet;
    invK[3] = (cy * 0.0f - 0.0f * 1.0f) * invdet;
    invK[4] = (1.0f * fx - 0.0f * cx) * invdet;
    invK[5] = (cx * 0.0f - fx * cy) * invdet;
    invK[6] = (0.0f * 0.0f - fy * 0.0f) * invdet;
    invK[7] = (0.0f * 0.0f - 0.0f * fx) * invdet;
    invK[8] = (fx * fy - 0.0f * 0.0f) * invdet;
}

// Depth and back-projection of Frame::computePoint3d (frame.cc:70-85).
__global__ void __launch_bounds__(256)
points3d_kernel(Pose T, Intr4 K, float bf, const int2 *__restrict__ xy0, int n0,
                const int2 *__restrict__ xy1, const int *__restrict__ minloc,
                unsigned char *__restrict__ good, float *__restrict__ Pw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n0) return;
    good[i] = 0;
    Pw[3 * (int64_t)i] = 0.f; Pw[3 * (int64_t)i + 1] = 0.f; Pw[3 * (int64_t)i + 2] = 0.f;
    const int loc = minloc[i];
    if (loc < 0) return;
    const int2 l = xy0[i];
    const float disparity = (float)(l.x - xy1[loc].x);
    if (disparity <= 1) return;
    const float depth = bf / disparity;
    float invK[9];
    inv_k3(K, invK);
    const float ip[3] = { (float)l.x, (float)l.y, 1.0f };
    float pc[3];
    for (int r = 0; r < 3; r++)
        pc[r] = invK[3 * r] * ip[0] + (invK[3 * r + 1] * ip[1] + invK[3 * r + 2] * ip[2]);
    float v[3] = { pc[0] / pc[2], 1.0f, -pc[1] / pc[2] };
    for (int k = 0; k < 3; k++) v[k] = v[k] * depth;
    const Pose inv = rigid_inverse_f(T);
    float o[3];
    rigid_apply_f(inv, v, o);
    Pw[3 * (int64_t)i] = o[0]; Pw[3 * (int64_t)i + 1] = o[1]; Pw[3 * (int64_t)i + 2] = o[2];
    good[i] = 1;
}

// ---- Tracking::updateTrackedPoints (tracking.cc:378-451): two-view linear triangulation ---------
// Eigen Quaternionf::normalized().toRotationMatrix() in float.
__device__ __forceinline__ void quat_to_rot_f(const float *q, float *R)
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

// Eigen Quaternionf::inverse(): conjugate coefficients / squaredNorm
__device__ __forceinline__ void quat_inverse_f(const float *q, float *o)
{
    const float n2 = (q[1] * q[1] + q[2] * q[2]) + (q[3] * q[3] + q[0] * q[0]);
    o[0] = q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = -q[3] / n2;
}

// Right singular vector of the smallest singular value of the 4 x 4 matrix A (what
// JacobiSVD::matrixV().col(3) holds up to sign), by one-sided Jacobi rotations on the columns of A
// in double: the columns are made mutually orthogonal, V collects the rotations, and the column
// of smallest norm marks the wanted column of V.
__device__ __forceinline__ void null_vector4(double A[4][4], double *v)
{
    double V[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) V[r][c] = r == c ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q];
                }
                if (ga * ga > 1e-32 * al * be && ga != 0.0) {
                    rotated = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const double ap = A[k][p], aq = A[k][q];
                        A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
                        const double vp = V[k][p], vq = V[k][q];
                        V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                    }
                }
            }
        if (!rotated) break;
    }
    double best = 0;
    int m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        double nn = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) nn += A[k][c] * A[k][c];
        if (c == 0 || nn < best) { best = nn; m = c; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = m == 0 ? V[k][0] : m == 1 ? V[k][1] : m == 2 ? V[k][2] : V[k][3];
}

// One thread per tracked point.
__global__ void __launch_bounds__(128)
triangulate_kernel(Pose T1, Pose T2, Intr4 K, const int2 *__restrict__ keys1, int n_keys1,
                   const int *__restrict__ tracks1, const int2 *__restrict__ keys2, int n_keys2,
                   const int *__restrict__ tracks2, int n_tp, float *__restrict__ Pw,
                   unsigned char *__restrict__ updated, int *__restrict__ count)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n_tp) return;
    updated[i] = 0;
    const int id1 = tracks1[i], id2 = tracks2[i];
    if (id1 < 0 || id2 < 0 || id1 >= n_keys1 || id2 >= n_keys2) return;
    float M1[3][4], M2[3][4], R[9];
    quat_to_rot_f(T1.q, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M1[r][c] = R[3 * r + c]; M1[r][3] = T1.t[r]; }
    quat_to_rot_f(T2.q, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M2[r][c] = R[3 * r + c]; M2[r][3] = T2.t[r]; }
    const int2 k1 = keys1[id1], k2 = keys2[id2];
    const float invfx = 1.0f / K.fx, invfy = 1.0f / K.fy;
    const float xn1[3] = { ((float)k1.x - K.cx) * invfx, 1.0f, -((float)k1.y - K.cy) * invfy };
    const float xn2[3] = { ((float)k2.x - K.cx) * invfx, 1.0f, -((float)k2.y - K.cy) * invfy };
    float qi[4], ray1[3], ray2[3];
    quat_inverse_f(T1.q, qi); quat_rotate_f(qi, xn1, ray1);
    quat_inverse_f(T2.q, qi); quat_rotate_f(qi, xn2, ray2);
    const float dot = (ray1[0] * ray2[0] + ray1[1] * ray2[1]) + ray1[2] * ray2[2];
    const float n1 = sqrtf((ray1[0] * ray1[0] + ray1[1] * ray1[1]) + ray1[2] * ray1[2]);
    const float n2 = sqrtf((ray2[0] * ray2[0] + ray2[1] * ray2[1]) + ray2[2] * ray2[2]);
    const float cosp = dot / (n1 * n2);
    if (cosp < 0 || (double)cosp > 0.9998) return;
    double A[4][4];
    for (int c = 0; c < 4; c++) {
        A[0][c] = (double)(xn1[0] * M1[1][c] - M1[0][c]);
        A[1][c] = (double)(xn1[2] * M1[1][c] - M1[2][c]);
        A[2][c] = (double)(xn2[0] * M2[1][c] - M2[0][c]);
        A[3][c] = (double)(xn2[2] * M2[1][c] - M2[2][c]);
    }
    double v[4];
    null_vector4(A, v);
    const float x3[4] = { (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
    if (x3[3] == 0) return;
    const float P[3] = { x3[0] / x3[3], x3[1] / x3[3], x3[2] / x3[3] };
    float Pc[3];
    rigid_apply_f(T1, P, Pc);
    if ((double)Pc[1] <= 0.2) return;
    rigid_apply_f(T2, P, Pc);
    if ((double)Pc[1] <= 0.2) return;
    Pw[3 * (int64_t)i] = P[0]; Pw[3 * (int64_t)i + 1] = P[1]; Pw[3 * (int64_t)i + 2] = P[2];
    updated[i] = 1;
    atomicAdd(count, 1);
}

Pose make_pose(const float *p)
{
    Pose T;
    for (int k = 0; k < 3; k++) T.t[k] = p[k];
    for (int k = 0; k < 4; k++) T.q[k] = p[3 + k];
    return T;
}

int launch_ok(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

int require_device()
{
    int ndev = 0;
    VS_HIP(hipGetDeviceCount(&ndev));
    VS_CHECK(ndev > 0, VS_ERR_HIP, "no HIP device available");
    return VS_OK;
}

// Small by-value arguments (poses, intrinsics) may arrive as host or device pointers like every
// other argument: they are copied into host temporaries before the host reads them.
struct HostArgs {
    float pose[2][7];
    float K4[4];
};
int stage_args(HostArgs &a, const float *pose1, const float *pose2, const float *K4)
{
    VS_HIP(hipMemcpy(a.pose[0], pose1, 7 * sizeof(float), hipMemcpyDefault));
    if (pose2) VS_HIP(hipMemcpy(a.pose[1], pose2, 7 * sizeof(float), hipMemcpyDefault));
    VS_HIP(hipMemcpy(a.K4, K4, 4 * sizeof(float), hipMemcpyDefault));
    return VS_OK;
}

template <typename T>
int fill_output(T *dst, size_t count, T value)
{
    std::vector<T> h(count, value);
    if (count) VS_HIP(hipMemcpy(dst, h.data(), count * sizeof(T), hipMemcpyDefault));
    return VS_OK;
}

template <typename T>
int upload(DevBuf<T> &buf, const void *src, size_t count)
{
    VS_TRY(buf.alloc(std::max<size_t>(count, 1)));
    if (count) VS_HIP(hipMemcpy(buf.p, src, count * sizeof(T), hipMemcpyDefault));
    return VS_OK;
}

}  // namespace

extern "C" {

int vs_reproject_points(const float *pose, const float *K4, const float *xyz, int32_t n,
                        int32_t *xy)
{
    VS_CHECK(pose && K4 && n >= 0 && (n == 0 || (xyz && xy)), VS_ERR_INVALID, "bad arguments");
    if (n == 0) return VS_OK;
    VS_TRY(require_device());
    DevBuf<float> d_xyz;
    DevBuf<int2> d_xy;
    VS_TRY(upload(d_xyz, xyz, (size_t)n * 3));
    VS_TRY(d_xy.alloc(n));
    HostArgs ha;
    VS_TRY(stage_args(ha, pose, nullptr, K4));
    const Intr4 K{ ha.K4[0], ha.K4[1], ha.K4[2], ha.K4[3] };
    hipLaunchKernelGGL(reproject_kernel, dim3(div_up(n, 256)), dim3(256), 0, 0, make_pose(ha.pose[0]), K, d_xyz.p, n, d_xy.p);
    VS_TRY(launch_ok("reproject_kernel"));
    VS_HIP(hipMemcpy(xy, d_xy.p, (size_t)n * sizeof(int2), hipMemcpyDefault));
    return VS_OK;
}

int vs_search_projection(const float *pose, const float *K4, const float *tp_xyz,
                         const float *tp_desc, int32_t n_tp, const int32_t *key_xy,
                         const float *key_desc, int32_t n_keys, float radius, double accept,
                         uint8_t *claimed, int32_t *tracks)
{
    VS_CHECK(pose && K4 && n_tp >= 0 && n_keys >= 0, VS_ERR_INVALID, "bad arguments");
    VS_CHECK(n_tp == 0 || (tp_xyz && tp_desc && tracks), VS_ERR_INVALID, "null tracked-point arrays");
    VS_CHECK(n_keys == 0 || (key_xy && key_desc && claimed), VS_ERR_INVALID, "null keypoint arrays");
    if (n_tp == 0) return VS_OK;
    VS_TRY(require_device());
    if (n_keys == 0) return fill_output<int32_t>(tracks, (size_t)n_tp, -1);
    DevBuf<float> d_xyz, d_tdesc, d_kdesc;
    DevBuf<int2> d_re, d_kxy;
    DevBuf<unsigned long long> d_cand;
    DevBuf<int> d_ncand, d_claimed, d_owner, d_done, d_tracks;
    DevBuf<unsigned char> d_over;
    VS_TRY(upload(d_xyz, tp_xyz, (size_t)n_tp * 3));
    VS_TRY(upload(d_tdesc, tp_desc, (size_t)n_tp * kDescDim));
    VS_TRY(upload(d_kdesc, key_desc, (size_t)n_keys * kDescDim));
    VS_TRY(upload(d_kxy, key_xy, (size_t)n_keys));
    VS_TRY(d_re.alloc(n_tp)); VS_TRY(d_cand.alloc((size_t)n_tp * kCandCap)); VS_TRY(d_ncand.alloc(n_tp));
    VS_TRY(d_over.alloc(n_tp)); VS_TRY(d_owner.alloc(n_keys)); VS_TRY(d_done.alloc(n_tp)); VS_TRY(d_tracks.alloc(n_tp));
    std::vector<uint8_t> b_claimed(n_keys);
    VS_HIP(hipMemcpy(b_claimed.data(), claimed, (size_t)n_keys, hipMemcpyDefault));
    std::vector<int> h_claimed(n_keys);
    for (int j = 0; j < n_keys; j++) h_claimed[j] = b_claimed[j] ? 1 : 0;
    VS_TRY(upload(d_claimed, h_claimed.data(), (size_t)n_keys));
    HostArgs ha;
    VS_TRY(stage_args(ha, pose, nullptr, K4));
    const Intr4 K{ ha.K4[0], ha.K4[1], ha.K4[2], ha.K4[3] };
    hipLaunchKernelGGL(reproject_kernel, dim3(div_up(n_tp, 256)), dim3(256), 0, 0, make_pose(ha.pose[0]), K, d_xyz.p, n_tp, d_re.p);
    hipLaunchKernelGGL(gated_nn_kernel<kWindow>, dim3(n_tp), dim3(64), 0, 0, d_tdesc.p, d_re.p, n_tp, d_kdesc.p, d_kxy.p,
                       n_keys, 2.0f * ha.K4[2], 2.0f * ha.K4[3], radius, accept, d_claimed.p, d_cand.p, d_ncand.p, d_over.p, (int *)nullptr);
    hipLaunchKernelGGL(greedy_resolve_kernel, dim3(1), dim3(1024), 0, 0, n_tp, n_keys, d_cand.p, d_ncand.p, d_over.p,
                       d_tdesc.p, d_re.p, d_kdesc.p, d_kxy.p, radius, accept, d_claimed.p, d_owner.p, d_done.p, d_tracks.p);
    VS_TRY(launch_ok("search projection kernels"));
    VS_HIP(hipMemcpy(tracks, d_tracks.p, (size_t)n_tp * sizeof(int), hipMemcpyDefault));
    VS_HIP(hipMemcpy(h_claimed.data(), d_claimed.p, (size_t)n_keys * sizeof(int), hipMemcpyDefault));
    for (int j = 0; j < n_keys; j++) b_claimed[j] = h_claimed[j] ? 1 : 0;
    VS_HIP(hipMemcpy(claimed, b_claimed.data(), (size_t)n_keys, hipMemcpyDefault));
    return VS_OK;
}

int vs_remove_outliers(const float *pose, const float *K4, const float *tp_xyz, int32_t n_tp,
                       const int32_t *key_xy, int32_t n_keys, double max_dist, int32_t *tracks,
                       int32_t *n_removed)
{
    VS_CHECK(pose && K4 && n_tp >= 0 && n_keys >= 0, VS_ERR_INVALID, "bad arguments");
    VS_CHECK(n_tp == 0 || (tp_xyz && tracks), VS_ERR_INVALID, "null tracked-point arrays");
    if (n_removed) *n_removed = 0;
    if (n_tp == 0 || n_keys == 0) return VS_OK;
    VS_CHECK(key_xy != nullptr, VS_ERR_INVALID, "null keypoint array");
    VS_TRY(require_device());
    DevBuf<float> d_xyz;
    DevBuf<int2> d_kxy;
    DevBuf<int> d_tracks, d_removed;
    VS_TRY(upload(d_xyz, tp_xyz, (size_t)n_tp * 3));
    VS_TRY(upload(d_kxy, key_xy, (size_t)n_keys));
    VS_TRY(upload(d_tracks, tracks, (size_t)n_tp));
    VS_TRY(d_removed.alloc(1));
    VS_HIP(hipMemset(d_removed.p, 0, sizeof(int)));
    HostArgs ha;
    VS_TRY(stage_args(ha, pose, nullptr, K4));
    const Intr4 K{ ha.K4[0], ha.K4[1], ha.K4[2], ha.K4[3] };
    hipLaunchKernelGGL(remove_outliers_kernel, dim3(div_up(n_tp, 256)), dim3(256), 0, 0, make_pose(ha.pose[0]), K, d_xyz.p, n_tp,
                       d_kxy.p, n_keys, max_dist, d_tracks.p, d_removed.p);
    VS_TRY(launch_ok("remove_outliers_kernel"));
    VS_HIP(hipMemcpy(tracks, d_tracks.p, (size_t)n_tp * sizeof(int), hipMemcpyDefault));
    int removed = 0;
    VS_HIP(hipMemcpy(&removed, d_removed.p, sizeof(int), hipMemcpyDefault));
    if (n_removed) *n_removed = removed;
    return VS_OK;
}

int vs_compute_points3d(const float *desc0, const int32_t *xy0, int32_t n0, const float *desc1,
                        const int32_t *xy1, int32_t n1, const float *K4, float baseline_fx,
                        const float *pose, uint8_t *good, float *xyz, int32_t *minloc)
{
    VS_CHECK(pose && K4 && n0 >= 0 && n1 >= 0, VS_ERR_INVALID, "bad arguments");
    VS_CHECK(n0 == 0 || (desc0 && xy0 && good && xyz), VS_ERR_INVALID, "null left arrays");
    VS_CHECK(n1 == 0 || (desc1 && xy1), VS_ERR_INVALID, "null right arrays");
    if (n0 == 0) return VS_OK;
    VS_TRY(require_device());
    DevBuf<float> d_d0, d_d1, d_pw;
    DevBuf<int2> d_xy0, d_xy1;
    DevBuf<int> d_loc;
    DevBuf<unsigned char> d_good;
    VS_TRY(upload(d_d0, desc0, (size_t)n0 * kDescDim));
    VS_TRY(upload(d_d1, desc1, (size_t)n1 * kDescDim));
    VS_TRY(upload(d_xy0, xy0, (size_t)n0));
    VS_TRY(upload(d_xy1, xy1, (size_t)n1));
    VS_TRY(d_loc.alloc(n0)); VS_TRY(d_good.alloc(n0)); VS_TRY(d_pw.alloc((size_t)n0 * 3));
    HostArgs ha;
    VS_TRY(stage_args(ha, pose, nullptr, K4));
    const Intr4 K{ ha.K4[0], ha.K4[1], ha.K4[2], ha.K4[3] };
    hipLaunchKernelGGL(gated_nn_kernel<kStereo>, dim3(n0), dim3(64), 0, 0, d_d0.p, d_xy0.p, n0, d_d1.p, d_xy1.p, n1,
                       0.f, 0.f, 0.f, 0.0, (const int *)nullptr, (unsigned long long *)nullptr, (int *)nullptr, (unsigned char *)nullptr, d_loc.p);
    hipLaunchKernelGGL(points3d_kernel, dim3(div_up(n0, 256)), dim3(256), 0, 0, make_pose(ha.pose[0]), K, baseline_fx, d_xy0.p,
                       n0, d_xy1.p, d_loc.p, d_good.p, d_pw.p);
    VS_TRY(launch_ok("computePoint3d kernels"));
    VS_HIP(hipMemcpy(good, d_good.p, (size_t)n0, hipMemcpyDefault));
    VS_HIP(hipMemcpy(xyz, d_pw.p, (size_t)n0 * 3 * sizeof(float), hipMemcpyDefault));
    if (minloc) VS_HIP(hipMemcpy(minloc, d_loc.p, (size_t)n0 * sizeof(int), hipMemcpyDefault));
    return VS_OK;
}

int vs_triangulate_tracks(const float *pose1, const float *pose2, const float *K4,
                          const int32_t *key_xy1, int32_t n_keys1, const int32_t *tracks1,
                          const int32_t *key_xy2, int32_t n_keys2, const int32_t *tracks2,
                          int32_t n_tp, float *xyz, uint8_t *updated, int32_t *n_updated)
{
    VS_CHECK(pose1 && pose2 && K4 && n_tp >= 0 && n_keys1 >= 0 && n_keys2 >= 0, VS_ERR_INVALID, "bad arguments");
    VS_CHECK(n_tp == 0 || (tracks1 && tracks2 && xyz), VS_ERR_INVALID, "null tracked-point arrays");
    VS_CHECK((n_keys1 == 0 || key_xy1) && (n_keys2 == 0 || key_xy2), VS_ERR_INVALID, "null keypoint arrays");
    if (n_updated) *n_updated = 0;
    if (n_tp == 0) return VS_OK;
    VS_TRY(require_device());
    if (n_keys1 == 0 || n_keys2 == 0) return updated ? fill_output<uint8_t>(updated, (size_t)n_tp, 0) : VS_OK;
    DevBuf<float> d_xyz;
    DevBuf<int2> d_k1, d_k2;
    DevBuf<int> d_t1, d_t2, d_count;
    DevBuf<unsigned char> d_upd;
    VS_TRY(upload(d_xyz, xyz, (size_t)n_tp * 3));
    VS_TRY(upload(d_k1, key_xy1, (size_t)n_keys1));
    VS_TRY(upload(d_k2, key_xy2, (size_t)n_keys2));
    VS_TRY(upload(d_t1, tracks1, (size_t)n_tp));
    VS_TRY(upload(d_t2, tracks2, (size_t)n_tp));
    VS_TRY(d_upd.alloc(n_tp)); VS_TRY(d_count.alloc(1));
    VS_HIP(hipMemset(d_count.p, 0, sizeof(int)));
    HostArgs ha;
    VS_TRY(stage_args(ha, pose1, pose2, K4));
    const Intr4 K{ ha.K4[0], ha.K4[1], ha.K4[2], ha.K4[3] };
    hipLaunchKernelGGL(triangulate_kernel, dim3(div_up(n_tp, 128)), dim3(128), 0, 0, make_pose(ha.pose[0]), make_pose(ha.pose[1]), K,
                       d_k1.p, n_keys1, d_t1.p, d_k2.p, n_keys2, d_t2.p, n_tp, d_xyz.p, d_upd.p, d_count.p);
    VS_TRY(launch_ok("triangulate_kernel"));
    VS_HIP(hipMemcpy(xyz, d_xyz.p, (size_t)n_tp * 3 * sizeof(float), hipMemcpyDefault));
    if (updated) VS_HIP(hipMemcpy(updated, d_upd.p, (size_t)n_tp, hipMemcpyDefault));
    int count = 0;
    VS_HIP(hipMemcpy(&count, d_count.p, sizeof(int), hipMemcpyDefault));
    if (n_updated) *n_updated = count;
    return VS_OK;
}

}  // extern "C"
