// Device kernels of the bundle-adjustment path (gfx950, wave64).
//
// Data layout (all device-resident, fp64 values, int32 indices):
//   cams     Nc x CamData   R (row-major, R = Rz*Ry*Rx), t, sin/cos(yaw): built by prep_cameras
//   pts      Np x 3
//   landmark-major observation stream (observations sorted by landmark, "pos" order):
//     pt_ptr Np+1, lcam[pos], luv[2*pos..], lobs[pos] (caller's index), lpt[pos]
//   camera-major view: cam_ptr Nc+1, cobs[k] = pos
//   off-diagonal blocks of S: brow/bcol per block, bptr, bent[e] = landmark seen by both cameras
//   work items: (block, [begin,end)) chunks, one wave each; partial sums per item
//
// No kernel uses global atomics: every sum is a per-wave register accumulation, a wave-level
// transposed reduction through LDS, a per-item partial in global memory and a fixed-order
// finalize pass, so results are bitwise reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

typedef double real;

struct CamData {
    real R[9];
    real t[3];
    real sy, cy;
    real pad[2];
};
static_assert(sizeof(CamData) == 128, "CamData is one 128-byte line");

struct Intr {
    real fx, fy, cx, cy;
};

constexpr int kWave = 64;
constexpr int kLmThreads = 256;   // landmark-major kernels: one thread per landmark
constexpr int kItemWaves = 4;     // block-major kernels: one wave per work item
constexpr int kRedStride = 65;    // padded row of the transposed wave reduction (conflict-free)
constexpr int kDiagVals = 39;     // S_cc sym 21 | rhs 6 | diag(U) 6 | g_c 6
constexpr int kOffVals = 36;      // S_ab 6x6

// Per-observation linearisation of the reference's reprojection residual
// (reference: src/core/reprojection_error.h:22-51; camera looks along +Y).
struct ObsLin {
    real r0, r1;             // residual
    real a00, a01, a11, a12; // d res / d Pc = [[a00 a01 0],[0 a11 a12]]  (= d res / d t)
    real q[3];               // R * Pw
};

__device__ __forceinline__ void project_obs(const CamData &c, const real *P, real u, real v,
                                            const Intr &K, ObsLin &o)
{
    o.q[0] = c.R[0] * P[0] + c.R[1] * P[1] + c.R[2] * P[2];
    o.q[1] = c.R[3] * P[0] + c.R[4] * P[1] + c.R[5] * P[2];
    o.q[2] = c.R[6] * P[0] + c.R[7] * P[1] + c.R[8] * P[2];
    const real x = o.q[0] + c.t[0], y = o.q[1] + c.t[1], z = o.q[2] + c.t[2];
    const real iy = real(1) / y;
    const real xi = x * iy, zi = z * iy;
    o.r0 = K.fx * xi + K.cx - u;
    o.r1 = -K.fy * zi + K.cy - v;
    o.a00 = K.fx * iy;
    o.a01 = -K.fx * xi * iy;
    o.a11 = K.fy * zi * iy;
    o.a12 = -K.fy * iy;
}

// Ceres HuberLoss(a) corrector for a block with rho'' <= 0: residual and Jacobian rows are
// scaled by sqrt(rho'), cost contribution is rho/2.  Returns the scale; *half_rho = rho/2.
__device__ __forceinline__ real huber_weight(real a, real s, real *half_rho)
{
    if (a > real(0) && s > a * a) {
        const real rs = sqrt(s);
        *half_rho = real(0.5) * (real(2) * a * rs - a * a);
        return sqrt(a / rs);
    }
    *half_rho = real(0.5) * s;
    return real(1);
}

__device__ __forceinline__ void scale_lin(ObsLin &o, real w)
{
    o.r0 *= w; o.r1 *= w;
    o.a00 *= w; o.a01 *= w; o.a11 *= w; o.a12 *= w;
}

// Jp = A * R (2x3 row-major).
__device__ __forceinline__ void point_jac(const CamData &c, const ObsLin &o, real *Jp)
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Jp[k] = o.a00 * c.R[k] + o.a01 * c.R[3 + k];
        Jp[3 + k] = o.a11 * c.R[3 + k] + o.a12 * c.R[6 + k];
    }
}

// Jc = [A | A*M] (2x6 row-major), M = dPc/d(roll,pitch,yaw):
//   dPc/droll = R*(x^ x Pw), dPc/dpitch = (Rz y^) x q, dPc/dyaw = z^ x q.
__device__ __forceinline__ void camera_jac(const CamData &c, const ObsLin &o, const real *P,
                                           real *Jc)
{
    real m0[3], m1[3], m2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) m0[i] = c.R[3 * i + 2] * P[1] - c.R[3 * i + 1] * P[2];
    m1[0] = c.cy * o.q[2];
    m1[1] = c.sy * o.q[2];
    m1[2] = -c.sy * o.q[1] - c.cy * o.q[0];
    m2[0] = -o.q[1];
    m2[1] = o.q[0];
    m2[2] = real(0);
    Jc[0] = o.a00; Jc[1] = o.a01; Jc[2] = real(0);
    Jc[6] = real(0); Jc[7] = o.a11; Jc[8] = o.a12;
    Jc[3] = o.a00 * m0[0] + o.a01 * m0[1];
    Jc[4] = o.a00 * m1[0] + o.a01 * m1[1];
    Jc[5] = o.a00 * m2[0] + o.a01 * m2[1];
    Jc[9] = o.a11 * m0[1] + o.a12 * m0[2];
    Jc[10] = o.a11 * m1[1] + o.a12 * m1[2];
    Jc[11] = o.a11 * m2[1];
}

// W = Jc^T Jp (6x3 row-major)
__device__ __forceinline__ void w_block(const real *Jc, const real *Jp, real *W)
{
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) W[3 * p + k] = Jc[p] * Jp[k] + Jc[6 + p] * Jp[3 + k];
}

// Z = W * Vinv, Vinv symmetric stored (xx,xy,xz,yy,yz,zz)
__device__ __forceinline__ void z_block(const real *W, const real *vi, real *Z)
{
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const real w0 = W[3 * p], w1 = W[3 * p + 1], w2 = W[3 * p + 2];
        Z[3 * p] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        Z[3 * p + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        Z[3 * p + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
    }
}

__device__ __forceinline__ void load_cam(const CamData *cams, int c, CamData &out)
{
    const double2 *src = reinterpret_cast<const double2 *>(cams + c);
    double2 *dst = reinterpret_cast<double2 *>(&out);
#pragma unroll
    for (int i = 0; i < 7; i++) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
__global__ void prep_cameras_kernel(int Nc, const real *__restrict__ t,
                                    const real *__restrict__ rpy, CamData *__restrict__ cams)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nc) return;
    real sr, cr, sp, cp, sy, cy;
    sincos(rpy[3 * c], &sr, &cr);
    sincos(rpy[3 * c + 1], &sp, &cp);
    sincos(rpy[3 * c + 2], &sy, &cy);
    CamData d;
    d.R[0] = cy * cp; d.R[1] = cy * sp * sr - sy * cr; d.R[2] = cy * sp * cr + sy * sr;
    d.R[3] = sy * cp; d.R[4] = sy * sp * sr + cy * cr; d.R[5] = sy * sp * cr - cy * sr;
    d.R[6] = -sp;     d.R[7] = cp * sr;                d.R[8] = cp * cr;
    d.t[0] = t[3 * c]; d.t[1] = t[3 * c + 1]; d.t[2] = t[3 * c + 2];
    d.sy = sy; d.cy = cy; d.pad[0] = d.pad[1] = real(0);
    cams[c] = d;
}

// Block-wide sum of one value per thread (256 threads), result valid in thread 0.
__device__ __forceinline__ real block_sum_256(real v, real *scratch /* >= 4 reals */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    real s = real(0);
    if (threadIdx.x == 0) s = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    return s;
}

// Damping mode of the landmark kernel.
//   lam_p != NULL : explicit per-parameter damping (3*Np)
//   scale_p != NULL : LM diagonal  lam = clamp(s^2 Vdd, min_d, max_d) / (radius s^2)
//   both NULL : none
struct PointDamping {
    const real *lam_p;
    const real *scale_p;
    real radius, min_d, max_d;
};

struct HuberObs {
    int pos;        // landmark-major position of the Huber-weighted observation or -1
    int cam, pt;    // its camera and landmark
    real a;         // Huber parameter
};

// (a)+(b), landmark side.  One thread per landmark j: V_j = sum Jp^T Jp, g_p_j = sum Jp^T r,
// cost, then Vinv_j = (V_j + damping)^-1.  Cameras come from LDS when they fit (STAGE_CAMS).
template <bool STAGE_CAMS>
__global__ void __launch_bounds__(kLmThreads)
landmark_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams,
                const real *__restrict__ pts, const int *__restrict__ pt_ptr,
                const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
                PointDamping damp, real *__restrict__ Vinv6, real *__restrict__ gp,
                real *__restrict__ Vdiag, real *__restrict__ hub_w, real *__restrict__ cost_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CamData *scams = reinterpret_cast<CamData *>(smem_raw);
    __shared__ real red[4];
    if (STAGE_CAMS) {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = threadIdx.x; i < Nc * 8; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    real cost = real(0);
    if (j < Np) {
        const int b = pt_ptr[j], e = pt_ptr[j + 1];
        const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
        real V[6] = { 0, 0, 0, 0, 0, 0 }, g[3] = { 0, 0, 0 };
        for (int pos = b; pos < e; pos++) {
            const int c = lcam[pos];
            CamData cd;
            if (STAGE_CAMS) cd = scams[c]; else load_cam(cams, c, cd);
            ObsLin o;
            project_obs(cd, P, luv[2 * pos], luv[2 * pos + 1], K, o);
            real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
            if (pos == hub.pos) {
                const real w = huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
                scale_lin(o, w);
                *hub_w = w;
            }
            cost += hr;
            real Jp[6];
            point_jac(cd, o, Jp);
            V[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
            V[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
            V[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
            V[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
            V[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
            V[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
            g[0] += Jp[0] * o.r0 + Jp[3] * o.r1;
            g[1] += Jp[1] * o.r0 + Jp[4] * o.r1;
            g[2] += Jp[2] * o.r0 + Jp[5] * o.r1;
        }
        if (Vdiag) { Vdiag[3 * j] = V[0]; Vdiag[3 * j + 1] = V[3]; Vdiag[3 * j + 2] = V[5]; }
        real vi[6] = { 0, 0, 0, 0, 0, 0 };
        if (e > b) {
            real l0 = 0, l1 = 0, l2 = 0;
            if (damp.lam_p) {
                l0 = damp.lam_p[3 * j]; l1 = damp.lam_p[3 * j + 1]; l2 = damp.lam_p[3 * j + 2];
            } else if (damp.scale_p) {
                const real d[3] = { V[0], V[3], V[5] };
                real l[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const real s = damp.scale_p[3 * j + k], s2 = s * s;
                    real x = s2 * d[k];
                    x = x < damp.min_d ? damp.min_d : x;
                    x = x > damp.max_d ? damp.max_d : x;
                    l[k] = x / damp.radius / s2;
                }
                l0 = l[0]; l1 = l[1]; l2 = l[2];
            }
            const real a = V[0] + l0, bb = V[1], cc = V[2], d = V[3] + l1, ee = V[4], f = V[5] + l2;
            const real c00 = d * f - ee * ee, c01 = cc * ee - bb * f, c02 = bb * ee - cc * d;
            const real id = real(1) / (a * c00 + bb * c01 + cc * c02);
            vi[0] = c00 * id; vi[1] = c01 * id; vi[2] = c02 * id;
            vi[3] = (a * f - cc * cc) * id; vi[4] = (bb * cc - a * ee) * id;
            vi[5] = (a * d - bb * bb) * id;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) Vinv6[6 * j + k] = vi[k];
#pragma unroll
        for (int k = 0; k < 3; k++) gp[3 * j + k] = g[k];
    }
    const real s = block_sum_256(cost, red);
    if (threadIdx.x == 0) cost_partial[blockIdx.x] = s;
}

// Cost only (candidate evaluation inside LM): one thread per landmark.
template <bool STAGE_CAMS>
__global__ void __launch_bounds__(kLmThreads)
cost_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams, const real *__restrict__ pts,
            const int *__restrict__ pt_ptr, const int *__restrict__ lcam,
            const real *__restrict__ luv, HuberObs hub, real *__restrict__ cost_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CamData *scams = reinterpret_cast<CamData *>(smem_raw);
    __shared__ real red[4];
    if (STAGE_CAMS) {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = threadIdx.x; i < Nc * 8; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    real cost = real(0);
    if (j < Np) {
        const int b = pt_ptr[j], e = pt_ptr[j + 1];
        const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
        for (int pos = b; pos < e; pos++) {
            const int c = lcam[pos];
            CamData cd;
            if (STAGE_CAMS) cd = scams[c]; else load_cam(cams, c, cd);
            ObsLin o;
            project_obs(cd, P, luv[2 * pos], luv[2 * pos + 1], K, o);
            real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
            if (pos == hub.pos) (void)huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
            cost += hr;
        }
    }
    const real s = block_sum_256(cost, red);
    if (threadIdx.x == 0) cost_partial[blockIdx.x] = s;
}

// (a) standalone: per-observation residual and Jacobians written out in the caller's order.
__global__ void __launch_bounds__(256)
obs_jacobian_kernel(int No, Intr K, const CamData *__restrict__ cams, const real *__restrict__ pts,
                    const int *__restrict__ lcam, const int *__restrict__ lpt,
                    const real *__restrict__ luv, const int *__restrict__ lobs, HuberObs hub,
                    real *__restrict__ res, real *__restrict__ Jc_out, real *__restrict__ Jp_out)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= No) return;
    const int c = lcam[pos], j = lpt[pos];
    CamData cd;
    load_cam(cams, c, cd);
    const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
    ObsLin o;
    project_obs(cd, P, luv[2 * pos], luv[2 * pos + 1], K, o);
    if (pos == hub.pos) {
        real hr;
        scale_lin(o, huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr));
    }
    const int64_t dst = lobs[pos];
    if (res) { res[2 * dst] = o.r0; res[2 * dst + 1] = o.r1; }
    if (Jc_out) {
        real Jc[12];
        camera_jac(cd, o, P, Jc);
#pragma unroll
        for (int k = 0; k < 12; k++) Jc_out[12 * dst + k] = Jc[k];
    }
    if (Jp_out) {
        real Jp[6];
        point_jac(cd, o, Jp);
#pragma unroll
        for (int k = 0; k < 6; k++) Jp_out[6 * dst + k] = Jp[k];
    }
}

// Transposed wave reduction: each lane holds NV partial values; lane v < NV ends up with the
// sum over the 64 lanes of value v, summed in lane order (fixed, reproducible).  Every thread of
// the block must call it (it synchronises the block before reusing the LDS rows).
template <int NV>
__device__ __forceinline__ real wave_reduce_transposed(const real *acc, real *lds_wave, int lane)
{
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; v++) lds_wave[v * kRedStride + lane] = acc[v];
    __syncthreads();
    real s = real(0);
    if (lane < NV) {
        const real *row = lds_wave + lane * kRedStride;
#pragma unroll 8
        for (int l = 0; l < kWave; l++) s += row[l];
    }
    return s;
}

// (a)+(b), camera side.  One wave per work item = a slice [begin,end) of one camera's
// observations (camera-major view).  Per observation: recompute r, Jc, Jp, W = Jc^T Jp and
// accumulate  S_cc += Jc^T Jc - W Vinv W^T (sym, 21), rhs += -Jc^T r + W Vinv g_p,
// diag(U) += diag(Jc^T Jc), g_c += Jc^T r.  With fix_points the Schur terms are skipped.
__global__ void __launch_bounds__(kItemWaves *kWave)
diag_kernel(int n_items, const int *__restrict__ item_cam, const int *__restrict__ item_begin,
            const int *__restrict__ item_end, Intr K, const CamData *__restrict__ cams,
            const real *__restrict__ pts, const int *__restrict__ cobs,
            const int *__restrict__ lpt, const real *__restrict__ luv, HuberObs hub,
            const real *__restrict__ hub_w, int fix_points, const real *__restrict__ Vinv6,
            const real *__restrict__ gp, real *__restrict__ partial /* n_items x kDiagVals */)
{
    __shared__ real red[kItemWaves * 21 * kRedStride];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * kItemWaves + wave;
    real acc[kDiagVals];
#pragma unroll
    for (int v = 0; v < kDiagVals; v++) acc[v] = real(0);
    if (item < n_items) {
        const int c = item_cam[item];
        const int b = item_begin[item], e = item_end[item];
        CamData cd;
        load_cam(cams, c, cd);
        for (int k = b + lane; k < e; k += kWave) {
            const int pos = cobs[k];
            const int j = lpt[pos];
            const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
            ObsLin o;
            project_obs(cd, P, luv[2 * pos], luv[2 * pos + 1], K, o);
            if (pos == hub.pos) scale_lin(o, *hub_w);
            real Jc[12];
            camera_jac(cd, o, P, Jc);
            // S_cc upper triangle (21), starting from U
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = p; q < 6; q++) acc[idx++] += Jc[p] * Jc[q] + Jc[6 + p] * Jc[6 + q];
            real gc[6];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                gc[p] = Jc[p] * o.r0 + Jc[6 + p] * o.r1;
                acc[21 + p] -= gc[p];
                acc[27 + p] += Jc[p] * Jc[p] + Jc[6 + p] * Jc[6 + p];
                acc[33 + p] += gc[p];
            }
            if (!fix_points) {
                real Jp[6], W[18], Z[18], vi[6];
                point_jac(cd, o, Jp);
                w_block(Jc, Jp, W);
#pragma unroll
                for (int m = 0; m < 6; m++) vi[m] = Vinv6[6 * j + m];
                z_block(W, vi, Z);
                const real g0 = gp[3 * j], g1 = gp[3 * j + 1], g2 = gp[3 * j + 2];
                idx = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) {
#pragma unroll
                    for (int q = p; q < 6; q++)
                        acc[idx++] -= Z[3 * p] * W[3 * q] + Z[3 * p + 1] * W[3 * q + 1] +
                                      Z[3 * p + 2] * W[3 * q + 2];
                    acc[21 + p] += Z[3 * p] * g0 + Z[3 * p + 1] * g1 + Z[3 * p + 2] * g2;
                }
            }
        }
    }
    real *lds_wave = red + wave * 21 * kRedStride;
    const real s1 = wave_reduce_transposed<21>(acc, lds_wave, lane);
    const real s2 = wave_reduce_transposed<18>(acc + 21, lds_wave, lane);
    if (item < n_items) {
        if (lane < 21) partial[(int64_t)item * kDiagVals + lane] = s1;
        if (lane < 18) partial[(int64_t)item * kDiagVals + 21 + lane] = s2;
    }
}

// (b) off-diagonal camera-pair blocks.  One wave per work item = a slice of the landmarks seen
// by both cameras (a < b) of one block:  S_ab -= W_a Vinv W_b^T, with W recomputed from the
// landmark position and the two (wave-uniform) cameras.
__global__ void __launch_bounds__(kItemWaves *kWave)
offdiag_kernel(int n_items, const int *__restrict__ item_block, const int *__restrict__ item_begin,
               const int *__restrict__ item_end, const int *__restrict__ brow,
               const int *__restrict__ bcol, Intr K, const CamData *__restrict__ cams,
               const real *__restrict__ pts, const int *__restrict__ bent, HuberObs hub,
               const real *__restrict__ hub_w, const real *__restrict__ Vinv6,
               real *__restrict__ partial /* n_items x kOffVals */)
{
    __shared__ real red[kItemWaves * 18 * kRedStride];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * kItemWaves + wave;
    real acc[kOffVals];
#pragma unroll
    for (int v = 0; v < kOffVals; v++) acc[v] = real(0);
    if (item < n_items) {
        const int blk = item_block[item];
        const int ca = brow[blk], cb = bcol[blk];
        const int b = item_begin[item], e = item_end[item];
        CamData A, B;
        load_cam(cams, ca, A);
        load_cam(cams, cb, B);
        for (int k = b + lane; k < e; k += kWave) {
            const int j = bent[k];
            const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
            real vi[6];
#pragma unroll
            for (int m = 0; m < 6; m++) vi[m] = Vinv6[6 * j + m];
            ObsLin oa, ob;
            // W does not depend on (u,v): pass zeros; only the Huber-weighted edge rescales.
            project_obs(A, P, real(0), real(0), K, oa);
            project_obs(B, P, real(0), real(0), K, ob);
            if (j == hub.pt) {
                if (ca == hub.cam) scale_lin(oa, *hub_w);
                if (cb == hub.cam) scale_lin(ob, *hub_w);
            }
            real Jc[12], Jp[6], Wa[18], Wb[18], Z[18];
            camera_jac(A, oa, P, Jc);
            point_jac(A, oa, Jp);
            w_block(Jc, Jp, Wa);
            camera_jac(B, ob, P, Jc);
            point_jac(B, ob, Jp);
            w_block(Jc, Jp, Wb);
            z_block(Wa, vi, Z);
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = 0; q < 6; q++)
                    acc[6 * p + q] -= Z[3 * p] * Wb[3 * q] + Z[3 * p + 1] * Wb[3 * q + 1] +
                                      Z[3 * p + 2] * Wb[3 * q + 2];
        }
    }
    real *lds_wave = red + wave * 18 * kRedStride;
    const real s1 = wave_reduce_transposed<18>(acc, lds_wave, lane);
    const real s2 = wave_reduce_transposed<18>(acc + 18, lds_wave, lane);
    if (item < n_items && lane < 18) {
        partial[(int64_t)item * kOffVals + lane] = s1;
        partial[(int64_t)item * kOffVals + 18 + lane] = s2;
    }
}

// Fixed-order sum of the per-item partials into the reduce buffer
//   [ S_diag Nc x 36 | S_off nb_off x 36 | rhs 6Nc | diag(U) 6Nc | g_c 6Nc | cost ].
// nb_off fixes the buffer layout; only the first nb_sum off-diagonal blocks are summed (0 when
// points are fixed).  Diagonal damping is NOT applied here (it is added after the cross-GPU sum).
__global__ void finalize_kernel(int Nc, int nb_off, int nb_sum,
                                const int *__restrict__ diag_item_ptr,
                                const real *__restrict__ diag_partial,
                                const int *__restrict__ off_item_ptr,
                                const real *__restrict__ off_partial, int n_cost_partials,
                                const real *__restrict__ cost_partial, real *__restrict__ red)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    real *S_diag = red;
    real *S_off = S_diag + (int64_t)Nc * 36;
    real *rhs = S_off + (int64_t)nb_off * 36;
    real *udiag = rhs + 6 * (int64_t)Nc;
    real *gc = udiag + 6 * (int64_t)Nc;
    real *cost = gc + 6 * (int64_t)Nc;
    const int64_t n_diag = (int64_t)Nc * kDiagVals;
    if (tid < n_diag) {
        const int c = (int)(tid / kDiagVals), v = (int)(tid % kDiagVals);
        real s = real(0);
        for (int it = diag_item_ptr[c]; it < diag_item_ptr[c + 1]; it++)
            s += diag_partial[(int64_t)it * kDiagVals + v];
        if (v < 21) {
            int p = 0, rem = v;
            while (rem >= 6 - p) { rem -= 6 - p; p++; }
            const int q = p + rem;
            S_diag[(int64_t)c * 36 + 6 * p + q] = s;
            S_diag[(int64_t)c * 36 + 6 * q + p] = s;
        } else if (v < 27) rhs[6 * c + v - 21] = s;
        else if (v < 33) udiag[6 * c + v - 27] = s;
        else gc[6 * c + v - 33] = s;
    } else if (tid < n_diag + (int64_t)nb_sum * kOffVals) {
        const int64_t u = tid - n_diag;
        const int blk = (int)(u / kOffVals), v = (int)(u % kOffVals);
        real s = real(0);
        for (int it = off_item_ptr[blk]; it < off_item_ptr[blk + 1]; it++)
            s += off_partial[(int64_t)it * kOffVals + v];
        S_off[(int64_t)blk * 36 + v] = s;
    } else if (tid == n_diag + (int64_t)nb_sum * kOffVals) {
        real s = real(0);
        for (int i = 0; i < n_cost_partials; i++) s += cost_partial[i];
        *cost = s;
    }
}

// Sum of `n` partials -> out[0] (single block).
__global__ void __launch_bounds__(256)
sum_partials_kernel(int n, const real *__restrict__ in, real *__restrict__ out)
{
    __shared__ real red[4];
    real s = real(0);
    for (int i = threadIdx.x; i < n; i += 256) s += in[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) *out = s;
}

// LM diagonal for the cameras: lam = clamp(s^2 Udd, min_d, max_d) / (radius s^2), or an explicit
// lam array; written to lam_out and added to the diagonal of S_cc in place.
__global__ void camera_damping_kernel(int Nc, const real *__restrict__ udiag,
                                      const real *__restrict__ scale_c,
                                      const real *__restrict__ lam_in, real radius, real min_d,
                                      real max_d, real *__restrict__ lam_out,
                                      real *__restrict__ S_diag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * Nc) return;
    real l = real(0);
    if (lam_in) l = lam_in[i];
    else if (scale_c) {
        const real s = scale_c[i], s2 = s * s;
        real x = s2 * udiag[i];
        x = x < min_d ? min_d : x;
        x = x > max_d ? max_d : x;
        l = x / radius / s2;
    }
    if (lam_out) lam_out[i] = l;
    const int c = i / 6, p = i % 6;
    S_diag[(int64_t)c * 36 + 7 * p] += l;
}

// Jacobi scaling s = 1 / (1 + sqrt(H_ii)).
__global__ void jacobi_scale_kernel(int n, const real *__restrict__ hdiag, real *__restrict__ scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scale[i] = real(1) / (real(1) + sqrt(hdiag[i]));
}

// Back-substitution and model cost change, one thread per landmark:
//   dp_j = -Vinv_j (g_p_j + sum_c W_cj^T dc_c), then  mcc -= m (r + m/2), m = Jc dc + Jp dp
// per residual row.  With fix_points dp = 0.
template <bool STAGE_CAMS>
__global__ void __launch_bounds__(kLmThreads)
backsub_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams,
               const real *__restrict__ pts, const int *__restrict__ pt_ptr,
               const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
               const real *__restrict__ hub_w, int fix_points, const real *__restrict__ Vinv6,
               const real *__restrict__ gp, const real *__restrict__ dcam,
               real *__restrict__ dpt, real *__restrict__ mcc_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CamData *scams = reinterpret_cast<CamData *>(smem_raw);
    __shared__ real red[4];
    if (STAGE_CAMS) {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = threadIdx.x; i < Nc * 8; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    real mcc = real(0);
    if (j < Np) {
        const int b = pt_ptr[j], e = pt_ptr[j + 1];
        const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
        real dp[3] = { 0, 0, 0 };
        if (!fix_points && e > b) {
            real acc[3] = { gp[3 * j], gp[3 * j + 1], gp[3 * j + 2] };
            for (int pos = b; pos < e; pos++) {
                const int c = lcam[pos];
                CamData cd;
                if (STAGE_CAMS) cd = scams[c]; else load_cam(cams, c, cd);
                ObsLin o;
                project_obs(cd, P, real(0), real(0), K, o);
                if (pos == hub.pos) scale_lin(o, *hub_w);
                real Jc[12], Jp[6];
                camera_jac(cd, o, P, Jc);
                point_jac(cd, o, Jp);
                real m0 = 0, m1 = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const real d = dcam[6 * c + k];
                    m0 += Jc[k] * d; m1 += Jc[6 + k] * d;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) acc[k] += Jp[k] * m0 + Jp[3 + k] * m1;
            }
            const real *vi = Vinv6 + 6 * (int64_t)j;
            dp[0] = -(vi[0] * acc[0] + vi[1] * acc[1] + vi[2] * acc[2]);
            dp[1] = -(vi[1] * acc[0] + vi[3] * acc[1] + vi[4] * acc[2]);
            dp[2] = -(vi[2] * acc[0] + vi[4] * acc[1] + vi[5] * acc[2]);
        }
        dpt[3 * j] = dp[0]; dpt[3 * j + 1] = dp[1]; dpt[3 * j + 2] = dp[2];
        for (int pos = b; pos < e; pos++) {
            const int c = lcam[pos];
            CamData cd;
            if (STAGE_CAMS) cd = scams[c]; else load_cam(cams, c, cd);
            ObsLin o;
            project_obs(cd, P, luv[2 * pos], luv[2 * pos + 1], K, o);
            if (pos == hub.pos) scale_lin(o, *hub_w);
            real Jc[12], Jp[6];
            camera_jac(cd, o, P, Jc);
            point_jac(cd, o, Jp);
            real m0 = 0, m1 = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const real d = dcam[6 * c + k];
                m0 += Jc[k] * d; m1 += Jc[6 + k] * d;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { m0 += Jp[k] * dp[k]; m1 += Jp[3 + k] * dp[k]; }
            mcc -= m0 * (o.r0 + real(0.5) * m0) + m1 * (o.r1 + real(0.5) * m1);
        }
    }
    const real s = block_sum_256(mcc, red);
    if (threadIdx.x == 0) mcc_partial[blockIdx.x] = s;
}

// Candidate parameters x + d and the squared norms Ceres' parameter-tolerance test needs.
// Only parameters that appear in a residual block are part of the problem (mask arrays).
// out_partial: per block {|x|^2, |d|^2}.
__global__ void __launch_bounds__(256)
candidate_kernel(int Nc, int Np, int fix_points, const real *__restrict__ t,
                 const real *__restrict__ rpy, const real *__restrict__ pts,
                 const unsigned char *__restrict__ cam_used,
                 const unsigned char *__restrict__ pt_used, int count_cams,
                 const real *__restrict__ dcam, const real *__restrict__ dpt,
                 real *__restrict__ t2, real *__restrict__ rpy2, real *__restrict__ pts2,
                 real *__restrict__ out_partial)
{
    __shared__ real red[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    real xn = 0, dn = 0;
    if (i < 6 * (int64_t)Nc) {
        const int c = (int)(i / 6), k = (int)(i % 6);
        const bool used = cam_used[c] != 0;
        const real x = k < 3 ? t[3 * c + k] : rpy[3 * c + k - 3];
        const real d = used ? dcam[i] : real(0);
        if (k < 3) t2[3 * c + k] = x + d; else rpy2[3 * c + k - 3] = x + d;
        if (used && count_cams) { xn = x * x; dn = d * d; }
    } else if (i < 6 * (int64_t)Nc + 3 * (int64_t)Np) {
        const int64_t u = i - 6 * (int64_t)Nc;
        const int j = (int)(u / 3);
        const bool used = !fix_points && pt_used[j] != 0;
        const real x = pts[u];
        const real d = used ? dpt[u] : real(0);
        pts2[u] = x + d;
        if (used) { xn = x * x; dn = d * d; }
    }
    const real s0 = block_sum_256(xn, red);
    const real s1 = block_sum_256(dn, red);
    if (threadIdx.x == 0) { out_partial[2 * blockIdx.x] = s0; out_partial[2 * blockIdx.x + 1] = s1; }
}

// max |g| over camera gradient (6Nc) and, unless fix_points, the landmark gradients (3Np).
__global__ void __launch_bounds__(256)
gradient_max_kernel(int64_t nc6, const real *__restrict__ gc, int64_t np3,
                    const real *__restrict__ gp, real *__restrict__ out)
{
    __shared__ real red[4];
    real m = 0;
    for (int64_t i = threadIdx.x; i < nc6; i += 256) m = fmax(m, fabs(gc[i]));
    for (int64_t i = threadIdx.x; i < np3; i += 256) m = fmax(m, fabs(gp[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

}  // namespace vs
