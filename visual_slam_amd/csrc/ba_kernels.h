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
constexpr int kRedStride = 67;    // row of the transposed wave reduction: 64 lanes + 2 zero cells, odd stride
constexpr int kRedSeg = 22;       // elements one lane sums in the reduction (3 segments per row)
constexpr int kDiagVals = 39;     // S_cc sym 21 | rhs 6 | diag(U) 6 | g_c 6
constexpr int kOffVals = 36;      // S_ab 6x6
#ifndef VS_BLK_WPE1
#define VS_BLK_WPE1 3                // waves per SIMD the 1-wave block kernel is compiled for
#endif
#ifndef VS_BLK_WPE4
#define VS_BLK_WPE4 2                // same, 4-wave block kernel
#endif
#ifndef VS_EPT1
#define VS_EPT1 2
#endif
#ifndef VS_EPT4
#define VS_EPT4 4
#endif
constexpr int kEpt4 = VS_EPT4;          // entries per thread of a 4-wave work item
constexpr int kEpt1 = VS_EPT1;          // entries per thread of a 1-wave work item

// Per-observation linearisation of the reference's reprojection residual
// (reference: src/core/reprojection_error.h:22-51; camera looks along +Y).
struct ObsLin {
    real r0, r1;             // residual
    real a00, a01, a11, a12; // d res / d Pc = [[a00 a01 0],[0 a11 a12]]  (= d res / d t)
    real q[3];               // R * Pw
};

// 1 / d from the hardware estimate and one cubically convergent step (4 instructions, within an
// ulp of the correctly rounded quotient) instead of the ~12-instruction IEEE division.
__device__ __forceinline__ real fast_rcp(real d)
{
    const real y0 = __builtin_amdgcn_rcp(d);
    const real e = fma(-d, y0, real(1));
    return fma(y0, fma(e, e, e), y0);
}

__device__ __forceinline__ void project_obs(const CamData &c, const real *P, real u, real v,
                                            const Intr &K, ObsLin &o)
{
    o.q[0] = c.R[0] * P[0] + c.R[1] * P[1] + c.R[2] * P[2];
    o.q[1] = c.R[3] * P[0] + c.R[4] * P[1] + c.R[5] * P[2];
    o.q[2] = c.R[6] * P[0] + c.R[7] * P[1] + c.R[8] * P[2];
    const real x = o.q[0] + c.t[0], y = o.q[1] + c.t[1], z = o.q[2] + c.t[2];
    const real iy = fast_rcp(y);
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
// Camera record of camera c from its parameters: R = Rz(yaw) Ry(pitch) Rx(roll)
// (reference: src/core/reprojection_error.h:32-35).
__device__ __forceinline__ CamData cam_from_sincos(const real *__restrict__ t, int c, real sr, real cr,
                                                   real sp, real cp, real sy, real cy)
{
    CamData d;
    d.R[0] = cy * cp; d.R[1] = cy * sp * sr - sy * cr; d.R[2] = cy * sp * cr + sy * sr;
    d.R[3] = sy * cp; d.R[4] = sy * sp * sr + cy * cr; d.R[5] = sy * sp * cr - cy * sr;
    d.R[6] = -sp;     d.R[7] = cp * sr;                d.R[8] = cp * cr;
    d.t[0] = t[3 * c]; d.t[1] = t[3 * c + 1]; d.t[2] = t[3 * c + 2];
    d.sy = sy; d.cy = cy; d.pad[0] = d.pad[1] = real(0);
    return d;
}

__device__ __forceinline__ CamData make_cam(const real *__restrict__ t, const real *__restrict__ rpy, int c)
{
    real sr, cr, sp, cp, sy, cy;
    sincos(rpy[3 * c], &sr, &cr);
    sincos(rpy[3 * c + 1], &sp, &cp);
    sincos(rpy[3 * c + 2], &sy, &cy);
    return cam_from_sincos(t, c, sr, cr, sp, cp, sy, cy);
}

__global__ void prep_cameras_kernel(int Nc, const real *__restrict__ t,
                                    const real *__restrict__ rpy, CamData *__restrict__ cams)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Nc) cams[c] = make_cam(t, rpy, c);
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
    const int *lam_map;     // lam_p is in the caller's landmark numbering: entry of landmark j at lam_map[j] (NULL: j)
    const real *radius_dev; // when set, the radius is read from device memory (a step enqueued before the host knows it)
};

struct HuberObs {
    int pos;        // landmark-major position of the Huber-weighted observation or -1
    int cam, pt;    // its camera and landmark
    real a;         // Huber parameter
};

// Per-landmark record written by the landmark kernel and read by the block and
// back-substitution kernels: P[3], inverse of the damped V (xx,xy,xz,yy,yz,zz), g_p[3].
constexpr int kRec = 12;
#ifndef VS_LPL
#define VS_LPL 2
#endif
constexpr int kLpl = VS_LPL;                          // lanes cooperating on one landmark (power of two)
constexpr int kLmPerBlock = kLmThreads / kLpl;        // landmarks per workgroup

// Stores the landmark records a wave holds (lane 0 of each group of kLpl lanes has one, for the
// consecutive landmarks jw, jw + 1, ...) as whole lines: the records of a wave are contiguous in
// `lmrec`, so they pass through a per-wave LDS staging area (kRecStage reals, half the wave's
// records at a time) and leave as 16-byte chunks in lane order instead of six scattered 16-byte
// stores per record.  Only LDS operations of one wave touch the area, and those execute in order.
constexpr int kRecsPerWave = kWave / kLpl;
constexpr int kRecStage = (kRecsPerWave / 2) * kRec;
__device__ __forceinline__ void store_records(real *stage, int lane, int jw, int Np, const double2 *rec,
                                              real *__restrict__ lmrec)
{
    static_assert(kRec == 12 && kRecsPerWave % 2 == 0, "six 16-byte chunks per record");
    constexpr int kHalf = kRecsPerWave / 2, kChunks = kHalf * 6;
    const int r = lane / kLpl;
    double2 *st = reinterpret_cast<double2 *>(stage);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        __builtin_amdgcn_wave_barrier();
        if ((lane & (kLpl - 1)) == 0 && r / kHalf == half) {
#pragma unroll
            for (int q = 0; q < 6; q++) st[(r - half * kHalf) * 6 + q] = rec[q];
        }
        __builtin_amdgcn_wave_barrier();
        double2 *dst = reinterpret_cast<double2 *>(lmrec + (int64_t)kRec * (jw + half * kHalf));
        const int n_valid = min(kHalf, Np - (jw + half * kHalf)) * 6;      // chunks of landmarks that exist
#pragma unroll
        for (int c0 = 0; c0 < kChunks; c0 += kWave) {
            const int c = c0 + lane;
            if (c < kChunks && c < n_valid) dst[c] = st[c];
        }
    }
}

// The work of the kLpl lanes that share landmark j (sl = lane within the group, [b, e) = the
// landmark's observations, P its position): V_j, g_p_j and the cost over its observations,
// Vinv_j = (V_j + damping)^-1; lane 0 of the group gets the landmark record in rec[0..6) (the
// caller stores it, see store_records) and stores diag(V_j) when asked to.
// (c_first, uv_first) = camera index and pixel of observation b + sl (unused when b + sl >= e),
// `fetch(c, cd)` delivers camera c's record.  Returns the landmark's cost in lane 0 of the group,
// zero elsewhere.
template <bool EARLY_DAMP, typename CamFetch>
__device__ __forceinline__ real landmark_lanes(int j, int sl, int Np, int b, int e, const real *P, const Intr &K,
                                               const int *__restrict__ lcam, const real *__restrict__ luv,
                                               const HuberObs &hub, const PointDamping &damp,
                                               double2 *rec, real *__restrict__ Vdiag,
                                               real *__restrict__ hub_w, int c_first, double2 uv_first, CamFetch fetch,
                                               int jd = -1)
{
    real acc[10];                                      // V (6), g (3), cost
#pragma unroll
    for (int v = 0; v < 10; v++) acc[v] = real(0);
    // With EARLY_DAMP the landmark's damping input (explicit values or Jacobi scales) is asked for
    // before the observation loop, so that its memory latency passes behind the loop instead of
    // after it (three more live registers: not for the kernel that sits at its register limit).
    real dmp[3] = { real(0), real(0), real(0) };
    auto load_damping = [&]() {
        const real *src = damp.lam_p ? damp.lam_p : damp.scale_p;
        // jd >= 0: the caller has already fetched this landmark's index in the damping array
        const int64_t jj = jd >= 0 ? jd : (damp.lam_p && damp.lam_map ? damp.lam_map[j] : j);
        if (src) { dmp[0] = src[3 * jj]; dmp[1] = src[3 * jj + 1]; dmp[2] = src[3 * jj + 2]; }
    };
    if (EARLY_DAMP && j < Np && sl == 0) load_damping();
    if (j < Np) {
        // the next observation's camera index and pixel are fetched while this one is processed
        // (the lane's first observation arrives from the caller, who may have fetched it early)
        int c_next = c_first;
        double2 uv_next = uv_first;
        for (int pos = b + sl; pos < e; pos += kLpl) {
            const int c = c_next;
            const double2 uv = uv_next;
            if (pos + kLpl < e) {
                c_next = lcam[pos + kLpl];
                uv_next = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)(pos + kLpl));
            }
            CamData cd;
            fetch(c, cd);
            ObsLin o;
            project_obs(cd, P, uv.x, uv.y, K, o);
            real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
            if (pos == hub.pos) {
                const real w = huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
                scale_lin(o, w);
                *hub_w = w;
            }
            acc[9] += hr;
            real Jp[6];
            point_jac(cd, o, Jp);
            acc[0] = fma(Jp[0], Jp[0], fma(Jp[3], Jp[3], acc[0]));
            acc[1] = fma(Jp[0], Jp[1], fma(Jp[3], Jp[4], acc[1]));
            acc[2] = fma(Jp[0], Jp[2], fma(Jp[3], Jp[5], acc[2]));
            acc[3] = fma(Jp[1], Jp[1], fma(Jp[4], Jp[4], acc[3]));
            acc[4] = fma(Jp[1], Jp[2], fma(Jp[4], Jp[5], acc[4]));
            acc[5] = fma(Jp[2], Jp[2], fma(Jp[5], Jp[5], acc[5]));
            acc[6] = fma(Jp[0], o.r0, fma(Jp[3], o.r1, acc[6]));
            acc[7] = fma(Jp[1], o.r0, fma(Jp[4], o.r1, acc[7]));
            acc[8] = fma(Jp[2], o.r0, fma(Jp[5], o.r1, acc[8]));
        }
    }
    // fixed-order sum over the cooperating lanes (every lane ends with the same totals)
#pragma unroll
    for (int v = 0; v < 10; v++)
#pragma unroll
        for (int off = 1; off < kLpl; off <<= 1) acc[v] += __shfl_xor(acc[v], off, 64);
    real cost = real(0);
    if (j < Np && sl == 0) {
        cost = acc[9];
        const real *V = acc;
        if (Vdiag) { Vdiag[3 * j] = V[0]; Vdiag[3 * j + 1] = V[3]; Vdiag[3 * j + 2] = V[5]; }
        real vi[6] = { 0, 0, 0, 0, 0, 0 };
        if (e > b) {
            if (!EARLY_DAMP) load_damping();
            real l0 = 0, l1 = 0, l2 = 0;
            if (damp.lam_p) {
                l0 = dmp[0]; l1 = dmp[1]; l2 = dmp[2];
            } else if (damp.scale_p) {
                const real d[3] = { V[0], V[3], V[5] };
                const real lm_radius = damp.radius_dev ? *damp.radius_dev : damp.radius;
                real l[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const real s = dmp[k], s2 = s * s;
                    real x = s2 * d[k];
                    x = x < damp.min_d ? damp.min_d : x;
                    x = x > damp.max_d ? damp.max_d : x;
                    l[k] = x * fast_rcp(lm_radius * s2);
                }
                l0 = l[0]; l1 = l[1]; l2 = l[2];
            }
            const real a = V[0] + l0, bb = V[1], cc = V[2], d = V[3] + l1, ee = V[4], f = V[5] + l2;
            const real c00 = d * f - ee * ee, c01 = cc * ee - bb * f, c02 = bb * ee - cc * d;
            const real id = fast_rcp(a * c00 + bb * c01 + cc * c02);
            vi[0] = c00 * id; vi[1] = c01 * id; vi[2] = c02 * id;
            vi[3] = (a * f - cc * cc) * id; vi[4] = (bb * cc - a * ee) * id;
            vi[5] = (a * d - bb * bb) * id;
        }
        rec[0] = make_double2(P[0], P[1]);
        rec[1] = make_double2(P[2], vi[0]);
        rec[2] = make_double2(vi[1], vi[2]);
        rec[3] = make_double2(vi[3], vi[4]);
        rec[4] = make_double2(vi[5], acc[6]);
        rec[5] = make_double2(acc[7], acc[8]);
    }
    return cost;
}

// Every camera record into the workgroup's LDS array `scams` (all threads call it; it ends with a
// barrier).  With prep_t / prep_rpy the records are built from the camera parameters (the array
// is followed by 6 Nc reals for the sines and cosines: one sincos per thread, 3 Nc of them, instead
// of three per camera in a single wave) and workgroup 0 also stores them to cams_out; otherwise
// they are copied from `cams`.
__device__ __forceinline__ void stage_all_cameras(CamData *scams, int Nc, const CamData *__restrict__ cams,
                                                  const real *__restrict__ prep_t, const real *__restrict__ prep_rpy,
                                                  CamData *__restrict__ cams_out)
{
    if (prep_t) {
        real *sc = reinterpret_cast<real *>(scams + Nc);
        for (int i = threadIdx.x; i < 3 * Nc; i += blockDim.x) sincos(prep_rpy[i], &sc[2 * i], &sc[2 * i + 1]);
        __syncthreads();
        for (int c = threadIdx.x; c < Nc; c += blockDim.x) {
            const real *q = sc + 6 * c;
            const CamData d = cam_from_sincos(prep_t, c, q[0], q[1], q[2], q[3], q[4], q[5]);
            scams[c] = d;
            if (blockIdx.x == 0 && cams_out) cams_out[c] = d;
        }
    } else {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = threadIdx.x; i < Nc * 8; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
}

// (a)+(b), landmark side.  kLpl lanes per landmark j share its observations:
// V_j = sum Jp^T Jp, g_p_j = sum Jp^T r, cost; then Vinv_j = (V_j + damping)^-1.
// Cameras come from LDS when they all fit (STAGE 1), or when the cameras seen by the landmarks of
// one workgroup span a short index range (STAGE 2, cam_lo[workgroup] = first camera of the range,
// cam_span = longest range: keyframes in time order).  With prep_t / prep_rpy given (STAGE 1 only)
// every workgroup builds the camera records itself, straight into LDS (which then also holds
// 6 Nc reals of sines and cosines behind the records), and workgroup 0 also stores them to
// `cams_out` for the kernels that follow: no separate camera pass.
template <int STAGE>   // 0: cameras gathered from global memory, 1: all cameras in LDS, 2: this workgroup's camera range in LDS
__global__ void __launch_bounds__(kLmThreads)
landmark_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams,
                const real *__restrict__ prep_t, const real *__restrict__ prep_rpy,
                CamData *__restrict__ cams_out,
                const real *__restrict__ pts, const int *__restrict__ pt_ptr,
                const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
                const int *__restrict__ cam_lo, int cam_span, PointDamping damp, real *__restrict__ lmrec, real *__restrict__ Vdiag,
                real *__restrict__ hub_w, real *__restrict__ cost_partial, int tile0)
{
    // a landmark shard launches only the tiles that hold its landmarks: workgroup b is tile tile0 + b
    const int tile = tile0 + (int)blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CamData *scams = reinterpret_cast<CamData *>(smem_raw);
    __shared__ real red[4];
    __shared__ __attribute__((aligned(16))) real rec_stage[(kLmThreads / kWave) * kRecStage];
    // the landmark's own data first: these loads are in flight while the cameras are staged
    const int j = tile * kLmPerBlock + (int)threadIdx.x / kLpl;
    const int sl = threadIdx.x & (kLpl - 1);
    real P[3] = { 0, 0, 0 };
    int b = 0, e = 0;
    if (j < Np) {
        b = pt_ptr[j]; e = pt_ptr[j + 1];
        P[0] = pts[3 * j]; P[1] = pts[3 * j + 1]; P[2] = pts[3 * j + 2];
    }
    if (STAGE == 1) stage_all_cameras(scams, Nc, cams, prep_t, prep_rpy, cams_out);
    int lo = 0;
    if (STAGE == 2) {
        lo = cam_lo[tile];
        const int cnt = min(cam_span, Nc - lo);
        const double2 *src = reinterpret_cast<const double2 *>(cams + lo);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = threadIdx.x; i < cnt * 8; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    int c_first = 0;
    double2 uv_first = make_double2(0, 0);
    if (b + sl < e) { c_first = lcam[b + sl]; uv_first = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)(b + sl)); }
    double2 rec[6] = {};
    const real cost = landmark_lanes<true>(j, sl, Np, b, e, P, K, lcam, luv, hub, damp, rec, Vdiag, hub_w, c_first, uv_first,
                                     [&](int c, CamData &cd) { if (STAGE == 0) load_cam(cams, c, cd); else cd = scams[c - lo]; });
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        store_records(rec_stage + wave * kRecStage, lane, tile * kLmPerBlock + wave * kRecsPerWave, Np, rec, lmrec);
    }
    const real s = block_sum_256(cost, red);
    if (threadIdx.x == 0) cost_partial[tile] = s;
}

// The landmark kernel for camera counts whose records fit LDS only as a whole-CU array (up to
// about 1300 cameras): one persistent workgroup of 1024 threads per CU stages every camera once
// (14 reals each; the 112-byte stride keeps 16-byte reads of consecutive cameras on distinct
// banks; built from the camera parameters by the workgroup itself) and then walks tiles of
// landmarks.  Without it every observation gathers its camera's
// 112 bytes through the vector cache, whose access rate then bounds the kernel.  Each group of
// 256 threads handles the landmarks landmark_kernel's workgroup of the same index would, and
// writes that workgroup's cost partial, so the partials and their sum are the same.
constexpr int kCamLdsReals = 14;
constexpr int kLmBigThreads = 1024;
template <bool MAPPED>   // explicit per-landmark damping in the caller's numbering (read through damp.lam_map)
__global__ void __launch_bounds__(kLmBigThreads)
landmark_lds_kernel(int Np, int Nc, int tile0, int n_tiles, Intr K, const real *__restrict__ prep_t,
                    const real *__restrict__ prep_rpy, CamData *__restrict__ cams_out,
                    const real *__restrict__ pts, const int *__restrict__ pt_ptr,
                    const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
                    PointDamping damp, real *__restrict__ lmrec, real *__restrict__ Vdiag,
                    real *__restrict__ hub_w, real *__restrict__ cost_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *scam = reinterpret_cast<double2 *>(smem_raw);
    __shared__ real red[kLmBigThreads / kWave];
    __shared__ __attribute__((aligned(16))) real rec_stage[(kLmBigThreads / kWave) * kRecStage];
    // every workgroup builds the camera records itself, straight into LDS (one camera per thread
    // and pass); workgroup 0 also stores them for the kernels that follow: no separate camera pass
    for (int c = threadIdx.x; c < Nc; c += kLmBigThreads) {
        const CamData d = make_cam(prep_t, prep_rpy, c);
        const double2 *src = reinterpret_cast<const double2 *>(&d);
#pragma unroll
        for (int q = 0; q < 7; q++) scam[7 * c + q] = src[q];
        if (blockIdx.x == 0) cams_out[c] = d;
    }
    __syncthreads();
    const int group = threadIdx.x / kLmThreads, tg = threadIdx.x % kLmThreads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = tg & (kLpl - 1);
    constexpr int kGroups = kLmBigThreads / kLmThreads;
    // Two memory latencies precede a landmark's arithmetic (its observation range and position,
    // then its first observation).  Both are taken off the critical path: the range and position
    // are fetched two tiles ahead, the first observation one tile ahead.
    struct Own { int b, e; real P[3]; };                 // the landmark index itself follows from the tile
    auto landmark_of = [&](int t0) {
        const int tile = t0 + group;
        return tile < n_tiles ? tile * kLmPerBlock + tg / kLpl : Np;
    };
    auto load_own = [&](int t0, Own &o) {
        const int j = landmark_of(t0);
        o.b = 0; o.e = 0; o.P[0] = o.P[1] = o.P[2] = real(0);
        if (j < Np) {
            o.b = pt_ptr[j]; o.e = pt_ptr[j + 1];
            o.P[0] = pts[3 * j]; o.P[1] = pts[3 * j + 1]; o.P[2] = pts[3 * j + 2];
        }
    };
    auto load_first = [&](const Own &o, int &c, double2 &uv) {
        c = 0; uv = make_double2(0, 0);
        if (o.b + sl < o.e) { c = lcam[o.b + sl]; uv = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)(o.b + sl)); }
    };
    // MAPPED: the landmark's index in the caller's damping array, fetched one tile ahead
    int jd_cur = -1, jd_nxt = -1;
    auto load_map = [&](int t0, int &jd) {
        const int j = landmark_of(t0);
        jd = -1;
        if (MAPPED && j < Np && sl == 0) jd = damp.lam_map[j];
    };
    const int stride = gridDim.x * kGroups;
    Own cur, nxt;
    int c_cur, c_nxt;
    double2 uv_cur, uv_nxt;
    // tiles [tile0, n_tiles): a landmark shard walks only the tiles that hold its landmarks
    const int t_first = tile0 + blockIdx.x * kGroups;
    load_own(t_first, cur);
    load_own(t_first + stride, nxt);
    load_first(cur, c_cur, uv_cur);
    load_map(t_first, jd_cur);
    for (int t0 = t_first; t0 < n_tiles; t0 += stride) {
        Own nn;
        load_own(t0 + 2 * stride, nn);
        load_first(nxt, c_nxt, uv_nxt);
        load_map(t0 + stride, jd_nxt);
        double2 rec[6] = {};
        real v = landmark_lanes<false>(landmark_of(t0), sl, Np, cur.b, cur.e, cur.P, K, lcam, luv, hub, damp, rec, Vdiag, hub_w, c_cur, uv_cur,
                                [&](int c, CamData &cd) {
                                    const double2 *src = scam + 7 * c;
                                    double2 *dst = reinterpret_cast<double2 *>(&cd);
#pragma unroll
                                    for (int q = 0; q < 7; q++) dst[q] = src[q];
                                }, jd_cur);
        if (t0 + group < n_tiles)
            store_records(rec_stage + wave * kRecStage, lane, (t0 + group) * kLmPerBlock + (tg / kWave) * kRecsPerWave, Np, rec, lmrec);
        // per-group sum in the order block_sum_256 takes it
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tg == 0 && t0 + group < n_tiles)
            cost_partial[t0 + group] = red[4 * group] + red[4 * group + 1] + red[4 * group + 2] + red[4 * group + 3];
        cur = nxt; nxt = nn; c_cur = c_nxt; uv_cur = uv_nxt; jd_cur = jd_nxt;
    }
}

// Cost only (candidate evaluation inside LM): one thread per landmark.  With STAGE_CAMS and prep_t /
// prep_rpy given, the workgroups build the camera records themselves (as landmark_kernel<1> does).
template <bool STAGE_CAMS>
__global__ void __launch_bounds__(kLmThreads)
cost_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams, const real *__restrict__ prep_t,
            const real *__restrict__ prep_rpy, CamData *__restrict__ cams_out, const real *__restrict__ pts,
            const int *__restrict__ pt_ptr, const int *__restrict__ lcam,
            const real *__restrict__ luv, HuberObs hub, real *__restrict__ cost_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CamData *scams = reinterpret_cast<CamData *>(smem_raw);
    __shared__ real red[4];
    if (STAGE_CAMS) stage_all_cameras(scams, Nc, cams, prep_t, prep_rpy, cams_out);
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
// sum over the 64 lanes of value v.  Three lanes share a value (22 row cells each, four
// independent chains per lane, cells 64 and 65 of a row hold zeros), then the three partial
// sums are added in segment order: the order is fixed, so results are reproducible, and the
// longest dependent chain is 9 additions instead of 64.  Every thread of the block must call it
// (it synchronises the block before reusing the LDS rows); clear_reduce_pads must have run.
template <int NV>
__device__ __forceinline__ real wave_reduce_transposed(const real *acc, real *lds_wave, int lane)
{
    static_assert(3 * NV <= kWave && 3 * kRedSeg <= kRedStride - 1, "three segments per row");
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; v++) lds_wave[v * kRedStride + lane] = acc[v];
    __syncthreads();
    real p = real(0);
    if (lane < 3 * NV) {
        const int seg = lane / NV, v = lane - seg * NV;
        const real *row = lds_wave + v * kRedStride + kRedSeg * seg;
        real c[4] = { row[0], row[1], row[2], row[3] };
#pragma unroll
        for (int l = 4; l < kRedSeg; l++) c[l & 3] += row[l];
        p = (c[0] + c[1]) + (c[2] + c[3]);
    }
    const real p1 = __shfl_down(p, NV, kWave), p2 = __shfl_down(p, 2 * NV, kWave);
    return (p + p1) + p2;
}

// Zeroes the two pad cells of every reduction row of this wave (once per kernel, before the
// first reduction; the reductions never write them).
__device__ __forceinline__ void clear_reduce_pads(real *lds_wave, int lane, int rows)
{
    if (lane < rows) {
        lds_wave[lane * kRedStride + kWave] = real(0);
        lds_wave[lane * kRedStride + kWave + 1] = real(0);
    }
}

// Views into the reduce buffer
//   [ S_diag Nc x 36 | S_off nb_off x 36 | rhs 6Nc | diag(U) 6Nc | g_c 6Nc | cost ].
struct RedView {
    real *S_diag, *S_off, *rhs, *udiag, *gc, *cost;
    const real *lam_c;     // optional camera damping (6 Nc) added to the diagonal of S_cc as it is stored
};
__host__ __device__ __forceinline__ RedView red_view(real *red, int Nc, int nb_off)
{
    RedView r;
    r.S_diag = red;
    r.S_off = r.S_diag + (int64_t)Nc * 36;
    r.rhs = r.S_off + (int64_t)nb_off * 36;
    r.udiag = r.rhs + 6 * (int64_t)Nc;
    r.gc = r.udiag + 6 * (int64_t)Nc;
    r.cost = r.gc + 6 * (int64_t)Nc;
    r.lam_c = nullptr;
    return r;
}

// value v of camera c's 39 sums: [0,21) upper triangle of S_cc, [21,27) rhs, [27,33) diag(U),
// [33,39) g_c
__device__ __forceinline__ void store_diag_value(const RedView &rv, int c, int v, real s)
{
    if (v < 21) {
        int p = 0, rem = v;
        while (rem >= 6 - p) { rem -= 6 - p; p++; }
        const int q = p + rem;
        if (p == q && rv.lam_c) s += rv.lam_c[6 * c + p];
        rv.S_diag[(int64_t)c * 36 + 6 * p + q] = s;
        rv.S_diag[(int64_t)c * 36 + 6 * q + p] = s;
    } else if (v < 27) rv.rhs[6 * c + v - 21] = s;
    else if (v < 33) rv.udiag[6 * c + v - 27] = s;
    else rv.gc[6 * c + v - 33] = s;
}

// Workgroup-wide fixed-order sum of NV per-thread values: per-wave transposed reduction, then
// the waves' sums are added in wave order.  Result valid in wave 0, lanes < NV.
template <int NV, int WAVES>
__device__ __forceinline__ real group_reduce(const real *acc, real *lds_wave, real *xw, int lane,
                                             int wave)
{
    real s = wave_reduce_transposed<NV>(acc, lds_wave, lane);
    if (WAVES > 1) {
        if (lane < NV) xw[wave * 24 + lane] = s;
        __syncthreads();
        if (wave == 0 && lane < NV) {
            s = xw[lane];
#pragma unroll
            for (int w = 1; w < WAVES; w++) s += xw[w * 24 + lane];
        }
    }
    return s;
}

// One work item of the block kernel (32 bytes, read through the scalar cache).
struct BlockItem {
    int blk;        // camera (diagonal item) or off-diagonal block id
    int begin, end; // entry range: camera-major observations, or camera-pair entries
    int ca, cb;     // the block's cameras (ca == cb for a diagonal item)
    int single;     // the block consists of this item only: write the final values directly
    int slot;       // row of diag_partial / off_partial this item writes when not single
    int pad1;
};
static_assert(sizeof(BlockItem) == 32, "BlockItem is 32 bytes");

__device__ __forceinline__ void load_rec(const real *__restrict__ lmrec, int j, real *r, int n2)
{
    const double2 *src = reinterpret_cast<const double2 *>(lmrec + (int64_t)kRec * j);
#pragma unroll
    for (int i = 0; i < 6; i++)
        if (i < n2) { const double2 v = src[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
}

// (a)+(b), block side: one workgroup of WAVES waves per work item, walked in chunks of EPT entries
// per thread.  Workgroup 0 also adds up the landmark kernel's cost partials.
//   diagonal items (ca == cb): a slice of one camera's observations (camera-major view).  Per
//     observation, with N = I - Jp Vinv Jp^T (2x2; N = I when points are fixed):
//       S_cc += Jc^T N Jc,  rhs += Jc^T (Jp Vinv g_p - r),  diag(U) += diag(Jc^T Jc),  g_c += Jc^T r
//   off-diagonal items: a slice of the landmarks seen by both cameras a < b of one block:  S_ab -= Jc_a^T (Jp_a Vinv Jp_b^T) Jc_b.
// r, Jc, Jp are recomputed from the parameters (J is never stored).  A thread's landmark indices
// are all loaded up front (one memory latency for the whole item instead of one per entry), the
// landmark record of the next entry is loaded while the current one is processed, and the entry
// loop is unrolled so the two record buffers alternate without copies.  A block made of a single
// item is written straight into the reduce buffer, otherwise a per-item partial is written and
// finalize_kernel adds the partials in item order.
template <int WAVES, int EPT>
__global__ void __launch_bounds__(WAVES *kWave) __attribute__((amdgpu_waves_per_eu(WAVES == 1 ? VS_BLK_WPE1 : VS_BLK_WPE4, WAVES == 1 ? VS_BLK_WPE1 : VS_BLK_WPE4)))
blocks_kernel(int n_items, const BlockItem *__restrict__ items, Intr K,
              const CamData *__restrict__ cams, const real *__restrict__ lmrec,
              const int *__restrict__ cpt, const real *__restrict__ cuv,
              const int *__restrict__ bent, HuberObs hub, const real *__restrict__ hub_w,
              int fix_points, real *__restrict__ diag_partial, real *__restrict__ off_partial,
              real *red, int Nc, int nb_off, const real *__restrict__ cost_partial, int n_cost_partials,
              const real *__restrict__ lam_c)
{
    __shared__ real lds[WAVES * 21 * kRedStride + WAVES * 24];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real *lds_wave = lds + wave * 21 * kRedStride;
    real *xw = lds + WAVES * 21 * kRedStride;
    clear_reduce_pads(lds_wave, lane, 21);
    // XCD-aware order: workgroups sharing an XCD (blockIdx.x % 8) take one contiguous eighth of
    // the item list, i.e. a band of cameras, so the landmark records that band's blocks share
    // stay in that XCD's L2.
    int item = blockIdx.x;
    {
        const int q = n_items >> 3, r = n_items & 7, x = item & 7, w = item >> 3;
        item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + w;
    }
    const BlockItem it = items[item];
    RedView rv = red_view(red, Nc, nb_off);
    rv.lam_c = lam_c;
    if (blockIdx.x == 0) {
        // total cost = fixed-order sum of the landmark kernel's per-workgroup partials
        real c = real(0);
        for (int i = threadIdx.x; i < n_cost_partials; i += WAVES * kWave) c += cost_partial[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
        if (WAVES > 1) {
            if (lane == 0) xw[wave] = c;
            __syncthreads();
            if (threadIdx.x == 0) {
                c = xw[0];
#pragma unroll
                for (int w = 1; w < WAVES; w++) c += xw[w];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) *rv.cost = c;
    }
    constexpr int kStride = WAVES * kWave;
    // a 4-wave item is walked in chunks of EPT entries per thread, the sums stay in registers across
    // chunks; a 1-wave item is one chunk (the loop costs that form registers it does not have)
    constexpr int kChunk = kStride * EPT;
    constexpr bool kLoop = WAVES > 1;
    if (it.ca == it.cb) {
        real acc[kDiagVals];
#pragma unroll
        for (int v = 0; v < kDiagVals; v++) acc[v] = real(0);
        CamData cd;
        load_cam(cams, it.ca, cd);
        int cb0 = it.begin;
        do {
        const int k0 = cb0 + (int)threadIdx.x;
        int jx[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) jx[e] = k0 + e * kStride < it.end ? cpt[k0 + e * kStride] : -1;
        real rec[2][kRec], uv[2][2];
        auto fetch = [&](int e, int buf) {
            if (jx[e] >= 0) {
                const double2 m = *reinterpret_cast<const double2 *>(cuv + 2 * (int64_t)(k0 + e * kStride));
                uv[buf][0] = m.x; uv[buf][1] = m.y;
                load_rec(lmrec, jx[e], rec[buf], 6);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            if (e + 1 < EPT) fetch(e + 1, (e + 1) & 1);
            if (jx[e] < 0) continue;
            const real *P = rec[e & 1], *vi = P + 3, *g = P + 9;
            ObsLin o;
            project_obs(cd, P, uv[e & 1][0], uv[e & 1][1], K, o);
            if (jx[e] == hub.pt && it.ca == hub.cam) scale_lin(o, *hub_w);
            real Jc[12];
            camera_jac(cd, o, P, Jc);
            real n00 = real(1), n01 = real(0), n11 = real(1), e0 = -o.r0, e1 = -o.r1;
            if (!fix_points) {
                real Jp[6], T[6];
                point_jac(cd, o, Jp);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    T[3 * r] = Jp[3 * r] * vi[0] + Jp[3 * r + 1] * vi[1] + Jp[3 * r + 2] * vi[2];
                    T[3 * r + 1] = Jp[3 * r] * vi[1] + Jp[3 * r + 1] * vi[3] + Jp[3 * r + 2] * vi[4];
                    T[3 * r + 2] = Jp[3 * r] * vi[2] + Jp[3 * r + 1] * vi[4] + Jp[3 * r + 2] * vi[5];
                }
                n00 -= T[0] * Jp[0] + T[1] * Jp[1] + T[2] * Jp[2];
                n01 -= T[0] * Jp[3] + T[1] * Jp[4] + T[2] * Jp[5];
                n11 -= T[3] * Jp[3] + T[4] * Jp[4] + T[5] * Jp[5];
                e0 += T[0] * g[0] + T[1] * g[1] + T[2] * g[2];
                e1 += T[3] * g[0] + T[4] * g[1] + T[5] * g[2];
            }
            // Jc row 0 is structurally zero at column 2, row 1 at column 0 (see camera_jac)
            real B0[6], B1[6];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                B0[q] = q == 2 ? n01 * Jc[8] : (q == 0 ? n00 * Jc[0] : fma(n00, Jc[q], n01 * Jc[6 + q]));
                B1[q] = q == 2 ? n11 * Jc[8] : (q == 0 ? n01 * Jc[0] : fma(n01, Jc[q], n11 * Jc[6 + q]));
            }
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 6; p++) {
#pragma unroll
                for (int q = p; q < 6; q++) {
                    real t = acc[idx];
                    if (p != 0) t = fma(Jc[6 + p], B1[q], t);
                    if (p != 2) t = fma(Jc[p], B0[q], t);
                    acc[idx++] = t;
                }
                real t1 = acc[21 + p], t2 = acc[27 + p], t3 = acc[33 + p];
                if (p != 0) { t1 = fma(Jc[6 + p], e1, t1); t2 = fma(Jc[6 + p], Jc[6 + p], t2); t3 = fma(Jc[6 + p], o.r1, t3); }
                if (p != 2) { t1 = fma(Jc[p], e0, t1); t2 = fma(Jc[p], Jc[p], t2); t3 = fma(Jc[p], o.r0, t3); }
                acc[21 + p] = t1; acc[27 + p] = t2; acc[33 + p] = t3;
            }
        }
        cb0 += kChunk;
        } while (kLoop && cb0 < it.end);   // chunks
        const real s1 = group_reduce<21, WAVES>(acc, lds_wave, xw, lane, wave);
        const real s2 = group_reduce<18, WAVES>(acc + 21, lds_wave, xw, lane, wave);
        if (wave == 0) {
            if (it.single) {
                if (lane < 21) store_diag_value(rv, it.blk, lane, s1);
                if (lane < 18) store_diag_value(rv, it.blk, 21 + lane, s2);
            } else {
                if (lane < 21) diag_partial[(int64_t)it.slot * kDiagVals + lane] = s1;
                if (lane < 18) diag_partial[(int64_t)it.slot * kDiagVals + 21 + lane] = s2;
            }
        }
    } else {
        real acc[kOffVals];
#pragma unroll
        for (int v = 0; v < kOffVals; v++) acc[v] = real(0);
        CamData A, B;
        load_cam(cams, it.ca, A);
        load_cam(cams, it.cb, B);
        int cb0 = it.begin;
        do {
        const int k0 = cb0 + (int)threadIdx.x;
        int jx[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) jx[e] = k0 + e * kStride < it.end ? bent[k0 + e * kStride] : -1;
        real rec[2][10];
        if (jx[0] >= 0) load_rec(lmrec, jx[0], rec[0], 5);
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            if (e + 1 < EPT) { if (jx[e + 1] >= 0) load_rec(lmrec, jx[e + 1], rec[(e + 1) & 1], 5); }
            if (jx[e] < 0) continue;
            const real *P = rec[e & 1], *vi = P + 3;
            ObsLin oa, ob;
            // the Jacobians do not depend on (u,v): pass zeros; only the Huber edge rescales
            project_obs(A, P, real(0), real(0), K, oa);
            project_obs(B, P, real(0), real(0), K, ob);
            if (jx[e] == hub.pt) {
                if (it.ca == hub.cam) scale_lin(oa, *hub_w);
                if (it.cb == hub.cam) scale_lin(ob, *hub_w);
            }
            real Jca[12], Jcb[12], Jpa[6], Jpb[6], T[6];
            camera_jac(A, oa, P, Jca);
            point_jac(A, oa, Jpa);
            camera_jac(B, ob, P, Jcb);
            point_jac(B, ob, Jpb);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                T[3 * r] = Jpa[3 * r] * vi[0] + Jpa[3 * r + 1] * vi[1] + Jpa[3 * r + 2] * vi[2];
                T[3 * r + 1] = Jpa[3 * r] * vi[1] + Jpa[3 * r + 1] * vi[3] + Jpa[3 * r + 2] * vi[4];
                T[3 * r + 2] = Jpa[3 * r] * vi[2] + Jpa[3 * r + 1] * vi[4] + Jpa[3 * r + 2] * vi[5];
            }
            // M = -(Jp_a Vinv Jp_b^T), so the block update is a pure accumulation
            const real m00 = -(T[0] * Jpb[0] + T[1] * Jpb[1] + T[2] * Jpb[2]);
            const real m01 = -(T[0] * Jpb[3] + T[1] * Jpb[4] + T[2] * Jpb[5]);
            const real m10 = -(T[3] * Jpb[0] + T[4] * Jpb[1] + T[5] * Jpb[2]);
            const real m11 = -(T[3] * Jpb[3] + T[4] * Jpb[4] + T[5] * Jpb[5]);
            // Jc row 0 is structurally zero at column 2, row 1 at column 0 (see camera_jac)
            real B0[6], B1[6];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                B0[q] = q == 2 ? m01 * Jcb[8] : (q == 0 ? m00 * Jcb[0] : fma(m00, Jcb[q], m01 * Jcb[6 + q]));
                B1[q] = q == 2 ? m11 * Jcb[8] : (q == 0 ? m10 * Jcb[0] : fma(m10, Jcb[q], m11 * Jcb[6 + q]));
            }
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    real t = acc[6 * p + q];
                    if (p != 0) t = fma(Jca[6 + p], B1[q], t);
                    if (p != 2) t = fma(Jca[p], B0[q], t);
                    acc[6 * p + q] = t;
                }
        }
        cb0 += kChunk;
        } while (kLoop && cb0 < it.end);   // chunks
        const real s1 = group_reduce<18, WAVES>(acc, lds_wave, xw, lane, wave);
        const real s2 = group_reduce<18, WAVES>(acc + 18, lds_wave, xw, lane, wave);
        if (wave == 0 && lane < 18) {
            real *dst = it.single ? rv.S_off + (int64_t)it.blk * 36
                                  : off_partial + (int64_t)it.slot * kOffVals;
            dst[lane] = s1;
            dst[18 + lane] = s2;
        }
    }
}

// Off-diagonal blocks of at most L (16 or 32) entries, 64 / L to a wave: lane = (block lane / L,
// entry lane % L).  The cameras differ between the blocks of a wave, so they live in vector
// registers here.  The 36 sums of a block are reduced over its L lanes through LDS in a fixed order
// and written straight into the reduce buffer (such a block is always a single item).
template <int L>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 4)))
blocks_small_kernel(int n_small, const BlockItem *__restrict__ items, Intr K,
                    const CamData *__restrict__ cams, const real *__restrict__ lmrec,
                    const int *__restrict__ bent, HuberObs hub, const real *__restrict__ hub_w,
                    real *red, int Nc, int nb_off)
{
    __shared__ real lds[18 * kRedStride];
    const int lane = threadIdx.x, q = lane / L, e = lane % L;
    const int idx = blockIdx.x * (kWave / L) + q;
    const BlockItem it = items[min(idx, n_small - 1)];
    const bool have = idx < n_small && it.begin + e < it.end;
    CamData A, B;
    load_cam(cams, it.ca, A);
    load_cam(cams, it.cb, B);
    const int j = bent[have ? it.begin + e : it.begin];
    real rec[10];
    load_rec(lmrec, j, rec, 5);
    const real *P = rec, *vi = rec + 3;
    ObsLin oa, ob;
    project_obs(A, P, real(0), real(0), K, oa);
    project_obs(B, P, real(0), real(0), K, ob);
    if (j == hub.pt) {
        if (it.ca == hub.cam) scale_lin(oa, *hub_w);
        if (it.cb == hub.cam) scale_lin(ob, *hub_w);
    }
    real Jca[12], Jcb[12], Jpa[6], Jpb[6], T[6];
    camera_jac(A, oa, P, Jca);
    point_jac(A, oa, Jpa);
    camera_jac(B, ob, P, Jcb);
    point_jac(B, ob, Jpb);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        T[3 * r] = Jpa[3 * r] * vi[0] + Jpa[3 * r + 1] * vi[1] + Jpa[3 * r + 2] * vi[2];
        T[3 * r + 1] = Jpa[3 * r] * vi[1] + Jpa[3 * r + 1] * vi[3] + Jpa[3 * r + 2] * vi[4];
        T[3 * r + 2] = Jpa[3 * r] * vi[2] + Jpa[3 * r + 1] * vi[4] + Jpa[3 * r + 2] * vi[5];
    }
    // M = -(Jp_a Vinv Jp_b^T); a lane without an entry contributes zeros
    const real live = have ? real(-1) : real(0);
    const real m00 = live * (T[0] * Jpb[0] + T[1] * Jpb[1] + T[2] * Jpb[2]);
    const real m01 = live * (T[0] * Jpb[3] + T[1] * Jpb[4] + T[2] * Jpb[5]);
    const real m10 = live * (T[3] * Jpb[0] + T[4] * Jpb[1] + T[5] * Jpb[2]);
    const real m11 = live * (T[3] * Jpb[3] + T[4] * Jpb[4] + T[5] * Jpb[5]);
    real B0[6], B1[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        B0[c] = c == 2 ? m01 * Jcb[8] : (c == 0 ? m00 * Jcb[0] : fma(m00, Jcb[c], m01 * Jcb[6 + c]));
        B1[c] = c == 2 ? m11 * Jcb[8] : (c == 0 ? m10 * Jcb[0] : fma(m10, Jcb[c], m11 * Jcb[6 + c]));
    }
    real acc[kOffVals];
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            real t = real(0);
            if (p != 0) t = fma(Jca[6 + p], B1[c], t);
            if (p != 2) t = fma(Jca[p], B0[c], t);
            acc[6 * p + c] = t;
        }
    const RedView rv = red_view(red, Nc, nb_off);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 18; v++) lds[v * kRedStride + lane] = acc[18 * pass + v];
        __syncthreads();
        for (int v = e; v < 18; v += L) {
            const real *cell = lds + v * kRedStride + L * q;
            real c4[4] = { cell[0], cell[1], cell[2], cell[3] };
#pragma unroll
            for (int l = 4; l < L; l++) c4[l & 3] += cell[l];
            if (idx < n_small) rv.S_off[(int64_t)it.blk * 36 + 18 * pass + v] = (c4[0] + c4[1]) + (c4[2] + c4[3]);
        }
    }
}

// Sum of p[it * stride] for it in [i0, i1), added in item order; the loads go out four at a time
// (a dependent load per addition would cost one memory latency per item).
__device__ __forceinline__ real sum_partials_strided(const real *__restrict__ p, int stride, int i0, int i1)
{
    real s = real(0);
    int it = i0;
    for (; it + 4 <= i1; it += 4) {
        const real a = p[(int64_t)it * stride], b = p[(int64_t)(it + 1) * stride],
                   c = p[(int64_t)(it + 2) * stride], d = p[(int64_t)(it + 3) * stride];
        s += a; s += b; s += c; s += d;
    }
    const int left = i1 - it;
    if (left > 0) {
        const real a = p[(int64_t)it * stride];
        const real b = left > 1 ? p[(int64_t)(it + 1) * stride] : real(0);
        const real c = left > 2 ? p[(int64_t)(it + 2) * stride] : real(0);
        s += a;
        if (left > 1) s += b;
        if (left > 2) s += c;
    }
    return s;
}

// Fixed-order sum of the per-item partials of multi-item blocks into the reduce buffer, and of
// the landmark kernel's cost partials.  nb_off fixes the buffer layout; only the first nb_sum
// off-diagonal blocks are touched (0 when points are fixed).  Diagonal damping is NOT applied
// here (it is added after the cross-GPU sum).
__global__ void __launch_bounds__(256)
finalize_kernel(int Nc, int nb_off, int nb_sum, const int *__restrict__ diag_item_ptr,
                const real *__restrict__ diag_partial, const int *__restrict__ off_item_ptr,
                const real *__restrict__ off_partial, int n_cost_partials,
                const real *__restrict__ cost_partial, real *__restrict__ red, const real *__restrict__ lam_c)
{
    __shared__ real scratch[4];
    RedView rv = red_view(red, Nc, nb_off);
    rv.lam_c = lam_c;
    if (blockIdx.x == gridDim.x - 1) {                 // last workgroup: cost (when asked to)
        if (n_cost_partials < 0) return;
        real s = real(0);
        for (int i = threadIdx.x; i < n_cost_partials; i += 256) s += cost_partial[i];
        s = block_sum_256(s, scratch);
        if (threadIdx.x == 0) *rv.cost = s;
        return;
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_diag = (int64_t)Nc * kDiagVals;
    if (tid < n_diag) {
        const int c = (int)(tid / kDiagVals), v = (int)(tid % kDiagVals);
        const int i0 = diag_item_ptr[c], i1 = diag_item_ptr[c + 1];
        if (i1 - i0 == 1) return;                      // written by blocks_kernel
        store_diag_value(rv, c, v, sum_partials_strided(diag_partial + v, kDiagVals, i0, i1));
    } else if (tid < n_diag + (int64_t)nb_sum * kOffVals) {
        const int64_t u = tid - n_diag;
        const int blk = (int)(u / kOffVals), v = (int)(u % kOffVals);
        const int i0 = off_item_ptr[blk], i1 = off_item_ptr[blk + 1];
        if (i1 - i0 == 1) return;
        rv.S_off[(int64_t)blk * 36 + v] = sum_partials_strided(off_partial + v, kOffVals, i0, i1);
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

// The scalars of one LM step in one launch (single block, fixed order): out[0] = sum of the model
// cost change partials, out[1], out[2] = sums of the interleaved (|x|^2, |d|^2) partials, out[3] =
// sum of the candidate cost partials (when n_cost > 0), out[6] = the solver's status word.
__global__ void __launch_bounds__(256)
step_scalars_kernel(int n_mcc, const real *__restrict__ mcc_partial, int n_pairs,
                    const real *__restrict__ pair_partial, int n_cost,
                    const real *__restrict__ cost_partial, const int *__restrict__ info,
                    real *__restrict__ out)
{
    __shared__ real red[4];
    // four loads in flight per thread and sum (a dependent load per addition would cost one memory
    // latency per partial); the order is fixed
    auto strided_sum = [&](const real *__restrict__ p, int n, int stride, int offset) {
        real s0 = real(0), s1 = real(0), s2 = real(0), s3 = real(0);
        int i = threadIdx.x;
        for (; i + 768 < n; i += 1024) {
            const real v0 = p[(int64_t)i * stride + offset], v1 = p[(int64_t)(i + 256) * stride + offset],
                       v2 = p[(int64_t)(i + 512) * stride + offset], v3 = p[(int64_t)(i + 768) * stride + offset];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; i < n; i += 256) s0 += p[(int64_t)i * stride + offset];
        return (s0 + s1) + (s2 + s3);
    };
    real m = strided_sum(mcc_partial, n_mcc, 1, 0);
    real a = strided_sum(pair_partial, n_pairs, 2, 0);
    real b = strided_sum(pair_partial, n_pairs, 2, 1);
    real c = strided_sum(cost_partial, n_cost, 1, 0);
    m = block_sum_256(m, red);
    a = block_sum_256(a, red);
    b = block_sum_256(b, red);
    c = block_sum_256(c, red);
    if (threadIdx.x == 0) {
        out[0] = m; out[1] = a; out[2] = b;
        if (n_cost > 0) out[3] = c;
        out[6] = (real)*info;
    }
}

// LM diagonal for the cameras: lam = clamp(s^2 Udd, min_d, max_d) / (radius s^2), or an explicit
// lam array; written to lam_out and added to the diagonal of S_cc in place.
__global__ void camera_damping_kernel(int Nc, const real *__restrict__ udiag,
                                      const real *__restrict__ scale_c,
                                      const real *__restrict__ lam_in, real radius, real min_d,
                                      real max_d, real *__restrict__ lam_out,
                                      real *__restrict__ S_diag, const real *__restrict__ rhs,
                                      real *__restrict__ rhs_copy, int *__restrict__ info)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * Nc) return;
    // set-up of the solve that follows, folded in: right-hand side into the solution vector, status cleared
    if (rhs_copy) rhs_copy[i] = rhs[i];
    if (info && i == 0) *info = 0;
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

// Back-substitution for landmark j (observations [b, e), position P):
//   dp_j = -(V_j + D)^-1 (g_p_j + sum_o Jp_o^T u_o),   u_o = Jc_o dc_cam(o),
// written to dpt, and the landmark's share of the model cost change
//   -sum_o m_o . (r_o + m_o / 2),   m_o = u_o + Jp_o dp_j,
// which is returned.  One pass over the observations: with q = sum_o Jp_o^T u_o and V = sum_o
// Jp_o^T Jp_o,  sum_o m.r = sum u.r + dp.g_p  and  sum_o m.m = sum u.u + 2 dp.q + dp^T V dp, so the
// Jacobians are built once and the camera record and camera step are fetched once per observation.
// `fetch(c, cd)` delivers camera c's record.
template <typename CamFetch>
__device__ __forceinline__ real backsub_landmark(int j, int b, int e, const real *P, const Intr &K,
                                                 const int *__restrict__ lcam, const real *__restrict__ luv,
                                                 const HuberObs &hub, const real *__restrict__ hub_w, int fix_points,
                                                 const real *__restrict__ lmrec, const real *__restrict__ dcam,
                                                 real *__restrict__ dpt, CamFetch fetch, real *dp_out = nullptr)
{
    real dp[3] = { 0, 0, 0 };
    real mcc = real(0);
    if (e > b) {
        real q[3] = { 0, 0, 0 }, V[6] = { 0, 0, 0, 0, 0, 0 }, s_ur = real(0), s_uu = real(0);
        int c_next = lcam[b];
        double2 uv_next = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)b);
        for (int pos = b; pos < e; pos++) {
            const int c = c_next;
            const double2 uv = uv_next;
            if (pos + 1 < e) {
                c_next = lcam[pos + 1];
                uv_next = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)(pos + 1));
            }
            const double2 *dsrc = reinterpret_cast<const double2 *>(dcam + 6 * (int64_t)c);
            const double2 d01 = dsrc[0], d23 = dsrc[1], d45 = dsrc[2];
            CamData cd;
            fetch(c, cd);
            ObsLin o;
            project_obs(cd, P, uv.x, uv.y, K, o);
            if (pos == hub.pos) scale_lin(o, *hub_w);
            real Jc[12], Jp[6];
            camera_jac(cd, o, P, Jc);
            point_jac(cd, o, Jp);
            const real d[6] = { d01.x, d01.y, d23.x, d23.y, d45.x, d45.y };
            real u0 = 0, u1 = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) { u0 += Jc[k] * d[k]; u1 += Jc[6 + k] * d[k]; }
            s_ur += u0 * o.r0 + u1 * o.r1;
            s_uu += u0 * u0 + u1 * u1;
#pragma unroll
            for (int k = 0; k < 3; k++) q[k] += Jp[k] * u0 + Jp[3 + k] * u1;
            V[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3]; V[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
            V[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5]; V[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
            V[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5]; V[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
        }
        // (the record is fetched here, not before the loop, so that it holds no registers across it)
        real rec[kRec];
        load_rec(lmrec, j, rec, 6);
        const real *vi = rec + 3, *gp = rec + 9;
        if (!fix_points) {
            const real a0 = gp[0] + q[0], a1 = gp[1] + q[1], a2 = gp[2] + q[2];
            dp[0] = -(vi[0] * a0 + vi[1] * a1 + vi[2] * a2);
            dp[1] = -(vi[1] * a0 + vi[3] * a1 + vi[4] * a2);
            dp[2] = -(vi[2] * a0 + vi[4] * a1 + vi[5] * a2);
        }
        const real Vd0 = V[0] * dp[0] + V[1] * dp[1] + V[2] * dp[2];
        const real Vd1 = V[1] * dp[0] + V[3] * dp[1] + V[4] * dp[2];
        const real Vd2 = V[2] * dp[0] + V[4] * dp[1] + V[5] * dp[2];
        const real dp_g = dp[0] * gp[0] + dp[1] * gp[1] + dp[2] * gp[2];
        const real dp_q = dp[0] * q[0] + dp[1] * q[1] + dp[2] * q[2];
        const real dp_V_dp = dp[0] * Vd0 + dp[1] * Vd1 + dp[2] * Vd2;
        mcc = -((s_ur + dp_g) + real(0.5) * (s_uu + real(2) * dp_q + dp_V_dp));
    }
    dpt[3 * j] = dp[0]; dpt[3 * j + 1] = dp[1]; dpt[3 * j + 2] = dp[2];
    if (dp_out) { dp_out[0] = dp[0]; dp_out[1] = dp[1]; dp_out[2] = dp[2]; }
    return mcc;
}

// Back-substitution, one thread per landmark; mcc_partial gets one sum per 256 landmarks.
template <bool STAGE_CAMS>
__global__ void __launch_bounds__(kLmThreads)
backsub_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams,
               const real *__restrict__ pts, const int *__restrict__ pt_ptr,
               const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
               const real *__restrict__ hub_w, int fix_points, const real *__restrict__ lmrec,
               const real *__restrict__ dcam, real *__restrict__ dpt,
               real *__restrict__ mcc_partial)
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
        mcc = backsub_landmark(j, b, e, P, K, lcam, luv, hub, hub_w, fix_points, lmrec, dcam, dpt,
                               [&](int c, CamData &cd) { if (STAGE_CAMS) cd = scams[c]; else load_cam(cams, c, cd); });
    }
    const real s = block_sum_256(mcc, red);
    if (threadIdx.x == 0) mcc_partial[blockIdx.x] = s;
}

// The same with every camera record in LDS (see landmark_lds_kernel): one persistent workgroup
// of 1024 threads per CU; each group of 256 threads handles the landmarks backsub_kernel's
// workgroup of the same index would and writes that workgroup's partial.
__global__ void __launch_bounds__(kLmBigThreads)
backsub_lds_kernel(int Np, int Nc, int n_tiles, Intr K, const CamData *__restrict__ cams,
                   const real *__restrict__ pts, const int *__restrict__ pt_ptr,
                   const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub,
                   const real *__restrict__ hub_w, int fix_points, const real *__restrict__ lmrec,
                   const real *__restrict__ dcam, real *__restrict__ dpt,
                   real *__restrict__ mcc_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *scam = reinterpret_cast<double2 *>(smem_raw);
    __shared__ real red[kLmBigThreads / kWave];
    {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        for (int i = threadIdx.x; i < Nc * 7; i += kLmBigThreads) {
            const int c = i / 7, q = i - 7 * c;
            scam[i] = src[8 * c + q];
        }
    }
    __syncthreads();
    const int group = threadIdx.x / kLmThreads, tg = threadIdx.x % kLmThreads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kGroups = kLmBigThreads / kLmThreads;
    for (int t0 = blockIdx.x * kGroups; t0 < n_tiles; t0 += gridDim.x * kGroups) {
        const int tile = t0 + group;
        const int j = tile * kLmThreads + tg;
        real v = real(0);
        if (tile < n_tiles && j < Np) {
            const int b = pt_ptr[j], e = pt_ptr[j + 1];
            const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
            v = backsub_landmark(j, b, e, P, K, lcam, luv, hub, hub_w, fix_points, lmrec, dcam, dpt,
                                 [&](int c, CamData &cd) {
                                     const double2 *src = scam + 7 * c;
                                     double2 *dst = reinterpret_cast<double2 *>(&cd);
#pragma unroll
                                     for (int q = 0; q < 7; q++) dst[q] = src[q];
                                 });
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tg == 0 && tile < n_tiles)
            mcc_partial[tile] = red[4 * group] + red[4 * group + 1] + red[4 * group + 2] + red[4 * group + 3];
    }
}

// out = max of in[0..n) (single wave).
__global__ void max_of_kernel(int n, const real *__restrict__ in, real *__restrict__ out)
{
    real m = in[0];
    for (int i = threadIdx.x; i < n; i += 64) m = fmax(m, in[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (threadIdx.x == 0) *out = m;
}

// Landmark renumbering at the boundary (the library numbers landmarks by the first camera that
// sees them; map[j] = the caller's index of internal landmark j).  Rows of `width` reals:
//   gather:  dst[j] (stride width)            = src[map[j]] (stride width)
//   scatter: dst[map[j]] (stride width)       = src[j] (stride src_stride, starting at src_off)
__global__ void gather_rows_kernel(int Np, int width, const int *__restrict__ map, const real *__restrict__ src,
                                   real *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)width * Np) return;
    const int j = (int)(i / width), k = (int)(i - (int64_t)j * width);
    dst[i] = src[(int64_t)width * map[j] + k];
}
__global__ void scatter_rows_kernel(int Np, int width, const int *__restrict__ map, const real *__restrict__ src,
                                    int src_stride, int src_off, real *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)width * Np) return;
    const int j = (int)(i / width), k = (int)(i - (int64_t)j * width);
    dst[(int64_t)width * map[j] + k] = src[(int64_t)src_stride * j + src_off + k];
}

// Multi-GPU exchange of the reduce buffer.  The landmarks of shard r see the cameras
// [cam_lo, cam_hi] only (landmarks are numbered by first camera), so its partial buffer is zero
// outside the diagonal blocks, right-hand side, diag(U) and g_c of those cameras, the off-diagonal
// blocks of those rows [blk_lo, blk_hi) and the cost: that is what a rank sends.
struct ShardSeg {
    int cam_lo, cam_hi;      // inclusive; cam_hi < cam_lo: the shard is empty
    int blk_lo, blk_hi;      // off-diagonal blocks of the rows [cam_lo, cam_hi]
};
__host__ __device__ __forceinline__ int64_t seg_len(const ShardSeg &g)
{
    const int64_t nc = g.cam_hi >= g.cam_lo ? g.cam_hi - g.cam_lo + 1 : 0, nb = g.blk_hi - g.blk_lo;
    return 54 * nc + 36 * nb + 1;      // 36 + 18 per camera, 36 per block, the cost
}

// This rank's segment -> out[0, seg_len): [S_diag rows | S_off blocks | rhs | diag(U) | g_c | cost]
__global__ void pack_segment_kernel(ShardSeg g, int Nc, int nb_off, const real *__restrict__ red, real *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nc = g.cam_hi >= g.cam_lo ? g.cam_hi - g.cam_lo + 1 : 0, nb = g.blk_hi - g.blk_lo;
    const int64_t n = 54 * nc + 36 * nb + 1;
    if (i >= n) return;
    const RedView rv = red_view(const_cast<real *>(red), Nc, nb_off);
    int64_t k = i;
    real v;
    if (k < 36 * nc) v = rv.S_diag[36 * (int64_t)g.cam_lo + k];
    else if ((k -= 36 * nc) < 36 * nb) v = rv.S_off[36 * (int64_t)g.blk_lo + k];
    else if ((k -= 36 * nb) < 6 * nc) v = rv.rhs[6 * (int64_t)g.cam_lo + k];
    else if ((k -= 6 * nc) < 6 * nc) v = rv.udiag[6 * (int64_t)g.cam_lo + k];
    else if ((k -= 6 * nc) < 6 * nc) v = rv.gc[6 * (int64_t)g.cam_lo + k];
    else v = *rv.cost;
    out[i] = v;
}

// red[i] = sum over the ranks whose segment covers element i, in rank order (every rank forms the
// same sums from the same gathered pieces: bitwise identical buffers).  recv: world x stride.
__global__ void sum_segments_kernel(int world, const ShardSeg *__restrict__ segs, int64_t stride, int Nc, int nb_off,
                                    const real *__restrict__ recv, real *__restrict__ red)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t red_len = 54 * (int64_t)Nc + 36 * (int64_t)nb_off + 1;
    if (i >= red_len) return;
    // element i of the reduce buffer: part (0 S_diag, 1 S_off, 2 rhs, 3 diag(U), 4 g_c, 5 cost), owner
    // (camera or block), offset inside the owner's values
    int part, owner, sub;
    int64_t k = i;
    if (k < 36 * (int64_t)Nc) { part = 0; owner = (int)(k / 36); sub = (int)(k % 36); }
    else if ((k -= 36 * (int64_t)Nc) < 36 * (int64_t)nb_off) { part = 1; owner = (int)(k / 36); sub = (int)(k % 36); }
    else if ((k -= 36 * (int64_t)nb_off) < 6 * (int64_t)Nc) { part = 2; owner = (int)(k / 6); sub = (int)(k % 6); }
    else if ((k -= 6 * (int64_t)Nc) < 6 * (int64_t)Nc) { part = 3; owner = (int)(k / 6); sub = (int)(k % 6); }
    else if ((k -= 6 * (int64_t)Nc) < 6 * (int64_t)Nc) { part = 4; owner = (int)(k / 6); sub = (int)(k % 6); }
    else { part = 5; owner = 0; sub = 0; }
    real sum = real(0);
    for (int r = 0; r < world; r++) {
        const ShardSeg g = segs[r];
        const int64_t nc = g.cam_hi >= g.cam_lo ? g.cam_hi - g.cam_lo + 1 : 0, nb = g.blk_hi - g.blk_lo;
        int64_t off;
        if (part == 1) {
            if (owner < g.blk_lo || owner >= g.blk_hi) continue;
            off = 36 * nc + 36 * (int64_t)(owner - g.blk_lo) + sub;
        } else if (part == 5) {
            off = 54 * nc + 36 * nb;
        } else {
            if (owner < g.cam_lo || owner > g.cam_hi) continue;
            const int64_t c = owner - g.cam_lo;
            off = part == 0 ? 36 * c + sub : 36 * nc + 36 * nb + 6 * nc * (part - 2) + 6 * c + sub;
        }
        sum += recv[(int64_t)r * stride + off];
    }
    red[i] = sum;
}

// Zeroes the points outside [pt_lo, pt_hi) (the shard sum that follows gathers all points).
__global__ void mask_points_kernel(int Np, int pt_lo, int pt_hi, real *__restrict__ pts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * (int64_t)Np) return;
    const int j = (int)(i / 3);
    if (j < pt_lo || j >= pt_hi) pts[i] = real(0);
}

// Candidate parameters x + d and the squared norms Ceres' parameter-tolerance test needs.
// Only parameters that appear in a residual block are part of the problem (mask arrays).
// out_partial: per block {|x|^2, |d|^2}.
__global__ void __launch_bounds__(256)
candidate_kernel(int Nc, int Np, int fix_points, const real *__restrict__ t,
                 const real *__restrict__ rpy, const real *__restrict__ pts,
                 const unsigned char *__restrict__ cam_used,
                 const unsigned char *__restrict__ pt_used, int count_cams, int pt_lo, int pt_hi,
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
        if (used && j >= pt_lo && j < pt_hi) { xn = x * x; dn = d * d; }   // a shard counts its own landmarks
    }
    const real s0 = block_sum_256(xn, red);
    const real s1 = block_sum_256(dn, red);
    if (threadIdx.x == 0) { out_partial[2 * blockIdx.x] = s0; out_partial[2 * blockIdx.x + 1] = s1; }
}

// max |g| over the camera gradient (nc6 = 6 Nc values) and the gradients of np landmarks (np = 0
// with fixed points; g_p sits at reals 9..11 of a landmark record): one partial maximum per
// workgroup, grid-strided; max_of_kernel finishes.
constexpr int kGradMaxBlocks = 2048;
__global__ void __launch_bounds__(256)
gradient_max_kernel(int64_t nc6, const real *__restrict__ gc, int64_t np,
                    const real *__restrict__ lmrec, real *__restrict__ partial)
{
    __shared__ real red[4];
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    real m = 0;
    for (int64_t i = i0; i < nc6; i += stride) m = fmax(m, fabs(gc[i]));
    for (int64_t j = i0; j < np; j += stride) {
        const real *g = lmrec + j * kRec + 9;
        m = fmax(fmax(m, fabs(g[0])), fmax(fabs(g[1]), fabs(g[2])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

}  // namespace vs
