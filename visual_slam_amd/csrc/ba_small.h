// Kernels for small windows (included by ba.hip only): the motion-only LM loop in ONE launch, and the
// fused set-up (small_scales_kernel, small_prepare_kernel) and tail (small_tail_kernel) of a full-BA LM
// step, which leave the general landmark / block / factorisation kernels six launches per iteration
// instead of fourteen.
//
// Motion-only bundle adjustment of a small window in one launch:
// The reference's live mode is Tracking::solveCamera (src/core/tracking.cc:79-120): every point is
// constant (fix_camera_or_point = true, src/core/optimizer.cc:134-137), a handful of keyframes, a
// few thousand observations, ceres::Solve with its default Levenberg-Marquardt loop -- twice per
// frame.  With the points fixed the normal equations are one 6x6 block per camera, so the whole
// loop (linearise, damp, solve, candidate cost, accept / reject, trust-region update, the three
// termination tests) fits one workgroup: no launch, copy or host synchronisation per iteration.
// The general path spends ~60 us per iteration on a dozen launches at this size; this kernel
// spends a few.  The loop is the one vs_ba_solve runs on the host (same tests in the same order).
#pragma once
#include "ba_kernels.h"

namespace vs {

constexpr int kSmallThreads = 512;     // two waves per SIMD: 256 registers per lane for the accumulators, a batch of observations and the 6x6 solve
constexpr int kSmallMaxCams = 128;
constexpr int kSmallMaxObs = 16384;

struct SmallOut {            // written by thread 0, straight into coherent page-locked host memory;
    real initial_cost, final_cost, final_radius;     // `seq` last: the host polls for it
    int iterations, successful_steps, termination, seq;
};

// sum over the wave of NV per-lane values, every lane ends with the totals (butterfly order: fixed)
template <int NV>
__device__ __forceinline__ void wave_sum_all(real *v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < NV; i++) v[i] += __shfl_xor(v[i], off, 64);
}

// block-wide sum, result returned to every thread (scratch: >= 16 reals; two barriers)
__device__ __forceinline__ real block_sum_all(real v, real *scratch)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    real s = real(0);
#pragma unroll
    for (int i = 0; i < kSmallThreads / 64; i++) s += scratch[i];
    return s;
}

__global__ void __launch_bounds__(kSmallThreads)
motion_only_solve_kernel(int Nc, Intr K, real *__restrict__ t_io, real *__restrict__ rpy_io,
                         const real *__restrict__ pts, const int *__restrict__ cam_ptr,
                         const int *__restrict__ cpt, const real *__restrict__ cuv,
                         const unsigned char *__restrict__ cam_used, HuberObs hub, vs_lm_options opt,
                         real *__restrict__ scale_out, SmallOut *out, real *cams_host, int seq)
{
    __shared__ CamData cams[kSmallMaxCams], cams2[kSmallMaxCams];
    __shared__ real x[6 * kSmallMaxCams], x2[6 * kSmallMaxCams], scale[6 * kSmallMaxCams], dc[6 * kSmallMaxCams];
    __shared__ real S[kSmallMaxCams][21], gc[6 * kSmallMaxCams], U[6 * kSmallMaxCams];
    __shared__ real scratch[kSmallThreads / 64], mccv[kSmallMaxCams];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kSmallThreads / 64;

    for (int i = tid; i < 6 * Nc; i += kSmallThreads) {
        const int c = i / 6, k = i % 6;
        x[i] = k < 3 ? t_io[3 * c + k] : rpy_io[3 * c + k - 3];
    }
    __syncthreads();
    auto build_cams = [&](const real *par, CamData *dst) {
        for (int c = tid; c < Nc; c += kSmallThreads) {
            real sr, cr, sp, cp, sy, cy;
            sincos(par[6 * c + 3], &sr, &cr);
            sincos(par[6 * c + 4], &sp, &cp);
            sincos(par[6 * c + 5], &sy, &cy);
            dst[c] = cam_from_sincos(par + 6 * c, 0, sr, cr, sp, cp, sy, cy);      // the translation leads a camera's six parameters
        }
        __syncthreads();
    };

    // A lane's observations of one camera are fetched kBatch at a time before any of them is used
    // (landmark index, then position and pixel): two memory latencies per batch instead of two per
    // observation -- at this size the kernel is a chain of latencies, not of arithmetic.
    constexpr int kBatch = 6;
    struct Obs { real P[3]; double2 uv; int j; };
    auto fetch_batch = [&](int k0, int kend, Obs *ob) {
#pragma unroll
        for (int u = 0; u < kBatch; u++) { const int k = k0 + 64 * u; ob[u].j = k < kend ? cpt[k] : -1; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const int k = k0 + 64 * u;
            if (ob[u].j >= 0) {
                const int j = ob[u].j;
                ob[u].P[0] = pts[3 * j]; ob[u].P[1] = pts[3 * j + 1]; ob[u].P[2] = pts[3 * j + 2];
                ob[u].uv = *reinterpret_cast<const double2 *>(cuv + 2 * (int64_t)k);
            }
        }
    };

    // cost 0.5 sum rho(|r|^2) at camera records `cd` (every thread gets the total)
    auto cost_at = [&](const CamData *cd) -> real {
        real acc = real(0);
        for (int c = wave; c < Nc; c += kWaves) {
            const CamData cam = cd[c];
            const int kend = cam_ptr[c + 1];
            for (int k0 = cam_ptr[c] + lane; k0 < kend; k0 += 64 * kBatch) {
                Obs ob[kBatch];
                fetch_batch(k0, kend, ob);
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    if (ob[u].j < 0) continue;
                    ObsLin o;
                    project_obs(cam, ob[u].P, ob[u].uv.x, ob[u].uv.y, K, o);
                    real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
                    if (c == hub.cam && ob[u].j == hub.pt) (void)huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
                    acc += hr;
                }
            }
        }
        return block_sum_all(acc, scratch);
    };

    // J_c^T J_c (upper triangle, 21: its diagonal is diag(U)) and g_c = J_c^T r (6) per camera, and the cost, at `cams`
    auto linearise = [&]() -> real {
        real cost_acc = real(0);
        for (int c = wave; c < Nc; c += kWaves) {
            const CamData cam = cams[c];
            real a[27];
#pragma unroll
            for (int v = 0; v < 27; v++) a[v] = real(0);
            const int kend = cam_ptr[c + 1];
            for (int k0 = cam_ptr[c] + lane; k0 < kend; k0 += 64 * kBatch) {
                Obs ob[kBatch];
                fetch_batch(k0, kend, ob);
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    if (ob[u].j < 0) continue;
                    ObsLin o;
                    project_obs(cam, ob[u].P, ob[u].uv.x, ob[u].uv.y, K, o);
                    real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
                    if (c == hub.cam && ob[u].j == hub.pt) {
                        const real w = huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
                        scale_lin(o, w);
                    }
                    cost_acc += hr;
                    real Jc[12];
                    camera_jac(cam, o, ob[u].P, Jc);
                    int idx = 0;
#pragma unroll
                    for (int p = 0; p < 6; p++) {
#pragma unroll
                        for (int q = p; q < 6; q++) { a[idx] = fma(Jc[p], Jc[q], fma(Jc[6 + p], Jc[6 + q], a[idx])); idx++; }
                        a[21 + p] = fma(Jc[p], o.r0, fma(Jc[6 + p], o.r1, a[21 + p]));
                    }
                }
            }
            wave_sum_all<27>(a);
            if (lane == 0) {
                int idx = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    U[6 * c + p] = a[idx];
#pragma unroll
                    for (int q = p; q < 6; q++) { S[c][idx] = a[idx]; idx++; }
                    gc[6 * c + p] = a[21 + p];
                }
            }
        }
        return block_sum_all(cost_acc, scratch);
    };
    auto gradient_max = [&]() -> real {
        real m = real(0);
        for (int i = tid; i < 6 * Nc; i += kSmallThreads) m = fmax(m, fabs(gc[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        __syncthreads();
        if (lane == 0) scratch[wave] = m;
        __syncthreads();
        real r = real(0);
#pragma unroll
        for (int i = 0; i < kWaves; i++) r = fmax(r, scratch[i]);
        return r;
    };

    build_cams(x, cams);
    real cost = linearise();
    const real cost0 = cost;
    for (int i = tid; i < 6 * Nc; i += kSmallThreads) {
        scale[i] = real(1) / (real(1) + sqrt(U[i]));
        scale_out[i] = scale[i];
    }
    real gmax = gradient_max();
    real radius = opt.initial_trust_region_radius, decrease_factor = real(2);
    int it = 0, term = 1, nsucc = 0, invalid = 0;
    bool running = gmax > opt.gradient_tolerance;
    if (!running) term = 0;
    // every thread carries the same loop state (all inputs of the decisions are block-wide sums)
    while (running) {
        if (it >= opt.max_num_iterations || radius < opt.min_trust_region_radius) {
            term = it >= opt.max_num_iterations ? 1 : 0;
            break;
        }
        it++;
        // damped 6x6 solve per camera
        if (tid == 0) s_bad = 0;
        __syncthreads();
        for (int c = tid; c < Nc; c += kSmallThreads) {
            real L[6][6], b[6];
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 6; p++) {
#pragma unroll
                for (int q = p; q < 6; q++) { L[p][q] = S[c][idx]; L[q][p] = S[c][idx]; idx++; }
                const real s = scale[6 * c + p], s2 = s * s;
                real xd = s2 * U[6 * c + p];
                xd = xd < opt.min_lm_diagonal ? opt.min_lm_diagonal : xd;
                xd = xd > opt.max_lm_diagonal ? opt.max_lm_diagonal : xd;
                L[p][p] += xd / radius / s2;
                b[p] = -gc[6 * c + p];
            }
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                real d = L[j][j];
#pragma unroll
                for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
                if (!(d > real(0))) { bad = true; d = real(1); }
                d = sqrt(d);
                L[j][j] = d;
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    real s = L[i][j];
#pragma unroll
                    for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
                    L[i][j] = s / d;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                real s = b[i];
#pragma unroll
                for (int k = 0; k < i; k++) s -= L[i][k] * b[k];
                b[i] = s / L[i][i];
            }
#pragma unroll
            for (int i = 5; i >= 0; i--) {
                real s = b[i];
#pragma unroll
                for (int k = i + 1; k < 6; k++) s -= L[k][i] * b[k];
                b[i] = s / L[i][i];
            }
            if (bad) s_bad = 1;
            // model cost change of this camera: -(d . g + d^T (J^T J) d / 2), step norms, candidate
            real dg = real(0), dSd = real(0);
            idx = 0;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                dg += b[p] * gc[6 * c + p];
#pragma unroll
                for (int q = p; q < 6; q++) { dSd += (p == q ? real(1) : real(2)) * b[p] * S[c][idx] * b[q]; idx++; }
            }
            const bool used = cam_used[c] != 0;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const real d = used ? b[p] : real(0);
                dc[6 * c + p] = d;
                x2[6 * c + p] = x[6 * c + p] + d;
            }
            mccv[c] = -(dg + real(0.5) * dSd);
        }
        __syncthreads();
        real mcc_l = real(0), xn_l = real(0), sn_l = real(0);
        for (int c = tid; c < Nc; c += kSmallThreads) {
            mcc_l += mccv[c];
            if (cam_used[c]) {
#pragma unroll
                for (int p = 0; p < 6; p++) { xn_l += x[6 * c + p] * x[6 * c + p]; sn_l += dc[6 * c + p] * dc[6 * c + p]; }
            }
        }
        const real mcc = block_sum_all(mcc_l, scratch);
        const real xnorm2 = block_sum_all(xn_l, scratch);
        const real snorm2 = block_sum_all(sn_l, scratch);
        const bool fail = s_bad != 0;
        if (fail || !(mcc > real(0))) {
            if (++invalid >= opt.max_consecutive_invalid_steps) { term = 2; break; }
            radius *= real(0.5);
            continue;
        }
        invalid = 0;
        build_cams(x2, cams2);
        const real cost_new = cost_at(cams2);
        const real xn = sqrt(xnorm2), sn = sqrt(snorm2);
        if (sn <= opt.parameter_tolerance * (xn + opt.parameter_tolerance)) { term = 0; break; }
        const real change = cost - cost_new;
        if (fabs(change) <= opt.function_tolerance * cost) { term = 0; break; }
        const real rho = change / mcc;
        if (rho > opt.min_relative_decrease) {
            __syncthreads();
            for (int i = tid; i < 6 * Nc; i += kSmallThreads) x[i] = x2[i];
            for (int c = tid; c < Nc; c += kSmallThreads) cams[c] = cams2[c];
            __syncthreads();
            cost = cost_new; nsucc++;
            const real y = real(2) * rho - real(1);
            real f = real(1) - y * y * y;
            if (f < real(1) / real(3)) f = real(1) / real(3);
            radius = fmin(radius / f, (real)opt.max_trust_region_radius);
            decrease_factor = real(2);
            (void)linearise();
            gmax = gradient_max();
            if (gmax <= opt.gradient_tolerance) { term = 0; break; }
        } else {
            radius /= decrease_factor;
            decrease_factor *= real(2);
        }
    }
    __syncthreads();
    for (int i = tid; i < 6 * Nc; i += kSmallThreads) {
        const int c = i / 6, k = i % 6;
        if (k < 3) t_io[3 * c + k] = x[i]; else rpy_io[3 * c + k - 3] = x[i];
        // ... and straight into host memory ([t | rpy]): reading the cameras back costs no GPU call
        if (cams_host) cams_host[k < 3 ? 3 * c + k : 3 * Nc + 3 * c + k - 3] = x[i];
    }
    if (tid == 0) {
        out->initial_cost = cost0; out->final_cost = cost; out->final_radius = radius;
        out->iterations = it; out->successful_steps = nsucc; out->termination = term;
    }
    // the cameras above must be in memory before the host sees the sequence number
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// Set-up of the reduced camera solve of a small window (6 Nc <= kSmallSolveN) in ONE launch of one
// workgroup: what gradient_max_kernel + max_of_kernel, camera_damping_kernel, the clearing of the
// dense matrix and assemble_lower_kernel do in five launches for a system of any size.  At ten
// keyframes (the reference's tracking window, src/core/tracking.cc:597-601) those launches cost
// several times the work in them.
//   1. (gmax_out != nullptr) max |gradient| of the linearisation the system belongs to
//   2. LM diagonal of the cameras, added to S_cc in place; right-hand side copied into the solution
//      vector; status word cleared (as camera_damping_kernel does)
//   3. the dense lower triangle A[i * ld + k] (k <= i) for cholesky_solve, the rest of the buffer
//      (a_elems reals) zero
// (A factorisation inside this kernel was built and measured: one workgroup, L D L^T with the
// elements in registers, one barrier per column -- 1.05 us per column, 65 us at 60 unknowns,
// against 27 us for the panel kernels of dense_solve.h; see tools/micro/small_solve_time.hip.)
constexpr int kSmallSolveN = 192;           // 32 cameras; the system is always factored on dense storage (one workgroup, dense_solve.h)
constexpr int kSmallPrepMaxPoints = 16384;  // landmark gradients scanned by one workgroup
constexpr int kSmallSolveThreads = 1024;    // the kernel is a few dependent memory round trips: more threads, fewer turns per loop

__global__ void __launch_bounds__(kSmallSolveThreads)
small_prepare_kernel(int Nc, int nb_off, const int *__restrict__ brow, const int *__restrict__ bcol,
                     real *__restrict__ S_diag, const real *__restrict__ S_off, const real *__restrict__ rhs,
                     const real *__restrict__ udiag, const real *__restrict__ scale_c, real radius, real min_d,
                     real max_d, real *__restrict__ lam_out, real *__restrict__ rhs_copy, int *__restrict__ info,
                     real *__restrict__ A, int ld, int a_elems, const real *__restrict__ gc, int np,
                     const real *__restrict__ lmrec, real *__restrict__ gmax_out)
{
    __shared__ real red[kSmallSolveThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = 6 * Nc;
    if (gmax_out) {
        real m = 0;
        for (int i = tid; i < n; i += kSmallSolveThreads) m = fmax(m, fabs(gc[i]));
        for (int j = tid; j < np; j += kSmallSolveThreads) {
            const real *g = lmrec + (int64_t)j * kRec + 9;
            m = fmax(fmax(m, fabs(g[0])), fmax(fabs(g[1]), fabs(g[2])));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        if (lane == 0) red[wave] = m;
    }
    // every element of the buffer is written exactly once: block elements by the loops below, the
    // rest (upper triangle, padding) with zero here
    for (int i = tid; i < a_elems; i += kSmallSolveThreads) {
        const int r = i / ld, k = i - r * ld;
        if (!(r < n && k <= r)) A[i] = real(0);
    }
    if (tid == 0) *info = 0;
    __syncthreads();
    if (gmax_out && tid == 0) {
        real m = red[0];
        for (int w = 1; w < kSmallSolveThreads / 64; w++) m = fmax(m, red[w]);
        *gmax_out = m;
    }
    // all blocks are present in a window this small only if every pair of cameras shares a landmark:
    // zero the lower triangle first where a block may be missing
    if (nb_off != Nc * (Nc - 1) / 2) {
        for (int i = tid; i < n * ld; i += kSmallSolveThreads) {
            const int r = i / ld, k = i - r * ld;
            if (k <= r) A[i] = real(0);
        }
        __syncthreads();
    }
    for (int e = tid; e < Nc * 36; e += kSmallSolveThreads) {
        const int c = e / 36, p = (e % 36) / 6, q = e % 6;
        if (q > p) continue;
        real v = S_diag[e];
        if (p == q) {
            const int i = 6 * c + p;
            real l = real(0);
            if (scale_c) {
                const real s = scale_c[i], s2 = s * s;
                real x = s2 * udiag[i];
                x = x < min_d ? min_d : x;
                x = x > max_d ? max_d : x;
                l = x / radius / s2;
            }
            if (lam_out) lam_out[i] = l;
            v += l;
            S_diag[e] = v;
            rhs_copy[i] = rhs[i];
        }
        A[(6 * c + p) * ld + 6 * c + q] = v;
    }
    for (int e = tid; e < nb_off * 36; e += kSmallSolveThreads) {
        const int blk = e / 36, p = (e % 36) / 6, q = e % 6;
        const int a = brow[blk], b = bcol[blk];          // a < b: S_ab[p][q] is element (6 b + q, 6 a + p)
        A[(6 * b + q) * ld + 6 * a + p] = S_off[e];
    }
}

// ---------------------------------------------------------------------------------------------
// After the first (undamped) pass of a solve over a small window: the Jacobi scales of cameras and
// landmarks, s = 1 / (1 + sqrt(H_ii)), and max |gradient|, in ONE launch of one workgroup instead of
// the four of jacobi_scale_kernel (twice), gradient_max_kernel and max_of_kernel.
__global__ void __launch_bounds__(1024)
small_scales_kernel(int nc6, const real *__restrict__ udiag, real *__restrict__ scale_c, int np,
                    const real *__restrict__ vdiag, real *__restrict__ scale_p, const real *__restrict__ gc,
                    int np_grad, const real *__restrict__ lmrec, real *__restrict__ gmax_out)
{
    __shared__ real red[16];
    const int tid = threadIdx.x;
    real m = 0;
    for (int i = tid; i < nc6; i += 1024) {
        scale_c[i] = real(1) / (real(1) + sqrt(udiag[i]));
        m = fmax(m, fabs(gc[i]));
    }
    for (int i = tid; i < 3 * np; i += 1024) scale_p[i] = real(1) / (real(1) + sqrt(vdiag[i]));
    for (int j = tid; j < np_grad; j += 1024) {
        const real *g = lmrec + (int64_t)j * kRec + 9;
        m = fmax(fmax(m, fabs(g[0])), fmax(fabs(g[1]), fabs(g[2])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) m = fmax(m, red[w]);
        *gmax_out = m;
    }
}

// ---------------------------------------------------------------------------------------------
// The rest of an LM step of a small window in ONE launch: what backsub_kernel, candidate_kernel,
// cost_kernel and step_scalars_kernel do in four.  Two lanes take a landmark through all of it --
// back-substitution (step of the point, its share of the model cost change), candidate position,
// cost of its observations at the candidate -- because nothing in that chain needs another
// landmark; every workgroup builds the candidate cameras into its LDS first.  The four sums of the
// step (model cost change, |x|^2, |d|^2, candidate cost) leave each workgroup as partials; the
// workgroup that finishes last (a ticket counter; the grid is a handful of workgroups, so the
// device-scope fences cost nothing that matters) adds them in workgroup order and writes the
// step's scalars -- with `mail` also straight into host memory, followed by the sequence number
// `seq`: the host polls for it instead of paying a copy and a stream synchronisation per iteration.
constexpr int kSmallTailThreads = 256;       // 128 landmarks per workgroup
constexpr int kSmallTailMaxGroups = 64;
constexpr int kSmallTailMaxPoints = kSmallTailMaxGroups * (kSmallTailThreads / 2);

// What the last workgroup needs to take the LM loop's accept / reject decision itself (the host takes
// it again from the same numbers): the next radius goes to *radius_next, which the linearisation
// enqueued behind this kernel -- at the candidate, before the host knows whether the step is
// accepted -- reads as its damping radius.  radius_next == nullptr: no decision wanted.
struct TailDecision {
    real cost_cur, radius, min_relative_decrease, max_radius, parameter_tolerance, function_tolerance;
    real *radius_next;
};

__device__ __forceinline__ real block_sum_tail(real v, real *scratch /* >= 4 reals */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    real s = real(0);
    if (threadIdx.x == 0)
        for (int i = 0; i < kSmallTailThreads / 64; i++) s += scratch[i];
    return s;
}

__global__ void __launch_bounds__(kSmallTailThreads)
small_tail_kernel(int Np, int Nc, Intr K, const CamData *__restrict__ cams, const real *__restrict__ t,
                  const real *__restrict__ rpy, const real *__restrict__ pts, const int *__restrict__ pt_ptr,
                  const int *__restrict__ lcam, const real *__restrict__ luv, HuberObs hub, const real *__restrict__ hub_w,
                  const real *__restrict__ lmrec, const real *__restrict__ dcam, const unsigned char *__restrict__ cam_used,
                  const unsigned char *__restrict__ pt_used, real *__restrict__ dpt, real *__restrict__ t2,
                  real *__restrict__ rpy2, real *__restrict__ pts2, int want_cost, const int *__restrict__ info,
                  real *partial, int *ticket, real *__restrict__ out, unsigned long long *mail, unsigned long long seq,
                  TailDecision dec)
{
    __shared__ CamData scams[kSmallSolveN / 6], scams2[kSmallSolveN / 6];
    __shared__ real x2[kSmallSolveN], sc2[kSmallSolveN], red[kSmallTailThreads / 64];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    real xn = 0, dn = 0, mcc = 0, cost = 0;
    {
        const double2 *src = reinterpret_cast<const double2 *>(cams);
        double2 *dst = reinterpret_cast<double2 *>(scams);
        for (int i = tid; i < Nc * 8; i += kSmallTailThreads) dst[i] = src[i];
    }
    if (tid < 6 * Nc) {
        const int c = tid / 6, k = tid % 6;
        const bool used = cam_used[c] != 0;
        const real x = k < 3 ? t[3 * c + k] : rpy[3 * c + k - 3];
        const real d = used ? dcam[tid] : real(0);
        x2[tid] = x + d;
        if (blockIdx.x == 0) {
            if (k < 3) t2[3 * c + k] = x + d; else rpy2[3 * c + k - 3] = x + d;
            if (used) { xn = x * x; dn = d * d; }
        }
    }
    __syncthreads();
    if (tid < 3 * Nc) {                                   // one angle per thread: the three sincos of a camera side by side
        const int c = tid / 3, k = tid % 3;
        sincos(x2[6 * c + 3 + k], &sc2[2 * tid], &sc2[2 * tid + 1]);
    }
    __syncthreads();
    if (tid < Nc) {
        // cam_from_sincos reads the translation as t[3 c + k]
        const real tt[3] = { x2[6 * tid], x2[6 * tid + 1], x2[6 * tid + 2] };
        const real *q = sc2 + 6 * tid;
        scams2[tid] = cam_from_sincos(tt, 0, q[0], q[1], q[2], q[3], q[4], q[5]);
    }
    __syncthreads();
    // two lanes per landmark: each takes every other observation in both passes, the sums of the
    // back-substitution are united with one exchange between the neighbours
    const int sub = tid & 1;
    for (int j = blockIdx.x * (kSmallTailThreads / 2) + (tid >> 1); j < Np; j += gridDim.x * (kSmallTailThreads / 2)) {
        const int b = pt_ptr[j], e = pt_ptr[j + 1];
        const real P[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
        real acc[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // q[3] | V[6] | sum u.r | sum u.u
        for (int pos = b + sub; pos < e; pos += 2) {
            const int c = lcam[pos];
            const double2 uv = *reinterpret_cast<const double2 *>(luv + 2 * (int64_t)pos);
            const double2 *dsrc = reinterpret_cast<const double2 *>(dcam + 6 * (int64_t)c);
            const double2 d01 = dsrc[0], d23 = dsrc[1], d45 = dsrc[2];
            const CamData cd = scams[c];
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
            acc[9] += u0 * o.r0 + u1 * o.r1;
            acc[10] += u0 * u0 + u1 * u1;
#pragma unroll
            for (int k = 0; k < 3; k++) acc[k] += Jp[k] * u0 + Jp[3 + k] * u1;
            acc[3] += Jp[0] * Jp[0] + Jp[3] * Jp[3]; acc[4] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
            acc[5] += Jp[0] * Jp[2] + Jp[3] * Jp[5]; acc[6] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
            acc[7] += Jp[1] * Jp[2] + Jp[4] * Jp[5]; acc[8] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
        }
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const real other = __shfl_xor(acc[k], 1, 64);
            acc[k] = sub == 0 ? acc[k] + other : other + acc[k];     // both lanes: even lane's share first
        }
        const real *q = acc, *V = acc + 3;
        real rec[kRec];
        load_rec(lmrec, j, rec, 6);
        const real *vi = rec + 3, *gp = rec + 9;
        real dp[3] = { 0, 0, 0 };
        if (e > b) {
            const real a0 = gp[0] + q[0], a1 = gp[1] + q[1], a2 = gp[2] + q[2];
            dp[0] = -(vi[0] * a0 + vi[1] * a1 + vi[2] * a2);
            dp[1] = -(vi[1] * a0 + vi[3] * a1 + vi[4] * a2);
            dp[2] = -(vi[2] * a0 + vi[4] * a1 + vi[5] * a2);
        }
        const bool used = pt_used[j] != 0;
        real P2[3];
#pragma unroll
        for (int k = 0; k < 3; k++) P2[k] = P[k] + (used ? dp[k] : real(0));
        if (sub == 0) {
            if (e > b) {
                const real Vd0 = V[0] * dp[0] + V[1] * dp[1] + V[2] * dp[2];
                const real Vd1 = V[1] * dp[0] + V[3] * dp[1] + V[4] * dp[2];
                const real Vd2 = V[2] * dp[0] + V[4] * dp[1] + V[5] * dp[2];
                const real dp_g = dp[0] * gp[0] + dp[1] * gp[1] + dp[2] * gp[2];
                const real dp_q = dp[0] * q[0] + dp[1] * q[1] + dp[2] * q[2];
                const real dp_V_dp = dp[0] * Vd0 + dp[1] * Vd1 + dp[2] * Vd2;
                mcc += -((acc[9] + dp_g) + real(0.5) * (acc[10] + real(2) * dp_q + dp_V_dp));
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dpt[3 * j + k] = dp[k];
                pts2[3 * j + k] = P2[k];
                if (used) { xn += P[k] * P[k]; dn += dp[k] * dp[k]; }
            }
        }
        if (want_cost)
            for (int pos = b + sub; pos < e; pos += 2) {
                const CamData cd = scams2[lcam[pos]];
                ObsLin o;
                project_obs(cd, P2, luv[2 * pos], luv[2 * pos + 1], K, o);
                real hr = real(0.5) * (o.r0 * o.r0 + o.r1 * o.r1);
                if (pos == hub.pos) (void)huber_weight(hub.a, o.r0 * o.r0 + o.r1 * o.r1, &hr);
                cost += hr;
            }
    }
    mcc = block_sum_tail(mcc, red);
    xn = block_sum_tail(xn, red);
    dn = block_sum_tail(dn, red);
    cost = block_sum_tail(cost, red);
    if (tid == 0) {
        real *p = partial + 4 * blockIdx.x;
        p[0] = mcc; p[1] = xn; p[2] = dn; p[3] = cost;
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || tid != 0) return;
    __threadfence();
    *ticket = 0;                                        // ready for the next launch
    mcc = xn = dn = cost = real(0);
    for (int g = 0; g < (int)gridDim.x; g++) {
        const volatile real *p = partial + 4 * g;
        mcc += p[0]; xn += p[1]; dn += p[2]; cost += p[3];
    }
    out[0] = mcc; out[1] = xn; out[2] = dn;
    if (want_cost) out[3] = cost;
    out[6] = (real)*info;
    // the decision vs_ba_solve's loop takes from these numbers, in its order of tests
    real accepted = real(0), radius_next = dec.radius;
    if (dec.radius_next) {
        if (*info == 0 && mcc > real(0)) {
            const real xnorm = sqrt(xn), snorm = sqrt(dn);
            const real change = dec.cost_cur - cost;
            if (!(snorm <= dec.parameter_tolerance * (xnorm + dec.parameter_tolerance)) &&
                !(fabs(change) <= dec.function_tolerance * dec.cost_cur)) {
                const real rho = change / mcc;
                if (rho > dec.min_relative_decrease) {
                    accepted = real(1);
                    const real y = real(2) * rho - real(1);
                    real f = real(1) - y * y * y;
                    if (f < real(1) / real(3)) f = real(1) / real(3);
                    radius_next = fmin(dec.radius / f, dec.max_radius);
                }
            }
        }
        *dec.radius_next = radius_next;
    }
    if (mail) {
        real *m = reinterpret_cast<real *>(mail);
        m[0] = mcc; m[1] = xn; m[2] = dn; m[3] = want_cost ? cost : out[3];
        m[4] = out[4]; m[5] = radius_next; m[6] = (real)*info; m[7] = accepted;
        __threadfence_system();
        __hip_atomic_store(mail + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace vs
