// Symmetric positive-definite solve of the reduced camera system on the device: blocked
// right-looking Cholesky (lower, in place) + triangular solves, on dense or BAND storage.
//
// Storage: element (i, j), j <= i, i - j <= bw, lives at A[i * ld + j].  Dense: ld = n, bw = n.
// Band (lower half-bandwidth bw): ld = bw and A points bw elements into an n * (bw + 1) array, so
// that row i occupies the bw + 1 consecutive slots ending at its diagonal; Cholesky fill-in stays
// inside the band, and every kernel below touches in-band elements only.
#pragma once
#include <hip/hip_runtime.h>
#include "ba_kernels.h"

namespace vs {

constexpr int kNB = 32;    // Cholesky panel width
constexpr int kTile = 64;  // trailing-update tile

// Scatter the stored upper-triangular 6x6 blocks into the LOWER triangle.  One thread per block
// element.
__global__ void assemble_lower_kernel(int Nc, int nb_off, const int *__restrict__ brow,
                                      const int *__restrict__ bcol,
                                      const real *__restrict__ S_diag,
                                      const real *__restrict__ S_off, real *__restrict__ A, int64_t ld)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nd = (int64_t)Nc * 36;
    if (tid < nd) {
        const int c = (int)(tid / 36), p = (int)(tid % 36) / 6, q = (int)(tid % 6);
        if (q <= p) A[(int64_t)(6 * c + p) * ld + 6 * c + q] = S_diag[tid];
    } else if (tid < nd + (int64_t)nb_off * 36) {
        const int64_t u = tid - nd;
        const int blk = (int)(u / 36), p = (int)(u % 36) / 6, q = (int)(u % 6);
        const int a = brow[blk], b = bcol[blk];   // a < b: S_ab[p][q] = S_ba[q][p] (lower)
        A[(int64_t)(6 * b + q) * ld + 6 * a + p] = S_off[u];
    }
}

// Cholesky of the 32 x 32 block held in LDS by a whole workgroup of 32 x 32 threads (thread
// (i, j) = element (i, j)); W is the working copy (identity-padded beyond nb), F receives the
// factor.  Column c of W is final after step c - 1 and is never rewritten, so a step reads it,
// updates the columns to its right and emits column c of F with ONE barrier.
// Returns the first failing pivot + 1 (pivot replaced by 1) or 0; uniform over the workgroup.
// 1 / sqrt(d) to double precision: hardware estimate + two Newton steps (d > 0, normal range).
__device__ __forceinline__ real rsqrt_refined(real d)
{
    real y = __builtin_amdgcn_rsq(d);
    const real hd = real(0.5) * d;
    y = y * fma(-hd * y, y, real(1.5));
    y = y * fma(-hd * y, y, real(1.5));
    return y;
}

__device__ __forceinline__ int potrf32_block(real (*W)[kNB + 1], real (*F)[kNB + 1], int nb, int i, int j)
{
    int bad = 0;
    for (int c = 0; c < nb; c++) {
        real d = W[c][c];
        if (!(d > real(0))) { if (bad == 0) bad = c + 1; d = real(1); }
        const real rs = rsqrt_refined(d);
        const real la = W[i][c] * rs, lb = W[j][c] * rs;     // column c of the factor at rows i, j
        if (j > c && j <= i && i < nb) W[i][j] = fma(-la, lb, W[i][j]);
        if (j == c && i >= c) F[i][c] = i == c ? d * rs : (i < nb ? la : real(0));
        __syncthreads();
    }
    return bad;
}

// Inverse of the lower-triangular 32 x 32 factor F (identity-padded) by a workgroup of 32 x 32
// threads: forward substitution on the identity held in X, one barrier per row.  Row c of the
// inverse is X[c][:] / F[c][c] once rows < c are eliminated; it goes straight to `out` (row-major
// 32 x 32) and X keeps the unnormalised row, which no later step reads.
__device__ __forceinline__ void trinv32_block(real (*F)[kNB + 1], real (*X)[kNB + 1], int nb, int i, int j,
                                              real *__restrict__ out)
{
    X[i][j] = i == j ? real(1) : real(0);
    __syncthreads();
    for (int c = 0; c < kNB; c++) {
        const real dc = c < nb ? F[c][c] : real(1);
        const real xc = X[c][j] / dc;
        if (i == c) out[c * kNB + j] = xc;
        else if (i > c && i < nb) X[i][j] = fma(-F[i][c], xc, X[i][j]);
        __syncthreads();
    }
}

// Factor the nb x nb diagonal block at k0 and write the inverse of its Cholesky factor
// (identity-padded to 32 x 32, row-major) to Linv.  One workgroup of 32 x 32 threads.
__global__ void __launch_bounds__(kNB *kNB)
potrf_diag_kernel(real *__restrict__ A, int64_t ld, int k0, int nb, int *__restrict__ info,
                  real *__restrict__ Linv)
{
    __shared__ real W[kNB][kNB + 1];
    __shared__ real F[kNB][kNB + 1];
    const int i = threadIdx.x / kNB, j = threadIdx.x % kNB;
    W[i][j] = (i < nb && j <= i) ? A[(int64_t)(k0 + i) * ld + k0 + j] : (i == j ? real(1) : real(0));
    F[i][j] = real(0);
    __syncthreads();
    const int bad = potrf32_block(W, F, nb, i, j);
    if (bad && threadIdx.x == 0 && *info == 0) *info = k0 + bad;
    if (i < nb && j <= i) A[(int64_t)(k0 + i) * ld + k0 + j] = F[i][j];
    __syncthreads();
    trinv32_block(F, W, nb, i, j, Linv);
}

// Inverses of all diagonal blocks of a finished factor, one workgroup per 32-column panel.
__global__ void __launch_bounds__(kNB *kNB)
trinv_panels_kernel(const real *__restrict__ A, int n, int64_t ld, real *__restrict__ Linv)
{
    __shared__ real F[kNB][kNB + 1];
    __shared__ real X[kNB][kNB + 1];
    const int k0 = blockIdx.x * kNB, nb = min(kNB, n - k0);
    const int i = threadIdx.x / kNB, j = threadIdx.x % kNB;
    F[i][j] = (i < nb && j <= i) ? A[(int64_t)(k0 + i) * ld + k0 + j] : (i == j ? real(1) : real(0));
    __syncthreads();
    trinv32_block(F, X, nb, i, j, Linv + (int64_t)blockIdx.x * kNB * kNB);
}

// Panel solve: rows i >= k0+nb:  X[i][:] = A[i][k0:k0+nb] * L_kk^-T = A_row * Linv^T.
// 32 x 32 threads: thread (c, r) forms X[row r][c] = sum_{m <= c} A[r][m] * Linv[c][m].
__global__ void __launch_bounds__(kNB *kNB)
trsm_panel_kernel(real *__restrict__ A, int n, int64_t ld, int bw, int k0, int nb,
                  const real *__restrict__ Linv)
{
    __shared__ real Li[kNB][kNB + 1];
    __shared__ real Ar[kNB][kNB + 1];
    const int c = threadIdx.x, r = threadIdx.y;
    const int i = k0 + nb + blockIdx.x * kNB + r;
    const bool in = i < n && c < nb && i - (k0 + c) <= bw;
    Li[r][c] = Linv[r * kNB + c];
    Ar[r][c] = in ? A[(int64_t)i * ld + k0 + c] : real(0);
    __syncthreads();
    real s = real(0);
    for (int m = 0; m <= c; m++) s += Ar[r][m] * Li[c][m];
    if (in) A[(int64_t)i * ld + k0 + c] = s;
}

// Trailing update of the lower triangle: A[i][j] -= sum_m X[i][m] X[j][m] for
// i,j >= k0+nb, j <= i.  64x64 tiles, 16x16 threads, 4x4 outputs per thread.
__global__ void __launch_bounds__(256)
syrk_update_kernel(real *__restrict__ A, int n, int64_t ld, int bw, int k0, int nb)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    __shared__ real Xi[kTile][kNB + 1];
    __shared__ real Xj[kTile][kNB + 1];
    const int base = k0 + nb;
    const int i0 = base + ti * kTile, j0 = base + tj * kTile;
    for (int e = threadIdx.x; e < kTile * kNB; e += 256) {
        const int r = e / kNB, c = e % kNB;
        Xi[r][c] = (i0 + r < n && c < nb && i0 + r - (k0 + c) <= bw) ? A[(int64_t)(i0 + r) * ld + k0 + c] : real(0);
        Xj[r][c] = (j0 + r < n && c < nb && j0 + r - (k0 + c) <= bw) ? A[(int64_t)(j0 + r) * ld + k0 + c] : real(0);
    }
    __syncthreads();
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    real acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = real(0);
#pragma unroll 4
    for (int m = 0; m < kNB; m++) {
        real xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { xi[a] = Xi[ty + 16 * a][m]; xj[a] = Xj[tx + 16 * a][m]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] += xi[a] * xj[b];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
            if (i < n && j <= i && i - j <= bw) A[(int64_t)i * ld + j] -= acc[a][b];
        }
}

// Whole factorisation in ONE workgroup (1024 threads) for systems whose trailing window is
// narrow (band) or small (few cameras): per 32-column panel, the workgroup factors the diagonal
// block, one thread per row forms the panel X = A_panel * L_kk^-T by forward substitution into
// LDS, and all threads apply the trailing update to the bw x bw window.  No launches between
// panels; the window lives in L2.  The block inverses the triangular solves use are formed
// afterwards for all panels at once (trinv_panels_kernel).
constexpr int kFusedMaxRows = 448;   // panel rows held in LDS (>= min(bw, n))

__global__ void __launch_bounds__(1024)
chol_fused_kernel(real *__restrict__ A, int n, int64_t ld, int bw, int *__restrict__ info)
{
    __shared__ real W[kNB][kNB + 1];
    __shared__ real F[kNB][kNB + 1];
    __shared__ real invd[kNB];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    real (*X)[kNB + 1] = reinterpret_cast<real (*)[kNB + 1]>(dyn_lds);
    const int tid = threadIdx.x;
    const int bi = tid / kNB, bj = tid % kNB;
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = min(kNB, n - k0);
        W[bi][bj] = (bi < nb && bj <= bi) ? A[(int64_t)(k0 + bi) * ld + k0 + bj] : (bi == bj ? real(1) : real(0));
        F[bi][bj] = real(0);
        __syncthreads();
        const int bad = potrf32_block(W, F, nb, bi, bj);
        if (bad && tid == 0 && *info == 0) *info = k0 + bad;
        if (bi < nb && bj <= bi) A[(int64_t)(k0 + bi) * ld + k0 + bj] = F[bi][bj];
        if (tid < kNB) invd[tid] = tid < nb ? real(1) / F[tid][tid] : real(1);
        __syncthreads();
        const int base = k0 + nb;
        const int rem = min(n - base, bw);
        if (rem <= 0) break;                         // uniform: last panel
        // panel row r: x[c] = (a[c] - sum_{m < c} x[m] F[c][m]) / F[c][c]; out-of-band a[m] = 0
        if (tid < rem) {
            const int i = base + tid;
            const real *row = A + (int64_t)i * ld + k0;
            const int m_lo = max(0, i - k0 - bw);
            real *xr = X[tid];                       // the thread's own LDS row
            for (int c = 0; c < kNB; c++) {
                const bool in = c >= m_lo && c < nb;
                xr[c] = in ? row[c] : real(0);
            }
            for (int c = 0; c < nb; c++) {
                real sum = xr[c];
                const real *fc = F[c];
#pragma unroll 8
                for (int m = 0; m < c; m++) sum = fma(-xr[m], fc[m], sum);
                sum *= invd[c];
                xr[c] = sum;
                if (c >= m_lo) A[(int64_t)i * ld + k0 + c] = sum;
            }
        }
        __syncthreads();
        // trailing update, 4 x 4 outputs per thread over 64 x 64 tiles of the lower triangle
        const int nt = (rem + kTile - 1) / kTile;
        const int n_tiles = nt * (nt + 1) / 2;
        const int sub = tid & 255, ty = sub / 16, tx = sub % 16;
        for (int t = tid >> 8; t < n_tiles; t += 4) {
            int ti = 0, acc_t = 0;
            while (acc_t + ti + 1 <= t) { acc_t += ti + 1; ti++; }
            const int tj = t - acc_t;
            real acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = real(0);
            int ri[4], rj[4];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                ri[a] = min(ti * kTile + ty + 16 * a, rem - 1);
                rj[a] = min(tj * kTile + tx + 16 * a, rem - 1);
            }
#pragma unroll 4
            for (int m = 0; m < kNB; m++) {
                real xi[4], xj[4];
#pragma unroll
                for (int a = 0; a < 4; a++) { xi[a] = X[ri[a]][m]; xj[a] = X[rj[a]][m]; }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) acc[a][b] = fma(xi[a], xj[b], acc[a][b]);
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int r = ti * kTile + ty + 16 * a, c = tj * kTile + tx + 16 * b;
                    if (r < rem && c <= r) A[(int64_t)(base + r) * ld + base + c] -= acc[a][b];
                }
        }
        __syncthreads();
    }
}

// Solve L y = b then L^T x = y in place (single workgroup, 1024 threads), L = lower of A.
// Linv holds the inverse of every 32 x 32 diagonal block of L (panel p at Linv + 1024 p), so a
// block solve is a 32 x 32 matrix-vector product.
__global__ void __launch_bounds__(1024)
chol_trsv_kernel(const real *__restrict__ A, int n, int64_t ld, int bw, const real *__restrict__ Linv,
                 real *__restrict__ b)
{
    __shared__ real y[kNB];
    __shared__ real bp[kNB];
    const int tid = threadIdx.x;
    // forward
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = min(kNB, n - k0);
        const real *Li = Linv + (int64_t)(k0 / kNB) * kNB * kNB;
        if (tid < kNB) bp[tid] = tid < nb ? b[k0 + tid] : real(0);
        __syncthreads();
        if (tid < kNB) {
            real s = real(0);
#pragma unroll
            for (int m = 0; m < kNB; m++) s = fma(Li[tid * kNB + m], bp[m], s);   // zero above the diagonal
            y[tid] = s;
            if (tid < nb) b[k0 + tid] = s;
        }
        __syncthreads();
        const int i_end = min(n, k0 + nb + bw);
        for (int i = k0 + nb + tid; i < i_end; i += 1024) {
            const real *row = A + (int64_t)i * ld + k0;
            const int m_lo = max(0, i - k0 - bw);                                     // in-band part
            real s = real(0);
#pragma unroll
            for (int m = 0; m < kNB; m++) {
                const real v = row[max(m, m_lo)];
                s = fma(m >= m_lo ? v : real(0), y[m], s);                             // y[m] = 0 for m >= nb
            }
            b[i] -= s;
        }
        __syncthreads();
    }
    // backward
    for (int k0 = ((n - 1) / kNB) * kNB; k0 >= 0; k0 -= kNB) {
        const int nb = min(kNB, n - k0);
        const real *Li = Linv + (int64_t)(k0 / kNB) * kNB * kNB;
        if (tid < kNB) bp[tid] = tid < nb ? b[k0 + tid] : real(0);
        __syncthreads();
        if (tid < kNB) {
            real s = real(0);
#pragma unroll
            for (int m = 0; m < kNB; m++) s = fma(Li[m * kNB + tid], bp[m], s);   // (L^-1)^T
            y[tid] = s;
            if (tid < nb) b[k0 + tid] = s;
        }
        __syncthreads();
        for (int i = max(0, k0 - bw) + tid; i < k0; i += 1024) {
            real s = real(0);
            const int m_end = min(nb, bw + i - k0 + 1);                                // in-band part
#pragma unroll
            for (int m = 0; m < kNB; m++) {
                const real v = A[(int64_t)(k0 + min(m, m_end - 1)) * ld + i];
                s = fma(m < m_end ? v : real(0), y[m], s);
            }
            b[i] -= s;
        }
        __syncthreads();
    }
}

// Motion-only BA: S is block diagonal; one thread per camera solves its 6x6 system by Cholesky.
__global__ void solve6_kernel(int Nc, const real *__restrict__ S_diag, const real *__restrict__ rhs,
                              real *__restrict__ x, int *__restrict__ info)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nc) return;
    real L[6][6], b[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        b[p] = rhs[6 * c + p];
#pragma unroll
        for (int q = 0; q < 6; q++) L[p][q] = S_diag[(int64_t)c * 36 + 6 * p + q];
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
#pragma unroll
    for (int p = 0; p < 6; p++) x[6 * c + p] = b[p];
    if (bad && *info == 0) *info = c + 1;
}

// Factor + solve.  A, ld, bw: storage as described at the top of this file, lower triangle
// filled; b: rhs, overwritten with the solution; info (device int, zeroed by the caller): first
// failing pivot + 1, or 0; Linv: device workspace of ceil(n / 32) * 32 * 32 reals.
static inline int cholesky_solve(real *A, real *b, int n, int64_t ld, int bw, int *info, real *Linv,
                                 hipStream_t s)
{
    const int window = bw < n ? bw : n;
    if (window <= kFusedMaxRows) {
        const size_t lds = (size_t)window * (kNB + 1) * sizeof(real);
        static bool lds_limit_raised = false;     // dynamic LDS beyond the default limit
        if (!lds_limit_raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(chol_fused_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kFusedMaxRows * (kNB + 1) * (int)sizeof(real)) != hipSuccess) {
                set_error("cannot raise the dynamic LDS limit of the fused Cholesky kernel");
                return VS_ERR_HIP;
            }
            lds_limit_raised = true;
        }
        hipLaunchKernelGGL(chol_fused_kernel, dim3(1), dim3(1024), lds, s, A, n, ld, bw, info);
        hipLaunchKernelGGL(trinv_panels_kernel, dim3((n + kNB - 1) / kNB), dim3(kNB * kNB), 0, s, A, n, ld, Linv);
    } else
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = n - k0 < kNB ? n - k0 : kNB;
        real *Lp = Linv + (int64_t)(k0 / kNB) * kNB * kNB;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(kNB * kNB), 0, s, A, ld, k0, nb, info, Lp);
        int rem = n - k0 - nb;
        if (rem > bw) rem = bw;   // rows further down hold no in-band element of this panel
        if (rem > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel, dim3((rem + kNB - 1) / kNB), dim3(kNB, kNB), 0, s, A, n, ld, bw, k0, nb, Lp);
            const int nt = (rem + kTile - 1) / kTile;
            hipLaunchKernelGGL(syrk_update_kernel, dim3(nt, nt), dim3(256), 0, s, A, n, ld, bw, k0, nb);
        }
    }
    hipLaunchKernelGGL(chol_trsv_kernel, dim3(1), dim3(1024), 0, s, A, n, ld, bw, Linv, b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Cholesky launch failed: %s", hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

}  // namespace vs
