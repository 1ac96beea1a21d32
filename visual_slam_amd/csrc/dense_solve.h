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

// Factor the nb x nb diagonal block at k0 (one workgroup of 32x32 threads) and write the
// inverse of its Cholesky factor (identity-padded to 32 x 32, row-major) to Linv.
__global__ void __launch_bounds__(kNB *kNB)
potrf_diag_kernel(real *__restrict__ A, int64_t ld, int k0, int nb, int *__restrict__ info,
                  real *__restrict__ Linv)
{
    __shared__ real L[kNB][kNB + 1];
    __shared__ real Li[kNB][kNB + 1];
    const int i = threadIdx.y, j = threadIdx.x;
    L[i][j] = (i < nb && j <= i) ? A[(int64_t)(k0 + i) * ld + k0 + j] : (i == j ? real(1) : real(0));
    Li[i][j] = real(0);
    __syncthreads();
    for (int c = 0; c < nb; c++) {
        if (i == c && j == c) {
            const real d = L[c][c];
            if (!(d > real(0))) { if (*info == 0) *info = k0 + c + 1; L[c][c] = real(1); }
            else L[c][c] = sqrt(d);
        }
        __syncthreads();
        if (j == c && i > c && i < nb) L[i][c] /= L[c][c];
        __syncthreads();
        if (i > c && j > c && j <= i && i < nb) L[i][j] -= L[i][c] * L[j][c];
        __syncthreads();
    }
    if (i < nb && j <= i) A[(int64_t)(k0 + i) * ld + k0 + j] = L[i][j];
    // column j of L^-1 by forward substitution (thread row 0 only; 32 independent columns)
    if (i == 0) {
        Li[j][j] = real(1) / L[j][j];
        for (int r = j + 1; r < kNB; r++) {
            real s = real(0);
            for (int m = j; m < r; m++) s += L[r][m] * Li[m][j];
            Li[r][j] = -s / L[r][r];
        }
    }
    __syncthreads();
    Linv[i * kNB + j] = Li[i][j];
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

// Solve L y = b then L^T x = y in place (single workgroup, 1024 threads), L = lower of A.
__global__ void __launch_bounds__(1024)
chol_trsv_kernel(const real *__restrict__ A, int n, int64_t ld, int bw, real *__restrict__ b)
{
    __shared__ real y[kNB];
    const int tid = threadIdx.x;
    // forward
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = min(kNB, n - k0);
        if (tid == 0) {
            for (int c = 0; c < nb; c++) {
                real s = b[k0 + c];
                const real *row = A + (int64_t)(k0 + c) * ld + k0;
                for (int m = 0; m < c; m++) s -= row[m] * y[m];
                y[c] = s / row[c];
            }
            for (int c = 0; c < nb; c++) b[k0 + c] = y[c];
        }
        __syncthreads();
        const int i_end = min(n, k0 + nb + bw);
        for (int i = k0 + nb + tid; i < i_end; i += 1024) {
            const real *row = A + (int64_t)i * ld + k0;
            real s = real(0);
            for (int m = 0; m < nb; m++)
                if (i - (k0 + m) <= bw) s += row[m] * y[m];
            b[i] -= s;
        }
        __syncthreads();
    }
    // backward
    for (int k0 = ((n - 1) / kNB) * kNB; k0 >= 0; k0 -= kNB) {
        const int nb = min(kNB, n - k0);
        if (tid == 0) {
            for (int c = nb - 1; c >= 0; c--) {
                real s = b[k0 + c];
                for (int m = c + 1; m < nb; m++) s -= A[(int64_t)(k0 + m) * ld + k0 + c] * y[m];
                y[c] = s / A[(int64_t)(k0 + c) * ld + k0 + c];
            }
            for (int c = 0; c < nb; c++) b[k0 + c] = y[c];
        }
        __syncthreads();
        for (int i = max(0, k0 - bw) + tid; i < k0; i += 1024) {
            real s = real(0);
            for (int m = 0; m < nb; m++)
                if (k0 + m - i <= bw) s += A[(int64_t)(k0 + m) * ld + i] * y[m];
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
// failing pivot + 1, or 0; Linv: device workspace of 32 x 32 reals.
static inline int cholesky_solve(real *A, real *b, int n, int64_t ld, int bw, int *info, real *Linv,
                                 hipStream_t s)
{
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = n - k0 < kNB ? n - k0 : kNB;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(kNB, kNB), 0, s, A, ld, k0, nb, info, Linv);
        int rem = n - k0 - nb;
        if (rem > bw) rem = bw;   // rows further down hold no in-band element of this panel
        if (rem > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel, dim3((rem + kNB - 1) / kNB), dim3(kNB, kNB), 0, s, A, n, ld, bw, k0, nb, Linv);
            const int nt = (rem + kTile - 1) / kTile;
            hipLaunchKernelGGL(syrk_update_kernel, dim3(nt, nt), dim3(256), 0, s, A, n, ld, bw, k0, nb);
        }
    }
    hipLaunchKernelGGL(chol_trsv_kernel, dim3(1), dim3(1024), 0, s, A, n, ld, bw, b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Cholesky launch failed: %s", hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

}  // namespace vs
