// Symmetric positive-definite solve of the reduced camera system on the device: blocked
// right-looking Cholesky (lower, in place) + triangular solves, on dense or BAND storage.
//
// Storage: element (i, j), j <= i, i - j <= bw, lives at A[i * ld + j].  Dense: ld >= n, bw = n.
// Band (lower half-bandwidth bw): ld >= bw and A points ld elements into the array, so that row i
// occupies the bw + 1 slots ending at A[i * (ld + 1)]; Cholesky fill-in stays inside the band, and
// every kernel below touches in-band elements only.  The caller chooses ld as a multiple of 16
// elements so that the 16-element row segments of the update tiles are aligned 128-byte lines.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <mutex>
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

// ---------------------------------------------------------------------------------------------
// Single-workgroup factorisation and solve for systems whose trailing window is narrow (band) or
// small (few cameras).  Matrix products run on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64, lane l holds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15],
// the result registers hold D[row (l >> 4) + 4 v][col l & 15], v = 0..3.  A 32-deep product takes 8
// instructions; the order in which k is consumed is free as long as both operands agree, so lane
// (l15, lk) takes k = 8 lk + s in step s and fetches its 8 values as four 16-byte LDS reads.
constexpr int kFusedMaxRows = 448;   // widest trailing window (rows below the diagonal block)
constexpr int kLookaheadMaxRows = kFusedMaxRows / 2;   // two panel buffers fit in LDS
constexpr int kXS = 34;              // LDS row stride in doubles: 16-byte aligned rows, and the
                                     // 16 rows of one operand read start 4 banks apart
constexpr int kWin = 512;            // circular LDS window of the right-hand side (>= 448 + 32)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_frag(const real *row, int lk, real *f)
{
    const double2 *p = reinterpret_cast<const double2 *>(row + 8 * lk);
#pragma unroll
    for (int q = 0; q < 4; q++) { const double2 v = p[q]; f[2 * q] = v.x; f[2 * q + 1] = v.y; }
}

// 16 x 16 tile of  P Q^T  over k = 0..31 from two fragments (two accumulators, summed at the end)
__device__ __forceinline__ d4 frag_product(const real *a, const real *b)
{
    d4 e = { 0, 0, 0, 0 }, o = { 0, 0, 0, 0 };
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], e, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s + 1], b[s + 1], o, 0, 0, 0);
    }
    return e + o;
}

// Value of `v` in lane `src` (uniform), as a scalar broadcast.
__device__ __forceinline__ real lane_value(real v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1 / d to double precision: hardware estimate (2^-24) + one cubically convergent step.
__device__ __forceinline__ real rcp_cubic(real d)
{
    const real y = __builtin_amdgcn_rcp(d);
    const real e = fma(-d, y, real(1));
    return fma(y, fma(e, e, e), y);                     // y (1 + e + e^2)
}

constexpr int kHalf = kNB / 2;
constexpr int kSS = kHalf + 2;                          // row stride of the half-block images in LDS

// Cholesky factor of the 32 x 32 diagonal block AND the inverse of that factor, by ONE wave with
// the block in registers (no barriers): lane i < 32 holds row i of the block (identity beyond nb),
// lane 32 + r holds row r of the identity.  A right-looking step c reads the pivot from lane c, and
// every lane updates its row with the same instructions,
//   w[j] -= (w[c] / d) * w_j[c]      (w_j[c] = column c at row j, read from lane j),
// so the identity rows ride along exactly like panel rows below the block and finish as
// I * L^-T: lane 32 + r ends up holding row r of L^-T = column r of L^-1.
// The chain from one pivot to the next is a lane read, the reciprocal of the pivot, one product
// and one fma (a dependent fp64 operation costs ~44 cycles); a step first updates column c + 1 and
// starts the next pivot's reciprocal, then updates the other columns while that chain is in flight.
// The block is processed as two halves of 16 columns: the steps of a half only update the columns
// of that half, and the update of columns 16..31 by the first 16 steps (the bulk of the lane-read
// traffic in an unblocked sweep) is ONE 32 x 16 x 16 product on the fp64 matrix cores, staged
// through wave-private LDS images.  Columns are scaled at the end (L[i][c] = w[c] / sqrt(d_c)),
// with the 32 reciprocal square roots computed side by side in the lanes.  A non-positive pivot is
// replaced by 1 with selects; the first such column + 1 is returned (0 = none).
// Outputs (LDS): Ld[c][i] = L[i][c] (entries above the diagonal are meaningless), Li = L^-1.
// img: wave-private LDS scratch of 48 * kSS reals, 16-byte aligned.
__device__ __forceinline__ int potrf_inverse_wave(const real (*Wn)[kNB + 1], int lane, real *img,
                                                  real (*Ld)[kNB + 1], real (*Li)[34])
{
    const int row = lane & 31;
    const bool ident = lane >= 32;
    real w[kNB], mm[kHalf];
#pragma unroll
    for (int j = 0; j < kNB; j++)
        w[j] = ident ? (j == row ? real(1) : real(0)) : (j <= row ? Wn[row][j] : real(0));
    int bad = 0;
    real diag = real(1);                               // pivot of column `row` (block lanes)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int c0 = h * kHalf, c1 = c0 + kHalf;
        real d = lane_value(w[c0], c0);
        bool neg = !(d > real(0));
        bad = (neg && bad == 0) ? c0 + 1 : bad;
        d = neg ? real(1) : d;
        real rinv = rcp_cubic(d);
#pragma unroll
        for (int c = c0; c < c1; c++) {
            w[c] = (neg && lane == c) ? real(1) : w[c];
            diag = row == c ? d : diag;
            const real u = w[c];
            const real m = -(u * rinv);                // minus the unit-diagonal factor entry of this row
            if (h == 0) mm[c - c0] = m;
            if (c + 1 < c1) {
                w[c + 1] = fma(m, lane_value(u, c + 1), w[c + 1]);
                d = lane_value(w[c + 1], c + 1);
                neg = !(d > real(0));
                bad = (neg && bad == 0) ? c + 2 : bad;
                d = neg ? real(1) : d;
                rinv = rcp_cubic(d);
            }
#pragma unroll
            for (int j = c + 2; j < c1; j++) w[j] = fma(m, lane_value(u, j), w[j]);
        }
        if (h == 0) {
            // columns 16..31 += M U^T for block rows 16..31 (lanes 16..31) and identity rows 0..15
            // (lanes 32..47); the other lanes' columns 16..31 are not needed or do not change.
            // Images: M rows of lanes 16..47 at img[(lane - 16) * kSS], U = block rows 16..31,
            // columns 0..15 (unscaled), at img[(32 + lane - 16) * kSS].
            real *Mimg = img, *Uimg = img + 32 * kSS;
            if (lane >= 16 && lane < 48) {
                double2 *dst = reinterpret_cast<double2 *>(Mimg + (lane - 16) * kSS);
#pragma unroll
                for (int q = 0; q < kHalf / 2; q++) dst[q] = make_double2(mm[2 * q], mm[2 * q + 1]);
            }
            if (lane >= 16 && lane < 32) {
                double2 *dst = reinterpret_cast<double2 *>(Uimg + (lane - 16) * kSS);
#pragma unroll
                for (int q = 0; q < kHalf / 2; q++) dst[q] = make_double2(w[2 * q], w[2 * q + 1]);
            }
            const int l15 = lane & 15, lk = lane >> 4;
            real bfr[4], af[2][4];
            {
                const double2 *p = reinterpret_cast<const double2 *>(Uimg + l15 * kSS + 4 * lk);
                const double2 v0 = p[0], v1 = p[1];
                bfr[0] = v0.x; bfr[1] = v0.y; bfr[2] = v1.x; bfr[3] = v1.y;
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const double2 *p = reinterpret_cast<const double2 *>(Mimg + (16 * t + l15) * kSS + 4 * lk);
                const double2 v0 = p[0], v1 = p[1];
                af[t][0] = v0.x; af[t][1] = v0.y; af[t][2] = v1.x; af[t][3] = v1.y;
            }
            d4 acc[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
            for (int sft = 0; sft < 4; sft++)
#pragma unroll
                for (int t = 0; t < 2; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t][sft], bfr[sft], acc[t], 0, 0, 0);
            // D[row lk + 4 v][col l15] of tile t -> image rows 16 t + lk + 4 v (the M image is dead)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int v = 0; v < 4; v++) Mimg[(16 * t + lk + 4 * v) * kSS + l15] = acc[t][v];
            if (lane >= 16 && lane < 48) {
                const double2 *src = reinterpret_cast<const double2 *>(Mimg + (lane - 16) * kSS);
#pragma unroll
                for (int q = 0; q < kHalf / 2; q++) {
                    const double2 v = src[q];
                    w[kHalf + 2 * q] += v.x;
                    w[kHalf + 2 * q + 1] += v.y;
                }
            }
        }
    }
    // column scaling, all 32 reciprocal square roots at once (lane c and lane 32 + c hold pivot c)
    const real rs = rsqrt_refined(diag);
#pragma unroll
    for (int c = 0; c < kNB; c++) {
        real *dst = ident ? &Li[c][row] : &Ld[c][row];
        *dst = w[c] * lane_value(rs, c);
    }
    return bad;
}

// Window update  A[base + r][base + c] -= X[r] . X[c]  over the 16 x 16 tiles (ti, tj), ti >= tj >= t0,
// of the lower triangle of a window of `rem` rows, numbered t = 0, 1, ...; the calling wave takes
// tiles first, first + stride, ... two at a time, and fetches the old values of its next two
// tiles before it works on the current two (the read-modify-write of the window is bound by the
// latency of those loads, not by the products).  X: panel image in LDS (rows beyond rem zero).
__device__ __forceinline__ void update_lower_tiles(real *__restrict__ A, int64_t ld, const real (*X)[kXS],
                                                   int base, int rem, int t0, int first, int stride, int lane)
{
    const int l15 = lane & 15, lk = lane >> 4;
    const int nrt = (rem + 15) >> 4, m = nrt - t0;
    const int n_tiles = m > 0 ? m * (m + 1) / 2 : 0;
    const int ldi = (int)ld;                                  // row offsets inside a tile fit 32 bits
    // Index arithmetic is kept off the issue path: tiles are decoded incrementally (row a of the
    // triangle starts at tile a (a + 1) / 2), one 64-bit address per tile, and the loads are
    // unconditional (rows clamped into the window; the masks only guard the stores).
    struct Pair { int ti[2], tj[2]; real *p[2]; real old[2][4]; };
    int dec_a = 0, dec_tri = 0;
    auto fetch = [&](int t, Pair &pr) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int tq = min(t + q * stride, n_tiles - 1);      // a missing second tile repeats the last one
            while (tq >= dec_tri + dec_a + 1) { dec_tri += dec_a + 1; dec_a++; }
            pr.ti[q] = dec_a + t0;
            pr.tj[q] = tq - dec_tri + t0;
            const int r0 = 16 * pr.ti[q] + lk, r0c = min(r0, rem - 1);
            pr.p[q] = A + (int64_t)(base + r0c) * ld + base + 16 * pr.tj[q] + l15;
#pragma unroll
            for (int v = 0; v < 4; v++)                            // rows beyond the window read its last row
                pr.old[q][v] = pr.p[q][(min(r0 + 4 * v, rem - 1) - r0c) * ldi];
        }
    };
    Pair cur, nxt;
    if (first < n_tiles) fetch(first, cur);
    for (int t = first; t < n_tiles; t += 2 * stride) {
        const int tn = t + 2 * stride;
        if (tn < n_tiles) fetch(tn, nxt);
        real a[2][8], b[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            load_frag(&X[16 * cur.ti[q] + l15][0], lk, a[q]);
            load_frag(&X[16 * cur.tj[q] + l15][0], lk, b[q]);
        }
        d4 p[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
        for (int sft = 0; sft < 8; sft++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                p[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][sft], b[q][sft], p[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q == 1 && t + stride >= n_tiles) break;          // uniform: no second tile
            const int cc = 16 * cur.tj[q] + l15;
            const int r0 = 16 * cur.ti[q] + lk;
#pragma unroll
            for (int v = 0; v < 4; v++) {                          // for stored rows the clamps are no-ops
                const int r = r0 + 4 * v;
                if (r < rem && cc <= r) cur.p[q][4 * v * ldi] = cur.old[q][v] - p[q][v];
            }
        }
        cur = nxt;
    }
}

constexpr int kCholThreads = 512, kCholWaves = kCholThreads / 64;   // 8 waves: 256 VGPRs each

// Whole factorisation in one workgroup of 8 waves, for windows of up to kLookaheadMaxRows rows (two
// panel images fit in LDS).  Per 32-column panel k (diagonal block W_k, rem rows X_k below it,
// both resident in LDS):
//   A. wave 0 factors W_k and inverts the factor in registers (potrf_inverse_wave) while the other
//      waves apply panel k-1's update to the part of the trailing window that panel k does not
//      read (columns >= 32 of the window, global read-modify-write);
//   B. X_k <- X_k * L_kk^-T on the matrix cores, to LDS and to global memory;
//   C. panel k's update of the FIRST 32 columns of the window, LDS to LDS: rows 0..31 become
//      W_k+1, the rows below become X_k+1 (the rows that newly enter the window come from global
//      memory).  These columns never make a round trip through global memory.
// The trailing window lives in L2; the block inverses are kept for the triangular solves.
__global__ void __launch_bounds__(kCholThreads)
chol_window_kernel(real *__restrict__ A, int n, int64_t ld, int bw, int *__restrict__ info,
                   real *__restrict__ Linv, int buf_rows, int64_t batch_stride_A, int64_t batch_stride_Linv)
{
    // batched use (one matrix per workgroup): matrix blockIdx.x lives batch_stride_A reals further
    A += (int64_t)blockIdx.x * batch_stride_A;
    Linv += (int64_t)blockIdx.x * batch_stride_Linv;
    __shared__ real Wn[kNB][kNB + 1];                            // diagonal block of the next panel
    // Ld[c][i] = L[i][c] of the current block.  It lives in Wn's storage: the factoring wave has the
    // whole block in registers before it writes the factor, and the factor has gone to global
    // memory (one more barrier) before the next block is assembled there.  The 8 KB this saves
    // are what lets a dense system of 50 cameras keep both panel images in LDS.
    real (*Ld)[kNB + 1] = Wn;
    __shared__ __attribute__((aligned(16))) real wimg[48 * kSS];  // scratch of the factoring wave
    __shared__ __attribute__((aligned(16))) real Li[kNB][kXS];    // inverse of the block's factor
    __shared__ int bad_s;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    typedef real (*Panel)[kXS];
    Panel Xcur = reinterpret_cast<Panel>(dyn_lds);
    Panel Xoth = Xcur + buf_rows;                                 // second panel image
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    if (tid == 0) bad_s = 0;

    auto lower_tiles = [&](Panel X, int base, int rem, int t0, int first, int stride) {
        update_lower_tiles(A, ld, X, base, rem, t0, first, stride, lane);
    };
    auto panel_rows = [&](int k0) { return max(0, min(n - (k0 + min(kNB, n - k0)), bw)); };

    // panel 0: diagonal block and panel rows from global memory
    {
        const int nb = min(kNB, n), rem = panel_rows(0);
        for (int e = tid; e < kNB * kNB; e += kCholThreads) {
            const int bi = e / kNB, bj = e % kNB;
            Wn[bi][bj] = (bi < nb && bj <= bi) ? A[(int64_t)bi * ld + bj] : (bi == bj ? real(1) : real(0));
        }
        for (int e = tid; e < ((rem + 15) & ~15) * kNB; e += kCholThreads) {
            const int r = e / kNB, c = e % kNB;
            const int i = nb + r;
            Xcur[r][c] = (r < rem && c < nb && i - c <= bw) ? A[(int64_t)i * ld + c] : real(0);
        }
    }
    __syncthreads();
    int rem_prev = 0, base_prev = 0;
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = min(kNB, n - k0);
        const int base = k0 + nb;
        const int rem = panel_rows(k0);
        const int nrt = (rem + 15) >> 4;                          // 16-row tiles of the panel
        real *Lp = Linv + (int64_t)(k0 / kNB) * kNB * kNB;
#ifdef VS_CHOL_TIMING
        unsigned long long st0 = __builtin_readcyclecounter();
#endif
        // A. factor + invert the diagonal block (wave 0) | rest of the previous panel's update
        if (wave == 0) {
            const int bad = potrf_inverse_wave(Wn, lane, wimg, Ld, Li);
            if (lane == 0 && bad && bad_s == 0) bad_s = k0 + bad;
        }
        else if (k0 > 0) lower_tiles(Xoth, base_prev, rem_prev, 2, wave - 1, kCholWaves - 1);
#ifdef VS_CHOL_TIMING
        unsigned long long st1 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef VS_CHOL_TIMING
        unsigned long long st2 = __builtin_readcyclecounter();
#endif
        // the block's factor and its inverse go to global memory; then Wn (which held the factor)
        // is reset to the identity
        for (int e = tid; e < kNB * kNB; e += kCholThreads) {
            const int r = e / kNB, c = e % kNB;
            if (r >= c && r < nb) A[(int64_t)(k0 + r) * ld + k0 + c] = Ld[c][r];
            Lp[e] = Li[r][c];
        }
        if (rem <= 0) break;                                      // uniform: last panel
        __syncthreads();
        for (int e = tid; e < kNB * kNB; e += kCholThreads) Wn[e / kNB][e % kNB] = e / kNB == e % kNB ? real(1) : real(0);
        // Band storage: the rows that enter the window two panels from now are first touched here,
        // one cache line per thread, so that their first real use does not wait for HBM.  The value
        // is only consumed (by an empty asm) at the end of the panel.
        real warm = real(0);
        if (bw < n) {
            const int lines_per_row = ((bw + 1) * (int)sizeof(real) + 127) / 128 + 1;
            const int i = base + rem + kNB + tid / lines_per_row;
            if (tid < kNB * lines_per_row && i < n) {
                const real *q = A + (int64_t)i * ld + (i - bw) + (tid % lines_per_row) * 16;
                if (q <= A + (int64_t)i * ld + i) warm = *reinterpret_cast<const volatile real *>(q);
            }
        }
        // B. X <- X * Li^T
        {
            real b0[8], b1[8];
            load_frag(&Li[l15][0], lk, b0);
            load_frag(&Li[16 + l15][0], lk, b1);
            for (int ti = wave; ti < nrt; ti += kCholWaves) {
                real a[8];
                load_frag(&Xcur[16 * ti + l15][0], lk, a);
                const d4 p0 = frag_product(a, b0), p1 = frag_product(a, b1);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int r = 16 * ti + lk + 4 * v;
                    Xcur[r][l15] = p0[v];
                    Xcur[r][16 + l15] = p1[v];
                    const int i = base + r;
                    if (r < rem) {
                        if (l15 < nb && i - (k0 + l15) <= bw) A[(int64_t)i * ld + k0 + l15] = p0[v];
                        if (16 + l15 < nb && i - (k0 + 16 + l15) <= bw) A[(int64_t)i * ld + k0 + 16 + l15] = p1[v];
                    }
                }
            }
        }
        __syncthreads();
#ifdef VS_CHOL_TIMING
        unsigned long long st3 = __builtin_readcyclecounter();
#endif
        const int k0n = base, nbn = min(kNB, n - k0n);            // next panel (nbn > 0 as rem > 0)
        const int remn = panel_rows(k0n);
        {
            // C. first 32 columns of the window -> next diagonal block and next panel rows.
            //    Tile list: (0,0), (1,0), (1,1), (2,0), (2,1), ...
            const int n_first = nrt == 1 ? 1 : 2 * nrt - 1;
#pragma unroll 1
            for (int idx = wave; idx < n_first; idx += kCholWaves) {
                const int ti = idx == 0 ? 0 : (idx + 1) >> 1, tj = idx == 0 ? 0 : (idx + 1) & 1;
                const int cc = 16 * tj + l15;
                real old[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int r = 16 * ti + lk + 4 * v;
                    old[v] = (r < rem && cc <= r) ? A[(int64_t)(base + r) * ld + base + cc] : real(0);
                }
                real a[8], b[8];
                load_frag(&Xcur[16 * ti + l15][0], lk, a);
                load_frag(&Xcur[16 * tj + l15][0], lk, b);
                const d4 p = frag_product(a, b);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int r = 16 * ti + lk + 4 * v;
                    if (r >= rem) continue;
                    if (r < kNB) { if (cc <= r) Wn[r][cc] = old[v] - p[v]; }
                    else Xoth[r - kNB][cc] = cc < nbn ? old[v] - p[v] : real(0);
                }
            }
            // rows that enter the window (not touched by this panel's update) and zero padding
            const int r_lo = max(rem - kNB, 0), r_hi = (remn + 15) & ~15;
            for (int e = r_lo * kNB + tid; e < r_hi * kNB; e += kCholThreads) {
                const int r = e / kNB, c = e % kNB;
                const int i = k0n + nbn + r;
                Xoth[r][c] = (r < remn && c < nbn && i - (k0n + c) <= bw) ? A[(int64_t)i * ld + k0n + c] : real(0);
            }
        }
        asm volatile("" :: "v"(warm));
        __syncthreads();
#ifdef VS_CHOL_TIMING
        if (tid == 0 && k0 == 32 * 3)
            printf("panel rem %d: potrf wave %llu, phase A %llu, product %llu, next columns / window %llu\n", rem,
                   st1 - st0, st2 - st0, st3 - st2, (unsigned long long)__builtin_readcyclecounter() - st3);
#endif
        rem_prev = rem; base_prev = base;
        { Panel t = Xcur; Xcur = Xoth; Xoth = t; }
    }
    if (tid == 0 && bad_s && *info == 0) *info = bad_s;
}

// Single-workgroup factorisation for windows too wide for two panel buffers (8 waves, one panel
// image in LDS).  Per 32-column panel (its diagonal block already factored and inverted):
//   1. the factor block and its inverse go to global memory, the panel rows below the block to LDS
//      (zero outside the band and beyond the matrix);
//   2. X = A_panel * L_kk^-T  on the matrix cores (one 16-row tile per wave and turn);
//   3. the three 16 x 16 tiles of the window update that make up the NEXT diagonal block;
//   4. wave 0 reads that block back and factors + inverts it in registers (potrf_inverse_wave)
//      while the other seven waves apply the rest of the window update.
// No launches between panels; the window lives in L2.  The block inverses are kept for the
// triangular solves.
__global__ void __launch_bounds__(kCholThreads)
chol_fused_kernel(real *__restrict__ A, int n, int64_t ld, int bw, int *__restrict__ info,
                  real *__restrict__ Linv)
{
    __shared__ real Wn[kNB][kNB + 1];                            // diagonal block being factored
    __shared__ real Ld[kNB][kNB + 1];                            // Ld[c][i] = L[i][c]
    __shared__ __attribute__((aligned(16))) real wimg[48 * kSS];  // scratch of the factoring wave
    __shared__ __attribute__((aligned(16))) real Li[kNB][kXS];    // inverse of the block's factor
    __shared__ int bad_s;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    real (*X)[kXS] = reinterpret_cast<real (*)[kXS]>(dyn_lds);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    if (tid == 0) bad_s = 0;
    // wave 0: diagonal block at k0 (nb columns, identity beyond) from global memory, factor, inverse
    auto factor_block = [&](int k0, int nb) {
        // lane (bj = lane % 32, rows lane / 32 + 2 q): 16 unconditional loads in flight (rows and
        // columns clamped into the block), masked when stored
        const int bj = lane % kNB, r0 = lane / kNB;
        const real *col = A + (int64_t)k0 * ld + k0 + min(bj, nb - 1);
        real v[kNB / 2];
#pragma unroll
        for (int q = 0; q < kNB / 2; q++) v[q] = col[(int64_t)min(r0 + 2 * q, nb - 1) * ld];
#pragma unroll
        for (int q = 0; q < kNB / 2; q++) {
            const int bi = r0 + 2 * q;
            Wn[bi][bj] = (bi < nb && bj <= bi) ? v[q] : (bi == bj ? real(1) : real(0));
        }
        const int bad = potrf_inverse_wave(Wn, lane, wimg, Ld, Li);   // a wave's LDS operations are ordered
        if (lane == 0 && bad && bad_s == 0) bad_s = k0 + bad;
    };
    __syncthreads();
    if (wave == 0) factor_block(0, min(kNB, n));
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += kNB) {
        const int nb = min(kNB, n - k0);
        const int base = k0 + nb;
        const int rem = max(0, min(n - base, bw));
        const int nrt = (rem + 15) >> 4;                          // 16-row tiles of the panel
        real *Lp = Linv + (int64_t)(k0 / kNB) * kNB * kNB;
#ifdef VS_CHOL_TIMING
        unsigned long long st0 = __builtin_readcyclecounter();
#endif
        // 1. factor block and inverse out, panel rows in
        for (int e = tid; e < kNB * kNB; e += kCholThreads) {
            const int r = e / kNB, c = e % kNB;
            if (r >= c && r < nb) A[(int64_t)(k0 + r) * ld + k0 + c] = Ld[c][r];
            Lp[e] = Li[r][c];
        }
        if (rem <= 0) break;                                      // uniform: last panel
        {
            // thread (r0, c) reads column c of rows r0, r0 + 16, ...: four unconditional loads (rows
            // clamped into the panel) in flight, masked when stored
            const int c = tid % kNB, r0 = tid / kNB;
            const real *col = A + (int64_t)base * ld + k0 + min(c, nb - 1);
            for (int rb = 0; rb < nrt * 16; rb += 64) {
                real v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = col[(int64_t)min(rb + 16 * q + r0, rem - 1) * ld];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = rb + 16 * q + r0;
                    if (r < nrt * 16)
                        X[r][c] = (r < rem && c < nb && base + r - (k0 + c) <= bw) ? v[q] : real(0);
                }
            }
        }
        __syncthreads();
#ifdef VS_CHOL_TIMING
        unsigned long long st1 = __builtin_readcyclecounter();
#endif
        // 2. X <- X * Li^T
        {
            real b0[8], b1[8];
            load_frag(&Li[l15][0], lk, b0);
            load_frag(&Li[16 + l15][0], lk, b1);
            for (int ti = wave; ti < nrt; ti += kCholWaves) {
                real a[8];
                load_frag(&X[16 * ti + l15][0], lk, a);
                const d4 p0 = frag_product(a, b0), p1 = frag_product(a, b1);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int r = 16 * ti + lk + 4 * v;
                    X[r][l15] = p0[v];
                    X[r][16 + l15] = p1[v];
                    const int i = base + r;
                    if (r < rem) {
                        if (l15 < nb && i - (k0 + l15) <= bw) A[(int64_t)i * ld + k0 + l15] = p0[v];
                        if (16 + l15 < nb && i - (k0 + 16 + l15) <= bw) A[(int64_t)i * ld + k0 + 16 + l15] = p1[v];
                    }
                }
            }
        }
        __syncthreads();
#ifdef VS_CHOL_TIMING
        unsigned long long st2 = __builtin_readcyclecounter();
#endif
        // 3. tiles 0, 1, 2 = (0,0), (1,0), (1,1): the next diagonal block
        if (wave < 3) update_lower_tiles(A, ld, X, base, rem, 0, wave, 1 << 20, lane);
        __syncthreads();
#ifdef VS_CHOL_TIMING
        unsigned long long st3 = __builtin_readcyclecounter();
#endif
        // 4. next block's factorisation | rest of the window update
        if (wave == 0) factor_block(base, min(kNB, n - base));
        else if (wave != 4)                                       // wave 4 shares wave 0's SIMD: its MFMAs
            update_lower_tiles(A, ld, X, base, rem, 0, 3 + (wave < 4 ? wave - 1 : wave - 2), kCholWaves - 2, lane);   // would stretch the pivot chain
#ifdef VS_CHOL_TIMING
        unsigned long long st4 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef VS_CHOL_TIMING
        if ((tid == 0 || tid == 64) && k0 == 32 * 3)
            printf("fused panel rem %d wave %d: load %llu product %llu block tiles %llu own phase 4 %llu, phase 4 with barrier %llu\n", rem,
                   wave, st1 - st0, st2 - st1, st3 - st2, st4 - st3, (unsigned long long)__builtin_readcyclecounter() - st3);
#endif
    }
    if (tid == 0 && bad_s && *info == 0) *info = bad_s;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every
// outstanding global load and store of the wave, which puts the latency of loads fetched ahead of
// their use (and of fire-and-forget stores) on the critical path of each barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row.
template <int CTRL>
__device__ __forceinline__ real dpp_move(real v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ real sum32(real v)
{
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    v += __shfl_xor(v, 16, 64);  // the other row of this 32-lane half
    return v;
}

// Solve L y = b then L^T x = y in place for the narrow-window case (single workgroup, 1024
// threads = 32 x 32).  Linv holds the inverse of every 32 x 32 diagonal block of L, so a block
// solve is a 32 x 32 matrix-vector product: thread (r, m) multiplies one element and the 32 lanes
// of a row add up.  The active part of the right-hand side lives in a circular LDS window; block
// inverses and factor rows, which do not depend on the right-hand side, are fetched before the
// dependent work of a panel starts.
// Sum over groups of 8 consecutive lanes (result in every lane of the group): three DPP steps.
__device__ __forceinline__ real sum8(real v)
{
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    return v;
}

template <int QN>   // ceil(window / 128): turns of the forward update (128 rows per turn)
__global__ void __launch_bounds__(1024)
chol_trsv_window_kernel(const real *__restrict__ A, int n, int64_t ld, int bw,
                        const real *__restrict__ Linv, real *__restrict__ b)
{
    __shared__ real win[kWin];
    __shared__ __attribute__((aligned(16))) real ys[kNB];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int g = lane & 7, sub = lane >> 3;                       // forward pass: 8 lanes x 4 columns per row
    const int reach = kNB + min(bw, n);                            // rows a panel touches
    // ---- forward: L y = b.  A row of 32 factor or block-inverse entries is shared by 8 lanes, each
    // multiplying 4 consecutive columns in registers, so a row sum is three DPP steps.
    for (int i = tid; i < min(n, reach); i += 1024) win[i & (kWin - 1)] = b[i];
    // block inverse: waves 0..3, row 8 * wave + sub, columns 4 g .. 4 g + 3
    auto fetch_inverse = [&](int p0, real *out) {
        const int pc = min(p0, ((n - 1) / kNB) * kNB);
        const real *src = Linv + (int64_t)(pc / kNB) * kNB * kNB + (8 * (wave & 3) + sub) * kNB + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = src[e];
    };
    // Factor rows below the diagonal block of the panel at p0: turn q covers rows
    // 128 q + 8 wave + sub.  Loads are unconditional (addresses clamped into the factor) and their
    // values are masked only where they are used, one panel later: a fixed number of loads in
    // flight lets the waits for older loads leave the younger ones outstanding, and nothing at the
    // point of the fetch depends on the data.
    auto fetch_rows = [&](int p0, real (*out)[4]) {
        const int pc = min(p0, ((n - 1) / kNB) * kNB);             // clamp past-the-end panels
        const int nbp = min(kNB, n - pc), basep = pc + nbp;
        const int remp = p0 < n ? max(0, min(n - basep, bw)) : 0;
#pragma unroll
        for (int q = 0; q < QN; q++) {
            const int r = 128 * q + 8 * wave + sub;
            // columns are not clamped: the caller pads the factor storage by kNB elements
            const real *src = A + (int64_t)(remp > 0 ? basep + min(r, remp - 1) : basep - 1) * ld + pc + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; e++) out[q][e] = src[e];          // raw: masked where it is used
        }
    };
    // Two register sets take turns (the loop is unrolled by two panels), so a value fetched in one
    // panel is consumed in the next without a register copy at the loop edge, which would wait for it.
    auto forward_panel = [&](int k0, const real *li, const real (*raw)[4], real *li_n, real (*raw_n)[4]) {
        const int nb = min(kNB, n - k0);
        const int base = k0 + nb;
        const int rem = max(0, min(n - base, bw));
#ifdef VS_CHOL_TIMING
        unsigned long long tt0 = __builtin_readcyclecounter();
#endif
        // this panel's block inverse and factor rows were fetched one panel ahead (li, raw); fetch
        // the next panel's into the other register set now
        real av[QN][4];
#pragma unroll
        for (int q = 0; q < QN; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int r = 128 * q + 8 * wave + sub, c = 4 * g + e;
                av[q][e] = (r < rem && c < nb && base + r - (k0 + c) <= bw) ? raw[q][e] : real(0);
            }
        fetch_inverse(k0 + kNB, li_n);
        fetch_rows(k0 + kNB, raw_n);
        real incoming = real(0);
        const int irow = k0 + reach + tid;
        if (tid < kNB && irow < n) incoming = b[irow];
        // y = Linv * b_panel (waves 0..3)
        if (wave < 4) {
            real t0 = real(0), t1 = real(0);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int c0 = k0 + 4 * g + e;
                t0 = fma(li[e], c0 < n ? win[c0 & (kWin - 1)] : real(0), t0);
                t1 = fma(li[e + 1], c0 + 1 < n ? win[(c0 + 1) & (kWin - 1)] : real(0), t1);
            }
            const real yv = sum8(t0 + t1);
            const int r = 8 * wave + sub;
            if (g == 0) { ys[r] = yv; if (r < nb) b[k0 + r] = yv; }
        }
        lds_barrier();
#ifdef VS_CHOL_TIMING
        unsigned long long tt1 = __builtin_readcyclecounter();
#endif
        const double2 y01 = reinterpret_cast<const double2 *>(ys)[2 * g], y23 = reinterpret_cast<const double2 *>(ys)[2 * g + 1];
#pragma unroll
        for (int q = 0; q < QN; q++) {
            const real s = sum8(fma(av[q][0], y01.x, av[q][1] * y01.y) + fma(av[q][2], y23.x, av[q][3] * y23.y));
            const int r = 128 * q + 8 * wave + sub;
            if (g == 0 && r < rem) win[(base + r) & (kWin - 1)] -= s;
        }
        if (tid < kNB && irow < n) win[irow & (kWin - 1)] = incoming;
        lds_barrier();
#ifdef VS_CHOL_TIMING
        if (tid == 0 && k0 == 32 * 50) printf("trsv forward panel: stage 1 %llu, stage 2 %llu (rem %d)\n", tt1 - tt0, (unsigned long long)__builtin_readcyclecounter() - tt1, rem);
#endif
    };
    {
        real liA[4], rawA[QN][4], liB[4], rawB[QN][4];
        fetch_inverse(0, liA);
        fetch_rows(0, rawA);
        __syncthreads();
        for (int k0 = 0; k0 < n; k0 += 2 * kNB) {
            forward_panel(k0, liA, rawA, liB, rawB);
            if (k0 + kNB < n) forward_panel(k0 + kNB, liB, rawB, liA, rawA);
        }
    }
    // ---- backward: L^T x = y (one thread per element of the transposed block inverse, one thread
    // per updated row and column half)
    __syncthreads();                                   // the forward results in b are complete
    const int r32 = tid / kNB, m = tid % kNB;
    const int k_last = ((n - 1) / kNB) * kNB;
    for (int i = max(0, k_last - bw) + tid; i < n; i += 1024) win[i & (kWin - 1)] = b[i];
    real li = Linv[(int64_t)(k_last / kNB) * kNB * kNB + m * kNB + r32];   // transposed element
    __syncthreads();
    for (int k0 = k_last; k0 >= 0; k0 -= kNB) {
        const int nb = min(kNB, n - k0);
        const int lo = max(0, k0 - bw);
        // thread (row, column half): the two halves of the 1024 threads split the 32 columns
        const int i = lo + (tid & 511), ch = tid >> 9;
        const bool mine = i < k0;
        real av[kNB / 2];
#pragma unroll
        for (int q = 0; q < kNB / 2; q++) {
            const int c = 16 * ch + q;
            av[q] = (mine && c < nb && k0 + c - i <= bw) ? A[(int64_t)(k0 + c) * ld + i] : real(0);
        }
        const real bp = k0 + m < n ? win[(k0 + m) & (kWin - 1)] : real(0);
        const real xv = sum32(li * bp);                            // (Linv^T b_panel)[r32]
        if (m == 0) { ys[r32] = xv; if (r32 < nb) b[k0 + r32] = xv; }
        if (k0 > 0) li = Linv[(int64_t)(k0 / kNB - 1) * kNB * kNB + m * kNB + r32];
        real incoming = real(0);
        const int irow = k0 - kNB - bw + tid;                      // rows entering the next window
        const bool take = tid < kNB && irow >= 0 && irow < lo;
        if (take) incoming = b[irow];
        __syncthreads();
        real s = real(0);
#pragma unroll
        for (int q = 0; q < kNB / 2; q++) s = fma(av[q], ys[16 * ch + q], s);
        if (mine && ch == 0) win[i & (kWin - 1)] -= s;            // fixed order: low columns first
        __syncthreads();
        if (mine && ch == 1) win[i & (kWin - 1)] -= s;
        if (take) win[irow & (kWin - 1)] = incoming;
        __syncthreads();
    }
}

// Solve L y = b then L^T x = y in place for wide windows (single workgroup, 1024 threads); the
// right-hand side stays in global memory.
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

// Dynamic LDS beyond the default limit for the single-workgroup Cholesky kernels: a per-device
// function attribute, set once per GPU.
constexpr int kLdsPerWorkgroup = 160 * 1024;     // gfx950
// dynamic LDS the look-ahead kernel may take beside its static arrays (0 until the limit is raised)
static int g_window_max_dyn[64] = {};

static inline int chol_window_raise_lds_limit()
{
    static std::mutex lds_mutex;
    static bool lds_limit_raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("hipGetDevice failed"); return VS_ERR_HIP; }
    std::lock_guard<std::mutex> lock(lds_mutex);
    if (!lds_limit_raised[dev]) {
        const int max_lds = kFusedMaxRows * kXS * (int)sizeof(real);
        hipFuncAttributes fa;
        int window_dyn = max_lds;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(chol_window_kernel)) == hipSuccess &&
            (int)fa.sharedSizeBytes > 0 && (int)fa.sharedSizeBytes < kLdsPerWorkgroup)
            window_dyn = std::max(max_lds, kLdsPerWorkgroup - (int)fa.sharedSizeBytes - 256);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(chol_window_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, window_dyn) != hipSuccess) {
            (void)hipGetLastError();
            window_dyn = max_lds;
        }
        g_window_max_dyn[dev] = window_dyn;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(chol_window_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, window_dyn) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(chol_fused_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess) {
            set_error("cannot raise the dynamic LDS limit of the single-workgroup Cholesky kernels");
            return VS_ERR_HIP;
        }
        lds_limit_raised[dev] = true;
    }
    return VS_OK;
}

// Factor + solve.  A, ld, bw: storage as described at the top of this file, lower triangle
// filled; b: rhs, overwritten with the solution; info (device int, zeroed by the caller): first
// failing pivot + 1, or 0; Linv: device workspace of ceil(n / 32) * 32 * 32 reals.  The storage behind A
// must extend kNB elements past the last matrix element (the solves read whole 32-column groups).
static inline int cholesky_solve(real *A, real *b, int n, int64_t ld, int bw, int *info, real *Linv,
                                 hipStream_t s)
{
    const int window = bw < n ? bw : n;
    if (window <= kFusedMaxRows) {
        VS_TRY(chol_window_raise_lds_limit());
        const int buf_rows = (window + 15) & ~15;
        // Dense storage: the window shrinks by a panel per step, the first image holds n - 32 rows
        // and the second n - 64, so systems somewhat wider than kLookaheadMaxRows still keep both
        // images in LDS (300 unknowns = 50 cameras just fit beside the static arrays).
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int rows_a = (std::max(n - kNB, 16) + 15) & ~15, rows_b = (std::max(n - 2 * kNB, 16) + 15) & ~15;
        const size_t tight = (size_t)(rows_a + rows_b) * kXS * sizeof(real);
        if (window <= kLookaheadMaxRows)
            hipLaunchKernelGGL(chol_window_kernel, dim3(1), dim3(kCholThreads), (size_t)2 * buf_rows * kXS * sizeof(real), s,
                               A, n, ld, bw, info, Linv, buf_rows, (int64_t)0, (int64_t)0);
        else if (bw >= n && dev >= 0 && dev < 64 && tight <= (size_t)g_window_max_dyn[dev] && getenv("VS_BA_NO_WIDE_LOOKAHEAD") == nullptr)
            hipLaunchKernelGGL(chol_window_kernel, dim3(1), dim3(kCholThreads), tight, s,
                               A, n, ld, bw, info, Linv, rows_a, (int64_t)0, (int64_t)0);
        else
            hipLaunchKernelGGL(chol_fused_kernel, dim3(1), dim3(kCholThreads), (size_t)buf_rows * kXS * sizeof(real), s,
                               A, n, ld, bw, info, Linv);
        const int turns = (window + 127) / 128;
        if (turns <= 1) hipLaunchKernelGGL(chol_trsv_window_kernel<1>, dim3(1), dim3(1024), 0, s, A, n, ld, bw, Linv, b);
        else if (turns <= 2) hipLaunchKernelGGL(chol_trsv_window_kernel<2>, dim3(1), dim3(1024), 0, s, A, n, ld, bw, Linv, b);
        else if (turns <= 3) hipLaunchKernelGGL(chol_trsv_window_kernel<3>, dim3(1), dim3(1024), 0, s, A, n, ld, bw, Linv, b);
        else hipLaunchKernelGGL(chol_trsv_window_kernel<(kFusedMaxRows + 127) / 128>, dim3(1), dim3(1024), 0, s, A, n, ld, bw, Linv, b);
    } else {
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
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Cholesky launch failed: %s", hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

}  // namespace vs
