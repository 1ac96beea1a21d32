// Parallel exact solve of the block-banded reduced camera system: block cyclic reduction.
//
// With keyframes in time order the camera pairs of S stay within `gap` cameras of the diagonal.
// Grouping g >= gap consecutive cameras into one block of w = 6 g unknowns makes S block
// tridiagonal: diagonal blocks D_j and couplings A_{j+1,j}.  A Cholesky sweep down that chain is a
// sequence of 6 Nc dependent pivots; cyclic reduction eliminates every other block of the chain at
// once (they do not touch each other), which leaves a block tridiagonal system of half the length,
// and repeats: log2(Nb) levels, every level a batch of independent dense w x w problems.
// Block j leaves at level t(j) = number of trailing one bits of j; its neighbours then are
// l = j - 2^t and r = j + 2^t (when inside the chain).  Eliminating block c with D_c = L L^T:
//     G_l = L^-1 A_{c,l},  G_r = L^-1 A_{c,r},  z = L^-1 b_c
//     D_l -= G_l^T G_l,  D_r -= G_r^T G_r,  A_{r,l} = -G_r^T G_l,  b_l -= G_l^T z,  b_r -= G_r^T z
// and, on the way back,  x_c = L^-T (z - G_l x_l - G_r x_r).
// Every step is an exact Schur complement of a positive-definite matrix, so the result is the
// Cholesky solution up to rounding, like the reference's sparse normal Cholesky.
// Per level: batched single-workgroup Cholesky of the leaving blocks (chol_window_kernel), a
// blocked triangular solve with many right-hand sides, and the update products; three launches.
#pragma once
#include "dense_solve.h"

namespace vs {

constexpr int kBcrT = 32;          // tile edge of the triangular solve and of the update products
constexpr int kBcrMaxW = 192;      // widest block (the batched factorisation keeps two panel images in LDS)

// D (Nb x w x w, lower triangle), C (Nb x w x w: C[j] = A_{j+1,j}, rows of block j+1, columns of
// block j) and b (Nb x w) from the stored 6 x 6 blocks; D and C are zero beforehand.  Unknowns
// beyond 6 Nc (the last block's padding) get a unit diagonal and a zero right-hand side.
__global__ void bcr_assemble_kernel(int Nc, int nb_off, const int *__restrict__ brow,
                                    const int *__restrict__ bcol, const real *__restrict__ S_diag,
                                    const real *__restrict__ S_off, const real *__restrict__ rhs,
                                    int g, int w, int Nb, real *__restrict__ D, real *__restrict__ C,
                                    real *__restrict__ b)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = ((int64_t)Nc + nb_off) * 36;
    if (tid < ne) {
        const int64_t blk = tid / 36;
        const int e = (int)(tid % 36), p = e / 6, q = e % 6;
        int a, bb;
        real v;
        if (blk < Nc) { a = bb = (int)blk; v = S_diag[tid]; }
        else { a = brow[blk - Nc]; bb = bcol[blk - Nc]; v = S_off[tid - (int64_t)Nc * 36]; }
        // S_ab(p, q) with a <= b sits at row 6a+p, column 6b+q; the lower triangle holds its transpose
        int i = 6 * bb + q, j = 6 * a + p;
        if (a == bb && j > i) return;                         // upper half of a diagonal block
        const int R = i / w, Cb = j / w;
        if (R == Cb) D[((int64_t)R * w + (i - R * w)) * w + (j - Cb * w)] = v;
        else C[((int64_t)Cb * w + (i - R * w)) * w + (j - Cb * w)] = v;   // R == Cb + 1 by construction
        return;
    }
    const int64_t u = tid - ne;
    if (u < (int64_t)Nb * w) {
        const bool live = u < 6 * (int64_t)Nc;
        b[u] = live ? rhs[u] : real(0);
        if (!live) {
            const int R = (int)(u / w), ii = (int)(u % w);
            D[((int64_t)R * w + ii) * w + ii] = real(1);
        }
    }
}

// Triangular solves of one level: for every leaving block c (blockIdx.y) and every 32-column tile
// of its right-hand sides [A_{c,l} | A_{c,r} | b_c] (blockIdx.x): X = L^-1 RHS, blocked by 32 rows
// with the inverses of the factor's diagonal blocks.  256 threads, the tile's X (w x 32) in LDS.
__global__ void __launch_bounds__(256)
bcr_trsm_kernel(int level, int Nb, int w, const real *__restrict__ D, const real *__restrict__ Linv,
                const real *__restrict__ C, const real *__restrict__ b, real *__restrict__ Gl,
                real *__restrict__ Gr, real *__restrict__ z)
{
    __shared__ real X[kBcrMaxW][kBcrT + 1];
    __shared__ real Lt[kBcrT][kBcrT + 1];
    const int s = 1 << level;
    const int c = s - 1 + (int)blockIdx.y * 2 * s;
    const int l = c - s, r = c + s;
    const int nt = w / kBcrT, tile = blockIdx.x;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int64_t ww = (int64_t)w * w;
    int kind, col0;
    if (tile < nt) { kind = 0; col0 = tile * kBcrT; if (l < 0) return; }
    else if (tile < 2 * nt) { kind = 1; col0 = (tile - nt) * kBcrT; if (r >= Nb) return; }
    else { kind = 2; col0 = 0; }
    const real *Lc = D + c * ww;
    const real *Li = Linv + (int64_t)c * nt * kBcrT * kBcrT;
    // right-hand side tile
    for (int e = tid; e < w * kBcrT; e += 256) {
        const int i = e / kBcrT, j = e % kBcrT;
        real v;
        if (kind == 0) v = C[l * ww + (int64_t)i * w + col0 + j];           // A_{c,l}(i, k)
        else if (kind == 1) v = C[c * ww + (int64_t)(col0 + j) * w + i];    // A_{c,r}(i, k) = A_{r,c}(k, i)
        else v = j == 0 ? b[(int64_t)c * w + i] : real(0);
        X[i][j] = v;
    }
    __syncthreads();
    // right-looking sweep: solve panel q with its block inverse, then take its contribution out of
    // every row below.  The inverse of the next panel's diagonal block is fetched while this
    // panel's update runs.
    real lt_next[kBcrT * kBcrT / 256];
#pragma unroll
    for (int u = 0; u < kBcrT * kBcrT / 256; u++) lt_next[u] = Li[tid + u * 256];
    for (int q = 0; q < nt; q++) {
#pragma unroll
        for (int u = 0; u < kBcrT * kBcrT / 256; u++) { const int e = tid + u * 256; Lt[e / kBcrT][e % kBcrT] = lt_next[u]; }
        __syncthreads();
        if (q + 1 < nt) {
#pragma unroll
            for (int u = 0; u < kBcrT * kBcrT / 256; u++) lt_next[u] = Li[(int64_t)(q + 1) * kBcrT * kBcrT + tid + u * 256];
        }
        real out[4] = { 0, 0, 0, 0 };
#pragma unroll 8
        for (int k = 0; k < kBcrT; k++) {
            const real tv = X[q * kBcrT + k][tx];
#pragma unroll
            for (int a = 0; a < 4; a++) out[a] = fma(Lt[ty * 4 + a][k], tv, out[a]);   // zero above the diagonal
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; a++) X[q * kBcrT + ty * 4 + a][tx] = out[a];
        __syncthreads();
        // rows below the panel: 32 x 32 tiles of X, one wave each, a 4 x 4 patch per thread (the
        // factor's rows come from L2 as 16-byte loads, eight distinct rows per wave and load)
        {
            const int wv = tid >> 6, r4 = (tid & 63) >> 3, c4 = tid & 7;
            for (int rb = q + 1 + wv; rb < nt; rb += 4) {
                const int row0 = rb * kBcrT + r4 * 4, col = c4 * 4;
                real acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2++) acc[a][c2] = X[row0 + a][col + c2];
                const real *Lr = Lc + (int64_t)row0 * w + q * kBcrT;
#pragma unroll 4
                for (int k = 0; k < kBcrT; k += 2) {
                    double2 lv[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) lv[a] = *reinterpret_cast<const double2 *>(Lr + (int64_t)a * w + k);
                    real x0[4], x1[4];
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2++) { x0[c2] = X[q * kBcrT + k][col + c2]; x1[c2] = X[q * kBcrT + k + 1][col + c2]; }
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int c2 = 0; c2 < 4; c2++) acc[a][c2] = fma(-lv[a].y, x1[c2], fma(-lv[a].x, x0[c2], acc[a][c2]));
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2++) X[row0 + a][col + c2] = acc[a][c2];
            }
        }
        __syncthreads();
    }
    if (kind == 2) {
        for (int i = tid; i < w; i += 256) z[(int64_t)c * w + i] = X[i][0];
    } else {
        real *G = (kind == 0 ? Gl : Gr) + c * ww;
        for (int e = tid; e < w * kBcrT; e += 256) G[(int64_t)(e / kBcrT) * w + col0 + e % kBcrT] = X[e / kBcrT][e % kBcrT];
    }
}

constexpr int kBcrKC = 16;          // rows of P and Q staged per step of a tile product

// 32 x 32 tile of  P0^T Q0 (+ P1^T Q1)  (P, Q: w x w row-major, the sums run over their rows) by ONE
// wave: 16 rows of both operands at a time pass through the wave's own LDS area, every lane adds a
// 4 x 4 patch (rows r4*4.., columns c4*4.. of the tile): four 16-byte LDS reads per 16 FMAs.  The
// next 16 rows are fetched into registers while the current ones are multiplied (the products of a
// slab take less time than its trip to L2, and a level of the reduction is a few such tiles per
// wave: without the prefetch the kernel is a chain of exposed memory latencies).  Only this wave
// touches Ps / Qs, and a wave's LDS operations execute in order: no workgroup barrier.
__device__ __forceinline__ void bcr_tile_ptq(const real *__restrict__ P0, const real *__restrict__ Q0,
                                             const real *__restrict__ P1, const real *__restrict__ Q1, int w,
                                             int i0, int j0, real (*Ps)[kBcrT], real (*Qs)[kBcrT],
                                             int lane, real (*acc)[4])
{
    const int r4 = lane >> 3, c4 = lane & 7;
    constexpr int kPer = kBcrKC * kBcrT / kWave;            // elements of P (and of Q) a lane moves per slab
    const int slabs_one = w / kBcrKC, n_slabs = P1 ? 2 * slabs_one : slabs_one;
    real pn[kPer], qn[kPer];
    auto fetch = [&](int slab) {
        const bool second = slab >= slabs_one;
        const real *P = second ? P1 : P0, *Q = second ? Q1 : Q0;
        const int k0 = (second ? slab - slabs_one : slab) * kBcrKC;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = lane + q * kWave, k = e / kBcrT, m = e % kBcrT;
            pn[q] = P[(int64_t)(k0 + k) * w + i0 + m];
            qn[q] = Q[(int64_t)(k0 + k) * w + j0 + m];
        }
    };
    fetch(0);
    for (int slab = 0; slab < n_slabs; slab++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = lane + q * kWave, k = e / kBcrT, m = e % kBcrT;
            Ps[k][m] = pn[q];
            Qs[k][m] = qn[q];
        }
        __builtin_amdgcn_wave_barrier();
        if (slab + 1 < n_slabs) fetch(slab + 1);
#pragma unroll 4
        for (int k = 0; k < kBcrKC; k++) {
            const double2 p01 = *reinterpret_cast<const double2 *>(&Ps[k][r4 * 4]);
            const double2 p23 = *reinterpret_cast<const double2 *>(&Ps[k][r4 * 4 + 2]);
            const double2 q01 = *reinterpret_cast<const double2 *>(&Qs[k][c4 * 4]);
            const double2 q23 = *reinterpret_cast<const double2 *>(&Qs[k][c4 * 4 + 2]);
            const real pv[4] = { p01.x, p01.y, p23.x, p23.y }, qv[4] = { q01.x, q01.y, q23.x, q23.y };
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) acc[a][c2] = fma(pv[a], qv[c2], acc[a][c2]);
        }
    }
}

// Updates of one level, four 32 x 32 output tiles per workgroup (one per wave).
// blockIdx.y < n_rem: staying block t = 2s - 1 + 2s * blockIdx.y loses the Schur terms of the
// blocks leaving on its left (c = t - s, through G_r) and on its right (c = t + s, through G_l):
// lower tiles of D_t, and b_t (the workgroup with blockIdx.x == gridDim.x - 1).
// blockIdx.y >= n_rem: leaving block c with both neighbours creates the coupling A_{r,l} =
// -G_r^T G_l (nt * nt tiles).  Every output tile has one writer.
__global__ void __launch_bounds__(256)
bcr_update_kernel(int level, int Nb, int w, int n_rem, const real *__restrict__ Gl,
                  const real *__restrict__ Gr, const real *__restrict__ z, real *__restrict__ D,
                  real *__restrict__ C, real *__restrict__ b)
{
    __shared__ __attribute__((aligned(16))) real Ps[4][kBcrKC][kBcrT];
    __shared__ __attribute__((aligned(16))) real Qs[4][kBcrKC][kBcrT];
    const int s = 1 << level, nt = w / kBcrT, n_lower = nt * (nt + 1) / 2;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int lane = tid & 63, wv = tid >> 6;
    const int64_t ww = (int64_t)w * w;
    const int item = blockIdx.y;
    if (blockIdx.x == gridDim.x - 1) {
        if (item >= n_rem) return;
        // right-hand side: b_t -= G_r(cL)^T z(cL) + G_l(cR)^T z(cR)
        // thread (i mod 32, part): column i of the block, every eighth row
        const int t = 2 * s - 1 + item * 2 * s;
        const int cL = t - s, cR = t + s;
        real *red = &Ps[0][0][0];                   // 8 x 33 partial sums
        for (int i0 = 0; i0 < w; i0 += kBcrT) {
            real sum = real(0);
            const real *G = Gr + cL * ww, *zz = z + (int64_t)cL * w;
            for (int k = ty; k < w; k += 8) sum = fma(G[(int64_t)k * w + i0 + tx], zz[k], sum);
            if (cR < Nb) {
                G = Gl + cR * ww; zz = z + (int64_t)cR * w;
                for (int k = ty; k < w; k += 8) sum = fma(G[(int64_t)k * w + i0 + tx], zz[k], sum);
            }
            red[ty * (kBcrT + 1) + tx] = sum;
            __syncthreads();
            if (ty == 0) {
                real tot = real(0);
#pragma unroll
                for (int part = 0; part < 8; part++) tot += red[part * (kBcrT + 1) + tx];
                b[(int64_t)t * w + i0 + tx] -= tot;
            }
            __syncthreads();
        }
        return;
    }
    const int tile = (int)blockIdx.x * 4 + wv;
    const int r4 = lane >> 3, c4 = lane & 7;
    real acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) acc[a][c2] = real(0);
    if (item < n_rem) {
        const int t = 2 * s - 1 + item * 2 * s;
        const int cL = t - s, cR = t + s;
        if (tile >= n_lower) return;
        int ti = 0, rem = tile;
        while (rem > ti) { rem -= ti + 1; ti++; }   // tile -> (ti, tj), tj <= ti
        const int tj = rem;
        const real *G1 = cR < Nb ? Gl + cR * ww : nullptr;
        bcr_tile_ptq(Gr + cL * ww, Gr + cL * ww, G1, G1, w, ti * kBcrT, tj * kBcrT, Ps[wv], Qs[wv], lane, acc);
        real *out = D + t * ww + (int64_t)(ti * kBcrT + r4 * 4) * w + tj * kBcrT + c4 * 4;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) out[(int64_t)a * w + c2] -= acc[a][c2];
    } else {
        const int c = s - 1 + (item - n_rem) * 2 * s;
        const int l = c - s, r = c + s;
        if (l < 0 || r >= Nb || c >= Nb || tile >= nt * nt) return;
        const int ti = tile / nt, tj = tile % nt;   // rows of r, columns of l
        bcr_tile_ptq(Gr + c * ww, Gl + c * ww, nullptr, nullptr, w, ti * kBcrT, tj * kBcrT, Ps[wv], Qs[wv], lane, acc);
        real *out = C + l * ww + (int64_t)(ti * kBcrT + r4 * 4) * w + tj * kBcrT + c4 * 4;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) out[(int64_t)a * w + c2] = -acc[a][c2];
    }
}

// Way back, one level: x_c = L^-T (z_c - G_l x_l - G_r x_r) for every block c that left at this
// level (one workgroup each).
__global__ void __launch_bounds__(256)
bcr_backsub_kernel(int level, int Nb, int w, const real *__restrict__ D, const real *__restrict__ Linv,
                   const real *__restrict__ Gl, const real *__restrict__ Gr, const real *__restrict__ z,
                   real *__restrict__ x)
{
    __shared__ real v[kBcrMaxW], xs[kBcrMaxW], xl[kBcrMaxW], xr[kBcrMaxW];
    __shared__ real part_s[8][kBcrT], u_s[kBcrT];
    const int s = 1 << level;
    const int c = s - 1 + (int)blockIdx.x * 2 * s;
    if (c >= Nb) return;
    const int l = c - s, r = c + s, nt = w / kBcrT;
    const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    const int64_t ww = (int64_t)w * w;
    for (int i = tid; i < w; i += 256) {
        xl[i] = l >= 0 ? x[(int64_t)l * w + i] : real(0);
        xr[i] = r < Nb ? x[(int64_t)r * w + i] : real(0);
    }
    __syncthreads();
    for (int row0 = grp * 4; row0 < w; row0 += 32) {       // four rows per group and step: independent loads
        real sum[4] = { 0, 0, 0, 0 };
        if (l >= 0) {
            const real *G = Gl + c * ww + (int64_t)row0 * w;
            for (int k = lane; k < w; k += 32) {
                const real xv = xl[k];
#pragma unroll
                for (int a = 0; a < 4; a++) sum[a] = fma(G[(int64_t)a * w + k], xv, sum[a]);
            }
        }
        if (r < Nb) {
            const real *G = Gr + c * ww + (int64_t)row0 * w;
            for (int k = lane; k < w; k += 32) {
                const real xv = xr[k];
#pragma unroll
                for (int a = 0; a < 4; a++) sum[a] = fma(G[(int64_t)a * w + k], xv, sum[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) sum[a] += __shfl_xor(sum[a], off, 32);
            if (lane == 0) v[row0 + a] = z[(int64_t)c * w + row0 + a] - sum[a];
        }
    }
    __syncthreads();
    const real *Lc = D + c * ww;
    // Panels from the last to the first:  u_i = v[32p+i] - sum_{rows below the panel} L[row][32p+i] x[row],
    // then x_p = Linv_p^T u.  Thread (i, part) of 32 x 8 adds every eighth row / four of the 32
    // terms; the factor entries and block-inverse entries a panel needs do not depend on the
    // solution, so those of panel p - 1 are fetched while panel p is being solved (a panel is
    // otherwise two exposed trips to L2 and a chain of 32 dependent fmas).
    constexpr int kMaxT = (kBcrMaxW - kBcrT) / 8;
    const int i = tid & 31, part = tid >> 5;
    real lrow[kMaxT], lrow_next[kMaxT], li[4], li_next[4];
    auto fetch_panel = [&](int p, real *lr, real *lq) {
#pragma unroll
        for (int t = 0; t < kMaxT; t++) {
            const int row = (p + 1) * kBcrT + part + 8 * t;
            lr[t] = row < w ? Lc[(int64_t)row * w + p * kBcrT + i] : real(0);
        }
        const real *Li = Linv + ((int64_t)c * nt + p) * kBcrT * kBcrT;
#pragma unroll
        for (int m = 0; m < 4; m++) lq[m] = Li[(part * 4 + m) * kBcrT + i];     // (L^-1)^T: column i of the inverse
    };
    fetch_panel(nt - 1, lrow, li);
    for (int p = nt - 1; p >= 0; p--) {
        if (p > 0) fetch_panel(p - 1, lrow_next, li_next);
        real sum = real(0);
#pragma unroll
        for (int t = 0; t < kMaxT; t++) {
            const int row = (p + 1) * kBcrT + part + 8 * t;
            if (row < w) sum = fma(lrow[t], xs[row], sum);
        }
        part_s[part][i] = sum;
        __syncthreads();
        if (tid < kBcrT) {
            real u = v[p * kBcrT + tid];
#pragma unroll
            for (int q = 0; q < 8; q++) u -= part_s[q][tid];
            u_s[tid] = u;
        }
        __syncthreads();
        real ps = real(0);
#pragma unroll
        for (int m = 0; m < 4; m++) ps = fma(li[m], u_s[part * 4 + m], ps);
        part_s[part][i] = ps;
        __syncthreads();
        if (tid < kBcrT) {
            real tot = part_s[0][tid];
#pragma unroll
            for (int q = 1; q < 8; q++) tot += part_s[q][tid];
            xs[p * kBcrT + tid] = tot;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kMaxT; t++) lrow[t] = lrow_next[t];
#pragma unroll
        for (int m = 0; m < 4; m++) li[m] = li_next[m];
    }
    for (int i = tid; i < w; i += 256) x[(int64_t)c * w + i] = xs[i];
}

struct BcrPlan {
    int g = 0, w = 0, Nb = 0, levels = 0;
};

// Block size for a system of Nc cameras whose pairs are at most `gap` apart, or g = 0 when cyclic
// reduction does not apply (blocks wider than the batched factorisation takes, or a chain too
// short to be worth it).
static inline BcrPlan bcr_plan(int Nc, int gap)
{
    BcrPlan p;
    const int g = ((gap < 1 ? 1 : gap) + 15) / 16 * 16;      // 6 g is a multiple of 32
    if (6 * g > kBcrMaxW) return p;
    const int Nb = (Nc + g - 1) / g;
    if (Nb < 4) return p;
    p.g = g; p.w = 6 * g; p.Nb = Nb;
    p.levels = 0;
    while (((1 << p.levels) - 1) < Nb) p.levels++;           // blocks with t trailing ones exist for t < levels
    return p;
}

// Device workspace of the solver in reals: D, C, Gl, Gr (Nb w^2 each), b, z, x (Nb w each),
// block inverses (Nb w 32).
static inline size_t bcr_workspace_reals(const BcrPlan &p)
{
    const size_t ww = (size_t)p.w * p.w;
    return 4 * p.Nb * ww + 3 * (size_t)p.Nb * p.w + (size_t)p.Nb * p.w * kBcrT;
}

// Solves S x = rhs.  S as stored 6 x 6 blocks (diagonal blocks already damped); x: 6 Nc reals.
// info (device int, zeroed by the caller) receives a non-zero value when a block is not positive
// definite.
static inline int bcr_solve(const BcrPlan &p, real *ws, int Nc, int nb_off, const int *brow, const int *bcol,
                            const real *S_diag, const real *S_off, const real *rhs, real *x_out, int *info,
                            hipStream_t s)
{
    const size_t ww = (size_t)p.w * p.w, nbw = (size_t)p.Nb * p.w;
    real *D = ws, *C = D + p.Nb * ww, *Gl = C + p.Nb * ww, *Gr = Gl + p.Nb * ww;
    real *b = Gr + p.Nb * ww, *z = b + nbw, *x = z + nbw, *Linv = x + nbw;
    const int nt = p.w / kBcrT;
    if (hipMemsetAsync(D, 0, 2 * p.Nb * ww * sizeof(real), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return VS_ERR_HIP; }
    const int64_t n_asm = ((int64_t)Nc + nb_off) * 36 + (int64_t)nbw;
    hipLaunchKernelGGL(bcr_assemble_kernel, dim3(div_up(n_asm, 256)), dim3(256), 0, s, Nc, nb_off, brow, bcol, S_diag,
                       S_off, rhs, p.g, p.w, p.Nb, D, C, b);
    VS_TRY(chol_window_raise_lds_limit());
    const int buf_rows = (p.w + 15) & ~15;
    for (int level = 0; level < p.levels; level++) {
        const int st = 1 << level;
        const int n_elim = (p.Nb - (st - 1) + 2 * st - 1) / (2 * st);           // c = st - 1 + 2 st k < Nb
        const int n_rem = p.Nb >= 2 * st ? (p.Nb - (2 * st - 1) + 2 * st - 1) / (2 * st) : 0;
        if (n_elim <= 0) break;
        // factor the leaving blocks: block c = st - 1 + 2 st k is batch entry k
        hipLaunchKernelGGL(chol_window_kernel, dim3(n_elim), dim3(kCholThreads), (size_t)2 * buf_rows * kXS * sizeof(real), s,
                           D + (size_t)(st - 1) * ww, p.w, (int64_t)p.w, p.w, info, Linv + (size_t)(st - 1) * nt * kBcrT * kBcrT,
                           buf_rows, (int64_t)(2 * st) * (int64_t)ww, (int64_t)(2 * st) * nt * kBcrT * kBcrT);
        hipLaunchKernelGGL(bcr_trsm_kernel, dim3(2 * nt + 1, n_elim), dim3(256), 0, s, level, p.Nb, p.w, D, Linv, C, b, Gl, Gr, z);
        if (n_rem > 0)
            hipLaunchKernelGGL(bcr_update_kernel, dim3(div_up(std::max(nt * nt, nt * (nt + 1) / 2), 4) + 1, n_rem + n_elim), dim3(256), 0, s,
                               level, p.Nb, p.w, n_rem, Gl, Gr, z, D, C, b);
    }
    for (int level = p.levels - 1; level >= 0; level--) {
        const int st = 1 << level;
        const int n_elim = (p.Nb - (st - 1) + 2 * st - 1) / (2 * st);
        if (n_elim <= 0) continue;
        hipLaunchKernelGGL(bcr_backsub_kernel, dim3(n_elim), dim3(256), 0, s, level, p.Nb, p.w, D, Linv, Gl, Gr, z, x);
    }
    if (hipMemcpyAsync(x_out, x, 6 * (size_t)Nc * sizeof(real), hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("hipMemcpyAsync failed"); return VS_ERR_HIP; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("cyclic reduction launch failed: %s", hipGetErrorString(e)); return VS_ERR_HIP; }
    return VS_OK;
}

}  // namespace vs
