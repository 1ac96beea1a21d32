// Micro-benchmark and record of an experiment: the reduced camera system of a small window
// factorised and solved by ONE workgroup (L D L^T, elements in registers, one barrier per column).
// Measured on MI355X: 1.05 us per column whatever the size (3.0 us set-up + 50.4 us factorisation +
// 8.4 us substitution at 60 unknowns; 103 us at 96), against 27 us for the panel kernels of
// dense_solve.h at 60 unknowns -- a single wave per SIMD pays every instruction's full latency, so
// the column-to-column chain costs ~2000 cycles, not the ~400 of its LDS round trip and barrier.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ivisual_slam_amd/csrc tools/micro/small_solve_time.hip -o tools/micro/small_solve_time.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "ba_kernels.h"

// The experiment: right-looking L D L^T of the packed lower triangle in ONE workgroup.  Thread
// (ty, tx) of 16 x 16 keeps the elements (ty + 16 a, tx + 16 b) in registers; per column the threads
// read the column from a double-buffered LDS array in one batch, update their elements without
// branches and the owners of the next column publish it: one barrier and one LDS round trip per
// column; the reciprocal of the next pivot is computed one step ahead by every thread.  The
// right-hand side is row n of the bordered matrix; back-substitution by one wave.
namespace vs {
constexpr int kSsN = 96, kSsThreads = 256, kSsRows = kSsN / 16 + 1, kSsCols = kSsN / 16;
__device__ __forceinline__ int tri_index(int i, int k) { return ((i * (i + 1)) >> 1) + k; }
__device__ unsigned long long g_ss_stamps[16];
#define VS_SS_STAMP(k) do { if (threadIdx.x == 0) { g_ss_stamps[2 * (k)] = clock64(); g_ss_stamps[2 * (k) + 1] = wall_clock64(); } } while (0)

__global__ void __launch_bounds__(kSsThreads)
ldlt_one_workgroup_kernel(int Nc, int nb_off, const int *__restrict__ brow, const int *__restrict__ bcol,
                          const real *__restrict__ S_diag, const real *__restrict__ S_off, const real *__restrict__ rhs,
                          real *__restrict__ x, int *__restrict__ info)
{
    __shared__ real A[(kSsN + 1) * (kSsN + 2) / 2];
    __shared__ real colbuf[2][16 * kSsRows], invs[kSsN], dnext[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = 6 * Nc;
    VS_SS_STAMP(0);
    for (int i = tid; i < (n + 1) * (n + 2) / 2; i += kSsThreads) A[i] = real(0);
    for (int i = tid; i < 2 * 16 * kSsRows; i += kSsThreads) (&colbuf[0][0])[i] = real(0);
    __syncthreads();
    for (int e = tid; e < Nc * 36; e += kSsThreads) {
        const int c = e / 36, p = (e % 36) / 6, q = e % 6;
        if (q > p) continue;
        if (p == q) A[tri_index(n, 6 * c + p)] = rhs[6 * c + p];
        A[tri_index(6 * c + p, 6 * c + q)] = S_diag[e];
    }
    for (int e = tid; e < nb_off * 36; e += kSsThreads) {
        const int blk = e / 36, p = (e % 36) / 6, q = e % 6;
        A[tri_index(6 * bcol[blk] + q, 6 * brow[blk] + p)] = S_off[e];
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    real el[kSsRows][kSsCols];
#pragma unroll
    for (int a = 0; a < kSsRows; a++)
#pragma unroll
        for (int b = 0; b < kSsCols; b++) {
            const int i = ty + 16 * a, k = tx + 16 * b;
            el[a][b] = (b <= a && i <= n && k <= i) ? A[tri_index(i, k)] : real(0);
        }
    if (tid <= n) colbuf[0][tid] = A[tri_index(tid, 0)];
    if (tid == 0) dnext[0] = n > 1 ? A[tri_index(1, 1)] : real(1);
    __syncthreads();
    const int n_row_tiles = (n + 16) >> 4;
    VS_SS_STAMP(1);
    int bad = 0;
    real inv_j;
    {
        real d0 = colbuf[0][0];
        if (!(d0 > real(0))) { bad = 1; d0 = real(1); }
        inv_j = fast_rcp(d0);
    }
    for (int j = 0; j < n; j++) {
        const int cur = j & 1, nxt = cur ^ 1, base = j + 1;
        real inv_next = real(0);
        if (base < n) {
            const real c1 = colbuf[cur][base];
            real pn = fma(-(c1 * inv_j), c1, dnext[cur]);
            if (!(pn > real(0))) { if (bad == 0) bad = base + 1; pn = real(1); }
            inv_next = fast_rcp(pn);
        }
        if (tid == 0) invs[j] = inv_j;
        const int t0 = base >> 4;
        real fr[kSsRows], cc[kSsCols];
#pragma unroll
        for (int a = 0; a < kSsRows; a++) fr[a] = colbuf[cur][ty + 16 * a];
#pragma unroll
        for (int b = 0; b < kSsCols; b++) cc[b] = colbuf[cur][tx + 16 * b];
#pragma unroll
        for (int a = 0; a < kSsRows; a++) fr[a] = ty + 16 * a >= base ? fr[a] * inv_j : real(0);
#pragma unroll
        for (int b = 0; b < kSsCols; b++) cc[b] = tx + 16 * b >= base ? cc[b] : real(0);
#pragma unroll
        for (int a = 0; a < kSsRows; a++) {
            if (a < t0 || a >= n_row_tiles) continue;
#pragma unroll
            for (int b = 0; b < kSsCols; b++)
                if (b <= a) el[a][b] = fma(-fr[a], cc[b], el[a][b]);
        }
        const int pub_tile = base >> 4, dg_tile = (base + 1) >> 4;
        const bool pub = tx == (base & 15), dgo = ty == ((base + 1) & 15) && tx == ty;
#pragma unroll
        for (int b = 0; b < kSsCols; b++) {
            if (b != pub_tile) continue;
#pragma unroll
            for (int a = 0; a < kSsRows; a++)
                if (a >= b && pub && ty + 16 * a >= base) colbuf[nxt][ty + 16 * a] = el[a][b];
        }
#pragma unroll
        for (int b = 0; b < kSsCols; b++)
            if (b == dg_tile && dgo) dnext[nxt] = el[b][b];
        __syncthreads();
        inv_j = inv_next;
    }
    VS_SS_STAMP(2);
    if (tid == 0) *info = bad;
#pragma unroll
    for (int a = 0; a < kSsRows; a++)
#pragma unroll
        for (int b = 0; b < kSsCols; b++) {
            const int i = ty + 16 * a, k = tx + 16 * b;
            if (b <= a && i <= n && k <= i) A[tri_index(i, k)] = el[a][b];
        }
    __syncthreads();
    if (wave != 0) return;
    real s0 = lane < n ? A[tri_index(n, lane)] : real(0), s1 = lane + 64 < n ? A[tri_index(n, lane + 64)] : real(0);
    for (int j = n - 1; j >= 0; j--) {
        const real src = j < 64 ? s0 : s1;
        const int lo = __builtin_amdgcn_readlane(__double2loint(src), j & 63);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(src), j & 63);
        const real xj = invs[j] * __hiloint2double(hi, lo);
        if (lane == 0) x[j] = xj;
        if (lane < j) s0 = fma(-A[tri_index(j, lane)], xj, s0);
        if (lane + 64 < j) s1 = fma(-A[tri_index(j, lane + 64)], xj, s1);
    }
    VS_SS_STAMP(3);
}
}  // namespace vs
using namespace vs;

int main(int argc, char **argv)
{
    const int Nc = argc > 1 ? atoi(argv[1]) : 10, n = 6 * Nc;
    std::vector<double> M((size_t)n * n), S((size_t)n * n, 0.0), rhs(n);
    srand(1);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; i++) {
        rhs[i] = rand() / (double)RAND_MAX;
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += M[(size_t)i * n + k] * M[(size_t)j * n + k];
            S[(size_t)i * n + j] = s + (i == j ? 1.0 : 0.0);
        }
    }
    const int nb = Nc * (Nc - 1) / 2;
    std::vector<double> Sd((size_t)Nc * 36), So((size_t)nb * 36);
    std::vector<int> br(nb), bc(nb);
    for (int c = 0; c < Nc; c++)
        for (int p = 0; p < 6; p++)
            for (int q = 0; q < 6; q++) Sd[(size_t)c * 36 + 6 * p + q] = S[(size_t)(6 * c + p) * n + 6 * c + q];
    int k = 0;
    for (int a = 0; a < Nc; a++)
        for (int b = a + 1; b < Nc; b++, k++) {
            br[k] = a; bc[k] = b;
            for (int p = 0; p < 6; p++)
                for (int q = 0; q < 6; q++) So[(size_t)k * 36 + 6 * p + q] = S[(size_t)(6 * a + p) * n + 6 * b + q];
        }
    double *dSd, *dSo, *drhs, *dx, *dsave; int *dbr, *dbc, *dinfo;
    hipMalloc(&dSd, Sd.size() * 8); hipMalloc(&dsave, Sd.size() * 8); hipMalloc(&dSo, So.size() * 8); hipMalloc(&drhs, n * 8); hipMalloc(&dx, n * 8);
    hipMalloc(&dbr, nb * 4); hipMalloc(&dbc, nb * 4); hipMalloc(&dinfo, 4);
    hipMemcpy(dsave, Sd.data(), Sd.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dSo, So.data(), So.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(drhs, rhs.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dbr, br.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbc, bc.data(), nb * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 20; rep++) {
        hipMemcpy(dSd, dsave, Sd.size() * 8, hipMemcpyDeviceToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(ldlt_one_workgroup_kernel, dim3(1), dim3(kSsThreads), 0, 0, Nc, nb, dbr, dbc, dSd, dSo, drhs, dx, dinfo);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<double> x(n); int info = -1;
    hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost); hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; i++) {
        double s = -rhs[i];
        for (int j = 0; j < n; j++) s += S[(size_t)i * n + j] * x[j];
        worst = fmax(worst, fabs(s));
    }
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ss_stamps), sizeof(st));
    for (int k = 1; k < 4; k++)
        printf("  phase %d: %llu shader cycles, %.2f us (100 MHz clock)\n", k, st[2 * k] - st[2 * k - 2], (st[2 * k + 1] - st[2 * k - 1]) / 100.0);
    printf("Nc %d n %d: %.1f us, info %d, max |S x - rhs| = %.3e\n", Nc, n, best * 1e3, info, worst);
    return 0;
}
