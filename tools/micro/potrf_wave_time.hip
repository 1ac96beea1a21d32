// Micro-benchmark: cycles of potrf_inverse_wave (factor + inverse of a 32 x 32 block by one wave in
// registers), the sequential core of every reduced-system factorisation.
// Measured on MI355X: 15980 cycles = 499 per pivot.  Of these ~12500 remain when the updates of the
// columns right of the pivot's neighbour are left out (wrong results): the pivot-to-pivot chain
// (product, fma, lane read, select, v_rcp_f64 and its three refinement fmas: ~310 cycles) is what
// the time consists of.  Tried on top of it, results unchanged bit for bit: issuing the previous
// step's updates in the latency shadows of the chain with scheduling barriers (16924 cycles, the
// compiler's own interleaving is no worse); batching the lane reads ahead of their fmas (15748);
// fetching the column through a wave-private LDS line with broadcast reads instead of two
// v_readlane per entry (2654 instead of 3024 instructions, 15048 cycles).  None is kept.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ivisual_slam_amd/csrc tools/micro/potrf_wave_time.hip -o tools/micro/potrf_wave_time.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "vslam_hip.h"
#include "common.h"
#include "dense_solve.h"
namespace vs { void set_error(const char *, ...) {} }   // the library's error sink, not needed here
using namespace vs;

__global__ void __launch_bounds__(64) k(const real *A, real *Lout, real *Liout, unsigned long long *cycles, int *bad_out)
{
    __shared__ real Wn[kNB][kNB + 1];
    __shared__ real Ld[kNB][kNB + 1];
    __shared__ __attribute__((aligned(16))) real wimg[48 * kSS];
    __shared__ __attribute__((aligned(16))) real Li[kNB][kXS];
    const int lane = threadIdx.x;
    for (int e = lane; e < kNB * kNB; e += 64) Wn[e / kNB][e % kNB] = A[e];
    __syncthreads();
    const unsigned long long t0 = clock64();
    const int bad = potrf_inverse_wave(Wn, lane, wimg, Ld, Li);
    const unsigned long long t1 = clock64();
    __syncthreads();
    for (int e = lane; e < kNB * kNB; e += 64) { Lout[e] = Ld[e % kNB][e / kNB]; Liout[e] = Li[e / kNB][e % kNB]; }
    if (lane == 0) { *cycles = t1 - t0; *bad_out = bad; }
}

int main()
{
    const int n = kNB;
    std::vector<double> M(n * n), S(n * n);
    srand(3);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int q = 0; q < n; q++) s += M[i * n + q] * M[j * n + q];
            S[i * n + j] = s + (i == j ? 1.0 : 0.0);
        }
    double *dA, *dL, *dLi; unsigned long long *dc; int *db;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dLi, n * n * 8); hipMalloc(&dc, 8); hipMalloc(&db, 4);
    hipMemcpy(dA, S.data(), n * n * 8, hipMemcpyHostToDevice);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 10; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dL, dLi, dc, db);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
    }
    std::vector<double> L(n * n), Li(n * n); int bad;
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Li.data(), dLi, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0, t = 0;
            for (int q = 0; q <= j; q++) s += L[i * n + q] * L[j * n + q];
            for (int q = j; q <= i; q++) t += Li[i * n + q] * L[q * n + j];
            e1 = fmax(e1, fabs(s - S[i * n + j]));
            e2 = fmax(e2, fabs(t - (i == j ? 1.0 : 0.0)));
        }
    printf("potrf_inverse_wave: %llu cycles (%.0f per pivot), bad %d, |L L^T - A| %.2e, |Linv L - I| %.2e\n", best, best / 32.0, bad, e1, e2);
    return 0;
}
