// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on one SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma64_rate.hip -o tools/micro/mfma64_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_loop(double *out, int iters, unsigned long long *cyc)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{ 0, 0, 0, 0 };
    double a = threadIdx.x * 0.001 + 1.0, b = 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void fma_loop(double *out, int iters, unsigned long long *cyc)
{
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = i;
    double a = threadIdx.x * 1e-9 + 1.0, b = 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = fma(a, acc[i], b);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
#define RUN(K, NACC, THREADS)                                                          \
    hipLaunchKernelGGL((K<NACC>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);       \
    hipLaunchKernelGGL((K<NACC>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);       \
    hipEventRecord(e0);                                                                 \
    hipLaunchKernelGGL((K<NACC>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);       \
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);      \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                       \
    printf("%-10s acc=%d threads=%4d : %.1f ticks per instruction per wave, %.2f ns per instruction per wave (kernel %.1f us, %.0f ticks/us)\n", #K, NACC, THREADS, (double)h / (iters * NACC), ms * 1e6 / (iters * NACC), ms * 1e3, h / (ms * 1e3));
    RUN(mfma_loop, 1, 64) RUN(mfma_loop, 2, 64) RUN(mfma_loop, 4, 64) RUN(mfma_loop, 8, 64)
    RUN(mfma_loop, 4, 256) RUN(mfma_loop, 4, 512) RUN(mfma_loop, 4, 1024)
    RUN(fma_loop, 1, 64) RUN(fma_loop, 4, 64) RUN(fma_loop, 16, 64) RUN(fma_loop, 16, 256) RUN(fma_loop, 16, 512) RUN(fma_loop, 16, 1024)
    return 0;
}
