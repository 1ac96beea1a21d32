// Micro-benchmark: cost of the "last arriving workgroup adds the partials" pattern
// (store, device-scope fence, counter increment) against plain stores.
// hipcc -O3 --offload-arch=gfx950 tools/micro/fence_cost.hip -o tools/micro/fence_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(64) k(double *part, int *ctr, double *out, int group, int spin)
{
    // some work so that the launch is not empty
    double v = threadIdx.x * 1e-3 + blockIdx.x;
    for (int i = 0; i < spin; i++) v = fma(v, 1.0000001, 1e-9);
    if (threadIdx.x < 36) part[(size_t)blockIdx.x * 36 + threadIdx.x] = v;
    if (MODE == 0) return;
    __threadfence();
    int last = 0;
    if (threadIdx.x == 0) last = atomicAdd(ctr + blockIdx.x / group, 1) == group - 1;
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) ctr[blockIdx.x / group] = 0;
    if (threadIdx.x < 36) {
        double s = 0;
        const size_t b0 = (size_t)(blockIdx.x / group) * group;
        for (int i = 0; i < group; i++) s += part[(b0 + i) * 36 + threadIdx.x];
        out[(size_t)(blockIdx.x / group) * 36 + threadIdx.x] = s;
    }
}

int main()
{
    const int n = 32768, group = 4;
    double *part, *out; int *ctr;
    hipMalloc(&part, (size_t)n * 36 * 8); hipMalloc(&out, (size_t)n * 36 * 8); hipMalloc(&ctr, n * 4);
    hipMemset(ctr, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : { 0, 2000 }) {
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 10; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n), dim3(64), 0, 0, part, ctr, out, group, spin);
                else hipLaunchKernelGGL(k<1>, dim3(n), dim3(64), 0, 0, part, ctr, out, group, spin);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("spin %d mode %d : %.1f us for %d workgroups\n", spin, mode, best * 1e3, n);
        }
    }
    // correctness of the pattern: every group sum must equal the sum of its members' values
    std::vector<double> hp((size_t)n * 36), ho((size_t)n / group * 36);
    hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int g = 0; g < n / group; g++)
        for (int t = 0; t < 36; t++) {
            double s = 0;
            for (int i = 0; i < group; i++) s += hp[((size_t)g * group + i) * 36 + t];
            if (s != ho[(size_t)g * 36 + t]) bad++;
        }
    printf("mismatching sums: %d\n", bad);
    return 0;
}
