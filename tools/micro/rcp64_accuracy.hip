// Micro-benchmark: relative error of v_rcp_f64 / v_rsq_f64 and of the refined forms used by the
// Cholesky kernels.   hipcc -O3 --offload-arch=gfx950 tools/micro/rcp64_accuracy.hip -o tools/micro/rcp64_accuracy.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *rcp0, double *rcp1, double *rcp2, double *rsq0, double *rsq1, double *rsq2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double y = __builtin_amdgcn_rcp(d);
    rcp0[i] = y;
    y = fma(y, fma(-d, y, 1.0), y); rcp1[i] = y;
    y = fma(y, fma(-d, y, 1.0), y); rcp2[i] = y;
    double r = __builtin_amdgcn_rsq(d);
    rsq0[i] = r;
    const double hd = 0.5 * d;
    r = r * fma(-hd * r, r, 1.5); rsq1[i] = r;
    r = r * fma(-hd * r, r, 1.5); rsq2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;          // [0,1)
        x[i] = ldexp(1.0 + u, (int)(s % 80) - 40);
    }
    double *dx, *o[6];
    hipMalloc(&dx, n * 8); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (auto &p : o) hipMalloc(&p, n * 8);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, o[0], o[1], o[2], o[3], o[4], o[5], n);
    const char *names[6] = { "v_rcp_f64", "rcp + 1 Newton", "rcp + 2 Newton", "v_rsq_f64", "rsq + 1 Newton", "rsq + 2 Newton" };
    std::vector<double> h(n);
    for (int q = 0; q < 6; q++) {
        hipMemcpy(h.data(), o[q], n * 8, hipMemcpyDeviceToHost);
        long double worst = 0;
        for (int i = 0; i < n; i++) {
            long double ref = q < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            long double e = fabsl(((long double)h[i] - ref) / ref);
            if (e > worst) worst = e;
        }
        printf("%-16s max relative error %.3Le = 2^%.1Lf\n", names[q], worst, log2l(worst));
    }
    return 0;
}
