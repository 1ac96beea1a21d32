// Does one wave per SIMD run v_mfma_f32_32x32x16_f16 at its back-to-back rate while it also
// issues NV independent fp32 VALU instructions (v_max3 / v_med3 chains, as a top-2 update uses)
// per MFMA?  Prints cycles per MFMA for NV = 0..10.   hipcc --offload-arch=gfx950 -O3 -o ovl mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
ovl_kernel(const float *in, float *out, long long *cycles, int iters)
{
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 8 + i]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float v[12];
    for (int i = 0; i < 12; i++) v[i] = in[threadIdx.x + 16 + i];
    float b1[4] = { -1e30f, -1e30f, -1e30f, -1e30f }, b2[4] = { -1e30f, -1e30f, -1e30f, -1e30f };
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
            // NV VALU instructions in 4 independent chains (med3 / max3 / max, three per two keys)
#pragma unroll
            for (int n = 0; n < NV; n++) {
                const int c = n & 3;
                const float k1 = v[(n * 3 + q) % 12], k2 = v[(n * 3 + q + 5) % 12];
                if (n % 3 == 0) { const float t = __builtin_amdgcn_fmed3f(b1[c], k1, k2); b2[c] = __builtin_fmaxf(b2[c], t); }
                else if (n % 3 == 1) b1[c] = __builtin_fmaxf(__builtin_fmaxf(b1[c], k1), k2);
                else b2[c] = __builtin_amdgcn_fmed3f(b1[c], b2[c], k1);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x2, NV + 2, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int c = 0; c < 4; c++) s += b1[c] + b2[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NV>
void run(const float *in, float *out, long long *cyc, int iters)
{
    hipLaunchKernelGGL(ovl_kernel<NV>, dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ovl_kernel<NV>, dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double mfmas = 4.0 * iters;
    printf("NV=%2d  %.1f ns per MFMA (wall)  %.1f counter ticks per MFMA  -> %.0f TFLOP/s chip\n", NV, ms * 1e6 / mfmas,
           (double)c / mfmas, 1024.0 * mfmas * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
}

// Sustained behaviour: the same launch repeated for a few seconds, time per launch at intervals.
template <int NV>
void sustained(const float *in, float *out, long long *cyc, int iters, int launches)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int l = 0; l < launches; l++) {
        const bool timed = l % (launches / 10) == 0 || l == launches - 1;
        if (timed) hipEventRecord(e0);
        hipLaunchKernelGGL(ovl_kernel<NV>, dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
        if (timed) {
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("  NV=%d launch %5d: %.1f ns per MFMA\n", NV, l, ms * 1e6 / (4.0 * iters));
        }
    }
    hipDeviceSynchronize();
}

int main(int argc, char **argv)
{
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    float *in, *out; long long *cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    {   // random operands: the matrix cores draw more power on real data than on zeros
        static float h[4096];
        unsigned x = 12345u;
        for (int i = 0; i < 4096; i++) { x = x * 1664525u + 1013904223u; h[i] = (float)((int)(x >> 8) % 2001 - 1000) * (zeros ? 0.f : 1e-3f); }
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int iters = 20000;
    run<0>(in, out, cyc, iters); run<2>(in, out, cyc, iters); run<4>(in, out, cyc, iters); run<5>(in, out, cyc, iters);
    run<6>(in, out, cyc, iters); run<7>(in, out, cyc, iters); run<8>(in, out, cyc, iters); run<10>(in, out, cyc, iters);
    sustained<0>(in, out, cyc, iters, 2000);
    sustained<7>(in, out, cyc, iters, 2000);
    return 0;
}
