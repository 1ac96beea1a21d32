// Micro-benchmark: rate of fp64 additions into a tile of 6x6 blocks held in LDS, the operation a
// landmark-major accumulation of the reduced camera matrix would be made of (every lane owns one
// landmark and adds the 36 values of one camera-pair block; neighbouring landmarks share blocks).
//   mode 0: ds_add_f64 (LDS atomic, no return)      mode 1: plain ds_write_b64 (the floor)
//   g     : consecutive lanes that add into the SAME block (same 36 addresses); 0 = a random block per lane
// Prints LDS cycles per wave-level instruction per CU (2.4 GHz), with one and four waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomic_rate.hip -o tools/micro/lds_atomic_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kBlocks = 300;   // 300 blocks x 36 doubles = 86 KB, the size of a window tile

template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, int g, double *out)
{
    extern __shared__ double tile[];
    for (int i = threadIdx.x; i < kBlocks * 36; i += blockDim.x) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double val = 1e-3 * (lane + 1);
    unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    int blk = g > 0 ? ((lane / g) * 7 + wave * 31) % kBlocks : (int)(h % kBlocks);
    for (int it = 0; it < iters; it++) {
        double *p = tile + blk * 36;
#pragma unroll
        for (int v = 0; v < 36; v++) {
            if (MODE == 0) atomicAdd(p + v, val);
            else p[v] = val;
        }
        if (g > 0) { blk += 13; if (blk >= kBlocks) blk -= kBlocks; }
        else { h = h * 1664525u + 1013904223u; blk = (int)((h >> 8) % kBlocks); }
    }
    __syncthreads();
    double s = 0;
    for (int i = threadIdx.x; i < kBlocks * 36; i += blockDim.x) s += tile[i];
    if (s == 12345.678) out[blockIdx.x] = s;   // keeps the loop alive
}

int main()
{
    double *out;
    hipMalloc(&out, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400, n_wg = 256;
    const size_t lds = (size_t)kBlocks * 36 * 8;
    printf("mode g threads  us   cycles per wave-instruction per CU\n");
    for (int mode = 0; mode < 2; mode++)
        for (int threads : { 256, 1024 })
            for (int g : { 1, 2, 4, 8, 16, 64, 0 }) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n_wg), dim3(threads), lds, 0, iters, g, out);
                    else hipLaunchKernelGGL(k<1>, dim3(n_wg), dim3(threads), lds, 0, iters, g, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double instr_per_cu = (double)iters * 36 * (threads / 64);
                printf("%d %2d %4d %8.1f %8.2f\n", mode, g, threads, best * 1e3, best * 1e-3 * 2.4e9 / instr_per_cu);
            }
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    return 0;
}
