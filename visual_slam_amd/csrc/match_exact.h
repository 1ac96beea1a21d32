// Exact descriptor arithmetic shared by the matcher kernels: the reference's distance
//   d = sqrtf(2 - 2*clamp((float)dot64(a, b), -1, 1))       (superpoint_frontend.cc:181-185,
//   tracking.cc:24-30, frame.cc:61-63), replayed bit for bit.
#ifndef VS_MATCH_EXACT_H_
#define VS_MATCH_EXACT_H_
#include "common.h"

namespace vs {

constexpr int kDescDim = VS_DESC_DIM;   // 256

// The reference's distance, bit for bit: double accumulation in index order, narrowed to float,
// clamped, sqrtf(2 - 2 s).
__device__ __forceinline__ float exact_distance(const float *__restrict__ a,
                                                const float *__restrict__ b)
{
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < kDescDim; k += 4) {
        const float4 x = *reinterpret_cast<const float4 *>(a + k);
        const float4 y = *reinterpret_cast<const float4 *>(b + k);
        acc = fma((double)x.x, (double)y.x, acc);
        acc = fma((double)x.y, (double)y.y, acc);
        acc = fma((double)x.z, (double)y.z, acc);
        acc = fma((double)x.w, (double)y.w, acc);
    }
    float s = (float)acc;
    s = (s < -1.0f) ? -1.0f : (1.0f < s) ? 1.0f : s;
    return sqrtf(2.0f - 2.0f * s);
}

__device__ __forceinline__ bool stereo_masked(int2 p0, int2 p1)
{
    int dy = p0.y - p1.y;
    dy = dy < 0 ? -dy : dy;
    return dy > 20 || p0.x < p1.x;
}

constexpr int kPanel = 32;               // floats of each row staged per step
constexpr int kPanelStride = 36;         // LDS row stride in floats (conflict-free ds_read_b128)

// Cooperative exact dots: each of the 64 lanes owns one (row of desc0, row of desc1) pair whose
// base pointers sit in pA/pB (LDS).  Rows are staged panel by panel with coalesced 128-byte
// segments (the next panel is already in flight while the current one is consumed), then every
// lane accumulates ITS OWN pair in index order (fp64 fma of exact products = the reference's
// double accumulation).  UNIFORM_A: every lane's first row is pA[0]; it is staged once, whole,
// into tA (256 floats).
template <bool UNIFORM_A>
__device__ __forceinline__ double panel_dot64(const float *const *pA, const float *const *pB,
                                              float *tA, float *tB, int lane)
{
    double acc = 0.0;
    const int seg = lane & 7, sub = lane >> 3;
    if (UNIFORM_A)
        *reinterpret_cast<float4 *>(tA + lane * 4) = *reinterpret_cast<const float4 *>(pA[0] + lane * 4);
    float4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7, ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7;
#define VS_PANEL_LOAD(po)                                                                        \
    do {                                                                                         \
        const int o_ = (po) + seg * 4;                                                           \
        rb0 = *reinterpret_cast<const float4 *>(pB[sub] + o_);                                   \
        rb1 = *reinterpret_cast<const float4 *>(pB[8 + sub] + o_);                               \
        rb2 = *reinterpret_cast<const float4 *>(pB[16 + sub] + o_);                              \
        rb3 = *reinterpret_cast<const float4 *>(pB[24 + sub] + o_);                              \
        rb4 = *reinterpret_cast<const float4 *>(pB[32 + sub] + o_);                              \
        rb5 = *reinterpret_cast<const float4 *>(pB[40 + sub] + o_);                              \
        rb6 = *reinterpret_cast<const float4 *>(pB[48 + sub] + o_);                              \
        rb7 = *reinterpret_cast<const float4 *>(pB[56 + sub] + o_);                              \
        if (!UNIFORM_A) {                                                                        \
            ra0 = *reinterpret_cast<const float4 *>(pA[sub] + o_);                               \
            ra1 = *reinterpret_cast<const float4 *>(pA[8 + sub] + o_);                           \
            ra2 = *reinterpret_cast<const float4 *>(pA[16 + sub] + o_);                          \
            ra3 = *reinterpret_cast<const float4 *>(pA[24 + sub] + o_);                          \
            ra4 = *reinterpret_cast<const float4 *>(pA[32 + sub] + o_);                          \
            ra5 = *reinterpret_cast<const float4 *>(pA[40 + sub] + o_);                          \
            ra6 = *reinterpret_cast<const float4 *>(pA[48 + sub] + o_);                          \
            ra7 = *reinterpret_cast<const float4 *>(pA[56 + sub] + o_);                          \
        }                                                                                        \
    } while (0)
#define VS_PANEL_PUT(t, g, r) *reinterpret_cast<float4 *>((t) + ((g)*8 + sub) * kPanelStride + seg * 4) = (r)
    VS_PANEL_LOAD(0);
#pragma unroll 1
    for (int p = 0; p < kDescDim / kPanel; p++) {
        VS_PANEL_PUT(tB, 0, rb0); VS_PANEL_PUT(tB, 1, rb1); VS_PANEL_PUT(tB, 2, rb2); VS_PANEL_PUT(tB, 3, rb3);
        VS_PANEL_PUT(tB, 4, rb4); VS_PANEL_PUT(tB, 5, rb5); VS_PANEL_PUT(tB, 6, rb6); VS_PANEL_PUT(tB, 7, rb7);
        if (!UNIFORM_A) {
            VS_PANEL_PUT(tA, 0, ra0); VS_PANEL_PUT(tA, 1, ra1); VS_PANEL_PUT(tA, 2, ra2); VS_PANEL_PUT(tA, 3, ra3);
            VS_PANEL_PUT(tA, 4, ra4); VS_PANEL_PUT(tA, 5, ra5); VS_PANEL_PUT(tA, 6, ra6); VS_PANEL_PUT(tA, 7, ra7);
        }
        __syncthreads();
        if (p + 1 < kDescDim / kPanel) VS_PANEL_LOAD((p + 1) * kPanel);
#pragma unroll
        for (int q = 0; q < kPanel / 4; q++) {
            const float4 y = *reinterpret_cast<const float4 *>(tB + lane * kPanelStride + q * 4);
            float4 x;
            if (UNIFORM_A) x = *reinterpret_cast<const float4 *>(tA + p * kPanel + q * 4);
            else x = *reinterpret_cast<const float4 *>(tA + lane * kPanelStride + q * 4);
            acc = fma((double)x.x, (double)y.x, acc);
            acc = fma((double)x.y, (double)y.y, acc);
            acc = fma((double)x.z, (double)y.z, acc);
            acc = fma((double)x.w, (double)y.w, acc);
        }
        __syncthreads();
    }
#undef VS_PANEL_LOAD
#undef VS_PANEL_PUT
    return acc;
}

__device__ __forceinline__ float distance_from_dot(double acc)
{
    float s = (float)acc;
    s = (s < -1.0f) ? -1.0f : (1.0f < s) ? 1.0f : s;
    return sqrtf(2.0f - 2.0f * s);
}


}  // namespace vs
#endif  // VS_MATCH_EXACT_H_
