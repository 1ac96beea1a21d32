// SuperPoint descriptor matcher for gfx950 (reference interfaces: include/vslam_hip.h).
//
// Exactness plan.  The reference decides nearest neighbours on
//   d(i,j) = sqrtf(2 - 2*clamp((float)dot64(desc0_i, desc1_j), -1, 1)),  first minimum wins.
// The dense 4096 x 4096 x 256 contraction runs on the matrix cores with fp16 operands:
// every descriptor (scaled by 2^8) is split into hi + lo fp16 parts and the kernel accumulates
// hi*hi + hi*lo + lo*hi in fp32, i.e. an fp32-accurate dot s~ with |s~ - s| <= kEps for unit
// descriptors (split residue <= 3*2^-22, fp32 accumulation of 768 products <= 768*2^-23).  A fused
// epilogue keeps, per row and per column, the best and second-best s~.  Where best - second
// exceeds 2*kEps + kCollapse (kCollapse covers distinct dots that the fp32 clamp/sqrt chain maps
// to one float), the approximate winner provably is the reference's first minimum and only its
// distance is recomputed exactly; otherwise the row/column is re-decided by an exact scan that
// replays the reference arithmetic (sequential fp64 accumulation, narrowing, clamp, sqrtf).
// Indices and distances are therefore bit-identical to the reference path.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

using namespace vs;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = VS_DESC_DIM;        // 256
constexpr int kBM = 128, kBN = 128;    // workgroup tile
constexpr int kBK = 32;                // halfs per K-step
constexpr int kRowH = kBK + 8;         // padded LDS row (80 B): conflict-free ds_read_b128
constexpr int kSub = 64;               // rows/cols covered by one wave = granularity of partials
constexpr float kScale = 256.0f;       // descriptor prescale (2^8), dot prescale 2^16
constexpr float kDotScale = 65536.0f;
constexpr float kEps = 1.1e-4f;        // bound on |s~ - s| (fp16 split + fp32 accumulation 1e-4, index packing 8e-6)
constexpr float kCollapse = 1.0e-6f;   // dots closer than this may share one fp32 distance
constexpr float kBand = (2.0f * kEps + kCollapse) * kDotScale;
constexpr float kNegInf = -3.0e38f;
constexpr float kMaskedDot = 0.5f * kDotScale;   // distance 1.0  <=>  dot 0.5

// fp32 rows -> hi/lo fp16 rows (n_pad rows, rows >= n zero-filled).
__global__ void __launch_bounds__(256)
split_kernel(const float *__restrict__ src, int n, int n_pad, _Float16 *__restrict__ hi,
             _Float16 *__restrict__ lo)
{
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= (int64_t)n_pad * kD) return;
    const int row = (int)(e / kD);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) v = *reinterpret_cast<const float4 *>(src + e);
    const float x[4] = { v.x * kScale, v.y * kScale, v.z * kScale, v.w * kScale };
    _Float16 h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        h[k] = (_Float16)x[k];
        l[k] = (_Float16)(x[k] - (float)h[k]);
    }
    *reinterpret_cast<uint2 *>(hi + e) = *reinterpret_cast<uint2 *>(h);
    *reinterpret_cast<uint2 *>(lo + e) = *reinterpret_cast<uint2 *>(l);
}

// plain VALU max / median without the canonicalising extra instructions the compiler adds
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmed3(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct Top2 {
    float b1, b2;
    int i1;
};

__device__ __forceinline__ void top2_push(Top2 &t, float v, int idx)
{
    if (v > t.b1) { t.b2 = t.b1; t.b1 = v; t.i1 = idx; }
    else t.b2 = fmaxf(t.b2, v);
}

// merge (ob1, oi1, ob2) into t; equal bests leave b2 == b1 (gap 0 => decided exactly later)
__device__ __forceinline__ void top2_merge(Top2 &t, float ob1, int oi1, float ob2)
{
    if (ob1 > t.b1) { t.b2 = fmaxf(t.b1, ob2); t.b1 = ob1; t.i1 = oi1; }
    else t.b2 = fmaxf(t.b2, ob1);
}

struct PairDims {
    int n0, n1;
};

// One workgroup = one 128 x 128 tile of s~ = 2^16 * desc0 . desc1 for one frame pair; 4 waves,
// each a 64 x 64 sub-tile as 2 x 2 MFMA 32x32x16 f16 tiles.  Epilogue: per-column top-2 over the
// wave's 64 rows (in registers) and per-row top-2 over its 64 columns (transposed through LDS).
template <bool MASKED>
__global__ void __launch_bounds__(256)
dot_top2_kernel(const _Float16 *__restrict__ a_hi, const _Float16 *__restrict__ a_lo,
                const _Float16 *__restrict__ b_hi, const _Float16 *__restrict__ b_lo,
                const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                float *__restrict__ row_b1, int *__restrict__ row_i1, float *__restrict__ row_b2,
                float *__restrict__ col_b1, int *__restrict__ col_i1, float *__restrict__ col_b2)
{
    // staging: 4 operand tiles of 128 x kRowH halfs; epilogue: 4 waves x 32 x 65 floats
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * kBM * kRowH * 2];
    _Float16 *sA_hi = reinterpret_cast<_Float16 *>(smem);
    _Float16 *sA_lo = sA_hi + kBM * kRowH;
    _Float16 *sB_hi = sA_lo + kBM * kRowH;
    _Float16 *sB_lo = sB_hi + kBN * kRowH;

    const int pair = blockIdx.z;
    const int ntc = n1_pad / kBN, ntr = n0_pad / kBM;
    (void)ntc;
    // XCD-aware tile order: workgroups that land on one XCD (blockIdx.x % 8) take a band of
    // `rpx` tile rows and walk it column by column, so the band's A rows stay in that XCD's L2
    // and each B panel is fetched once per band instead of once per tile row.
    int tile_r, tile_c;
    {
        const int lin = blockIdx.x;
        if ((ntr & 7) == 0) {
            const int rpx = ntr >> 3;                  // tile rows per XCD
            const int q = lin >> 3;                    // position inside the XCD's band
            tile_r = (lin & 7) * rpx + q % rpx;
            tile_c = q / rpx;
        } else {
            tile_r = lin / ntc;
            tile_c = lin % ntc;
        }
    }
    const int n0 = dims[pair].n0, n1 = dims[pair].n1;
    const int i0 = tile_r * kBM, j0 = tile_c * kBN;
    if (i0 >= n0 || j0 >= n1) return;   // tile fully outside this pair: nothing reads its partials
    const _Float16 *gA_hi = a_hi + ((int64_t)pair * n0_pad + i0) * kD;
    const _Float16 *gA_lo = a_lo + ((int64_t)pair * n0_pad + i0) * kD;
    const _Float16 *gB_hi = b_hi + ((int64_t)pair * n1_pad + j0) * kD;
    const _Float16 *gB_lo = b_lo + ((int64_t)pair * n1_pad + j0) * kD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging map: 128 rows x 4 chunks of 16 B per operand tile; 2 chunks per thread, held in
    // named registers (an indexed array would be demoted to scratch memory)
    static_assert(kBK == 32, "staging map below is written for 32-half K-steps");
    const int c0 = tid, c1 = tid + 256;
    const int64_t g0 = (int64_t)(c0 >> 2) * kD + (c0 & 3) * 8, g1 = (int64_t)(c1 >> 2) * kD + (c1 & 3) * 8;
    const int l0 = (c0 >> 2) * kRowH + (c0 & 3) * 8, l1 = (c1 >> 2) * kRowH + (c1 & 3) * 8;
    uint4 ra0, ra1, rb0, rb1, rc0, rc1, rd0, rd1;
#define VS_LD4(r, base, ko)                                                                \
    r##0 = *reinterpret_cast<const uint4 *>(base + g0 + ko);                               \
    r##1 = *reinterpret_cast<const uint4 *>(base + g1 + ko)
#define VS_ST4(r, base)                                                                    \
    *reinterpret_cast<uint4 *>(base + l0) = r##0;                                          \
    *reinterpret_cast<uint4 *>(base + l1) = r##1
#define VS_LOAD_TILE(kt)                                                                   \
    do {                                                                                   \
        const int ko = (kt) * kBK;                                                         \
        VS_LD4(ra, gA_hi, ko); VS_LD4(rb, gA_lo, ko); VS_LD4(rc, gB_hi, ko); VS_LD4(rd, gB_lo, ko); \
    } while (0)
#define VS_STORE_TILE()                                                                    \
    do {                                                                                   \
        VS_ST4(ra, sA_hi); VS_ST4(rb, sA_lo); VS_ST4(rc, sB_hi); VS_ST4(rd, sB_lo);        \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    VS_LOAD_TILE(0);
    constexpr int kSteps = kD / kBK;
    for (int kt = 0; kt < kSteps; kt++) {
        __syncthreads();
        VS_STORE_TILE();
        __syncthreads();
        if (kt + 1 < kSteps) VS_LOAD_TILE(kt + 1);
        // keep the prefetch in front of the MFMA block (the scheduler otherwise sinks the loads
        // to the end of the iteration, exposing their whole latency at the next barrier)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ks++) {
            // fragment of a 32x32x16 MFMA: lane holds 8 consecutive k of row/col (lane & 31),
            // k offset 8 * (lane >> 5)
            const int koff = ks * 16 + (lane >> 5) * 8;
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int row = wr * 64 + m * 32 + (lane & 31);
                ah[m] = *reinterpret_cast<const half8 *>(sA_hi + row * kRowH + koff);
                al[m] = *reinterpret_cast<const half8 *>(sA_lo + row * kRowH + koff);
            }
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const int col = wc * 64 + n * 32 + (lane & 31);
                bh[n] = *reinterpret_cast<const half8 *>(sB_hi + col * kRowH + koff);
                bl[n] = *reinterpret_cast<const half8 *>(sB_lo + col * kRowH + koff);
            }
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }

#undef VS_LOAD_TILE
#undef VS_STORE_TILE
#undef VS_LD4
#undef VS_ST4
    // ---- epilogue --------------------------------------------------------------------------
    // element (m, n, r) of this lane: row gi = i0 + wr*64 + m*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
    //                                 col gj = j0 + wc*64 + n*32 + (lane&31)
    const int half = lane >> 5, l31 = lane & 31;
    const bool partial_tile = i0 + kBM > n0 || j0 + kBN > n1;      // workgroup-uniform
    if (MASKED || partial_tile) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int gj = j0 + wc * 64 + n * 32 + l31;
            int2 pj = make_int2(0, 0);
            if (MASKED && gj < n1) pj = xy1[(int64_t)pair * n1_pad + gj];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int gi = i0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[m][n][r];
                    if (MASKED && gi < n0 && gj < n1) {
                        const int2 pi = xy0[(int64_t)pair * n0_pad + gi];
                        int dy = pi.y - pj.y;
                        dy = dy < 0 ? -dy : dy;
                        if (dy > 20 || pi.x < pj.x) v = kMaskedDot;
                    }
                    if (gi >= n0 || gj >= n1) v = kNegInf;
                    acc[m][n][r] = v;
                }
        }
    }
    // running top-2 on PACKED scores: the low 6 mantissa bits of each score carry the element
    // number (relative perturbation <= 2^-17, inside kEps), so no index needs tracking:
    //   b2 <- median(b1, b2, key);  b1 <- max(b1, key)         (2 VALU + 1 pack per element)
#define VS_TOP2_STEP(v, e)                                                                  \
    do {                                                                                     \
        const float k_ = __uint_as_float((__float_as_uint(v) & 0xffffffc0u) | (unsigned)(e)); \
        b2 = vmed3(b1, b2, k_);                                                              \
        b1 = vmax(b1, k_);                                                                   \
    } while (0)
    // columns: top-2 over the wave's 64 rows
#pragma unroll
    for (int n = 0; n < 2; n++) {
        float b1 = kNegInf, b2 = kNegInf;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) VS_TOP2_STEP(acc[m][n][r], m * 16 + r);
        const int e1 = (int)(__float_as_uint(b1) & 63u);
        const int rr = e1 & 15;
        Top2 t{ b1, b2, i0 + wr * 64 + (e1 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half };
        const float ob1 = __shfl_xor(t.b1, 32, 64), ob2 = __shfl_xor(t.b2, 32, 64);
        const int oi1 = __shfl_xor(t.i1, 32, 64);
        top2_merge(t, ob1, oi1, ob2);
        if (half == 0) {
            const int64_t o = ((int64_t)pair * (n0_pad / kSub) + (i0 / kSub + wr)) * n1_pad + j0 + wc * 64 + n * 32 + l31;
            col_b1[o] = t.b1; col_i1[o] = t.i1; col_b2[o] = t.b2;
        }
    }
    // rows: transpose 32 rows x 64 cols at a time through LDS (row stride 68 floats keeps the
    // float4 reads aligned and conflict-free)
    constexpr int kTS = 68;
    float *sT = reinterpret_cast<float *>(smem) + wave * (32 * kTS);
    static_assert(4 * 32 * kTS * 4 <= (int)sizeof(smem), "transpose scratch fits the staging LDS");
#pragma unroll
    for (int m = 0; m < 2; m++) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                sT[((r & 3) + 8 * (r >> 2) + 4 * half) * kTS + n * 32 + l31] = acc[m][n][r];
        __syncthreads();
        float b1 = kNegInf, b2 = kNegInf;
        const float4 *rowp = reinterpret_cast<const float4 *>(sT + l31 * kTS + 32 * half);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float4 q = rowp[c];
            VS_TOP2_STEP(q.x, 4 * c); VS_TOP2_STEP(q.y, 4 * c + 1);
            VS_TOP2_STEP(q.z, 4 * c + 2); VS_TOP2_STEP(q.w, 4 * c + 3);
        }
        const int e1 = (int)(__float_as_uint(b1) & 63u);
        Top2 t{ b1, b2, j0 + wc * 64 + 32 * half + e1 };
        const float ob1 = __shfl_xor(t.b1, 32, 64), ob2 = __shfl_xor(t.b2, 32, 64);
        const int oi1 = __shfl_xor(t.i1, 32, 64);
        top2_merge(t, ob1, oi1, ob2);
        if (half == 0) {
            const int64_t o = ((int64_t)pair * (n1_pad / kSub) + (j0 / kSub + wc)) * n0_pad + i0 + wr * 64 + m * 32 + l31;
            row_b1[o] = t.b1; row_i1[o] = t.i1; row_b2[o] = t.b2;
        }
    }
#undef VS_TOP2_STEP
}

// Merge the per-64 partials of one row (dir 0) or column (dir 1); decide whether the approximate
// winner is certain.  One thread per item.
__global__ void __launch_bounds__(256)
merge_kernel(const PairDims *__restrict__ dims, int n_pad_items, int n_parts,
             int dir, const float *__restrict__ b1, const int *__restrict__ i1,
             const float *__restrict__ b2, int *__restrict__ nn_idx,
             unsigned char *__restrict__ uncertain, float *__restrict__ best, int force_uncertain)
{
    const int pair = blockIdx.y;
    const int it = blockIdx.x * 256 + threadIdx.x;
    const int n_items = dir == 0 ? dims[pair].n0 : dims[pair].n1;
    const int n_other = dir == 0 ? dims[pair].n1 : dims[pair].n0;
    if (it >= n_items) return;
    Top2 t{ kNegInf, kNegInf, 0 };
    const int parts = (n_other + kSub - 1) / kSub;
    for (int p = 0; p < parts; p++) {
        const int64_t o = ((int64_t)pair * n_parts + p) * n_pad_items + it;
        top2_merge(t, b1[o], i1[o], b2[o]);
    }
    nn_idx[(int64_t)pair * n_pad_items + it] = t.i1;
    best[(int64_t)pair * n_pad_items + it] = t.b1;
    uncertain[(int64_t)pair * n_pad_items + it] = (force_uncertain || !(t.b1 - t.b2 > kBand)) ? 1 : 0;
}

// The reference's distance, bit for bit: double accumulation in index order, narrowed to float,
// clamped, sqrtf(2 - 2 s).
__device__ __forceinline__ float exact_distance(const float *__restrict__ a,
                                                const float *__restrict__ b)
{
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < kD; k += 4) {
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
    for (int p = 0; p < kD / kPanel; p++) {
        VS_PANEL_PUT(tB, 0, rb0); VS_PANEL_PUT(tB, 1, rb1); VS_PANEL_PUT(tB, 2, rb2); VS_PANEL_PUT(tB, 3, rb3);
        VS_PANEL_PUT(tB, 4, rb4); VS_PANEL_PUT(tB, 5, rb5); VS_PANEL_PUT(tB, 6, rb6); VS_PANEL_PUT(tB, 7, rb7);
        if (!UNIFORM_A) {
            VS_PANEL_PUT(tA, 0, ra0); VS_PANEL_PUT(tA, 1, ra1); VS_PANEL_PUT(tA, 2, ra2); VS_PANEL_PUT(tA, 3, ra3);
            VS_PANEL_PUT(tA, 4, ra4); VS_PANEL_PUT(tA, 5, ra5); VS_PANEL_PUT(tA, 6, ra6); VS_PANEL_PUT(tA, 7, ra7);
        }
        __syncthreads();
        if (p + 1 < kD / kPanel) VS_PANEL_LOAD((p + 1) * kPanel);
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

// Certain items: exact distance of the winner.  Uncertain items: appended to the rescan list and
// their packed (distance, index) key reset.  One wave per 64 items over both directions:
// item < n0_pad -> row, else column.
template <bool MASKED>
__global__ void __launch_bounds__(64)
exact_winner_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                    const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                    const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                    const int *__restrict__ idx0, const int *__restrict__ idx1,
                    const unsigned char *__restrict__ unc0, const unsigned char *__restrict__ unc1,
                    float *__restrict__ dist0, float *__restrict__ dist1,
                    int *__restrict__ rescan_list, int *__restrict__ rescan_count,
                    unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float tA[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ const float *pA[64];
    __shared__ const float *pB[64];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x;
    const int it = blockIdx.x * 64 + lane;
    const int n0 = dims[pair].n0, n1 = dims[pair].n1;
    int i = 0, j = 0;
    bool valid = false, unc = false;
    if (it < n0_pad) {
        if (it < n0) { valid = true; i = it; j = idx0[(int64_t)pair * n0_pad + it]; unc = unc0[(int64_t)pair * n0_pad + it]; }
    } else if (it - n0_pad < n1) {
        const int c = it - n0_pad;
        valid = true; j = c; i = idx1[(int64_t)pair * n1_pad + c]; unc = unc1[(int64_t)pair * n1_pad + c];
    }
    bool masked = false;
    if (MASKED && valid && !unc)
        masked = stereo_masked(xy0[(int64_t)pair * n0_pad + i], xy1[(int64_t)pair * n1_pad + j]);
    pA[lane] = d0 + ((int64_t)pair * n0_pad + i) * kD;
    pB[lane] = d1 + ((int64_t)pair * n1_pad + j) * kD;
    __syncthreads();
    const double acc = panel_dot64<false>(pA, pB, tA, tB, lane);
    if (!valid) return;
    if (unc) {
        const int slot = atomicAdd(rescan_count, 1);
        rescan_list[2 * (int64_t)slot] = pair;
        rescan_list[2 * (int64_t)slot + 1] = it;
        keys[(int64_t)pair * (n0_pad + n1_pad) + it] = ~0ull;
        return;
    }
    const float d = masked ? 1.0f : distance_from_dot(acc);
    if (it < n0_pad) dist0[(int64_t)pair * n0_pad + it] = d;
    else dist1[(int64_t)pair * n1_pad + it - n0_pad] = d;
}

// Exact re-decision of the listed rows/columns, 64 candidates per wave task: exact distances
// from staged panels, wave minimum with lowest-index tie-break, one 64-bit atomic min of the
// packed (distance bits, index) key per task (non-negative floats order like their bit patterns).
template <bool MASKED>
__global__ void __launch_bounds__(64)
rescan_chunk_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                    const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                    const int2 *__restrict__ xy1, int n0_pad, int n1_pad, int chunks,
                    const int *__restrict__ rescan_list, const int *__restrict__ rescan_count,
                    const float *__restrict__ row_b1, const float *__restrict__ col_b1,
                    const float *__restrict__ best0, const float *__restrict__ best1,
                    unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tA[kD];
    __shared__ const float *pA[1];
    __shared__ const float *pB[64];
    const int lane = threadIdx.x;
    const int64_t n_tasks = (int64_t)(*rescan_count) * chunks;
    for (int64_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
        const int w = (int)(t / chunks), chunk = (int)(t % chunks);
        const int pair = rescan_list[2 * (int64_t)w], it = rescan_list[2 * (int64_t)w + 1];
        const int n0 = dims[pair].n0, n1 = dims[pair].n1;
        const bool is_row = it < n0_pad;
        const int self = is_row ? it : it - n0_pad;
        const int n_cand = is_row ? n1 : n0;
        if (chunk * 64 >= n_cand) continue;                          // uniform per workgroup
        // a chunk whose best approximate score is more than the band below the item's best
        // cannot contain the exact first minimum
        const float chunk_best = is_row ? row_b1[((int64_t)pair * (n1_pad / kSub) + chunk) * n0_pad + self]
                                        : col_b1[((int64_t)pair * (n0_pad / kSub) + chunk) * n1_pad + self];
        const float item_best = is_row ? best0[(int64_t)pair * n0_pad + self] : best1[(int64_t)pair * n1_pad + self];
        if (!(item_best - chunk_best <= kBand)) continue;
        const int c = chunk * 64 + lane;
        __syncthreads();
        if (lane == 0)
            pA[0] = is_row ? d0 + ((int64_t)pair * n0_pad + self) * kD : d1 + ((int64_t)pair * n1_pad + self) * kD;
        pB[lane] = is_row ? d1 + ((int64_t)pair * n1_pad + c) * kD : d0 + ((int64_t)pair * n0_pad + c) * kD;
        __syncthreads();
        const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
        float d = 3.0e38f;
        if (c < n_cand) {
            d = distance_from_dot(acc);
            if (MASKED) {
                const int i = is_row ? self : c, j = is_row ? c : self;
                if (stereo_masked(xy0[(int64_t)pair * n0_pad + i], xy1[(int64_t)pair * n1_pad + j])) d = 1.0f;
            }
        }
        unsigned long long key = c < n_cand ? ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)c : ~0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        if (lane == 0) atomicMin(&keys[(int64_t)pair * (n0_pad + n1_pad) + it], key);
    }
}

// Unpack the keys of the re-decided items into the index / distance arrays.
__global__ void __launch_bounds__(256)
unpack_kernel(int n0_pad, int n1_pad, const int *__restrict__ rescan_list,
              const int *__restrict__ rescan_count, const unsigned long long *__restrict__ keys,
              int *__restrict__ idx0, int *__restrict__ idx1, float *__restrict__ dist0,
              float *__restrict__ dist1)
{
    const int count = *rescan_count;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
        const int pair = rescan_list[2 * (int64_t)w], it = rescan_list[2 * (int64_t)w + 1];
        const unsigned long long key = keys[(int64_t)pair * (n0_pad + n1_pad) + it];
        const int idx = (int)(key & 0xffffffffu);
        const float d = __uint_as_float((unsigned)(key >> 32));
        if (it < n0_pad) { idx0[(int64_t)pair * n0_pad + it] = idx; dist0[(int64_t)pair * n0_pad + it] = d; }
        else { idx1[(int64_t)pair * n1_pad + it - n0_pad] = idx; dist1[(int64_t)pair * n1_pad + it - n0_pad] = d; }
    }
}

// Ratio-test mode only: exact scan of one whole row or column per wave (first minimum) that
// also yields the exact second-smallest distance of rows.
template <bool MASKED>
__global__ void __launch_bounds__(256)
rescan_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
              const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
              const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
              const int *__restrict__ rescan_list, const int *__restrict__ rescan_count,
              int *__restrict__ idx0, int *__restrict__ idx1, float *__restrict__ dist0,
              float *__restrict__ dist1, float *__restrict__ second0)
{
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const int count = *rescan_count;
    for (int w = wave_global; w < count; w += n_waves) {
        const int pair = rescan_list[2 * (int64_t)w], it = rescan_list[2 * (int64_t)w + 1];
        const int n0 = dims[pair].n0, n1 = dims[pair].n1;
        const bool is_row = it < n0_pad;
        const int self = is_row ? it : it - n0_pad;
        const int n_cand = is_row ? n1 : n0;
        const float *base_self = is_row ? d0 + ((int64_t)pair * n0_pad + self) * kD
                                        : d1 + ((int64_t)pair * n1_pad + self) * kD;
        float bd = 3.0e38f, sd = 3.0e38f;
        int bi = 0x7fffffff;
        for (int c = lane; c < n_cand; c += 64) {
            float d;
            const int i = is_row ? self : c, j = is_row ? c : self;
            if (MASKED && stereo_masked(xy0[(int64_t)pair * n0_pad + i], xy1[(int64_t)pair * n1_pad + j])) d = 1.0f;
            else {
                const float *other = is_row ? d1 + ((int64_t)pair * n1_pad + c) * kD
                                            : d0 + ((int64_t)pair * n0_pad + c) * kD;
                d = is_row ? exact_distance(base_self, other) : exact_distance(other, base_self);
            }
            if (d < bd) { sd = bd; bd = d; bi = c; }
            else if (d < sd) sd = d;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64), os = __shfl_xor(sd, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (od < bd || (od == bd && oi < bi)) { sd = fminf(bd, os); bd = od; bi = oi; }
            else sd = fminf(sd, od);
        }
        if (lane == 0) {
            if (is_row) {
                idx0[(int64_t)pair * n0_pad + self] = bi; dist0[(int64_t)pair * n0_pad + self] = bd;
                second0[(int64_t)pair * n0_pad + self] = sd;
            } else {
                idx1[(int64_t)pair * n1_pad + self] = bi; dist1[(int64_t)pair * n1_pad + self] = bd;
            }
        }
    }
}

// Mutual check, threshold (and optional ratio test), ordered compaction.  One workgroup per pair.
__global__ void __launch_bounds__(1024)
compact_kernel(const PairDims *__restrict__ dims, int n0_pad, int n1_pad,
               const int *__restrict__ idx0, const int *__restrict__ idx1,
               const float *__restrict__ dist0, const float *__restrict__ second0, float nn_thresh,
               float ratio, int *__restrict__ out_i, int *__restrict__ out_j,
               float *__restrict__ out_d, int *__restrict__ out_n)
{
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int pair = blockIdx.x;
    const int n0 = dims[pair].n0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n0; start += 1024) {
        const int i = start + tid;
        bool keep = false;
        int j = 0;
        float d = 0.f;
        if (i < n0) {
            j = idx0[(int64_t)pair * n0_pad + i];
            d = dist0[(int64_t)pair * n0_pad + i];
            keep = idx1[(int64_t)pair * n1_pad + j] == i && d < nn_thresh;
            if (keep && ratio < 1.0f) keep = d < ratio * second0[(int64_t)pair * n0_pad + i];
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (keep) {
            const int64_t o = (int64_t)pair * n0_pad + off + before;
            out_i[o] = i; out_j[o] = j; out_d[o] = d;
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int w = 0; w < 16; w++) s += wave_tot[w]; base_s += s; }
        __syncthreads();
    }
    if (tid == 0) out_n[pair] = base_s;
}

int launch_ok(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

}  // namespace

struct vs_matcher {
    int max_pairs = 0, n0_pad = 0, n1_pad = 0;
    hipStream_t stream = nullptr;
    std::vector<PairDims> h_dims;
    std::vector<unsigned char> h_masked;
    bool ran = false;
    hipEvent_t *prof_ev = nullptr;   // when set: 6 events recorded around vs_matcher_run's stages
    DevBuf<float> d0, d1;
    DevBuf<_Float16> a_hi, a_lo, b_hi, b_lo;
    DevBuf<int2> xy0, xy1;
    DevBuf<PairDims> dims;
    DevBuf<float> row_b1, row_b2, col_b1, col_b2, dist0, dist1, second0, out_d, best0, best1;
    DevBuf<int> row_i1, col_i1, idx0, idx1, rescan_list, rescan_count, out_i, out_j, out_n;
    DevBuf<unsigned char> unc0, unc1;
    DevBuf<unsigned long long> keys;
};

extern "C" {

int vs_matcher_create(vs_matcher_handle *out, int32_t max_pairs, int32_t max_n0, int32_t max_n1)
{
    VS_CHECK(out != nullptr, VS_ERR_INVALID, "null handle pointer");
    *out = nullptr;
    VS_CHECK(max_pairs > 0 && max_n0 > 0 && max_n1 > 0, VS_ERR_INVALID, "sizes must be positive");
    int ndev = 0;
    VS_HIP(hipGetDeviceCount(&ndev));
    VS_CHECK(ndev > 0, VS_ERR_HIP, "no HIP device available");
    vs_matcher *m = new (std::nothrow) vs_matcher();
    VS_CHECK(m != nullptr, VS_ERR_NOMEM, "out of host memory");
    m->max_pairs = max_pairs;
    m->n0_pad = (max_n0 + kBM - 1) / kBM * kBM;
    m->n1_pad = (max_n1 + kBN - 1) / kBN * kBN;
    const size_t P = (size_t)max_pairs, r0 = (size_t)m->n0_pad, r1 = (size_t)m->n1_pad;
    int rc = VS_OK;
    auto A = [&](int r) { if (rc == VS_OK) rc = r; };
    A(m->d0.alloc(P * r0 * kD)); A(m->d1.alloc(P * r1 * kD));
    A(m->a_hi.alloc(P * r0 * kD)); A(m->a_lo.alloc(P * r0 * kD));
    A(m->b_hi.alloc(P * r1 * kD)); A(m->b_lo.alloc(P * r1 * kD));
    A(m->xy0.alloc(P * r0)); A(m->xy1.alloc(P * r1)); A(m->dims.alloc(P));
    const size_t rp = P * (r1 / kSub) * r0, cp = P * (r0 / kSub) * r1;
    A(m->row_b1.alloc(rp)); A(m->row_b2.alloc(rp)); A(m->row_i1.alloc(rp));
    A(m->col_b1.alloc(cp)); A(m->col_b2.alloc(cp)); A(m->col_i1.alloc(cp));
    A(m->idx0.alloc(P * r0)); A(m->idx1.alloc(P * r1)); A(m->dist0.alloc(P * r0)); A(m->dist1.alloc(P * r1));
    A(m->second0.alloc(P * r0)); A(m->best0.alloc(P * r0)); A(m->best1.alloc(P * r1)); A(m->unc0.alloc(P * r0)); A(m->unc1.alloc(P * r1));
    A(m->rescan_list.alloc(2 * P * (r0 + r1))); A(m->rescan_count.alloc(4)); A(m->keys.alloc(P * (r0 + r1)));
    A(m->out_i.alloc(P * r0)); A(m->out_j.alloc(P * r0)); A(m->out_d.alloc(P * r0)); A(m->out_n.alloc(P));
    if (rc != VS_OK) { delete m; return rc; }
    m->h_dims.assign(P, PairDims{ 0, 0 });
    m->h_masked.assign(P, 0);
    *out = m;
    return VS_OK;
}

int vs_matcher_destroy(vs_matcher_handle m)
{
    if (m) { (void)hipStreamSynchronize(m->stream); delete m; }
    return VS_OK;
}

int vs_matcher_set_stream(vs_matcher_handle m, void *stream)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    m->stream = (hipStream_t)stream;
    return VS_OK;
}

int vs_matcher_set_pair(vs_matcher_handle m, int32_t pair, const float *desc0, int32_t n0,
                        const float *desc1, int32_t n1, const int32_t *xy0, const int32_t *xy1)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(pair >= 0 && pair < m->max_pairs, VS_ERR_INVALID, "pair %d out of range [0,%d)", pair, m->max_pairs);
    VS_CHECK(n0 >= 0 && n0 <= m->n0_pad && n1 >= 0 && n1 <= m->n1_pad, VS_ERR_INVALID,
             "pair sizes %d x %d exceed the matcher's capacity %d x %d", n0, n1, m->n0_pad, m->n1_pad);
    VS_CHECK((n0 == 0 || desc0) && (n1 == 0 || desc1), VS_ERR_INVALID, "null descriptor pointer");
    VS_CHECK((xy0 == nullptr) == (xy1 == nullptr), VS_ERR_INVALID, "xy0 and xy1 must both be given or both be NULL");
    hipStream_t s = m->stream;
    float *g0 = m->d0.p + (size_t)pair * m->n0_pad * kD, *g1 = m->d1.p + (size_t)pair * m->n1_pad * kD;
    if (n0) VS_HIP(hipMemcpyAsync(g0, desc0, (size_t)n0 * kD * sizeof(float), hipMemcpyDefault, s));
    if (n1) VS_HIP(hipMemcpyAsync(g1, desc1, (size_t)n1 * kD * sizeof(float), hipMemcpyDefault, s));
    if (xy0) {
        if (n0) VS_HIP(hipMemcpyAsync(m->xy0.p + (size_t)pair * m->n0_pad, xy0, (size_t)n0 * sizeof(int2), hipMemcpyDefault, s));
        if (n1) VS_HIP(hipMemcpyAsync(m->xy1.p + (size_t)pair * m->n1_pad, xy1, (size_t)n1 * sizeof(int2), hipMemcpyDefault, s));
    }
    const size_t e0 = (size_t)m->n0_pad * kD, e1 = (size_t)m->n1_pad * kD;
    hipLaunchKernelGGL(split_kernel, dim3(div_up((int64_t)e0 / 4, 256)), dim3(256), 0, s, g0, n0, m->n0_pad,
                       m->a_hi.p + (size_t)pair * e0, m->a_lo.p + (size_t)pair * e0);
    hipLaunchKernelGGL(split_kernel, dim3(div_up((int64_t)e1 / 4, 256)), dim3(256), 0, s, g1, n1, m->n1_pad,
                       m->b_hi.p + (size_t)pair * e1, m->b_lo.p + (size_t)pair * e1);
    VS_TRY(launch_ok("split_kernel"));
    m->h_dims[pair] = PairDims{ n0, n1 };
    m->h_masked[pair] = xy0 != nullptr;
    VS_HIP(hipMemcpyAsync(m->dims.p + pair, &m->h_dims[pair], sizeof(PairDims), hipMemcpyHostToDevice, s));
    VS_HIP(hipStreamSynchronize(s));
    m->ran = false;
    return VS_OK;
}

int vs_matcher_run(vs_matcher_handle m, int32_t n_pairs, float nn_thresh, float ratio)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(n_pairs > 0 && n_pairs <= m->max_pairs, VS_ERR_INVALID, "n_pairs %d out of range (1..%d)", n_pairs, m->max_pairs);
    bool masked = m->h_masked[0];
    for (int p = 1; p < n_pairs; p++)
        VS_CHECK((bool)m->h_masked[p] == masked, VS_ERR_INVALID, "a batch must be all two-way or all stereo pairs");
    hipStream_t s = m->stream;
    auto mark = [&](int i) { if (m->prof_ev) (void)hipEventRecord(m->prof_ev[i], s); };
    const int r0 = m->n0_pad, r1 = m->n1_pad;
    const int tiles = (r0 / kBM) * (r1 / kBN);
    mark(0);
    const dim3 grid(tiles, 1, n_pairs);
    if (masked)
        hipLaunchKernelGGL(dot_top2_kernel<true>, grid, dim3(256), 0, s, m->a_hi.p, m->a_lo.p, m->b_hi.p, m->b_lo.p,
                           m->dims.p, m->xy0.p, m->xy1.p, r0, r1, m->row_b1.p, m->row_i1.p, m->row_b2.p,
                           m->col_b1.p, m->col_i1.p, m->col_b2.p);
    else
        hipLaunchKernelGGL(dot_top2_kernel<false>, grid, dim3(256), 0, s, m->a_hi.p, m->a_lo.p, m->b_hi.p, m->b_lo.p,
                           m->dims.p, m->xy0.p, m->xy1.p, r0, r1, m->row_b1.p, m->row_i1.p, m->row_b2.p,
                           m->col_b1.p, m->col_i1.p, m->col_b2.p);
    VS_TRY(launch_ok("dot_top2_kernel"));
    mark(1);
    const int force = ratio < 1.0f ? 1 : 0;   // the ratio test needs every row's exact second best
    hipLaunchKernelGGL(merge_kernel, dim3(div_up(r0, 256), n_pairs), dim3(256), 0, s, m->dims.p, r0, r1 / kSub, 0,
                       m->row_b1.p, m->row_i1.p, m->row_b2.p, m->idx0.p, m->unc0.p, m->best0.p, force);
    hipLaunchKernelGGL(merge_kernel, dim3(div_up(r1, 256), n_pairs), dim3(256), 0, s, m->dims.p, r1, r0 / kSub, 1,
                       m->col_b1.p, m->col_i1.p, m->col_b2.p, m->idx1.p, m->unc1.p, m->best1.p, 0);
    VS_TRY(launch_ok("merge_kernel"));
    mark(2);
    VS_HIP(hipMemsetAsync(m->rescan_count.p, 0, 4 * sizeof(int), s));
    const dim3 egrid(div_up(r0 + r1, 64), n_pairs);
    const int chunks = std::max(r0, r1) / 64;
    if (masked)
        hipLaunchKernelGGL(exact_winner_kernel<true>, egrid, dim3(64), 0, s, m->d0.p, m->d1.p, m->dims.p, m->xy0.p,
                           m->xy1.p, r0, r1, m->idx0.p, m->idx1.p, m->unc0.p, m->unc1.p, m->dist0.p, m->dist1.p,
                           m->rescan_list.p, m->rescan_count.p, m->keys.p);
    else
        hipLaunchKernelGGL(exact_winner_kernel<false>, egrid, dim3(64), 0, s, m->d0.p, m->d1.p, m->dims.p, m->xy0.p,
                           m->xy1.p, r0, r1, m->idx0.p, m->idx1.p, m->unc0.p, m->unc1.p, m->dist0.p, m->dist1.p,
                           m->rescan_list.p, m->rescan_count.p, m->keys.p);
    if (force) {
        // ratio test: every row needs its exact second-best distance -> whole-row scans
        if (masked)
            hipLaunchKernelGGL(rescan_kernel<true>, dim3(2048), dim3(256), 0, s, m->d0.p, m->d1.p, m->dims.p, m->xy0.p,
                               m->xy1.p, r0, r1, m->rescan_list.p, m->rescan_count.p, m->idx0.p, m->idx1.p,
                               m->dist0.p, m->dist1.p, m->second0.p);
        else
            hipLaunchKernelGGL(rescan_kernel<false>, dim3(2048), dim3(256), 0, s, m->d0.p, m->d1.p, m->dims.p, m->xy0.p,
                               m->xy1.p, r0, r1, m->rescan_list.p, m->rescan_count.p, m->idx0.p, m->idx1.p,
                               m->dist0.p, m->dist1.p, m->second0.p);
    } else {
        if (masked)
            hipLaunchKernelGGL(rescan_chunk_kernel<true>, dim3(16384), dim3(64), 0, s, m->d0.p, m->d1.p, m->dims.p,
                               m->xy0.p, m->xy1.p, r0, r1, chunks, m->rescan_list.p, m->rescan_count.p, m->row_b1.p, m->col_b1.p,
                               m->best0.p, m->best1.p, m->keys.p);
        else
            hipLaunchKernelGGL(rescan_chunk_kernel<false>, dim3(16384), dim3(64), 0, s, m->d0.p, m->d1.p, m->dims.p,
                               m->xy0.p, m->xy1.p, r0, r1, chunks, m->rescan_list.p, m->rescan_count.p, m->row_b1.p, m->col_b1.p,
                               m->best0.p, m->best1.p, m->keys.p);
        hipLaunchKernelGGL(unpack_kernel, dim3(64), dim3(256), 0, s, r0, r1, m->rescan_list.p, m->rescan_count.p,
                           m->keys.p, m->idx0.p, m->idx1.p, m->dist0.p, m->dist1.p);
    }
    VS_TRY(launch_ok("exact kernels"));
    mark(3);
    hipLaunchKernelGGL(compact_kernel, dim3(n_pairs), dim3(1024), 0, s, m->dims.p, r0, r1, m->idx0.p, m->idx1.p,
                       m->dist0.p, m->second0.p, nn_thresh, ratio, m->out_i.p, m->out_j.p, m->out_d.p, m->out_n.p);
    VS_TRY(launch_ok("compact_kernel"));
    mark(4);
    m->ran = true;
    return VS_OK;
}

int vs_matcher_profile_run(vs_matcher_handle m, int32_t n_pairs, float nn_thresh, float ratio,
                           int32_t iters, float *avg_ms)
{
    VS_CHECK(m && avg_ms && iters > 0, VS_ERR_INVALID, "bad profiling arguments");
    hipEvent_t ev[5];
    for (auto &e : ev) VS_HIP(hipEventCreate(&e));
    for (int k = 0; k < 5; k++) avg_ms[k] = 0.f;
    int rc = vs_matcher_run(m, n_pairs, nn_thresh, ratio);
    for (int it = 0; it < iters && rc == VS_OK; it++) {
        m->prof_ev = ev;
        rc = vs_matcher_run(m, n_pairs, nn_thresh, ratio);
        m->prof_ev = nullptr;
        if (rc != VS_OK) break;
        if (hipEventSynchronize(ev[4]) != hipSuccess) { set_error("event sync failed"); rc = VS_ERR_HIP; break; }
        for (int k = 0; k < 4; k++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
            avg_ms[k] += ms;
        }
        float tot = 0.f;
        (void)hipEventElapsedTime(&tot, ev[0], ev[4]);
        avg_ms[4] += tot;
    }
    for (int k = 0; k < 5; k++) avg_ms[k] /= (float)iters;
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

int vs_matcher_get_matches(vs_matcher_handle m, int32_t pair, int32_t *out_i, int32_t *out_j,
                           float *out_d, int32_t *out_n)
{
    VS_CHECK(m && out_n, VS_ERR_INVALID, "null argument");
    VS_CHECK(m->ran, VS_ERR_STATE, "call vs_matcher_run first");
    VS_CHECK(pair >= 0 && pair < m->max_pairs, VS_ERR_INVALID, "pair %d out of range", pair);
    hipStream_t s = m->stream;
    int n = 0;
    VS_HIP(hipMemcpyAsync(&n, m->out_n.p + pair, sizeof(int), hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    const size_t o = (size_t)pair * m->n0_pad;
    if (n > 0) {
        if (out_i) VS_HIP(hipMemcpyAsync(out_i, m->out_i.p + o, (size_t)n * sizeof(int), hipMemcpyDefault, s));
        if (out_j) VS_HIP(hipMemcpyAsync(out_j, m->out_j.p + o, (size_t)n * sizeof(int), hipMemcpyDefault, s));
        if (out_d) VS_HIP(hipMemcpyAsync(out_d, m->out_d.p + o, (size_t)n * sizeof(float), hipMemcpyDefault, s));
        VS_HIP(hipStreamSynchronize(s));
    }
    *out_n = n;
    return VS_OK;
}

int vs_matcher_get_nn(vs_matcher_handle m, int32_t pair, int32_t *idx0, int32_t *idx1)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(m->ran, VS_ERR_STATE, "call vs_matcher_run first");
    VS_CHECK(pair >= 0 && pair < m->max_pairs, VS_ERR_INVALID, "pair %d out of range", pair);
    hipStream_t s = m->stream;
    const PairDims d = m->h_dims[pair];
    if (idx0 && d.n0) VS_HIP(hipMemcpyAsync(idx0, m->idx0.p + (size_t)pair * m->n0_pad, (size_t)d.n0 * sizeof(int), hipMemcpyDefault, s));
    if (idx1 && d.n1) VS_HIP(hipMemcpyAsync(idx1, m->idx1.p + (size_t)pair * m->n1_pad, (size_t)d.n1 * sizeof(int), hipMemcpyDefault, s));
    VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

int vs_matcher_get_rescans(vs_matcher_handle m, int64_t *n)
{
    VS_CHECK(m && n, VS_ERR_INVALID, "null argument");
    VS_CHECK(m->ran, VS_ERR_STATE, "call vs_matcher_run first");
    int c = 0;
    VS_HIP(hipMemcpyAsync(&c, m->rescan_count.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipStreamSynchronize(m->stream));
    *n = c;
    return VS_OK;
}

static int match_once(const float *d0, const int32_t *xy0, int32_t n0, const float *d1,
                      const int32_t *xy1, int32_t n1, float nn_thresh, float ratio, int32_t *out_i,
                      int32_t *out_j, float *out_d, int32_t *out_n)
{
    VS_CHECK(out_n != nullptr, VS_ERR_INVALID, "null out_n");
    VS_CHECK(n0 >= 0 && n1 >= 0, VS_ERR_INVALID, "negative size");
    *out_n = 0;
    if (n0 == 0 || n1 == 0) return VS_OK;   // the reference's loops do nothing on empty inputs
    vs_matcher_handle m = nullptr;
    VS_TRY(vs_matcher_create(&m, 1, n0, n1));
    int rc = vs_matcher_set_pair(m, 0, d0, n0, d1, n1, xy0, xy1);
    if (rc == VS_OK) rc = vs_matcher_run(m, 1, nn_thresh, ratio);
    if (rc == VS_OK) rc = vs_matcher_get_matches(m, 0, out_i, out_j, out_d, out_n);
    vs_matcher_destroy(m);
    return rc;
}

int vs_match_two_way(const float *desc0, int32_t n0, const float *desc1, int32_t n1,
                     float nn_thresh, float ratio, int32_t *out_i, int32_t *out_j, float *out_d,
                     int32_t *out_n)
{
    return match_once(desc0, nullptr, n0, desc1, nullptr, n1, nn_thresh, ratio, out_i, out_j, out_d, out_n);
}

int vs_match_stereo(const float *desc0, const int32_t *xy0, int32_t n0, const float *desc1,
                    const int32_t *xy1, int32_t n1, float nn_thresh, float ratio, int32_t *out_i,
                    int32_t *out_j, float *out_d, int32_t *out_n)
{
    if (out_n && (n0 == 0 || n1 == 0)) { *out_n = 0; return VS_OK; }
    VS_CHECK(xy0 && xy1, VS_ERR_INVALID, "stereo matching needs keypoint coordinates");
    return match_once(desc0, xy0, n0, desc1, xy1, n1, nn_thresh, ratio, out_i, out_j, out_d, out_n);
}

}  // extern "C"
