// SuperPoint descriptor matcher for gfx950 (reference interfaces: include/vslam_hip.h).
//
// Exactness plan.  The reference decides nearest neighbours on
//   d(i,j) = sqrtf(2 - 2*clamp((float)dot64(desc0_i, desc1_j), -1, 1)),  first minimum wins.
// The dense 4096 x 4096 x 256 contraction runs on the matrix cores with fp16 operands: every
// descriptor (scaled by 2^8) is rounded to fp16 and the kernel accumulates the products in fp32.
// The rounding kernel also measures, per frame pair, the largest rounding residue ||a - a16|| and
// the largest ||a16|| of each side, which give a certified bound eps on |s~ - s| for every entry
// of that pair (Cauchy-Schwarz on the residues + fp32 accumulation of 256 products + the index
// packing of the epilogue).  A fused epilogue keeps, per row and per column, the best and
// second-best s~.  Where best - second exceeds 2*eps + kCollapse (kCollapse covers distinct dots
// that the fp32 clamp/sqrt chain maps to one float), the approximate winner provably is the
// reference's first minimum and only its distance is recomputed exactly; otherwise the row/column
// is re-decided by an exact scan that replays the reference arithmetic (sequential fp64
// accumulation, narrowing, clamp, sqrtf).  Indices and distances are therefore bit-identical to
// that arithmetic.  cv::Mat::dot itself is not in the reference tree: builds of OpenCV that
// accumulate in float SIMD lanes round differently, so parity with OpenCV's dot is unpinned.
// Descriptors need not have unit norm (scores beyond the clamp are treated as the ties they are),
// but their entries must stay below 255 in magnitude (fp16 range after the 2^8 prescale).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "match_exact.h"

using namespace vs;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = VS_DESC_DIM;        // 256
#ifndef VS_WAVE_ROWS
#define VS_WAVE_ROWS 4
#endif
constexpr int kWR = VS_WAVE_ROWS, kWC = 4 / kWR;   // the 4 waves of a workgroup as kWR x kWC sub-tiles of 64 x 64
constexpr int kBM = 64 * kWR, kBN = 64 * kWC;      // workgroup tile
#ifndef VS_BK
#define VS_BK 128
#endif
constexpr int kBK = VS_BK;             // halfs per K-step (one ring stage)
#ifndef VS_RING
#define VS_RING 2
#endif
// Two stages (32 KB) + the transposition areas = 66 KB: two workgroups per CU, two waves per SIMD.
constexpr int kRing = VS_RING;         // LDS ring stages (kRing - 1 K-steps of lookahead)
constexpr int kSub = 64;               // rows/cols covered by one wave = granularity of partials
constexpr float kScale = 256.0f;       // descriptor prescale (2^8), dot prescale 2^16
constexpr float kDotScale = 65536.0f;
constexpr float kCollapse = 1.0e-6f;   // dots closer than this may share one fp32 distance
// relative (to ||a16|| ||b16||) error of the fp32 accumulation of 256 products (256 * 2^-23) and of
// the epilogue's index packing into the low 6 mantissa bits (2^-17)
constexpr float kAccumRel = 256.0f * 1.1920929e-7f + 7.6293945e-6f;
constexpr float kNegInf = -3.0e38f;
constexpr float kMaskedDot = 0.5f * kDotScale;   // distance 1.0  <=>  dot 0.5

struct PairDims {
    int n0, n1;
};

constexpr int kSplitRows = 32;          // rows per workgroup of split_kernel

// fp32 rows -> fp16 rows scaled by 2^8 (n_pad rows, rows >= n zero-filled).  One wave per row,
// kSplitRows rows per workgroup.
// stats[0] / stats[1] receive the maxima over rows of ||x - x16||^2 and ||x16||^2 (scaled units),
// as float bit patterns (non-negative floats order like unsigned integers).
__global__ void __launch_bounds__(256)
split_kernel(const float *__restrict__ src, int n, int n_pad, _Float16 *__restrict__ hi,
             unsigned *__restrict__ stats)
{
    __shared__ float wmax[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float res_max = 0.f, nrm_max = 0.f;
    for (int q = 0; q < kSplitRows / 4; q++) {
        const int row = blockIdx.x * kSplitRows + q * 4 + wave;
        if (row >= n_pad) break;
        const int64_t e = (int64_t)row * kD + lane * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) v = *reinterpret_cast<const float4 *>(src + e);
        const float x[4] = { v.x * kScale, v.y * kScale, v.z * kScale, v.w * kScale };
        _Float16 h[4];
        float res2 = 0.f, nrm2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            h[k] = (_Float16)x[k];
            const float hf = (float)h[k], r = x[k] - hf;    // the residue is exact in fp32
            res2 = fmaf(r, r, res2);
            nrm2 = fmaf(hf, hf, nrm2);
        }
        *reinterpret_cast<uint2 *>(hi + e) = *reinterpret_cast<uint2 *>(h);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            res2 += __shfl_xor(res2, off, 64);
            nrm2 += __shfl_xor(nrm2, off, 64);
        }
        res_max = fmaxf(res_max, res2);
        nrm_max = fmaxf(nrm_max, nrm2);
    }
    if (lane == 0) { wmax[0][wave] = res_max; wmax[1][wave] = nrm_max; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float m = fmaxf(fmaxf(wmax[threadIdx.x][0], wmax[threadIdx.x][1]),
                              fmaxf(wmax[threadIdx.x][2], wmax[threadIdx.x][3]));
        atomicMax(stats + threadIdx.x, __float_as_uint(m));
    }
}

// The same rounding for a run of frame pairs in one launch (vs_matcher_set_pair / set_pairs):
// grid = (row blocks, side, pair).  Side 0 reads pair p's rows at src0 + p * stride0 (floats), side 1
// at src1 + p * stride1; the rows are also copied into the handle's own fp32 arrays (copy0 / copy1,
// pair pitch n_pad * 256) unless they already are those arrays.  Instead of atomics on the
// statistics every workgroup stores its two maxima to `part` ([pair][side][block][2]);
// split_finish_kernel reduces them, so nothing has to be zeroed first.
__global__ void __launch_bounds__(256)
split_pairs_kernel(const float *__restrict__ src0, int64_t stride0, int n0, int n0_pad,
                   const float *__restrict__ src1, int64_t stride1, int n1, int n1_pad,
                   float *__restrict__ copy0, float *__restrict__ copy1, _Float16 *__restrict__ a16,
                   _Float16 *__restrict__ b16, int first_pair, int part_blocks, float *__restrict__ part)
{
    __shared__ float wmax[2][4];
    const int side = blockIdx.y, p = blockIdx.z, pair = first_pair + p;
    const int n = side ? n1 : n0, n_pad = side ? n1_pad : n0_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float res_max = 0.f, nrm_max = 0.f;
    if ((int)blockIdx.x * kSplitRows < n_pad) {
        const float *src = side ? src1 + (int64_t)p * stride1 : src0 + (int64_t)p * stride0;
        float *copy = (side ? copy1 : copy0) + (int64_t)pair * n_pad * kD;
        _Float16 *hi = (side ? b16 : a16) + (int64_t)pair * n_pad * kD;
        for (int q = 0; q < kSplitRows / 4; q++) {
            const int row = blockIdx.x * kSplitRows + q * 4 + wave;
            if (row >= n_pad) break;
            const int64_t e = (int64_t)row * kD + lane * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < n) {
                v = *reinterpret_cast<const float4 *>(src + e);
                if (src != copy) *reinterpret_cast<float4 *>(copy + e) = v;
            }
            const float x[4] = { v.x * kScale, v.y * kScale, v.z * kScale, v.w * kScale };
            _Float16 h[4];
            float res2 = 0.f, nrm2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                h[k] = (_Float16)x[k];
                const float hf = (float)h[k], r = x[k] - hf;    // the residue is exact in fp32
                res2 = fmaf(r, r, res2);
                nrm2 = fmaf(hf, hf, nrm2);
            }
            *reinterpret_cast<uint2 *>(hi + e) = *reinterpret_cast<uint2 *>(h);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                res2 += __shfl_xor(res2, off, 64);
                nrm2 += __shfl_xor(nrm2, off, 64);
            }
            res_max = fmaxf(res_max, res2);
            nrm_max = fmaxf(nrm_max, nrm2);
        }
    }
    if (lane == 0) { wmax[0][wave] = res_max; wmax[1][wave] = nrm_max; }
    __syncthreads();
    if (threadIdx.x < 2)
        part[(((int64_t)p * 2 + side) * part_blocks + blockIdx.x) * 2 + threadIdx.x] =
            fmaxf(fmaxf(wmax[threadIdx.x][0], wmax[threadIdx.x][1]), fmaxf(wmax[threadIdx.x][2], wmax[threadIdx.x][3]));
}

// One wave per pair: the maxima over the workgroups of split_pairs_kernel become the pair's four
// statistics, and the pair's sizes are written.
__global__ void __launch_bounds__(64)
split_finish_kernel(int first_pair, int part_blocks, const float *__restrict__ part, int n0, int n1,
                    unsigned *__restrict__ stats, PairDims *__restrict__ dims)
{
    const int p = blockIdx.x, pair = first_pair + p, lane = threadIdx.x;
    float m[4] = { 0.f, 0.f, 0.f, 0.f };   // side 0 residue, side 0 norm, side 1 residue, side 1 norm
    for (int b = lane; b < part_blocks; b += 64)
#pragma unroll
        for (int k = 0; k < 4; k++)
            m[k] = fmaxf(m[k], part[(((int64_t)p * 2 + (k >> 1)) * part_blocks + b) * 2 + (k & 1)]);
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], off, 64));
    if (lane < 4) stats[4 * (int64_t)pair + lane] = __float_as_uint(lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3]);
    if (lane == 0) { dims[pair].n0 = n0; dims[pair].n1 = n1; }
}

// Certainty band of one frame pair in units of s~ (= 2^16 * dot): twice the bound on |s~ - s|
// plus the collapse margin.  stats = { max||a-a16||^2, max||a16||^2, max||b-b16||^2, max||b16||^2 }.
__device__ __forceinline__ float pair_band(const unsigned *__restrict__ stats, int pair)
{
    const uint4 q = *reinterpret_cast<const uint4 *>(stats + 4 * (int64_t)pair);
    // 0.2 % inflation covers the fp32 rounding of the sums of squares and of this formula
    const float ea = sqrtf(__uint_as_float(q.x)) * 1.002f, na = sqrtf(__uint_as_float(q.y)) * 1.002f;
    const float eb = sqrtf(__uint_as_float(q.z)) * 1.002f, nb = sqrtf(__uint_as_float(q.w)) * 1.002f;
    const float eps = ea * (nb + eb) + na * eb + kAccumRel * na * nb;
    return 2.0f * eps * 1.002f + kCollapse * kDotScale;
}

// The reference clamps the dot to [-1, 1] before the distance, so every score beyond the clamp
// ties (the first index wins): certainty tests compare scores clamped the same way.
__device__ __forceinline__ float clamp_score(float s) { return fminf(fmaxf(s, -kDotScale), kDotScale); }

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
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// union of two (best, second) pairs of packed scores
__device__ __forceinline__ void top2_union(float &b1, float &b2, float ob1, float ob2)
{
    const float lo = fminf(b1, ob1);
    b1 = fmaxf(b1, ob1);
    b2 = fmaxf(lo, fmaxf(b2, ob2));
}

// LDS image of one streamed desc1 K-step: kBN rows (columns of the score tile) x kBK halfs in
// unpadded rows whose 16-byte chunks are XOR-swizzled with the row number so that the
// ds_read_b128 fragment reads are conflict-free (rows sharing a 256-byte bank line get different
// chunk slots).  The image is filled by LDS-DMA (global_load_lds_dwordx4: destination =
// wave-uniform base + 16 * lane), so the swizzle is applied to the per-lane SOURCE address and
// again to the read address.
constexpr int kStageBytes = kBN * kBK * 2;
constexpr int kChunksPerRow = kBK / 8;                  // 16-byte chunks per image row
constexpr int kRowsPerDma = 64 / kChunksPerRow;         // image rows per DMA wave-instruction
constexpr int kRowsPerWave = kBN / 4;                   // image rows one wave fills per stage
constexpr int kDmaPerWave = kRowsPerWave / kRowsPerDma; // DMA instructions per wave and stage
constexpr int kSwzShift = kBK == 128 ? 0 : (kBK == 64 ? 1 : 2);   // log2(rows per 256-byte bank line)
// DMAs allowed in flight behind stage s when it is read: those of the kRing - 2 younger stages
constexpr int kInflight = (kRing - 2) * kDmaPerWave;
static_assert(kBK == 32 || kBK == 64 || kBK == 128, "swizzle is written for 64-, 128- and 256-byte image rows");
__device__ __forceinline__ int stage_swizzle(int row) { return (row >> kSwzShift) & (kChunksPerRow - 1); }
constexpr int kTS = 68;                                 // transpose row stride in floats
#ifdef VS_EXP_NO_EPILOGUE
constexpr int kDotLds = kRing * kStageBytes;
#else
constexpr int kDotLds = kRing * kStageBytes + 4 * 32 * kTS * 4;
#endif
static_assert(kDotLds <= 160 * 1024, "the ring and the transposition areas fit the CU's LDS");

// One workgroup = one band: kBM rows of desc0 of one frame pair against `tiles_per_wg` consecutive
// kBN-column tiles of desc1.  4 waves, each a 64 x 64 sub-tile as 2 x 2 MFMA 32x32x16 f16 tiles.
// Every wave keeps the fp16 fragments of its 64 desc0 rows (all 256 k) in registers for the whole
// band; desc1 streams through a kRing-stage LDS ring, one barrier per K-step, the DMAs of the next
// kRing - 1 K-steps (also across tile boundaries) in flight behind the MFMAs and the epilogue of
// the current one.  Epilogue: per-column top-2 over the wave's 64 rows (in registers) and per-row top-2 over
// its 64 columns (transposed through a wave-private LDS area, no barrier).
#ifdef VS_MATCH_STAMPS      // timeline instrumentation build: cycle stamps of one workgroup's first tiles
__device__ unsigned long long g_match_stamps[4 * 64];
#define VS_STAMP()                                                                              \
    do {                                                                                        \
        if (stamp_on && n_stamp < 64) {                                                         \
            const unsigned long long t_ = __builtin_readcyclecounter();                         \
            if (lane == 0) s_stamps[wave * 64 + n_stamp] = t_;                                  \
            n_stamp++;                                                                          \
        }                                                                                       \
    } while (0)
#else
#define VS_STAMP() do { } while (0)
#endif

template <bool MASKED>
#ifndef VS_DOT_WPE
#define VS_DOT_WPE 2
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VS_DOT_WPE, VS_DOT_WPE)))
dot_top2_kernel(const _Float16 *__restrict__ a16, const _Float16 *__restrict__ b16,
                const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                const int2 *__restrict__ xy1, int n0_pad, int n1_pad, int tiles_per_wg,
                float2 *__restrict__ row_part, float2 *__restrict__ col_part, float4 *__restrict__ row_sum)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[kDotLds];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef VS_MATCH_STAMPS
    __shared__ unsigned long long s_stamps[4 * 64];
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 8;
    int n_stamp = 0;
#endif
    const int wr = wave / kWC, wc = wave % kWC;
    const int half = lane >> 5, l31 = lane & 31;

    // band decode; workgroups that share an XCD (blockIdx.x % 8) take a contiguous range of
    // bands, so an XCD's L2 holds the desc1 rows of one or two frame pairs at a time
    const int ntr = n0_pad / kBM, ntc = n1_pad / kBN;
    const int splits = (ntc + tiles_per_wg - 1) / tiles_per_wg;
    const int bands_per_pair = ntr * splits;
    int band;
    {
        const int total = gridDim.x, x = blockIdx.x & 7, w = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        band = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + w;
    }
    const int pair = band / bands_per_pair;
    const int tile_r = (band % bands_per_pair) / splits, split = band % splits;
    const int n0 = dims[pair].n0, n1 = dims[pair].n1;
    const int i0 = tile_r * kBM;
    const int c_begin = split * tiles_per_wg;
    const int c_end = min(min(ntc, c_begin + tiles_per_wg), (n1 + kBN - 1) / kBN);
    if (i0 >= n0 || c_begin >= c_end) return;   // band outside this pair: nothing reads its partials

    // desc0 fragments of a 32x32x16 MFMA: lane holds 8 consecutive k of row (lane & 31), k offset
    // 8 * (lane >> 5); 2 row tiles x 16 k-steps = 128 VGPRs
    half8 af[2][kD / 16];
    {
        const _Float16 *gA = a16 + ((int64_t)pair * n0_pad + i0 + wr * 64 + l31) * kD + half * 8;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int ks = 0; ks < kD / 16; ks++)
                af[m][ks] = *reinterpret_cast<const half8 *>(gA + (int64_t)m * 32 * kD + ks * 16);
    }

    // LDS-DMA map: wave w fills image rows [32 w, 32 w + 32) of a stage with kDmaPerWave
    // instructions of kRowsPerDma rows.  Lane l writes chunk slot l % kChunksPerRow of row
    // l / kChunksPerRow and therefore fetches source chunk slot ^ swizzle(row).
    int src_off[kDmaPerWave];
#pragma unroll
    for (int q = 0; q < kDmaPerWave; q++) {
        const int row = wave * kRowsPerWave + q * kRowsPerDma + lane / kChunksPerRow;
        src_off[q] = row * kD + (((lane % kChunksPerRow) ^ stage_swizzle(row)) * 8);
    }
    const _Float16 *gB = b16 + (int64_t)pair * n1_pad * kD;
    constexpr int kStepsPerTile = kD / kBK;
    const int n_stages = (c_end - c_begin) * kStepsPerTile;
#define VS_ISSUE_STAGE(s_)                                                                     \
    do {                                                                                       \
        const int st_ = (s_);                                                                  \
        const _Float16 *g_ = gB + (int64_t)(c_begin + st_ / kStepsPerTile) * kBN * kD + (st_ % kStepsPerTile) * kBK; \
        unsigned char *l_ = smem + (st_ % kRing) * kStageBytes + wave * (kRowsPerWave * kBK * 2);        \
        _Pragma("unroll") for (int q_ = 0; q_ < kDmaPerWave; q_++)                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g_ + src_off[q_]), \
                                             (__attribute__((address_space(3))) void *)(l_ + q_ * 1024), 16, 0, 0); \
    } while (0)

    // fragment read offsets inside a stage: image row wc*64 + n*32 + l31, chunk (2 ks + half) ^ swz
    const int swz = stage_swizzle(l31);
    const int frag_row = (wc * 64 + l31) * (kBK * 2);

    float *sT = reinterpret_cast<float *>(smem + kRing * kStageBytes) + wave * (32 * kTS);

#pragma unroll
    for (int st = 0; st < kRing - 1; st++)
        if (st < n_stages) VS_ISSUE_STAGE(st);
    // Running (best, second) of this lane's row over the tiles of the band, with the 64-column
    // chunk the best lies in: after a tile's row pass every lane holds the finished pair of ONE
    // row, so the band's union costs a handful of instructions per tile, and the merge kernel
    // reads one summary per row and band instead of one partial per row and tile.
    float band_b1 = kNegInf, band_b2 = kNegInf;
    int band_chunk = c_begin;
    for (int tc = c_begin; tc < c_end; tc++) {
    const int j0 = tc * kBN;
    VS_STAMP();
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

#pragma unroll
    for (int kt = 0; kt < kStepsPerTile; kt++) {
        const int s = (tc - c_begin) * kStepsPerTile + kt;
        // Wait until this wave's share of stage s has landed; after the barrier every wave's has,
        // and every wave has finished reading ring slot (s - 1) % kRing, which is refilled next.
        // Vector-memory operations retire in issue order, so allowing the DMAs of the kRing - 2
        // younger stages to stay in flight retires stage s; the first step of a tile (the
        // epilogue's two stores follow the DMAs; letting them stay in flight with a counted wait
        // was measured and gains nothing) and the tail of the band (fewer younger stages exist)
        // drain fully.
        if (kt == 0 || s + kRing - 2 >= n_stages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kInflight) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        VS_STAMP();
        if (s + kRing - 1 < n_stages) VS_ISSUE_STAGE(s + kRing - 1);
        // desc1 fragments are read two sub-steps ahead of the MFMAs that use them (three register
        // sets).  The reads are issued and waited for by hand: LDS returns data in issue order, so
        // before the MFMAs of sub-step ks all reads but those of the (at most two) younger
        // sub-steps must have come back -- a counted wait; the compiler would drain the whole
        // queue, which exposes one LDS latency per sub-step.
        const unsigned lbase = (unsigned)(uintptr_t)(smem) + (unsigned)((s % kRing) * kStageBytes + frag_row);
        half8 bf[3][2];
        // The outputs are early-clobber: without it the compiler may give the address the first
        // register of an output, and the second read then takes its address from a register the
        // first read (still in flight) is about to overwrite -- harmless while the second read
        // issues within the first one's latency, a garbage address (out-of-range LDS reads return
        // zero: 8 of a column's 256 products lost) when the wave is held up between the two.
#define VS_FRAG_OUT(x) "=&v"(x)
#define VS_FRAG_READS(set, ks_)                                                               \
        do {                                                                                  \
            const unsigned a_ = lbase + (unsigned)((((2 * (ks_) + half) ^ swz) * 16));        \
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"               \
                         : VS_FRAG_OUT(bf[set][0]), VS_FRAG_OUT(bf[set][1]) : "v"(a_), "n"(32 * kBK * 2) : "memory"); \
        } while (0)
        VS_FRAG_READS(0, 0);
        VS_FRAG_READS(1, 1);
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ks++) {
            if (ks + 2 < kBK / 16) VS_FRAG_READS((ks + 2) % 3, ks + 2);
            if (ks + 2 < kBK / 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (ks + 1 < kBK / 16) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#ifndef VS_EXP_NO_MFMA      // ablation build (wrong results) when defined: everything but the MFMAs
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m][kt * (kBK / 16) + ks], bf[ks % 3][n], acc[m][n], 0, 0, 0);
#else
                    acc[m][n][0] += (float)bf[ks % 3][n][0] + (float)af[m][kt * (kBK / 16) + ks][0];
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#undef VS_FRAG_READS
        __builtin_amdgcn_sched_barrier(0);
        VS_STAMP();
    }

#ifdef VS_EXP_NO_EPILOGUE   // ablation build (wrong results): times the K loop without the top-2 epilogue
    {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) sum += acc[m][n][r];
        if (sum == 12345.678f) col_part[0] = make_float2(sum, sum);
        continue;
    }
#endif
    // ---- epilogue --------------------------------------------------------------------------
    // element (m, n, r) of this lane: row gi = i0 + wr*64 + m*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
    //                                 col gj = j0 + wc*64 + n*32 + (lane&31)
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
    // running top-2 on PACKED scores: the low 6 mantissa bits of each score carry the element's
    // position inside the wave's 64 rows / columns (relative perturbation <= 2^-17, part of the
    // band), so no index needs tracking:
    //   b2 <- median(b1, b2, key);  b1 <- max(b1, key)         (2 VALU + 1 pack per element;
    //   the two-key form below needs 1.5 VALU + 1 pack per element)
    // Each lane packs a 5-bit position; bit 5 (= lane half) is set once on the lane's best.
#define VS_TOP2_STEP(v, e)                                                                  \
    do {                                                                                     \
        const float k_ = __uint_as_float((__float_as_uint(v) & 0xffffffc0u) | (unsigned)(e)); \
        b2 = vmed3(b1, b2, k_);                                                              \
        b1 = vmax(b1, k_);                                                                   \
    } while (0)
    // two keys at once (b1 >= b2 holds before and after): the largest of {b1, k1, k2} is the new
    // best, and since b2 <= b1 the new second is max(b2, median(b1, k1, k2)): 3 VALU for 2 keys
#define VS_TOP2_STEP2(v1, e1, v2, e2)                                                        \
    do {                                                                                     \
        const float k1_ = __uint_as_float((__float_as_uint(v1) & 0xffffffc0u) | (unsigned)(e1)); \
        const float k2_ = __uint_as_float((__float_as_uint(v2) & 0xffffffc0u) | (unsigned)(e2)); \
        const float t_ = vmed3(b1, k1_, k2_);                                                \
        b1 = vmax3(b1, k1_, k2_);                                                            \
        b2 = vmax(b2, t_);                                                                   \
    } while (0)
    const unsigned half_bit = (unsigned)half << 5;
    // The column partial leaves together with the row partial at the end of the tile: the compiler
    // cannot tell the transposition area from the LDS-DMA ring and drains the vector-memory counter
    // before the transposition reads, which would also wait for a store issued before them.
    float2 col_out;
    // columns: top-2 over the wave's 64 rows; position code e = 32*half + 16*m + r stands for
    // row m*32 + (r&3) + 8*(r>>2) + 4*half of the wave's 64 (decoded by col_position())
    {
        float c1[2], c2[2];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            float b1 = kNegInf, b2 = kNegInf;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) VS_TOP2_STEP2(acc[m][n][r], m * 16 + r, acc[m][n][r + 1], m * 16 + r + 1);
            c1[n] = __uint_as_float(__float_as_uint(b1) | half_bit);
            c2[n] = b2;
        }
        // Unite the two halves of the rows without an LDS round trip: one v_permlane32_swap of
        // (block 0, block 1) exchanges a.hi <-> b.lo, after which lanes 0-31 hold both halves'
        // pairs of block 0 and lanes 32-63 those of block 1.
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c1[0]), __float_as_uint(c1[1]), false, false);
        const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c2[0]), __float_as_uint(c2[1]), false, false);
        float b1 = __uint_as_float(s1[0]), b2 = __uint_as_float(s2[0]);
        top2_union(b1, b2, __uint_as_float(s1[1]), __uint_as_float(s2[1]));
        col_out = make_float2(b1, b2);
    }
    __builtin_amdgcn_sched_barrier(0);
    VS_STAMP();
    // rows: transpose 32 rows x 64 cols at a time through the wave's own LDS area (row stride 68
    // floats keeps the float4 reads aligned and conflict-free; a wave's LDS operations complete in
    // issue order, so no barrier is needed); position code = column inside the wave's 64
    {
        float r1[2], r2[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    sT[((r & 3) + 8 * (r >> 2) + 4 * half) * kTS + n * 32 + l31] = acc[m][n][r];
            float b1 = kNegInf, b2 = kNegInf;
            const float4 *rowp = reinterpret_cast<const float4 *>(sT + l31 * kTS + 32 * half);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float4 q = rowp[c];
                VS_TOP2_STEP2(q.x, 4 * c, q.y, 4 * c + 1);
                VS_TOP2_STEP2(q.z, 4 * c + 2, q.w, 4 * c + 3);
            }
            r1[m] = __uint_as_float(__float_as_uint(b1) | half_bit);
            r2[m] = b2;
        }
        // lanes 0-31 finish rows 0-31 of the wave's 64, lanes 32-63 rows 32-63 (see the column pass)
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1[0]), __float_as_uint(r1[1]), false, false);
        const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r2[0]), __float_as_uint(r2[1]), false, false);
        float b1 = __uint_as_float(s1[0]), b2 = __uint_as_float(s2[0]);
        top2_union(b1, b2, __uint_as_float(s1[1]), __uint_as_float(s2[1]));
        const int64_t o = ((int64_t)pair * (n1_pad / kSub) + (j0 / kSub + wc)) * n0_pad + i0 + wr * 64 + half * 32 + l31;
        row_part[o] = make_float2(b1, b2);
        if (b1 > band_b1) band_chunk = j0 / kSub + wc;     // strictly greater: the first chunk wins ties, as in the merge
        top2_union(band_b1, band_b2, b1, b2);
        const int64_t oc = ((int64_t)pair * (n0_pad / kSub) + (i0 / kSub + wr)) * n1_pad + j0 + wc * 64 + half * 32 + l31;
        col_part[oc] = col_out;
    }
    __builtin_amdgcn_sched_barrier(0);
    VS_STAMP();
    }   // column tiles of the band
    if (row_sum)
        row_sum[((int64_t)pair * splits + split) * n0_pad + i0 + wr * 64 + half * 32 + l31] =
            make_float4(band_b1, band_b2, __int_as_float(band_chunk), 0.f);
#ifdef VS_MATCH_STAMPS
    if (stamp_on && lane == 0)
        for (int i = 0; i < 64; i++) g_match_stamps[wave * 64 + i] = i < n_stamp ? s_stamps[wave * 64 + i] : 0ull;
#endif
#undef VS_TOP2_STEP
#undef VS_TOP2_STEP2
#undef VS_ISSUE_STAGE
}

// position code of a column partial -> row inside the wave's 64 rows
__device__ __forceinline__ int col_position(unsigned code)
{
    const int rr = code & 15, m = (code >> 4) & 1, half = (code >> 5) & 1;
    return m * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
}
// candidate index named by the packed best of chunk `p` (rows: a column; columns: a row)
__device__ __forceinline__ int partial_index(bool is_row, int p, float packed_b1)
{
    const unsigned code = __float_as_uint(packed_b1) & 63u;
    return p * kSub + (is_row ? (int)code : col_position(code));
}

// Partials of item `it` of a pair (it < n0_pad: row it; else column it - n0_pad): chunk p of the
// other side lives at part[p * stride].
struct ItemView {
    const float2 *part;
    int64_t stride;
    int n_cand, self;
    bool is_row;
};
__device__ __forceinline__ ItemView item_view(int pair, int it, PairDims d, int n0_pad, int n1_pad,
                                              const float2 *row_part, const float2 *col_part)
{
    ItemView v;
    v.is_row = it < n0_pad;
    v.self = v.is_row ? it : it - n0_pad;
    v.n_cand = v.is_row ? d.n1 : d.n0;
    v.stride = v.is_row ? n0_pad : n1_pad;
    v.part = v.is_row ? row_part + (int64_t)pair * (n1_pad / kSub) * n0_pad + v.self
                      : col_part + (int64_t)pair * (n0_pad / kSub) * n1_pad + v.self;
    return v;
}

// Largest three of a stream of (value, chunk, is-a-chunk-best) entries.
struct Top3 {
    float v[3];
    int tag[3];          // 2 * chunk + (1 if the value is its chunk's best)
};
__device__ __forceinline__ void top3_insert(Top3 &t, float x, int tag)
{
    if (x > t.v[0]) { t.v[2] = t.v[1]; t.tag[2] = t.tag[1]; t.v[1] = t.v[0]; t.tag[1] = t.tag[0]; t.v[0] = x; t.tag[0] = tag; }
    else if (x > t.v[1]) { t.v[2] = t.v[1]; t.tag[2] = t.tag[1]; t.v[1] = x; t.tag[1] = tag; }
    else if (x > t.v[2]) { t.v[2] = x; t.tag[2] = tag; }
}

// Merge the per-64 partials of every row and column; store the approximate winner and list the
// items whose winner is not certain.  One thread per item.
// With want_second (the ratio test is on) every ROW also gets the column of its second-best score
// where that is beyond doubt: the best and the second come from different chunks (so the second is
// a chunk's best and its column is known), and both the best and the third lie more than the band
// away from it (every score outside the partials is below its chunk's second, hence below the
// third).  Other rows get -1 and are listed for the exact second-best scan.
__global__ void __launch_bounds__(256)
merge_kernel(const PairDims *__restrict__ dims, int n0_pad, int n1_pad,
             const float2 *__restrict__ row_part, const float2 *__restrict__ col_part,
             const unsigned *__restrict__ stats, int *__restrict__ idx0, int *__restrict__ idx1,
             int *__restrict__ rescan_list, int *__restrict__ rescan_count, int want_second,
             int *__restrict__ sec_idx, int *__restrict__ sec_list,
             const float4 *__restrict__ row_sum, int splits, int tiles_per_band)
{
    // Items that need an exact pass are listed through two counters in global memory.  A single
    // word takes about 90 atomic adds per microsecond, and a 64-pair batch lists some ten thousand
    // items: the workgroup counts its items in LDS and draws its range with one add per list.
    __shared__ int wg_n[2], wg_base[2];
    if (threadIdx.x < 2) wg_n[threadIdx.x] = 0;
    __syncthreads();
    const int pair = blockIdx.y;
    const int it = blockIdx.x * 256 + threadIdx.x;
    const PairDims d = dims[pair];
    bool list_item = false, list_second = false;
    int my_slot = 0, my_slot2 = 0, self = 0;
    if (it < n0_pad + n1_pad) {
    const ItemView v = item_view(pair, it, d, n0_pad, n1_pad, row_part, col_part);
    self = v.self;
    if (v.self < (v.is_row ? d.n0 : d.n1)) {
    float b1 = kNegInf, b2 = kNegInf;
    int pbest = 0;
    const int parts = (v.n_cand + kSub - 1) / kSub;
    const bool second = want_second && v.is_row;
    Top3 t3;
#pragma unroll
    for (int k = 0; k < 3; k++) { t3.v[k] = kNegInf; t3.tag[k] = -1; }
    int p = 0;
    if (v.is_row && row_sum && !second) {
        // rows: the score kernel has already united the partials of each band (same order, same
        // tie rule), one summary per band
        for (int sp = 0; sp < splits && sp * tiles_per_band * kBN < d.n1; sp++) {   // bands beyond the pair's columns did not run
            const float4 q = row_sum[((int64_t)pair * splits + sp) * n0_pad + v.self];
            if (q.x > b1) pbest = __float_as_int(q.z);
            top2_union(b1, b2, q.x, q.y);
        }
        p = parts;
    }
    for (; p + 8 <= parts; p += 8) {                  // eight loads in flight per thread
        float2 q[8];
#pragma unroll
        for (int u = 0; u < 8; u++) q[u] = v.part[(p + u) * v.stride];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (q[u].x > b1) pbest = p + u;
            top2_union(b1, b2, q[u].x, q[u].y);
            if (second) { top3_insert(t3, q[u].x, 2 * (p + u) + 1); top3_insert(t3, q[u].y, 2 * (p + u)); }
        }
    }
    for (; p < parts; p++) {
        const float2 q = v.part[p * v.stride];
        if (q.x > b1) pbest = p;
        top2_union(b1, b2, q.x, q.y);
        if (second) { top3_insert(t3, q.x, 2 * p + 1); top3_insert(t3, q.y, 2 * p); }
    }
    const int win = parts > 0 ? partial_index(v.is_row, pbest, b1) : -1;
    if (v.is_row) idx0[(int64_t)pair * n0_pad + v.self] = win;
    else idx1[(int64_t)pair * n1_pad + v.self] = win;
    const float band = pair_band(stats, pair);
    const bool uncertain = !(clamp_score(b1) - clamp_score(b2) > band);
    if (uncertain && parts > 0) { list_item = true; my_slot = atomicAdd(&wg_n[0], 1); }
    if (second) {
        const bool plain = !uncertain && (t3.tag[1] & 1) && (t3.tag[1] >> 1) != (t3.tag[0] >> 1) &&
                           clamp_score(t3.v[1]) - clamp_score(t3.v[2]) > band;
        sec_idx[(int64_t)pair * n0_pad + v.self] = plain ? partial_index(true, t3.tag[1] >> 1, t3.v[1]) : -1;
        if (!plain) { list_second = true; my_slot2 = atomicAdd(&wg_n[1], 1); }
    }
    }
    }
    __syncthreads();
    if (threadIdx.x < 2 && wg_n[threadIdx.x] > 0) wg_base[threadIdx.x] = atomicAdd(rescan_count + threadIdx.x, wg_n[threadIdx.x]);
    __syncthreads();
    if (list_item) {
        const int64_t slot = wg_base[0] + my_slot;
        rescan_list[2 * slot] = pair;
        rescan_list[2 * slot + 1] = it;
    }
    if (list_second) {
        const int64_t slot = wg_base[1] + my_slot2;
        sec_list[2 * slot] = pair;
        sec_list[2 * slot + 1] = self;
    }
}

// Exact re-decision of the listed rows/columns, one wave per item.  Lanes first look at one
// 64-candidate chunk each: a chunk whose best approximate score lies more than the band below
// the item's best cannot hold the exact first minimum; a chunk whose SECOND best lies below the
// band contributes only its best entry; any other chunk is scanned whole.  Exact distances come
// from staged panels; the wave minimum of the packed (distance bits, index) keys is the
// reference's first minimum (non-negative floats order like their bit patterns).
template <bool MASKED>
__global__ void __launch_bounds__(64)
rescan_items_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                    const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                    const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                    const int *__restrict__ rescan_list, const int *__restrict__ rescan_count,
                    const float2 *__restrict__ row_part, const float2 *__restrict__ col_part,
                    const unsigned *__restrict__ stats, int *__restrict__ idx0,
                    int *__restrict__ idx1)
{
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tA[kD];
    __shared__ const float *pA[1];
    __shared__ const float *pB[64];
    const int lane = threadIdx.x;
    const int count = *rescan_count;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int pair = rescan_list[2 * (int64_t)w], it = rescan_list[2 * (int64_t)w + 1];
        const PairDims dm = dims[pair];
        const ItemView v = item_view(pair, it, dm, n0_pad, n1_pad, row_part, col_part);
        const int parts = (v.n_cand + kSub - 1) / kSub;
        const float band = pair_band(stats, pair);
        const float *self_row = v.is_row ? d0 + ((int64_t)pair * n0_pad + v.self) * kD
                                         : d1 + ((int64_t)pair * n1_pad + v.self) * kD;
        const float *other_base = v.is_row ? d1 + (int64_t)pair * n1_pad * kD : d0 + (int64_t)pair * n0_pad * kD;
        const int2 self_xy = MASKED ? (v.is_row ? xy0[(int64_t)pair * n0_pad + v.self] : xy1[(int64_t)pair * n1_pad + v.self])
                                    : make_int2(0, 0);
        const int2 *other_xy = v.is_row ? xy1 + (int64_t)pair * n1_pad : xy0 + (int64_t)pair * n0_pad;
        // the item's best approximate score
        float item_best = kNegInf;
        for (int p = lane; p < parts; p += 64) item_best = fmaxf(item_best, v.part[p * v.stride].x);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) item_best = fmaxf(item_best, __shfl_xor(item_best, off, 64));
        item_best = clamp_score(item_best);

        unsigned long long key = ~0ull;
        // exact distance of candidate `c` for this lane given the finished dot
        auto candidate_key = [&](int c, double acc) -> unsigned long long {
            float d = distance_from_dot(acc);
            if (MASKED) {
                const int2 o = other_xy[c];
                if (v.is_row ? stereo_masked(self_xy, o) : stereo_masked(o, self_xy)) d = 1.0f;
            }
            return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)c;
        };
        for (int pbase = 0; pbase < parts; pbase += 64) {
            const int p = pbase + lane;
            float2 q = make_float2(kNegInf, kNegInf);
            if (p < parts) q = v.part[p * v.stride];
            const bool in_band = p < parts && item_best - clamp_score(q.x) <= band;
            const bool whole = in_band && item_best - clamp_score(q.y) <= band;
            const bool single = in_band && !whole;
            if (__ballot(single)) {
                const int c = single ? partial_index(v.is_row, p, q.x) : 0;
                __syncthreads();
                if (lane == 0) pA[0] = self_row;
                pB[lane] = other_base + (int64_t)c * kD;
                __syncthreads();
                const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
                if (single) {
                    const unsigned long long k = candidate_key(c, acc);
                    key = k < key ? k : key;
                }
            }
            unsigned long long todo = __ballot(whole);
            while (todo) {
                const int pl = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int c = (pbase + pl) * kSub + lane;
                const int cc = c < v.n_cand ? c : 0;
                __syncthreads();
                if (lane == 0) pA[0] = self_row;
                pB[lane] = other_base + (int64_t)cc * kD;
                __syncthreads();
                const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
                if (c < v.n_cand) {
                    const unsigned long long k = candidate_key(c, acc);
                    key = k < key ? k : key;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        if (lane == 0) {
            const int win = (int)(key & 0xffffffffu);
            if (v.is_row) idx0[(int64_t)pair * n0_pad + v.self] = win;
            else idx1[(int64_t)pair * n1_pad + v.self] = win;
        }
    }
}

// Exact distance of every mutual row (idx1[idx0[i]] == i); other rows get +huge and never
// reach the output.  One wave per 64 rows.
template <bool MASKED>
__global__ void __launch_bounds__(64)
mutual_distance_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                       const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                       const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                       const int *__restrict__ idx0, const int *__restrict__ idx1,
                       float *__restrict__ dist0, const int *__restrict__ sec_idx, float *__restrict__ second0)
{
    __shared__ __attribute__((aligned(16))) float tA[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ const float *pA[64];
    __shared__ const float *pB[64];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    const int n0 = dims[pair].n0, n1 = dims[pair].n1;
    if (blockIdx.x * 64 >= n0) return;
    int j = -1;
    if (i < n0) j = idx0[(int64_t)pair * n0_pad + i];
    const bool mutual = j >= 0 && j < n1 && idx1[(int64_t)pair * n1_pad + j] == i;
    if (!__ballot(mutual)) {
        if (i < n0) dist0[(int64_t)pair * n0_pad + i] = 3.0e38f;
        return;
    }
    bool masked = false;
    if (MASKED && mutual) masked = stereo_masked(xy0[(int64_t)pair * n0_pad + i], xy1[(int64_t)pair * n1_pad + j]);
    pA[lane] = d0 + ((int64_t)pair * n0_pad + (mutual ? i : 0)) * kD;
    pB[lane] = d1 + ((int64_t)pair * n1_pad + (mutual ? j : 0)) * kD;
    __syncthreads();
    const double acc = panel_dot64<false>(pA, pB, tA, tB, lane);
    if (i < n0) dist0[(int64_t)pair * n0_pad + i] = !mutual ? 3.0e38f : masked ? 1.0f : distance_from_dot(acc);
    if (sec_idx == nullptr) return;
    // ratio test: exact distance of the second-best column of the mutual rows that have one on record
    int j2 = -1;
    if (mutual) j2 = sec_idx[(int64_t)pair * n0_pad + i];
    if (!__ballot(j2 >= 0)) return;
    bool masked2 = false;
    if (MASKED && j2 >= 0) masked2 = stereo_masked(xy0[(int64_t)pair * n0_pad + i], xy1[(int64_t)pair * n1_pad + j2]);
    __syncthreads();
    pB[lane] = d1 + ((int64_t)pair * n1_pad + (j2 >= 0 ? j2 : 0)) * kD;
    __syncthreads();
    const double acc2 = panel_dot64<false>(pA, pB, tA, tB, lane);
    if (j2 >= 0) second0[(int64_t)pair * n0_pad + i] = masked2 ? 1.0f : distance_from_dot(acc2);
}

// Ratio test, rows without a second-best column on record: exact second-smallest distance of the
// row, one wave per listed row (mutual rows only: the others never reach the output).  Every
// column whose approximate score lies within the band of the row's approximate second best can be
// the exact second; chunks are pruned as in rescan_items_kernel, the row's exact nearest
// neighbour (idx0) is left out, and the smallest remaining exact distance is the second.
template <bool MASKED>
__global__ void __launch_bounds__(64)
second_scan_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                   const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                   const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                   const int *__restrict__ sec_list, const int *__restrict__ sec_count,
                   const float2 *__restrict__ row_part, const float2 *__restrict__ col_part,
                   const unsigned *__restrict__ stats, const int *__restrict__ idx0,
                   const int *__restrict__ idx1, float *__restrict__ second0)
{
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tA[kD];
    __shared__ const float *pA[1];
    __shared__ const float *pB[64];
    const int lane = threadIdx.x;
    const int count = *sec_count;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int pair = sec_list[2 * (int64_t)w], i = sec_list[2 * (int64_t)w + 1];
        const PairDims dm = dims[pair];
        const int jbest = idx0[(int64_t)pair * n0_pad + i];
        if (jbest < 0 || jbest >= dm.n1 || idx1[(int64_t)pair * n1_pad + jbest] != i) continue;   // not mutual (uniform)
        const ItemView v = item_view(pair, i, dm, n0_pad, n1_pad, row_part, col_part);
        const int parts = (v.n_cand + kSub - 1) / kSub;
        const float band = pair_band(stats, pair);
        const float *self_row = d0 + ((int64_t)pair * n0_pad + i) * kD;
        const float *other_base = d1 + (int64_t)pair * n1_pad * kD;
        const int2 self_xy = MASKED ? xy0[(int64_t)pair * n0_pad + i] : make_int2(0, 0);
        const int2 *other_xy = xy1 + (int64_t)pair * n1_pad;
        // approximate second best of the row: the largest two of all partial values
        float t1 = kNegInf, t2 = kNegInf;
        for (int p = lane; p < parts; p += 64) { const float2 q = v.part[p * v.stride]; top2_union(t1, t2, q.x, q.y); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) top2_union(t1, t2, __shfl_xor(t1, off, 64), __shfl_xor(t2, off, 64));
        const float second_approx = clamp_score(t2);
        unsigned long long key = ~0ull;
        auto candidate_key = [&](int c, double acc) -> unsigned long long {
            float d = distance_from_dot(acc);
            if (MASKED && stereo_masked(self_xy, other_xy[c])) d = 1.0f;
            return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)c;
        };
        for (int pbase = 0; pbase < parts; pbase += 64) {
            const int p = pbase + lane;
            float2 q = make_float2(kNegInf, kNegInf);
            if (p < parts) q = v.part[p * v.stride];
            const bool in_band = p < parts && second_approx - clamp_score(q.x) <= band;
            const bool whole = in_band && second_approx - clamp_score(q.y) <= band;
            const bool single = in_band && !whole;
            if (__ballot(single)) {
                const int c = single ? partial_index(true, p, q.x) : 0;
                __syncthreads();
                if (lane == 0) pA[0] = self_row;
                pB[lane] = other_base + (int64_t)c * kD;
                __syncthreads();
                const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
                if (single && c != jbest) {
                    const unsigned long long k = candidate_key(c, acc);
                    key = k < key ? k : key;
                }
            }
            unsigned long long todo = __ballot(whole);
            while (todo) {
                const int pl = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int c = (pbase + pl) * kSub + lane;
                const int cc = c < v.n_cand ? c : 0;
                __syncthreads();
                if (lane == 0) pA[0] = self_row;
                pB[lane] = other_base + (int64_t)cc * kD;
                __syncthreads();
                const double acc = panel_dot64<true>(pA, pB, tA, tB, lane);
                if (c < v.n_cand && c != jbest) {
                    const unsigned long long k = candidate_key(c, acc);
                    key = k < key ? k : key;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        if (lane == 0) second0[(int64_t)pair * n0_pad + i] = key == ~0ull ? INFINITY : __uint_as_float((unsigned)(key >> 32));
    }
}

// Row-sharded pair: packed key (exact distance bits, global row) of every column's nearest row
// among this rank's rows; the element-wise minimum of the ranks' keys is the column's nearest
// row of the whole frame (equal distances: the smaller row, i.e. the first minimum).  One wave
// per 64 columns.
template <bool MASKED>
__global__ void __launch_bounds__(64)
column_key_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                  const PairDims *__restrict__ dims, const int2 *__restrict__ xy0,
                  const int2 *__restrict__ xy1, int n0_pad, int n1_pad,
                  const int *__restrict__ idx1, int row0, unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float tA[64 * kPanelStride];
    __shared__ __attribute__((aligned(16))) float tB[64 * kPanelStride];
    __shared__ const float *pA[64];
    __shared__ const float *pB[64];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane;
    const int n0 = dims[pair].n0, n1 = dims[pair].n1;
    if (blockIdx.x * 64 >= n1) return;
    int r = -1;
    if (j < n1) r = idx1[(int64_t)pair * n1_pad + j];
    const bool have = r >= 0 && r < n0;
    if (!__ballot(have)) {
        if (j < n1) keys[(int64_t)pair * n1_pad + j] = ~0ull;
        return;
    }
    bool masked = false;
    if (MASKED && have) masked = stereo_masked(xy0[(int64_t)pair * n0_pad + r], xy1[(int64_t)pair * n1_pad + j]);
    pA[lane] = d0 + ((int64_t)pair * n0_pad + (have ? r : 0)) * kD;
    pB[lane] = d1 + ((int64_t)pair * n1_pad + (have ? j : 0)) * kD;
    __syncthreads();
    const double acc = panel_dot64<false>(pA, pB, tA, tB, lane);
    const float d = masked ? 1.0f : distance_from_dot(acc);
    if (j < n1)
        keys[(int64_t)pair * n1_pad + j] = have ? ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(row0 + r) : ~0ull;
}

// After the minimum over ranks: idx1[j] = (global nearest row) - row0, which equals a local row
// index exactly when the winner lives on this rank.
__global__ void apply_keys_kernel(const PairDims *__restrict__ dims, int n1_pad,
                                  const unsigned long long *__restrict__ keys, int row0, int *__restrict__ idx1)
{
    const int pair = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dims[pair].n1) return;
    const unsigned long long k = keys[(int64_t)pair * n1_pad + j];
    idx1[(int64_t)pair * n1_pad + j] = k == ~0ull ? -1 : (int)(unsigned)(k & 0xffffffffull) - row0;
}

// Mutual check, threshold (and optional ratio test), ordered compaction.  One workgroup per pair.
// row0 is added to the row numbers written out (0 unless the pair is row-sharded).
__global__ void __launch_bounds__(1024)
compact_kernel(const PairDims *__restrict__ dims, int n0_pad, int n1_pad,
               const int *__restrict__ idx0, const int *__restrict__ idx1,
               const float *__restrict__ dist0, const float *__restrict__ second0, float nn_thresh,
               float ratio, int row0, int *__restrict__ out_i, int *__restrict__ out_j,
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
            keep = j >= 0 && idx1[(int64_t)pair * n1_pad + j] == i && d < nn_thresh;
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
            out_i[o] = row0 + i; out_j[o] = j; out_d[o] = d;
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
    DevBuf<_Float16> a16, b16;
    DevBuf<unsigned> stats;          // 4 per pair, see pair_band()
    DevBuf<float> split_part;        // per-workgroup maxima of split_pairs_kernel
    int part_blocks = 0;
    DevBuf<int2> xy0, xy1;
    DevBuf<PairDims> dims;
    DevBuf<float2> row_part, col_part;   // (best, second) packed scores per item and 64-chunk
    DevBuf<float4> row_sum;              // band summaries of the rows (see dot_top2_kernel)
    DevBuf<float> dist0, dist1, second0, out_d;
    DevBuf<int> idx0, idx1, rescan_list, rescan_count, out_i, out_j, out_n;
    DevBuf<int> sec_idx, sec_list;      // ratio test: second-best column per row, rows that need the exact scan
    // row-sharded pairs: packed column keys and how their minimum over ranks is taken
    DevBuf<unsigned long long> keys;
    vs_comm *comm = nullptr;
    vs_allreduce_min_u64_fn min_fn = nullptr;
    void *min_user = nullptr;
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
    A(m->a16.alloc(P * r0 * kD)); A(m->b16.alloc(P * r1 * kD)); A(m->stats.alloc(4 * P));
    A(m->xy0.alloc(P * r0)); A(m->xy1.alloc(P * r1)); A(m->dims.alloc(P));
    m->part_blocks = div_up(std::max(m->n0_pad, m->n1_pad), kSplitRows);
    A(m->split_part.alloc(P * 2 * (size_t)m->part_blocks * 2));
    const size_t rp = P * (r1 / kSub) * r0, cp = P * (r0 / kSub) * r1;
    A(m->row_part.alloc(rp)); A(m->col_part.alloc(cp));
    A(m->row_sum.alloc(P * r0 * (size_t)std::max<size_t>(1, (r1 / kBN) / 4)));   // bands of >= 4 tiles
    A(m->idx0.alloc(P * r0)); A(m->idx1.alloc(P * r1)); A(m->dist0.alloc(P * r0)); A(m->dist1.alloc(P * r1));
    A(m->second0.alloc(P * r0)); A(m->sec_idx.alloc(P * r0)); A(m->sec_list.alloc(2 * P * r0));
    A(m->rescan_list.alloc(2 * P * (r0 + r1))); A(m->rescan_count.alloc(4));
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

// Is `p` device memory (hipMalloc'd, or any pointer the device can read in place)?
static bool device_readable(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice;
}

// Descriptors (and keypoints) of pairs [first, first + n_pairs), every pair n0 x n1; pair p's rows
// at desc + p * stride floats (keypoints dense: 2 n ints per pair).  Device inputs are read in
// place by the rounding kernel, which also copies them into the handle, and the call returns
// without waiting for the stream; host inputs are copied first and the call waits, so that the
// caller's buffers are free on return.
static int set_pairs_impl(vs_matcher *m, int first, int n_pairs, const float *desc0, int64_t stride0, int n0,
                          const float *desc1, int64_t stride1, int n1, const int32_t *xy0, const int32_t *xy1)
{
    hipStream_t s = m->stream;
    const size_t e0 = (size_t)m->n0_pad * kD, e1 = (size_t)m->n1_pad * kD;
    float *g0 = m->d0.p + (size_t)first * e0, *g1 = m->d1.p + (size_t)first * e1;
    bool host_input = false;
    // read in place: device memory the kernel can fetch as aligned 16-byte pieces
    auto in_place = [&](const float *d, int64_t stride, int n) {
        if (n == 0) return true;
        const bool dev = device_readable(d);
        host_input = host_input || !dev;
        return dev && (reinterpret_cast<uintptr_t>(d) & 15) == 0 && stride % 4 == 0;
    };
    const bool dev0 = in_place(desc0, stride0, n0), dev1 = in_place(desc1, stride1, n1);
    if (!dev0)
        VS_HIP(hipMemcpy2DAsync(g0, e0 * sizeof(float), desc0, (size_t)stride0 * sizeof(float), (size_t)n0 * kD * sizeof(float),
                                (size_t)n_pairs, hipMemcpyDefault, s));
    if (!dev1)
        VS_HIP(hipMemcpy2DAsync(g1, e1 * sizeof(float), desc1, (size_t)stride1 * sizeof(float), (size_t)n1 * kD * sizeof(float),
                                (size_t)n_pairs, hipMemcpyDefault, s));
    if (xy0) {
        host_input = host_input || (n0 && !device_readable(xy0)) || (n1 && !device_readable(xy1));
        if (n0) VS_HIP(hipMemcpy2DAsync(m->xy0.p + (size_t)first * m->n0_pad, (size_t)m->n0_pad * sizeof(int2), xy0, (size_t)n0 * sizeof(int2),
                                        (size_t)n0 * sizeof(int2), (size_t)n_pairs, hipMemcpyDefault, s));
        if (n1) VS_HIP(hipMemcpy2DAsync(m->xy1.p + (size_t)first * m->n1_pad, (size_t)m->n1_pad * sizeof(int2), xy1, (size_t)n1 * sizeof(int2),
                                        (size_t)n1 * sizeof(int2), (size_t)n_pairs, hipMemcpyDefault, s));
    }
    // sources of the rounding kernel: the caller's device arrays, or the copies just made
    const float *s0 = dev0 && n0 ? desc0 : g0, *s1 = dev1 && n1 ? desc1 : g1;
    const int64_t st0 = dev0 && n0 ? stride0 : (int64_t)e0, st1 = dev1 && n1 ? stride1 : (int64_t)e1;
    float *part = m->split_part.p + (size_t)first * 2 * m->part_blocks * 2;
    hipLaunchKernelGGL(split_pairs_kernel, dim3(m->part_blocks, 2, n_pairs), dim3(256), 0, s, s0, st0, n0, m->n0_pad, s1, st1, n1,
                       m->n1_pad, m->d0.p, m->d1.p, m->a16.p, m->b16.p, first, m->part_blocks, part);
    hipLaunchKernelGGL(split_finish_kernel, dim3(n_pairs), dim3(64), 0, s, first, m->part_blocks, part, n0, n1, m->stats.p, m->dims.p);
    VS_TRY(launch_ok("split_pairs_kernel"));
    for (int p = first; p < first + n_pairs; p++) {
        m->h_dims[p] = PairDims{ n0, n1 };
        m->h_masked[p] = xy0 != nullptr;
    }
    if (host_input) VS_HIP(hipStreamSynchronize(s));
    m->ran = false;
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
    return set_pairs_impl(m, pair, 1, desc0, (int64_t)n0 * kD, n0, desc1, (int64_t)n1 * kD, n1, xy0, xy1);
}

int vs_matcher_set_pairs(vs_matcher_handle m, int32_t first_pair, int32_t n_pairs, const float *desc0,
                         int64_t stride0, int32_t n0, const float *desc1, int64_t stride1, int32_t n1,
                         const int32_t *xy0, const int32_t *xy1)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(n_pairs > 0 && first_pair >= 0 && first_pair <= m->max_pairs - n_pairs, VS_ERR_INVALID,
             "pairs [%d, %d) out of range [0,%d)", first_pair, first_pair + n_pairs, m->max_pairs);
    VS_CHECK(n0 >= 0 && n0 <= m->n0_pad && n1 >= 0 && n1 <= m->n1_pad, VS_ERR_INVALID,
             "pair sizes %d x %d exceed the matcher's capacity %d x %d", n0, n1, m->n0_pad, m->n1_pad);
    VS_CHECK((n0 == 0 || desc0) && (n1 == 0 || desc1), VS_ERR_INVALID, "null descriptor pointer");
    VS_CHECK(stride0 >= (int64_t)n0 * kD && stride1 >= (int64_t)n1 * kD, VS_ERR_INVALID,
             "pair strides %lld / %lld must cover %d / %d rows of 256 floats", (long long)stride0, (long long)stride1, n0, n1);
    VS_CHECK((xy0 == nullptr) == (xy1 == nullptr), VS_ERR_INVALID, "xy0 and xy1 must both be given or both be NULL");
    return set_pairs_impl(m, first_pair, n_pairs, desc0, stride0, n0, desc1, stride1, n1, xy0, xy1);
}

// Device buffers one matcher run works on (all pairs of a batch share the row pads).
struct MatchView {
    int n0_pad, n1_pad;
    const float *d0, *d1;            // fp32 descriptors, pair p at p * n_pad * 256
    _Float16 *a16, *b16;             // their fp16 images (split_kernel)
    unsigned *stats;                 // 4 per pair, see pair_band()
    const int2 *xy0, *xy1;           // keypoints (stereo mask) or unused
    PairDims *dims;
    float2 *row_part, *col_part;     // (best, second) packed scores per item and 64-chunk
    float4 *row_sum = nullptr;       // per row and band of the score kernel: (best, second, chunk of the best) or null
    size_t row_sum_len = 0;          // float4 elements available at row_sum
    float *dist0, *dist1, *second0;
    int *idx0, *idx1, *rescan_list, *rescan_count;   // rescan_count: [0] uncertain items, [1] rows for the exact second-best scan
    int *sec_idx, *sec_list;
    int *out_i, *out_j, *out_n;
    float *out_d;
};

// The matcher pipeline on pairs [0, n_pairs) of `v`: scores + fused top-2, merge, exact
// re-decisions, mutual check + compaction.  Launches only; allocates nothing.
// How a row-sharded run combines its column keys across ranks (nullptr for a whole pair).
struct ShardCtx {
    int row0;
    unsigned long long *keys;       // n_pairs x n1_pad
    vs_comm *comm;
    vs_allreduce_min_u64_fn fn;
    void *user;
};

static int run_match(const MatchView &v, int n_pairs, bool masked, float nn_thresh, float ratio,
                     hipStream_t s, hipEvent_t *prof_ev, const ShardCtx *shard = nullptr)
{
    auto mark = [&](int i) { if (prof_ev) (void)hipEventRecord(prof_ev[i], s); };
    const int r0 = v.n0_pad, r1 = v.n1_pad;
    // column tiles per workgroup: whole rows of tiles when that still gives every CU several
    // workgroups, otherwise split the tile rows
    const int ntr = r0 / kBM, ntc = r1 / kBN;
    const int want_splits = std::min(ntc, std::max(1, div_up(2048, n_pairs * ntr)));
    const int tiles_per_wg = div_up(ntc, want_splits);
    const int splits = div_up(ntc, tiles_per_wg);
    mark(0);
    const dim3 grid(n_pairs * ntr * splits);
    // band summaries of the rows: worth their 16 bytes per row and band when a band unites at least
    // four tiles (the merge then reads them instead of the per-tile partials)
    float4 *row_sum = nullptr;
    if (v.row_sum && tiles_per_wg >= 4 && (size_t)n_pairs * splits * r0 <= v.row_sum_len) row_sum = v.row_sum;
    if (masked)
        hipLaunchKernelGGL(dot_top2_kernel<true>, grid, dim3(256), 0, s, v.a16, v.b16, v.dims, v.xy0,
                           v.xy1, r0, r1, tiles_per_wg, v.row_part, v.col_part, row_sum);
    else
        hipLaunchKernelGGL(dot_top2_kernel<false>, grid, dim3(256), 0, s, v.a16, v.b16, v.dims, v.xy0,
                           v.xy1, r0, r1, tiles_per_wg, v.row_part, v.col_part, row_sum);
    VS_TRY(launch_ok("dot_top2_kernel"));
    mark(1);
    const int want_second = ratio < 1.0f ? 1 : 0;     // the ratio test needs the exact second-smallest distance of the output rows
    VS_HIP(hipMemsetAsync(v.rescan_count, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(merge_kernel, dim3(div_up(r0 + r1, 256), n_pairs), dim3(256), 0, s, v.dims, r0, r1,
                       v.row_part, v.col_part, v.stats, v.idx0, v.idx1, v.rescan_list,
                       v.rescan_count, want_second, v.sec_idx, v.sec_list, row_sum, splits, tiles_per_wg);
    VS_TRY(launch_ok("merge_kernel"));
    mark(2);
    if (masked)
        hipLaunchKernelGGL(rescan_items_kernel<true>, dim3(8192), dim3(64), 0, s, v.d0, v.d1, v.dims,
                           v.xy0, v.xy1, r0, r1, v.rescan_list, v.rescan_count, v.row_part,
                           v.col_part, v.stats, v.idx0, v.idx1);
    else
        hipLaunchKernelGGL(rescan_items_kernel<false>, dim3(8192), dim3(64), 0, s, v.d0, v.d1, v.dims,
                           v.xy0, v.xy1, r0, r1, v.rescan_list, v.rescan_count, v.row_part,
                           v.col_part, v.stats, v.idx0, v.idx1);
    if (shard) {
        // the column nearest neighbours so far are over this rank's rows only
        const dim3 kgrid(div_up(r1, 64), n_pairs);
        if (masked)
            hipLaunchKernelGGL(column_key_kernel<true>, kgrid, dim3(64), 0, s, v.d0, v.d1, v.dims, v.xy0, v.xy1,
                               r0, r1, v.idx1, shard->row0, shard->keys);
        else
            hipLaunchKernelGGL(column_key_kernel<false>, kgrid, dim3(64), 0, s, v.d0, v.d1, v.dims, v.xy0, v.xy1,
                               r0, r1, v.idx1, shard->row0, shard->keys);
        VS_TRY(launch_ok("column_key_kernel"));
        const int64_t nk = (int64_t)n_pairs * r1;
        if (shard->comm) {
            VS_TRY(comm_allreduce_min_u64(shard->comm, shard->keys, nk, s));
        } else if (shard->fn) {
            VS_HIP(hipStreamSynchronize(s));
            VS_CHECK(shard->fn(shard->user, shard->keys, nk) == 0, VS_ERR_STATE, "the minimum all-reduce callback failed");
        }
        hipLaunchKernelGGL(apply_keys_kernel, dim3(div_up(r1, 256), n_pairs), dim3(256), 0, s, v.dims, r1,
                           shard->keys, shard->row0, v.idx1);
        VS_TRY(launch_ok("apply_keys_kernel"));
    }
    {
        const dim3 dgrid(div_up(r0, 64), n_pairs);
        const int *sec = want_second ? v.sec_idx : nullptr;
        if (masked) {
            hipLaunchKernelGGL(mutual_distance_kernel<true>, dgrid, dim3(64), 0, s, v.d0, v.d1, v.dims,
                               v.xy0, v.xy1, r0, r1, v.idx0, v.idx1, v.dist0, sec, v.second0);
            if (want_second)
                hipLaunchKernelGGL(second_scan_kernel<true>, dim3(8192), dim3(64), 0, s, v.d0, v.d1, v.dims, v.xy0, v.xy1,
                                   r0, r1, v.sec_list, v.rescan_count + 1, v.row_part, v.col_part, v.stats, v.idx0,
                                   v.idx1, v.second0);
        } else {
            hipLaunchKernelGGL(mutual_distance_kernel<false>, dgrid, dim3(64), 0, s, v.d0, v.d1, v.dims,
                               v.xy0, v.xy1, r0, r1, v.idx0, v.idx1, v.dist0, sec, v.second0);
            if (want_second)
                hipLaunchKernelGGL(second_scan_kernel<false>, dim3(8192), dim3(64), 0, s, v.d0, v.d1, v.dims, v.xy0, v.xy1,
                                   r0, r1, v.sec_list, v.rescan_count + 1, v.row_part, v.col_part, v.stats, v.idx0,
                                   v.idx1, v.second0);
        }
    }
    VS_TRY(launch_ok("exact kernels"));
    mark(3);
    hipLaunchKernelGGL(compact_kernel, dim3(n_pairs), dim3(1024), 0, s, v.dims, r0, r1, v.idx0, v.idx1,
                       v.dist0, v.second0, nn_thresh, ratio, shard ? shard->row0 : 0, v.out_i, v.out_j, v.out_d, v.out_n);
    VS_TRY(launch_ok("compact_kernel"));
    mark(4);
    return VS_OK;
}

static MatchView handle_view(vs_matcher *m)
{
    MatchView v;
    v.n0_pad = m->n0_pad; v.n1_pad = m->n1_pad;
    v.d0 = m->d0.p; v.d1 = m->d1.p; v.a16 = m->a16.p; v.b16 = m->b16.p; v.stats = m->stats.p;
    v.xy0 = m->xy0.p; v.xy1 = m->xy1.p; v.dims = m->dims.p;
    v.row_part = m->row_part.p; v.col_part = m->col_part.p;
    v.row_sum = m->row_sum.p; v.row_sum_len = m->row_sum.n;
    v.dist0 = m->dist0.p; v.dist1 = m->dist1.p; v.second0 = m->second0.p;
    v.idx0 = m->idx0.p; v.idx1 = m->idx1.p; v.rescan_list = m->rescan_list.p; v.rescan_count = m->rescan_count.p;
    v.sec_idx = m->sec_idx.p; v.sec_list = m->sec_list.p;
    v.out_i = m->out_i.p; v.out_j = m->out_j.p; v.out_n = m->out_n.p; v.out_d = m->out_d.p;
    return v;
}

int vs_matcher_run(vs_matcher_handle m, int32_t n_pairs, float nn_thresh, float ratio)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(n_pairs > 0 && n_pairs <= m->max_pairs, VS_ERR_INVALID, "n_pairs %d out of range (1..%d)", n_pairs, m->max_pairs);
    bool masked = m->h_masked[0];
    for (int p = 1; p < n_pairs; p++)
        VS_CHECK((bool)m->h_masked[p] == masked, VS_ERR_INVALID, "a batch must be all two-way or all stereo pairs");
    VS_TRY(run_match(handle_view(m), n_pairs, masked, nn_thresh, ratio, m->stream, m->prof_ev));
    m->ran = true;
    return VS_OK;
}

int vs_matcher_set_comm(vs_matcher_handle m, vs_comm_handle c)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    m->comm = c;
    return VS_OK;
}

int vs_matcher_set_min_allreduce(vs_matcher_handle m, vs_allreduce_min_u64_fn fn, void *user)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    m->min_fn = fn;
    m->min_user = user;
    return VS_OK;
}

int vs_matcher_run_sharded(vs_matcher_handle m, int32_t n_pairs, int32_t row0, float nn_thresh, float ratio)
{
    VS_CHECK(m != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(n_pairs > 0 && n_pairs <= m->max_pairs, VS_ERR_INVALID, "n_pairs %d out of range (1..%d)", n_pairs, m->max_pairs);
    VS_CHECK(row0 >= 0, VS_ERR_INVALID, "negative first row");
    bool masked = m->h_masked[0];
    for (int p = 1; p < n_pairs; p++)
        VS_CHECK((bool)m->h_masked[p] == masked, VS_ERR_INVALID, "a batch must be all two-way or all stereo pairs");
    const size_t nk = (size_t)m->max_pairs * m->n1_pad;
    if (m->keys.n < nk) VS_TRY(m->keys.alloc(nk));
    const ShardCtx shard{ row0, m->keys.p, m->comm, m->comm ? nullptr : m->min_fn, m->min_user };
    VS_TRY(run_match(handle_view(m), n_pairs, masked, nn_thresh, ratio, m->stream, m->prof_ev, &shard));
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

// ---------------------------------------------------------------------------------------------
// Allocation-free one-shot two-way matcher.  Everything vs_match_two_way keeps in a freshly
// allocated handle (fp16 images, top-2 partials, nearest-neighbour and rescan lists) lives in one
// file-scope device array; the caller's descriptors are read in place.
constexpr int kNoAllocMaxN = VS_MATCH_NOALLOC_MAX_N;
constexpr size_t align256(size_t b) { return (b + 255) / 256 * 256; }
constexpr int kNaPad0 = (kNoAllocMaxN + kBM - 1) / kBM * kBM, kNaPad1 = (kNoAllocMaxN + kBN - 1) / kBN * kBN;
constexpr size_t kNaA16 = 0;
constexpr size_t kNaB16 = kNaA16 + align256((size_t)kNaPad0 * kD * 2);
constexpr size_t kNaRowPart = kNaB16 + align256((size_t)kNaPad1 * kD * 2);
constexpr size_t kNaColPart = kNaRowPart + align256((size_t)(kNaPad1 / kSub) * kNaPad0 * sizeof(float2));
constexpr size_t kNaIdx0 = kNaColPart + align256((size_t)(kNaPad0 / kSub) * kNaPad1 * sizeof(float2));
constexpr size_t kNaIdx1 = kNaIdx0 + align256((size_t)kNaPad0 * 4);
constexpr size_t kNaDist0 = kNaIdx1 + align256((size_t)kNaPad1 * 4);
constexpr size_t kNaDist1 = kNaDist0 + align256((size_t)kNaPad0 * 4);
constexpr size_t kNaSecond0 = kNaDist1 + align256((size_t)kNaPad1 * 4);
constexpr size_t kNaRescan = kNaSecond0 + align256((size_t)kNaPad0 * 4);
constexpr size_t kNaSecIdx = kNaRescan + align256((size_t)2 * (kNaPad0 + kNaPad1) * 4);
constexpr size_t kNaSecList = kNaSecIdx + align256((size_t)kNaPad0 * 4);
constexpr size_t kNaSmall = kNaSecList + align256((size_t)2 * kNaPad0 * 4);   // stats 16 B | dims 8 B | pad | rescan count 16 B
constexpr size_t kNaBytes = kNaSmall + 256;

static __device__ __align__(256) unsigned char scratch[kNaBytes];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each GPU, resolved once per device
std::mutex g_scratch_mutex;

// dims, zeroed rounding statistics of the pair
__global__ void noalloc_setup_kernel(PairDims *dims, unsigned *stats, int n0, int n1)
{
    if (threadIdx.x == 0) { dims->n0 = n0; dims->n1 = n1; }
    if (threadIdx.x < 4) stats[threadIdx.x] = 0u;
}

// Pointers a thread has already shown to be device memory are not asked about again: a serving
// loop or a graph capture that reuses its buffers then makes no runtime query per call.
int device_pointer_check(const void *p, const char *what)
{
    constexpr int kKnown = 8;
    static thread_local const void *known[kKnown] = {};
    static thread_local int next = 0;
    for (int i = 0; i < kKnown; i++)
        if (known[i] == p) return VS_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered) {
        (void)hipGetLastError();
        set_error("%s must be a device pointer", what);
        return VS_ERR_INVALID;
    }
    known[next] = p; next = (next + 1) % kKnown;
    return VS_OK;
}
}  // namespace

int vs_match_two_way_noalloc(const float *desc0, int32_t n0, const float *desc1, int32_t n1,
                             float nn_thresh, float ratio, int32_t *out_i, int32_t *out_j,
                             float *out_d, int32_t *out_n, void *hip_stream)
{
    VS_CHECK(out_n != nullptr, VS_ERR_INVALID, "null out_n");
    VS_CHECK(n0 >= 0 && n1 >= 0, VS_ERR_INVALID, "negative size");
    VS_CHECK(n0 <= kNoAllocMaxN && n1 <= kNoAllocMaxN, VS_ERR_INVALID,
             "%d x %d descriptors exceed the allocation-free matcher's limit of %d per side", n0, n1, kNoAllocMaxN);
    hipStream_t s = (hipStream_t)hip_stream;
    VS_TRY(device_pointer_check(out_n, "out_n"));
    if (n0 == 0 || n1 == 0) {            // the reference's loops do nothing on empty inputs
        VS_HIP(hipMemsetAsync(out_n, 0, sizeof(int32_t), s));
        return VS_OK;
    }
    VS_CHECK(desc0 && desc1 && out_i && out_j && out_d, VS_ERR_INVALID, "null argument");
    VS_TRY(device_pointer_check(desc0, "desc0")); VS_TRY(device_pointer_check(desc1, "desc1"));
    VS_TRY(device_pointer_check(out_i, "out_i")); VS_TRY(device_pointer_check(out_j, "out_j"));
    VS_TRY(device_pointer_check(out_d, "out_d"));
    // the descriptors are read in place as 16-byte vectors
    VS_CHECK(((uintptr_t)desc0 & 15) == 0 && ((uintptr_t)desc1 & 15) == 0, VS_ERR_INVALID,
             "descriptor arrays must be 16-byte aligned");
    int dev = 0;
    VS_HIP(hipGetDevice(&dev));
    VS_CHECK(dev >= 0 && dev < kMaxDevices, VS_ERR_UNSUPPORTED, "device ordinal %d beyond %d", dev, kMaxDevices);
    unsigned char *base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {       // each GPU holds its own copy of the array at its own address
            void *p = nullptr;
            VS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[dev] = (unsigned char *)p;
        }
        base = g_scratch_ptr[dev];
    }
    MatchView v;
    v.n0_pad = (n0 + kBM - 1) / kBM * kBM; v.n1_pad = (n1 + kBN - 1) / kBN * kBN;
    v.d0 = desc0; v.d1 = desc1;
    v.a16 = (_Float16 *)(base + kNaA16); v.b16 = (_Float16 *)(base + kNaB16);
    v.row_part = (float2 *)(base + kNaRowPart); v.col_part = (float2 *)(base + kNaColPart);
    v.idx0 = (int *)(base + kNaIdx0); v.idx1 = (int *)(base + kNaIdx1);
    v.dist0 = (float *)(base + kNaDist0); v.dist1 = (float *)(base + kNaDist1);
    v.second0 = (float *)(base + kNaSecond0);
    v.rescan_list = (int *)(base + kNaRescan);
    v.sec_idx = (int *)(base + kNaSecIdx); v.sec_list = (int *)(base + kNaSecList);
    v.stats = (unsigned *)(base + kNaSmall); v.dims = (PairDims *)(base + kNaSmall + 16);
    v.rescan_count = (int *)(base + kNaSmall + 64);
    v.xy0 = v.xy1 = nullptr;
    v.out_i = out_i; v.out_j = out_j; v.out_d = out_d; v.out_n = out_n;
    hipLaunchKernelGGL(noalloc_setup_kernel, dim3(1), dim3(64), 0, s, v.dims, v.stats, n0, n1);
    hipLaunchKernelGGL(split_kernel, dim3(div_up(v.n0_pad, kSplitRows)), dim3(256), 0, s, desc0, n0, v.n0_pad, v.a16, v.stats);
    hipLaunchKernelGGL(split_kernel, dim3(div_up(v.n1_pad, kSplitRows)), dim3(256), 0, s, desc1, n1, v.n1_pad, v.b16, v.stats + 2);
    VS_TRY(launch_ok("split_kernel"));
    return run_match(v, 1, false, nn_thresh, ratio, s, nullptr);
}

}  // extern "C"

namespace {
// bare chain of MFMAs on four accumulators, operands in registers (see vs_debug_mfma_rate)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mfma_rate_kernel(const float *__restrict__ in, float *__restrict__ out, int iters)
{
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)in[(threadIdx.x + i) & 1023]; b[i] = (_Float16)in[(threadIdx.x + 8 + i) & 1023]; }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

extern "C" int vs_debug_mfma_rate(int32_t millis, int32_t random_operands, float *tflops)
{
    VS_CHECK(tflops != nullptr && millis > 0 && millis <= 5000, VS_ERR_INVALID, "vs_debug_mfma_rate: bad arguments");
    int dev = 0, cus = 0;
    VS_HIP(hipGetDevice(&dev));
    VS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    DevBuf<float> in, out;
    VS_TRY(in.alloc(1024));
    VS_TRY(out.alloc((size_t)cus * 256));
    std::vector<float> h(1024, 0.f);
    if (random_operands) {
        unsigned x = 12345u;
        for (float &v : h) { x = x * 1664525u + 1013904223u; v = (float)((int)(x >> 8) % 2001 - 1000) * 1e-3f; }
    }
    VS_HIP(hipMemcpy(in.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    VS_HIP(hipEventCreate(&e0));
    VS_HIP(hipEventCreate(&e1));
    const int iters = 20000;                                   // about 1.5 ms per launch
    const int launches = std::max(2, (int)(millis / 1.5f));
    int rc = VS_OK;
    float ms = 0.f;
    for (int l = 0; l < launches && rc == VS_OK; l++) {      // the last launch, after the clock has settled, is the one timed
        if (l == launches - 1 && hipEventRecord(e0, nullptr) != hipSuccess) rc = VS_ERR_HIP;
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, nullptr, in.p, out.p, iters);
    }
    if (rc == VS_OK && (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = VS_ERR_HIP;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != VS_OK) { set_error("vs_debug_mfma_rate: HIP call failed"); return rc; }
    const double flops = (double)cus * 4 /*waves*/ * 4.0 * iters * (2.0 * 32 * 32 * 16);
    *tflops = (float)(flops / (ms * 1e-3) / 1e12);
    return VS_OK;
}

#ifdef VS_MATCH_STAMPS
extern "C" int vs_debug_match_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_stamps), sizeof(unsigned long long) * 4 * 64);
}
#endif
