// Device-side construction of the bundle-adjustment index structures (included by ba.hip only).
//
// From the caller's edge list (camera index, landmark index, pixel per observation; reference:
// the edge loop of src/core/optimizer.cc:102-143) the block kernels need: the observations sorted
// by landmark (cameras ascending inside a landmark) and by camera, the list of camera pairs that
// share a landmark (the off-diagonal blocks of the reduced system, sorted by row then column)
// and, per pair, the landmarks it shares in ascending order.  On the host these are counting
// sorts and table walks over every observation and every pair (tens of milliseconds at a million
// observations); here they are four device radix sorts, a few scans and element-wise kernels, and
// the host only sees the per-camera and per-pair counts it cuts into work items.
#pragma once
#include <climits>
#include <hipcub/hipcub.hpp>

namespace vs {
namespace build {

struct Scalars {
    int bad_cam, bad_pt;      // smallest observation with an index out of range, INT_MAX = none
    int dup;                  // smallest sorted position whose (landmark, camera) repeats the one before, INT_MAX = none
    int sorted;               // 1: landmarks already are in first-camera order
    int pt_lo, pt_hi;         // landmark range of this shard (internal numbering)
    int lo_pos, hi_pos;       // its range of sorted observations
    int n_pairs;              // camera-pair entries over all landmarks
    int nb_off, n_ent;        // distinct camera pairs; entries of this shard's landmarks
    int span;                 // longest camera range seen by one landmark tile
    int hub_pos, hub_cam, hub_pt;   // observation 0 in landmark-major position (-1: not on this shard)
    int pad;
};

__global__ void init_kernel(Scalars *sc, int Np, int Nc, int *first_cam)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < Np) first_cam[j] = Nc;
    if (j == 0) {
        sc->bad_cam = sc->bad_pt = sc->dup = INT_MAX;
        sc->sorted = 1;
        sc->pt_lo = 0; sc->pt_hi = Np; sc->lo_pos = 0; sc->hi_pos = 0;
        sc->n_pairs = 0; sc->nb_off = 0; sc->n_ent = 0; sc->span = 1;
        sc->hub_pos = sc->hub_cam = sc->hub_pt = -1;
    }
}

__global__ void first_cam_kernel(int Ng, int Nc, int Np, const int *__restrict__ cam, const int *__restrict__ pt,
                                 int *first_cam, unsigned char *cam_used, Scalars *sc)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Ng) return;
    const int c = cam[o], j = pt[o];
    if (c < 0 || c >= Nc) { atomicMin(&sc->bad_cam, o); return; }
    if (j < 0 || j >= Np) { atomicMin(&sc->bad_pt, o); return; }
    atomicMin(&first_cam[j], c);
    cam_used[c] = 1;
}

__global__ void sorted_check_kernel(int Np, const int *__restrict__ first_cam, Scalars *sc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > 0 && j < Np && first_cam[j - 1] > first_cam[j]) sc->sorted = 0;
}

__global__ void iota_kernel(int n, int *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// old2new from new2old (identity when `identity` is set: landmarks keep the caller's numbers)
__global__ void invert_kernel(int Np, const int *__restrict__ new2old, int *old2new, int identity)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Np) return;
    if (identity) old2new[q] = q; else old2new[new2old[q]] = q;
}

__global__ void obs_key_kernel(int Ng, int Nc, int Np, const int *__restrict__ cam, const int *__restrict__ pt,
                               const int *__restrict__ old2new, unsigned long long *keys, int *vals,
                               unsigned char *pt_used)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Ng) return;
    const int c = cam[o], j = pt[o];
    const bool ok = c >= 0 && c < Nc && j >= 0 && j < Np;     // bad observations are reported, not dereferenced
    const int jn = ok ? old2new[j] : 0;
    keys[o] = ((unsigned long long)(unsigned)jn << 32) | (unsigned)(ok ? c : 0);
    vals[o] = o;
    if (ok) pt_used[jn] = 1;
}

__global__ void dup_check_kernel(int Ng, const unsigned long long *__restrict__ keys, Scalars *sc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 0 && k < Ng && keys[k] == keys[k - 1]) atomicMin(&sc->dup, k);
}

// gptr[j] = first sorted position whose landmark is >= j, for j in [0, Np]
__global__ void landmark_ptr_kernel(int Ng, int Np, const unsigned long long *__restrict__ keys, int *gptr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > Np) return;
    int lo = 0, hi = Ng;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)(keys[mid] >> 32) < j) lo = mid + 1; else hi = mid;
    }
    gptr[j] = lo;
}

// Shard bounds (a contiguous landmark range balanced by observation count) and the number of
// camera-pair entries of each landmark; one thread finishes the scalars.
__global__ void shard_bounds_kernel(int Ng, int Np, int rank, int world, const int *__restrict__ gptr, Scalars *sc)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int pt_lo = 0, pt_hi = Np;
    if (world > 1) {
        auto bound = [&](int r) {
            const long long target = (long long)Ng * r / world;
            int lo = 0, hi = Np + 1;                     // first index of gptr[0..Np] with gptr >= target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (gptr[mid] < target) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        pt_lo = min(bound(rank), Np);
        pt_hi = rank == world - 1 ? Np : min(bound(rank + 1), Np);
        if (pt_hi < pt_lo) pt_hi = pt_lo;
    }
    sc->pt_lo = pt_lo; sc->pt_hi = pt_hi;
    sc->lo_pos = gptr[pt_lo]; sc->hi_pos = gptr[pt_hi];
}

__global__ void pair_count_kernel(int Np, const int *__restrict__ gptr, int *npairs)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    const long long k = gptr[j + 1] - gptr[j];
    npairs[j] = (int)(k * (k - 1) / 2);
}

// total of an exclusive scan: last offset + last count
__global__ void scan_total_kernel(int n, const int *__restrict__ off, const int *__restrict__ cnt, int *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = n > 0 ? off[n - 1] + cnt[n - 1] : 0;
}

__global__ void local_obs_kernel(int No, int lo_pos, const unsigned long long *__restrict__ keys,
                                 const int *__restrict__ gord, const double *__restrict__ uv, int *lcam, int *lpt,
                                 int *lobs, double *luv, int want_hub, Scalars *sc)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= No) return;
    const unsigned long long key = keys[lo_pos + pos];
    const int o = gord[lo_pos + pos];
    const int c = (int)(unsigned)(key & 0xffffffffull), j = (int)(key >> 32);
    lcam[pos] = c; lpt[pos] = j; lobs[pos] = o;
    *reinterpret_cast<double2 *>(luv + 2 * (size_t)pos) = *reinterpret_cast<const double2 *>(uv + 2 * (size_t)o);
    if (o == 0 && want_hub) { sc->hub_pos = pos; sc->hub_cam = c; sc->hub_pt = j; }
}

__global__ void local_ptr_kernel(int Np, int No, int lo_pos, const int *__restrict__ gptr, int *pt_ptr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > Np) return;
    pt_ptr[j] = min(max(gptr[j] - lo_pos, 0), No);
}

// Camera pairs (a < c) of every landmark, in the landmark's slot range of the entry list.
__global__ void pair_gen_kernel(int Np, const int *__restrict__ gptr, const int *__restrict__ poff,
                                const unsigned long long *__restrict__ okeys, unsigned long long *pkeys, int *pvals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    const int b = gptr[j], e = gptr[j + 1];
    int at = poff[j];
    for (int a = b; a < e; a++) {
        const unsigned long long ca = okeys[a] & 0xffffffffull;
        for (int c = a + 1; c < e; c++) {
            pkeys[at] = (ca << 32) | (okeys[c] & 0xffffffffull);
            pvals[at] = j;
            at++;
        }
    }
}

// head[i] = 1 where a new camera pair starts; loc[i] = 1 where the entry's landmark is on this shard
__global__ void pair_flags_kernel(int P, const unsigned long long *__restrict__ pkeys, const int *__restrict__ pvals,
                                  int pt_lo, int pt_hi, int *head, int *loc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    head[i] = i == 0 || pkeys[i] != pkeys[i - 1];
    const int j = pvals[i];
    loc[i] = j >= pt_lo && j < pt_hi;
}

// Blocks and entries from the scans: bidx = exclusive scan of head (so the block of entry i is
// bidx[i] + head[i] - 1), lsum = exclusive scan of loc.
__global__ void pair_scatter_kernel(int P, const unsigned long long *__restrict__ pkeys, const int *__restrict__ pvals,
                                    const int *__restrict__ head, const int *__restrict__ bidx,
                                    const int *__restrict__ loc, const int *__restrict__ lsum,
                                    int *brow, int *bcol, int *bptr, int *bent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (head[i]) {
        const int b = bidx[i];
        brow[b] = (int)(pkeys[i] >> 32);
        bcol[b] = (int)(unsigned)(pkeys[i] & 0xffffffffull);
        bptr[b] = lsum[i];
    }
    if (loc[i]) bent[lsum[i]] = pvals[i];
    if (i == P - 1) bptr[bidx[i] + head[i]] = lsum[i] + loc[i];
}

__global__ void camera_major_kernel(int No, const int *__restrict__ cpos, const int *__restrict__ lpt,
                                    const double *__restrict__ luv, int *cpt, double *cuv)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= No) return;
    const int pos = cpos[k];
    cpt[k] = lpt[pos];
    *reinterpret_cast<double2 *>(cuv + 2 * (size_t)k) = *reinterpret_cast<const double2 *>(luv + 2 * (size_t)pos);
}

// cam_ptr[c] = first camera-major position whose camera is >= c, for c in [0, Nc]
__global__ void camera_ptr_kernel(int No, int Nc, const unsigned *__restrict__ ccam, int *cam_ptr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > Nc) return;
    int lo = 0, hi = No;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)ccam[mid] < c) lo = mid + 1; else hi = mid;
    }
    cam_ptr[c] = lo;
}

// First camera and camera span of each landmark tile (tile_landmarks landmarks): one wave per tile.
__global__ void __launch_bounds__(64)
tile_span_kernel(int n_tiles, int Np, int Nc, int tile_landmarks, const int *__restrict__ pt_ptr,
                 const int *__restrict__ lcam, int *lo, Scalars *sc)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= n_tiles) return;
    const int j0 = w * tile_landmarks, j1 = min(Np, j0 + tile_landmarks);
    int mn = Nc, mx = -1;
    for (int p = pt_ptr[j0] + lane; p < pt_ptr[j1]; p += 64) { const int c = lcam[p]; mn = min(mn, c); mx = max(mx, c); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) {
        if (mx < 0) { mn = 0; mx = 0; }
        lo[w] = mn;
        atomicMax(&sc->span, mx - mn + 1);
    }
}

// Camera range seen by the landmarks of every shard (one workgroup per rank): the shard bounds
// as shard_bounds_kernel takes them, then min / max of the cameras of its sorted observations.
__global__ void __launch_bounds__(256)
rank_camera_range_kernel(int Ng, int Np, int Nc, int world, const int *__restrict__ gptr,
                         const unsigned long long *__restrict__ okeys, int *cam_lo, int *cam_hi)
{
    __shared__ int s_mn[4], s_mx[4];
    const int r = blockIdx.x;
    auto bound = [&](int q) {
        const long long target = (long long)Ng * q / world;
        int lo = 0, hi = Np + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (gptr[mid] < target) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    int pt_lo = min(bound(r), Np), pt_hi = r == world - 1 ? Np : min(bound(r + 1), Np);
    if (pt_hi < pt_lo) pt_hi = pt_lo;
    int mn = Nc, mx = -1;
    for (int p = gptr[pt_lo] + (int)threadIdx.x; p < gptr[pt_hi]; p += 256) {
        const int c = (int)(unsigned)(okeys[p] & 0xffffffffull);
        mn = min(mn, c); mx = max(mx, c);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cam_lo[r] = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        cam_hi[r] = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    }
}

inline int bits_for(long long n_values)      // bits that hold values in [0, n_values)
{
    int b = 1;
    while (((long long)1 << b) < n_values) b++;
    return b;
}

// Stable radix sort of (key, value) pairs on `s`; the temporary storage grows on demand.
template <typename K>
int sort_pairs(DevBuf<unsigned char> &tmp, const K *kin, K *kout, const int *vin, int *vout, int n, int end_bit,
               hipStream_t s)
{
    size_t bytes = 0;
    VS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, n, 0, end_bit, s));
    if (tmp.n < bytes) VS_TRY(tmp.alloc(bytes));
    VS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin, kout, vin, vout, n, 0, end_bit, s));
    return VS_OK;
}

inline int exclusive_sum(DevBuf<unsigned char> &tmp, const int *in, int *out, int n, hipStream_t s)
{
    size_t bytes = 0;
    VS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, n, s));
    if (tmp.n < bytes) VS_TRY(tmp.alloc(bytes));
    VS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, n, s));
    return VS_OK;
}

}  // namespace build
}  // namespace vs
