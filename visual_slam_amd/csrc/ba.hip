// Host side of the bundle-adjustment C ABI: index-structure construction, kernel sequencing,
// the Levenberg-Marquardt loop.  Reference interfaces replaced are cited in include/vslam_hip.h.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "ba_kernels.h"
#include "dense_solve.h"
#include "bcr_solve.h"
#include "ba_build.h"
#include "ba_small.h"

using namespace vs;

namespace {

// Open-addressing map from a camera pair (a < b) to a block id; grows at 50 % load.
struct PairMap {
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    uint64_t mask = 0;
    size_t count = 0;
    static constexpr uint64_t kEmpty = ~0ull;
    void init(size_t cap_hint)
    {
        size_t cap = 1024;
        while (cap < cap_hint) cap <<= 1;
        keys.assign(cap, kEmpty);
        vals.assign(cap, -1);
        mask = cap - 1;
        count = 0;
    }
    static uint64_t hash(uint64_t k)
    {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return k;
    }
    size_t slot(uint64_t k) const
    {
        size_t s = hash(k) & mask;
        while (keys[s] != kEmpty && keys[s] != k) s = (s + 1) & mask;
        return s;
    }
    void grow()
    {
        std::vector<uint64_t> ok;
        std::vector<int32_t> ov;
        ok.swap(keys); ov.swap(vals);
        init(ok.size() * 2);
        for (size_t i = 0; i < ok.size(); i++)
            if (ok[i] != kEmpty) { const size_t s = slot(ok[i]); keys[s] = ok[i]; vals[s] = ov[i]; count++; }
    }
    // true when the key was absent (it is inserted with value -1)
    bool insert(uint64_t k)
    {
        if ((count + 1) * 2 > keys.size()) grow();
        const size_t s = slot(k);
        if (keys[s] != kEmpty) return false;
        keys[s] = k; count++;
        return true;
    }
    void set(uint64_t k, int32_t v) { vals[slot(k)] = v; }
    int32_t get(uint64_t k) const
    {
        const size_t s = slot(k);
        return keys[s] == kEmpty ? -1 : vals[s];
    }
};

}  // namespace

struct vs_ba {
    int Nc = 0, Np = 0;
    int64_t No_global = 0;   // edges handed to set_observations
    int No = 0;              // edges this handle linearises (== No_global unless sharded)
    Intr K{};
    hipStream_t stream = nullptr;
    bool fix_points = false;
    double huber_a = 0.0;
    bool have_cams = false, have_pts = false, have_obs = false, linearized = false;
    bool have_scale = false;
    // vs_ba_solve linearises an accepted point once, with the damping of the next step, and takes the
    // gradient test from that pass: red/lmrec then already hold what the next step starts from
    bool prelin = false;
    double prelin_radius = 0.0;
    int rank = 0, world = 1;
    HuberObs hub{ -1, -1, -1, 0.0 };
    int max_pair_gap = 0;   // largest |a - b| over the camera pairs of the reduced system
    bool step_has_gmax = false;          // the last LM step's fused solve also wrote max |gradient|
    int nb_off = 0, n_diag_items = 0, n_off_items = 0, n_lm_blocks = 0, n_lm4_blocks = 0, item_waves = 4;
    bool stage_cams = false;
    int lds_cam_grid = 0;                 // > 0: landmark_lds_kernel (every camera in LDS, one workgroup per CU) with this grid
    int64_t red_len = 0;
    vs_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    vs_allgather_fn allgather = nullptr;
    void *allgather_user = nullptr;
    // what each shard sends when the reduce buffers are combined (exchange_reduced)
    std::vector<ShardSeg> segs;
    DevBuf<ShardSeg> d_segs;
    int64_t seg_stride = 0;
    DevBuf<real> seg_send, seg_recv;
    vs_comm *comm = nullptr;
    const void *known_dev_dc = nullptr, *known_dev_dp = nullptr;   // damping pointers already classified as device memory
    DevBuf<real> eval_res, eval_Jc, eval_Jp;                       // staging of vs_ba_evaluate, grown on demand
    int pt_lo = 0, pt_hi = 0;   // landmarks this handle linearises
    int tile_lo = 0, tile_hi = 0;   // landmark-kernel tiles (kLmPerBlock landmarks each) that hold them
    // Landmarks are numbered internally by the first camera that sees them (neighbouring lanes then
    // read neighbouring camera records and landmark records, and a shard touches a contiguous
    // camera range).  renum: the internal numbering differs from the caller's; new2old[j] = the
    // caller's index of internal landmark j.  pts_user / perm_stage: caller-order staging.
    bool renum = false;
    DevBuf<int> new2old;
    DevBuf<real> pts_user, perm_stage;
    DevBuf<real> rank_slots;    // one value per rank: maxima through a sum-only collective
    DevBuf<unsigned char> arena;             // small edge sets: every index array in one block (build_structure)
    PinnedBuf<unsigned char> arena_stage;
    // Small parameter uploads pass through page-locked staging owned by the handle, so the setters
    // need not wait for the copy: [t | rpy | points].  busy[r]: a copy out of region r may still be
    // in flight (r = 0 cameras, 1 points, 2 the arena staging).
    PinnedBuf<SmallOut, true> small_host;    // motion_only_solve_kernel's summary, written by the kernel, polled by the host
    int small_seq = 0;
    PinnedBuf<real, true> cams_host;         // [t | rpy] as the motion-only kernel left them (valid while cams_host_valid)
    bool cams_host_valid = false;
    bool init_has_gmax = false;              // the first pass of the solve left max |gradient| in scalars[4]
    PinnedBuf<real> param_stage;
    bool busy[3] = { false, false, false };

    std::vector<int32_t> h_brow, h_bcol;
    hipEvent_t *prof_ev = nullptr;   // when set: 6 events recorded around linearize()'s kernels

    DevBuf<real> t, rpy, pts, t2, rpy2, pts2;
    DevBuf<CamData> cams;
    DevBuf<int> pt_ptr, lcam, lpt, lobs, cam_ptr, cpt, brow, bcol, bent;
    DevBuf<int> ditem_ptr, oitem_ptr;
    DevBuf<int> lm_cam_lo;      // first camera seen by each landmark-kernel workgroup (range staging)
    int lm_cam_span = 0;        // longest such range, 0 = range staging not used
    int n_multi_diag = 0, n_multi_off = 0;
    int n_list_items = 0, n_small16 = 0, n_small32 = 0;   // items of the block kernel / of the small-block kernels
    DevBuf<BlockItem> small_items;              // blocks of 1..16 entries, then blocks of 17..32
    DevBuf<BlockItem> items, diag_items;
    DevBuf<unsigned char> cam_used, pt_used;
    DevBuf<real> luv, cuv, lmrec, Vdiag, hub_w, cost_partial, diag_partial, off_partial, red;
    // cost_kernel's partials (one per 256 landmarks): a buffer of their own, because the landmark
    // kernels' partials (one per tile of kLmPerBlock landmarks, only a shard's own tiles rewritten)
    // are summed over the whole of cost_partial by the block and finalize kernels
    DevBuf<real> ccost_partial;
    DevBuf<real> scale_c, scale_p, lam_c, lam_c_user, lam_p_user, dcam, dpt, mcc_partial,
        cand_partial, gmax_partial, scalars, dense, linv, bcr_ws;
    DevBuf<int> info;
    PinnedBuf<real> host_scalars;   // 16 reals: [0,8) step scalars, 8 gradient maximum, 9 cost
    PinnedBuf<unsigned long long, true> mail;   // 8 step scalars + sequence number, written by small_tail_kernel, polled by the host
    unsigned long long mail_seq = 0;
    DevBuf<real> tail_partial;               // small_tail_kernel: four sums per workgroup
    DevBuf<int> tail_ticket;                 // ... and its ticket counter (zero between launches)
    DevBuf<real> radius_next;                // radius the tail kernel derived for the speculative linearisation

    real *S_diag() { return red.p; }
    real *S_off() { return red.p + (int64_t)Nc * 36; }
    real *rhs() { return S_off() + (int64_t)nb_off * 36; }
    real *udiag() { return rhs() + 6 * (int64_t)Nc; }
    real *gc() { return udiag() + 6 * (int64_t)Nc; }
    real *cost() { return gc() + 6 * (int64_t)Nc; }
};

namespace {

int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    return VS_OK;
}

int prep_cameras(vs_ba *h, const real *t, const real *rpy)
{
    hipLaunchKernelGGL(prep_cameras_kernel, dim3(div_up(h->Nc, 64)), dim3(64), 0, h->stream, h->Nc,
                       t, rpy, h->cams.p);
    return launch_check("prep_cameras");
}

// camera records, plus the sines and cosines the landmark kernel keeps behind them
// Dynamic LDS beyond the default limit for landmark_lds_kernel / backsub_lds_kernel: a per-device
// function attribute.
// The camera array shares the CU's 160 KB with the kernel's 24.1 KB of static LDS.
constexpr size_t kLdsCamBytesMax = 134 * 1024;
int landmark_lds_raise_limit()
{
    static std::mutex m;
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("hipGetDevice failed"); return VS_ERR_HIP; }
    std::lock_guard<std::mutex> lock(m);
    if (!raised[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(landmark_lds_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCamBytesMax) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(landmark_lds_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCamBytesMax) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(backsub_lds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCamBytesMax) != hipSuccess) {
            set_error("cannot raise the dynamic LDS limit of the landmark kernel");
            return VS_ERR_HIP;
        }
        raised[dev] = true;
    }
    return VS_OK;
}

size_t cam_lds_bytes(const vs_ba *h) { return h->stage_cams ? (size_t)h->Nc * (sizeof(CamData) + 6 * sizeof(real)) : 0; }

// caller-order rows -> internal order (dst may not alias src)
int rows_to_internal(vs_ba *h, int width, const real *src, real *dst)
{
    const int64_t n = (int64_t)width * h->Np;
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, h->Np, width, h->new2old.p, src, dst);
    return launch_check("gather_rows_kernel");
}
// internal rows (stride src_stride, first column src_off) -> caller order, densely packed
int rows_to_caller(vs_ba *h, int width, const real *src, int src_stride, int src_off, real *dst)
{
    const int64_t n = (int64_t)width * h->Np;
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, h->Np, width, h->new2old.p, src,
                       src_stride, src_off, dst);
    return launch_check("scatter_rows_kernel");
}

// Segments of the reduce buffer every shard sends (see ShardSeg): block ranges from the sorted
// block rows, the common stride, and the table the summing kernel reads.
int finish_segments(vs_ba *h, const std::vector<int> &cam_lo, const std::vector<int> &cam_hi)
{
    h->segs.assign((size_t)h->world, ShardSeg{ 0, -1, 0, 0 });
    h->seg_stride = 0;
    if (h->world <= 1) return VS_OK;
    for (int r = 0; r < h->world; r++) {
        ShardSeg g{ cam_lo[r], cam_hi[r], 0, 0 };
        if (g.cam_hi >= g.cam_lo) {
            g.blk_lo = (int)(std::lower_bound(h->h_brow.begin(), h->h_brow.end(), g.cam_lo) - h->h_brow.begin());
            g.blk_hi = (int)(std::upper_bound(h->h_brow.begin(), h->h_brow.end(), g.cam_hi) - h->h_brow.begin());
        } else { g.cam_lo = 0; g.cam_hi = -1; }
        h->segs[r] = g;
        h->seg_stride = std::max(h->seg_stride, seg_len(g));
    }
    h->seg_stride = (h->seg_stride + 1) / 2 * 2;
    VS_TRY(h->d_segs.upload(h->segs.data(), h->segs.size(), h->stream));
    VS_HIP(hipStreamSynchronize(h->stream));
    return VS_OK;
}

// Work-item lists of the block kernels from the camera-major and camera-pair entry counts
// (cam_ptr: Nc + 1, bptr: nb_off + 1, both on the host); also the landmark-tile range of this shard.
struct ItemLists {
    std::vector<BlockItem> items, diag_items, small_items;
    std::vector<int> di_ptr, oi_ptr;
};
int make_items(vs_ba *h, const int *cam_ptr, const int *bptr, ItemLists &out)
{
    const int Nc = h->Nc, Np = h->Np, nb_off = h->nb_off, No = h->No, pt_lo = h->pt_lo, pt_hi = h->pt_hi;
    const int n_ent = bptr[nb_off];
    // work items: one workgroup each; 4 waves per item and 4 entries per thread when the typical
    // block is large, else 1 wave and 2 entries (measured optima at 50 / 1000 / 10000 cameras).  Blocks that fit one item are written by the block kernel directly,
    // the others go through partials + finalize.
    {
        int64_t nb_used = 0, nc_used = 0;
        for (int c = 0; c < Nc; c++) nc_used += cam_ptr[c + 1] > cam_ptr[c];
        for (int k = 0; k < nb_off; k++) nb_used += bptr[k + 1] > bptr[k];
        const double avg = nb_used ? double(n_ent) / double(nb_used) : (nc_used ? double(No) / double(nc_used) : 0.0);
        h->item_waves = avg >= 512.0 ? 4 : 1;
        // a window of a few keyframes cannot fill the chip either way; as 4-wave items every block
        // is a single item and the finalize launch disappears, which is worth more there
        if ((int64_t)n_ent + No <= 32768) h->item_waves = 4;
    }
    if (const char *e = getenv("VS_BA_ITEM_WAVES")) h->item_waves = atoi(e) == 4 ? 4 : 1;   // tuning knobs
    int ept = h->item_waves == 4 ? kEpt4 : kEpt1;          // entries per thread and item
    if (const char *e = getenv("VS_BA_ITEM_EPT")) ept = std::min(ept, std::max(1, atoi(e)));
    // a 4-wave item holds up to max_chunks chunks of entries (the block kernel walks them in a loop
    // with the sums in registers): fewer partials, and at 50 cameras every block is one item
    const int chunk = h->item_waves * kWave * ept;
    const int max_chunks = h->item_waves == 4 ? 24 / kEpt4 : 1;   // 6144 entries per 4-wave item
    const int ch = chunk * max_chunks;
    // Items are listed block row by block row (camera c's diagonal item(s), then its
    // off-diagonal blocks (c, c') in column order) so that a contiguous range of items is a band
    // of cameras.  With fixed points only the diagonal items are launched (diag_items).
    std::vector<BlockItem> &items = out.items, &diag_items = out.diag_items, &small_items = out.small_items;
    std::vector<BlockItem> small32;
    items.clear(); diag_items.clear(); small_items.clear();
    // Off-diagonal blocks of at most 32 entries go to the small-block kernels (several blocks per
    // wave) when they hold a tenth of all entries or more: otherwise two more launches cost more
    // than the idle lanes of those blocks in the block kernel.
    bool pack_small = h->item_waves == 1 && getenv("VS_BA_NO_SMALL_BLOCKS") == nullptr;
    if (pack_small) {
        int64_t small_entries = 0;
        for (int k = 0; k < nb_off; k++) {
            const int nk = bptr[k + 1] - bptr[k];
            if (nk <= 32) small_entries += nk;
        }
        pack_small = small_entries * 10 >= (int64_t)n_ent;
    }
    std::vector<int> &di_ptr = out.di_ptr, &oi_ptr = out.oi_ptr;
    di_ptr.assign((size_t)Nc + 1, 0); oi_ptr.assign((size_t)nb_off + 1, 0);
    int n_diag_items = 0, n_off_items = 0;
    {
        int k = 0;   // off-diagonal blocks are sorted by (row, col): walk them row by row
        for (int c = 0; c < Nc; c++) {
            const size_t first = items.size();
            for (int b = cam_ptr[c]; b < cam_ptr[c + 1]; b += ch)
                items.push_back(BlockItem{ c, b, std::min(b + ch, cam_ptr[c + 1]), c, c, 0, n_diag_items++, 0 });
            di_ptr[c + 1] = n_diag_items;
            if (items.size() - first == 1) items[first].single = 1;
            for (size_t i = first; i < items.size(); i++) diag_items.push_back(items[i]);
            for (; k < nb_off && h->h_brow[k] == c; k++) {
                const int nk = bptr[k + 1] - bptr[k];
                if (pack_small && nk >= 1 && nk <= 32) {   // four / two such blocks share a wave of the small-block kernel
                    (nk <= 16 ? small_items : small32).push_back(BlockItem{ k, bptr[k], bptr[k + 1], h->h_brow[k], h->h_bcol[k], 1, n_off_items++, 0 });
                    oi_ptr[k + 1] = n_off_items;
                    continue;
                }
                const size_t f2 = items.size();
                for (int b = bptr[k]; b < bptr[k + 1]; b += ch)
                    items.push_back(BlockItem{ k, b, std::min(b + ch, bptr[k + 1]), h->h_brow[k], h->h_bcol[k], 0, n_off_items++, 0 });
                oi_ptr[k + 1] = n_off_items;
                if (items.size() - f2 == 1) items[f2].single = 1;
            }
        }
        for (int kk = 0; kk < nb_off; kk++) oi_ptr[kk + 1] = std::max(oi_ptr[kk + 1], oi_ptr[kk]);
    }
    // The block kernel gives workgroups with equal blockIdx % 8 one contiguous eighth of the list
    // and the hardware starts workgroups in blockIdx order: put each eighth's share of the
    // (heavier) diagonal items at its front so that they never sit in the last round.  Only with
    // the few large items of the 4-wave form: with many small items the last round is short anyway
    // and keeping a camera's diagonal item next to its row of blocks is worth more (measured).
    if (h->item_waves == 4 && !items.empty()) {
        std::vector<BlockItem> dg, off;
        for (const BlockItem &it : items) (it.ca == it.cb ? dg : off).push_back(it);
        const size_t n = items.size(), q = n / 8, r = n % 8;
        size_t id = 0, io = 0, pos = 0;
        for (size_t x = 0; x < 8; x++) {
            const size_t len = q + (x < r ? 1 : 0);
            size_t nd = dg.size() / 8 + (x < dg.size() % 8 ? 1 : 0);
            nd = std::min(nd, len);
            // off-diagonal items left over when the diagonal share does not fit go to later eighths
            for (size_t i = 0; i < nd; i++) items[pos++] = dg[id++];
            for (size_t i = nd; i < len; i++) {
                if (io < off.size()) items[pos++] = off[io++];
                else items[pos++] = dg[id++];
            }
        }
    }
    h->n_diag_items = n_diag_items;
    h->n_off_items = n_off_items;
    h->n_list_items = (int)items.size();
    h->n_small16 = (int)small_items.size();
    h->n_small32 = (int)small32.size();
    small_items.insert(small_items.end(), small32.begin(), small32.end());
    h->n_multi_diag = h->n_multi_off = 0;          // blocks that are not exactly one item: finalize sums (or zeroes) them
    for (int c = 0; c < Nc; c++) h->n_multi_diag += di_ptr[c + 1] - di_ptr[c] != 1;   // empty blocks too:
    for (int k = 0; k < nb_off; k++) h->n_multi_off += oi_ptr[k + 1] - oi_ptr[k] != 1; // finalize zeroes them
    h->n_lm_blocks = div_up(Np, kLmThreads);
    h->n_lm4_blocks = div_up(Np, kLmPerBlock);
    // a shard's landmark kernel runs over its own tiles only; the records and cost partials of the
    // other tiles are zeroed once (below) and never written
    h->tile_lo = h->world > 1 ? pt_lo / kLmPerBlock : 0;
    h->tile_hi = h->world > 1 ? std::max(h->tile_lo + 1, div_up(pt_hi, kLmPerBlock)) : h->n_lm4_blocks;
    h->tile_hi = std::min(h->tile_hi, std::max(h->n_lm4_blocks, 1));
    if (h->tile_lo >= h->tile_hi) h->tile_lo = std::max(0, h->tile_hi - 1);

    h->stage_cams = (size_t)Nc * (sizeof(CamData) + 6 * sizeof(real)) <= 48 * 1024;
    if (items.empty()) items.push_back(BlockItem{ 0, 0, 0, 0, 0, 0, 0, 0 });
    if (diag_items.empty()) diag_items.push_back(BlockItem{ 0, 0, 0, 0, 0, 0, 0, 0 });
    if (small_items.empty()) small_items.push_back(BlockItem{ 0, 0, 0, 0, 0, 0, 0, 0 });
    if (di_ptr.empty()) di_ptr.push_back(0);
    if (oi_ptr.empty()) oi_ptr.push_back(0);
    return VS_OK;
}

// The all-cameras-in-a-CU's-LDS form of the landmark kernel, where it applies (after make_items).
int choose_lds_cameras(vs_ba *h)
{
    const int Nc = h->Nc;
    // Up to about 1200 cameras every camera fits the LDS of a CU as one array (one persistent
    // workgroup per CU, no separate camera pass): preferred over the range staging where both apply
    // (config 4 in keyframe order: 31 us against 33 + 4 us for range staging and its camera pass)
    h->lds_cam_grid = 0;
    if (!h->stage_cams && (size_t)Nc * kCamLdsReals * sizeof(real) <= kLdsCamBytesMax &&
        getenv("VS_BA_NO_LDS_CAMS") == nullptr) {
        int dev = 0, cus = 0;
        VS_HIP(hipGetDevice(&dev));
        VS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        VS_TRY(landmark_lds_raise_limit());
        h->lds_cam_grid = std::max(1, std::min(cus, div_up(h->tile_hi - h->tile_lo, kLmBigThreads / kLmThreads)));
    }

    return VS_OK;
}

// Buffers whose size follows from the item lists, zeroed where kernels rely on it.
int alloc_work_buffers(vs_ba *h)
{
    hipStream_t s = h->stream;
    const int Nc = h->Nc, nb_off = h->nb_off;
    VS_TRY(h->diag_partial.alloc((size_t)std::max(h->n_diag_items, 1) * kDiagVals));
    VS_TRY(h->off_partial.alloc((size_t)std::max(h->n_off_items, 1) * kOffVals));
    VS_TRY(h->cost_partial.alloc((size_t)std::max(h->n_lm4_blocks, 1)));
    VS_HIP(hipMemsetAsync(h->cost_partial.p, 0, (size_t)std::max(h->n_lm4_blocks, 1) * sizeof(real), s));
    if (h->lmrec.p) VS_HIP(hipMemsetAsync(h->lmrec.p, 0, h->lmrec.n * sizeof(real), s));
    if (h->Vdiag.p) VS_HIP(hipMemsetAsync(h->Vdiag.p, 0, h->Vdiag.n * sizeof(real), s));
    VS_TRY(h->ccost_partial.alloc((size_t)std::max(h->n_lm_blocks, 1)));
    VS_TRY(h->mcc_partial.alloc((size_t)std::max(h->n_lm_blocks, 1)));
    h->red_len = (int64_t)Nc * 36 + (int64_t)nb_off * 36 + 18 * (int64_t)Nc + 1;
    VS_TRY(h->red.alloc((size_t)h->red_len));
    VS_HIP(hipMemsetAsync(h->red.p, 0, (size_t)h->red_len * sizeof(real), s));
    return VS_OK;
}

// Builds every index structure from the caller's edge list (host side, once per edge set).
int build_structure(vs_ba *h, const int32_t *cam, const int32_t *pt_user, const double *uv)
{
    const int Nc = h->Nc, Np = h->Np;
    const int64_t Ng = h->No_global;
    h->have_obs = false;     // a call that fails part-way leaves no usable edge set behind
    h->linearized = false;
    // VS_BA_BUILD_TIMING=1 prints where the time of this function goes (developer knob)
    const bool timing = getenv("VS_BA_BUILD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "build_structure: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    for (int64_t o = 0; o < Ng; o++) {
        VS_CHECK(cam[o] >= 0 && cam[o] < Nc, VS_ERR_INVALID, "observation %lld: camera index %d out of range [0,%d)", (long long)o, cam[o], Nc);
        VS_CHECK(pt_user[o] >= 0 && pt_user[o] < Np, VS_ERR_INVALID, "observation %lld: point index %d out of range [0,%d)", (long long)o, pt_user[o], Np);
    }
    // Internal landmark numbering: by the first (lowest) camera that sees the landmark, the caller's
    // order kept among equals (a counting sort).  Landmarks the reference creates keyframe by
    // keyframe already come that way, and then nothing is renumbered.
    std::vector<int32_t> ptn_store;
    const int32_t *pt = pt_user;
    {
        // points held in the internal order of an earlier edge set go back to the caller's order first
        if (h->renum && h->have_pts) {
            VS_TRY(rows_to_caller(h, 3, h->pts.p, 3, 0, h->pts_user.p));
            VS_HIP(hipMemcpyAsync(h->pts.p, h->pts_user.p, 3 * (size_t)Np * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
            VS_HIP(hipStreamSynchronize(h->stream));
        }
        h->renum = false;
        std::vector<int32_t> first_cam((size_t)Np, Nc);
        for (int64_t o = 0; o < Ng; o++) first_cam[pt_user[o]] = std::min(first_cam[pt_user[o]], cam[o]);
        bool sorted = true;
        for (int j = 1; j < Np && sorted; j++) sorted = first_cam[j - 1] <= first_cam[j];
        if (!sorted && getenv("VS_BA_NO_RENUMBER") == nullptr) {
            std::vector<int32_t> start((size_t)Nc + 2, 0), new2old((size_t)Np), old2new((size_t)Np);
            for (int j = 0; j < Np; j++) start[first_cam[j] + 1]++;
            for (int c = 0; c <= Nc; c++) start[c + 1] += start[c];
            for (int j = 0; j < Np; j++) { const int32_t q = start[first_cam[j]]++; new2old[q] = j; old2new[j] = q; }
            ptn_store.resize((size_t)std::max<int64_t>(Ng, 1));
            for (int64_t o = 0; o < Ng; o++) ptn_store[o] = old2new[pt_user[o]];
            pt = ptn_store.data();
            VS_TRY(h->new2old.upload(new2old.data(), new2old.size(), h->stream));
            if (h->pts_user.n < 3 * (size_t)Np) VS_TRY(h->pts_user.alloc(3 * (size_t)std::max(Np, 1)));
            if (h->perm_stage.n < 9 * (size_t)Np) VS_TRY(h->perm_stage.alloc(9 * (size_t)std::max(Np, 1)));
            h->renum = true;
            if (h->have_pts) {   // the points were set before the edges: bring them into the internal order
                VS_HIP(hipMemcpyAsync(h->pts_user.p, h->pts.p, 3 * (size_t)Np * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
                VS_TRY(rows_to_internal(h, 3, h->pts_user.p, h->pts.p));
            }
            VS_HIP(hipStreamSynchronize(h->stream));
        }
    }
    lap("landmark numbering");
    // global landmark CSR
    std::vector<int64_t> gptr((size_t)Np + 1, 0);
    for (int64_t o = 0; o < Ng; o++) gptr[pt[o] + 1]++;
    for (int j = 0; j < Np; j++) gptr[j + 1] += gptr[j];
    std::vector<int64_t> gord((size_t)std::max<int64_t>(Ng, 1));
    {
        std::vector<int64_t> fill(gptr.begin(), gptr.end() - 1);
        for (int64_t o = 0; o < Ng; o++) gord[fill[pt[o]]++] = o;
    }
    lap("landmark CSR");
    // shard: contiguous landmark range balanced by observation count
    int pt_lo = 0, pt_hi = Np;
    if (h->world > 1) {
        auto bound = [&](int r) {
            const int64_t target = Ng * r / h->world;
            return (int)(std::lower_bound(gptr.begin(), gptr.end(), target) - gptr.begin());
        };
        pt_lo = std::min(bound(h->rank), Np);
        pt_hi = h->rank == h->world - 1 ? Np : std::min(bound(h->rank + 1), Np);
        if (pt_hi < pt_lo) pt_hi = pt_lo;
    }
    h->pt_lo = pt_lo; h->pt_hi = pt_hi;
    const int64_t lo_pos = gptr[pt_lo], hi_pos = gptr[pt_hi];
    VS_CHECK(hi_pos - lo_pos < (int64_t)1 << 31, VS_ERR_UNSUPPORTED, "more than 2^31 observations on one GPU");
    const int No = (int)(hi_pos - lo_pos);
    h->No = No;

    // landmark-major arrays (local), cameras ascending within a landmark
    std::vector<int> pt_ptr((size_t)Np + 1, 0), lcam((size_t)std::max(No, 1)), lpt((size_t)std::max(No, 1)), lobs((size_t)std::max(No, 1));
    std::vector<real> luv((size_t)std::max(No, 1) * 2);
    std::vector<unsigned char> cam_used((size_t)Nc, 0), pt_used((size_t)Np, 0);
    for (int64_t o = 0; o < Ng; o++) { cam_used[cam[o]] = 1; pt_used[pt[o]] = 1; }
    h->hub = HuberObs{ -1, -1, -1, (real)h->huber_a };
    std::vector<std::pair<int, int64_t>> tmp;
    // Global pattern of camera pairs + local entries.  Every pair met gets a slot number in order
    // of discovery: pairs (a, c) with c - a < kNear (keyframes in time order: nearly all of them)
    // through a table of Nc x kNear slots, the others through a hash map.  The slots of the local
    // landmarks' pairs are kept in the order they are met, so the two passes that bin the entries
    // by block below are linear walks with no lookups.
    constexpr int kNear = 64;
    std::vector<int32_t> near_tab((size_t)Nc * kNear, -1);
    PairMap pmap;
    pmap.init(1 << 12);
    std::vector<uint64_t> uniq;
    std::vector<int32_t> pair_slot;
    pair_slot.reserve((size_t)std::max<int64_t>(Ng, 16) * 2);
    auto slot_of = [&](int a, int c) -> int32_t {
        const uint64_t key = ((uint64_t)(uint32_t)a << 32) | (uint32_t)c;
        if (c - a < kNear) {
            int32_t &t = near_tab[(size_t)a * kNear + (c - a)];
            if (t < 0) { t = (int32_t)uniq.size(); uniq.push_back(key); }
            return t;
        }
        if (pmap.insert(key)) { pmap.set(key, (int32_t)uniq.size()); uniq.push_back(key); }
        return pmap.get(key);
    };
    // camera range seen by the landmarks of every shard (what a rank sends in exchange_reduced)
    std::vector<int> rank_first((size_t)h->world + 1, Np), rank_cam_lo((size_t)h->world, Nc), rank_cam_hi((size_t)h->world, -1);
    for (int r = 0; r < h->world; r++) {
        const int64_t target = Ng * r / h->world;
        rank_first[r] = std::min((int)(std::lower_bound(gptr.begin(), gptr.end(), target) - gptr.begin()), Np);
    }
    rank_first[0] = 0;
    int rank_of_j = 0;
    for (int j = 0; j < Np; j++) {
        const int64_t b = gptr[j], e = gptr[j + 1];
        tmp.clear();
        for (int64_t k = b; k < e; k++) tmp.emplace_back(cam[gord[k]], gord[k]);
        std::sort(tmp.begin(), tmp.end());
        while (rank_of_j + 1 < h->world && j >= rank_first[rank_of_j + 1]) rank_of_j++;
        if (!tmp.empty()) {
            rank_cam_lo[rank_of_j] = std::min(rank_cam_lo[rank_of_j], tmp.front().first);
            rank_cam_hi[rank_of_j] = std::max(rank_cam_hi[rank_of_j], tmp.back().first);
        }
        for (size_t a = 1; a < tmp.size(); a++)
            VS_CHECK(tmp[a].first != tmp[a - 1].first, VS_ERR_INVALID, "point %d is observed twice by camera %d", j, tmp[a].first);
        const bool local = j >= pt_lo && j < pt_hi;
        if (local) {
            pt_ptr[j] = (int)(b - lo_pos);
            for (size_t a = 0; a < tmp.size(); a++) {
                const int pos = (int)(b - lo_pos + a);
                const int64_t o = tmp[a].second;
                lcam[pos] = tmp[a].first; lpt[pos] = j; lobs[pos] = (int)o;
                luv[2 * (size_t)pos] = uv[2 * o]; luv[2 * (size_t)pos + 1] = uv[2 * o + 1];
                if (o == 0 && h->huber_a > 0.0) { h->hub.pos = pos; h->hub.cam = tmp[a].first; h->hub.pt = j; }
            }
        } else {
            pt_ptr[j] = j < pt_lo ? 0 : No;
        }
        for (size_t a = 0; a < tmp.size(); a++)
            for (size_t c = a + 1; c < tmp.size(); c++) {
                const int32_t sl = slot_of(tmp[a].first, tmp[c].first);
                if (local) pair_slot.push_back(sl);
            }
    }
    lap("landmark-major + pair pattern");
    pt_ptr[Np] = No;
    for (int j = pt_hi; j < Np; j++) pt_ptr[j] = No;
    VS_CHECK(Ng < ((int64_t)1 << 31), VS_ERR_UNSUPPORTED, "more than 2^31 observations");
    // blocks in (row, column) order: block number of every discovery slot
    const int nb_off = (int)uniq.size();
    std::vector<int32_t> slot_block((size_t)nb_off);
    {
        std::vector<std::pair<uint64_t, int32_t>> keyed((size_t)nb_off);
        for (int b = 0; b < nb_off; b++) keyed[b] = { uniq[b], b };
        std::sort(keyed.begin(), keyed.end());
        for (int b = 0; b < nb_off; b++) { uniq[b] = keyed[b].first; slot_block[keyed[b].second] = b; }
    }
    h->nb_off = nb_off;
    h->h_brow.resize((size_t)nb_off); h->h_bcol.resize((size_t)nb_off);
    h->max_pair_gap = 0;
    for (int b = 0; b < nb_off; b++) {
        h->h_brow[b] = (int32_t)(uniq[b] >> 32); h->h_bcol[b] = (int32_t)(uniq[b] & 0xffffffffu);
        h->max_pair_gap = std::max(h->max_pair_gap, h->h_bcol[b] - h->h_brow[b]);
    }
    // local entries per block (counting sort; landmarks ascending within a block)
    std::vector<int> bptr((size_t)nb_off + 1, 0);
    for (int32_t &sl : pair_slot) { sl = slot_block[sl]; bptr[sl + 1]++; }
    for (int b = 0; b < nb_off; b++) {
        VS_CHECK((int64_t)bptr[b + 1] + bptr[b] < ((int64_t)1 << 31), VS_ERR_UNSUPPORTED, "more than 2^31 camera-pair entries on one GPU");
        bptr[b + 1] += bptr[b];
    }
    const int n_ent = bptr[nb_off];
    std::vector<int> bent((size_t)std::max(n_ent, 1));
    {
        std::vector<int> fill(bptr.begin(), bptr.end() - 1);
        size_t at = 0;                                     // pair_slot follows the landmarks in order
        for (int j = pt_lo; j < pt_hi; j++) {
            const int64_t k = ((j + 1 < Np) ? pt_ptr[j + 1] : No) - pt_ptr[j];
            for (int64_t q = k * (k - 1) / 2; q > 0; q--) bent[fill[pair_slot[at++]]++] = j;
        }
    }
    lap("pair entries");
    // camera-major view of the local observations
    std::vector<int> cam_ptr((size_t)Nc + 1, 0), cpt((size_t)std::max(No, 1));
    std::vector<real> cuv((size_t)std::max(No, 1) * 2);
    for (int p = 0; p < No; p++) cam_ptr[lcam[p] + 1]++;
    for (int c = 0; c < Nc; c++) cam_ptr[c + 1] += cam_ptr[c];
    {
        std::vector<int> fill(cam_ptr.begin(), cam_ptr.end() - 1);
        for (int p = 0; p < No; p++) {
            const int k = fill[lcam[p]]++;
            cpt[k] = lpt[p]; cuv[2 * (size_t)k] = luv[2 * (size_t)p]; cuv[2 * (size_t)k + 1] = luv[2 * (size_t)p + 1];
        }
    }
    lap("camera-major view");
    VS_TRY(finish_segments(h, rank_cam_lo, rank_cam_hi));
    ItemLists lists;
    VS_TRY(make_items(h, cam_ptr.data(), bptr.data(), lists));
    std::vector<BlockItem> &items = lists.items, &diag_items = lists.diag_items, &small_items = lists.small_items;
    std::vector<int> &di_ptr = lists.di_ptr, &oi_ptr = lists.oi_ptr;
    // More cameras than fit LDS: when the cameras seen by the landmarks of one landmark-kernel
    // workgroup span a short index range (keyframes in time order), that range is staged instead.
    h->lm_cam_span = 0;
    if (!h->stage_cams && getenv("VS_BA_NO_RANGE_STAGING") == nullptr) {
        std::vector<int> lo((size_t)std::max(h->n_lm4_blocks, 1), 0);
        int span = 1;
        for (int w = 0; w < h->n_lm4_blocks; w++) {
            const int j0 = w * kLmPerBlock, j1 = std::min(Np, j0 + kLmPerBlock);
            int mn = Nc, mx = -1;
            for (int p = pt_ptr[j0]; p < pt_ptr[j1]; p++) { mn = std::min(mn, lcam[p]); mx = std::max(mx, lcam[p]); }
            if (mx < 0) { mn = 0; mx = 0; }
            lo[w] = mn;
            span = std::max(span, mx - mn + 1);
        }
        if ((size_t)span * sizeof(CamData) <= 32 * 1024) {
            h->lm_cam_span = span;
            VS_TRY(h->lm_cam_lo.upload(lo.data(), lo.size(), h->stream));
        }
    }

    VS_TRY(choose_lds_cameras(h));

    lap("work items");
    hipStream_t s = h->stream;
    std::vector<int> one(1, 0);
    // The index arrays go to the device as ONE copy when they are small (the reference's own
    // window sizes): they are packed into a page-locked staging block and the handle's buffers
    // become windows of one device arena.  Twenty separate copies from pageable memory cost more
    // than building the arrays.  Large edge sets are copied array by array, straight from the vectors.
    struct Part { void *view_buf; int kind; const void *src; size_t bytes, count, off; };
    std::vector<Part> parts;
    size_t total = 0;
    auto add = [&](void *buf, int kind, const void *src, size_t count, size_t elem) {
        parts.push_back(Part{ buf, kind, src, count * elem, count, total });
        total += (count * elem + 255) / 256 * 256;
    };
    add(&h->pt_ptr, 0, pt_ptr.data(), pt_ptr.size(), sizeof(int));
    add(&h->lcam, 0, lcam.data(), lcam.size(), sizeof(int));
    add(&h->lpt, 0, lpt.data(), lpt.size(), sizeof(int));
    add(&h->lobs, 0, lobs.data(), lobs.size(), sizeof(int));
    add(&h->luv, 1, luv.data(), luv.size(), sizeof(real));
    add(&h->cam_ptr, 0, cam_ptr.data(), cam_ptr.size(), sizeof(int));
    add(&h->cpt, 0, cpt.data(), cpt.size(), sizeof(int));
    add(&h->cuv, 1, cuv.data(), cuv.size(), sizeof(real));
    add(&h->cam_used, 2, cam_used.data(), cam_used.size(), 1);
    add(&h->pt_used, 2, pt_used.data(), pt_used.size(), 1);
    add(&h->brow, 0, nb_off ? h->h_brow.data() : one.data(), (size_t)std::max(nb_off, 1), sizeof(int));
    add(&h->bcol, 0, nb_off ? h->h_bcol.data() : one.data(), (size_t)std::max(nb_off, 1), sizeof(int));
    add(&h->bent, 0, bent.data(), bent.size(), sizeof(int));
    add(&h->ditem_ptr, 0, di_ptr.data(), di_ptr.size(), sizeof(int));
    add(&h->oitem_ptr, 0, oi_ptr.data(), oi_ptr.size(), sizeof(int));
    add(&h->small_items, 3, small_items.data(), small_items.size(), sizeof(BlockItem));
    add(&h->items, 3, items.data(), items.size(), sizeof(BlockItem));
    add(&h->diag_items, 3, diag_items.data(), diag_items.size(), sizeof(BlockItem));
    const size_t one_off = total;      // a double 1.0 for the Huber weight rides along
    total += 256;
    const bool packed = total <= ((size_t)2 << 20);
    if (packed) {
        if (h->busy[2]) {      // the staging block of an earlier edge set may still be on its way
            VS_HIP(hipStreamSynchronize(s));
            h->busy[0] = h->busy[1] = h->busy[2] = false;
        }
        if (h->arena.n < total || !h->arena.p) VS_TRY(h->arena.alloc(total));
        if (h->arena_stage.n < total) VS_TRY(h->arena_stage.alloc(total));
        for (const Part &q : parts) memcpy(h->arena_stage.p + q.off, q.src, q.bytes);
        const real one_r = 1.0;
        memcpy(h->arena_stage.p + one_off, &one_r, sizeof(real));
        VS_HIP(hipMemcpyAsync(h->arena.p, h->arena_stage.p, total, hipMemcpyHostToDevice, s));
        VS_HIP(hipMemcpyAsync(h->hub_w.p, h->arena_stage.p + one_off, sizeof(real), hipMemcpyHostToDevice, s));
        h->busy[2] = true;
    }
    for (const Part &q : parts) {
        unsigned char *at = packed ? h->arena.p + q.off : nullptr;
        switch (q.kind) {
        case 0: { auto *b = (DevBuf<int> *)q.view_buf; if (packed) b->view((int *)at, q.count); else { if (b->is_view) b->release(); VS_TRY(b->upload((const int *)q.src, q.count, s)); } break; }
        case 1: { auto *b = (DevBuf<real> *)q.view_buf; if (packed) b->view((real *)at, q.count); else { if (b->is_view) b->release(); VS_TRY(b->upload((const real *)q.src, q.count, s)); } break; }
        case 2: { auto *b = (DevBuf<unsigned char> *)q.view_buf; if (packed) b->view(at, q.count); else { if (b->is_view) b->release(); VS_TRY(b->upload((const unsigned char *)q.src, q.count, s)); } break; }
        default: { auto *b = (DevBuf<BlockItem> *)q.view_buf; if (packed) b->view((BlockItem *)at, q.count); else { if (b->is_view) b->release(); VS_TRY(b->upload((const BlockItem *)q.src, q.count, s)); } break; }
        }
    }
    VS_TRY(alloc_work_buffers(h));
    // Everything a packed edge set sent came out of the handle's own staging; otherwise the copies
    // read vectors that die with this call.
    if (!(packed && h->stage_cams)) {
        const real one_r = 1.0;
        VS_HIP(hipMemcpyAsync(h->hub_w.p, &one_r, sizeof(real), hipMemcpyHostToDevice, s));
        VS_HIP(hipStreamSynchronize(s));
        h->busy[0] = h->busy[1] = h->busy[2] = false;
    }
    lap("allocations + uploads");
    h->have_obs = true;
    h->linearized = false;
    h->have_scale = false; h->prelin = false;
    return VS_OK;
}

// The same structures as build_structure, built on the device (ba_build.h): the edge list is
// uploaded once, sorted by (landmark, camera) and by camera, the camera pairs are enumerated,
// sorted and run-length encoded there; the host reads back the scalars, the per-camera and
// per-pair counts and the pair list, cuts the work items and uploads them.
int build_structure_device(vs_ba *h, const int32_t *cam, const int32_t *pt_user, const double *uv)
{
    using namespace build;
    const int Nc = h->Nc, Np = h->Np;
    VS_CHECK(h->No_global < ((int64_t)1 << 31), VS_ERR_UNSUPPORTED, "more than 2^31 observations");
    const int Ng = (int)h->No_global;
    hipStream_t s = h->stream;
    h->have_obs = false;
    h->linearized = false;
    const bool timing = getenv("VS_BA_BUILD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "build_structure_device: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // points held in the internal order of an earlier edge set go back to the caller's order first
    if (h->renum && h->have_pts) {
        VS_TRY(rows_to_caller(h, 3, h->pts.p, 3, 0, h->pts_user.p));
        VS_HIP(hipMemcpyAsync(h->pts.p, h->pts_user.p, 3 * (size_t)Np * sizeof(real), hipMemcpyDeviceToDevice, s));
    }
    h->renum = false;
    auto grid = [](int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + 255) / 256)); };
    DevBuf<int> d_cam, d_pt, first_cam, first_sorted, iota, old2new, ovals, gord, gptr, npairs, poff;
    DevBuf<real> d_uv;
    DevBuf<unsigned long long> okeys_in, okeys;
    DevBuf<unsigned char> tmp;
    DevBuf<Scalars> sc;
    PinnedBuf<Scalars> hsc;
    VS_TRY(d_cam.upload(cam, Ng, s)); VS_TRY(d_pt.upload(pt_user, Ng, s)); VS_TRY(d_uv.upload(uv, 2 * (size_t)Ng, s));
    VS_TRY(sc.alloc(1)); VS_TRY(hsc.alloc(1));
    VS_TRY(first_cam.alloc(Np)); VS_TRY(first_sorted.alloc(Np)); VS_TRY(old2new.alloc(Np));
    VS_TRY(iota.alloc((size_t)std::max(Np, Ng)));
    VS_TRY(h->cam_used.alloc(Nc)); VS_TRY(h->pt_used.alloc(Np));
    VS_HIP(hipMemsetAsync(h->cam_used.p, 0, (size_t)Nc, s));
    VS_HIP(hipMemsetAsync(h->pt_used.p, 0, (size_t)Np, s));
    lap("upload");
    hipLaunchKernelGGL(init_kernel, grid(Np), dim3(256), 0, s, sc.p, Np, Nc, first_cam.p);
    hipLaunchKernelGGL(first_cam_kernel, grid(Ng), dim3(256), 0, s, Ng, Nc, Np, d_cam.p, d_pt.p, first_cam.p, h->cam_used.p, sc.p);
    hipLaunchKernelGGL(sorted_check_kernel, grid(Np), dim3(256), 0, s, Np, first_cam.p, sc.p);
    // internal landmark numbering: a stable sort by first camera (the identity when already sorted)
    const bool no_renum = getenv("VS_BA_NO_RENUMBER") != nullptr;
    if (h->new2old.n < (size_t)Np || !h->new2old.p) VS_TRY(h->new2old.alloc(Np));
    hipLaunchKernelGGL(iota_kernel, grid(Np), dim3(256), 0, s, Np, iota.p);
    VS_TRY(sort_pairs(tmp, (const unsigned *)first_cam.p, (unsigned *)first_sorted.p, iota.p, h->new2old.p, Np, bits_for((long long)Nc + 1), s));
    hipLaunchKernelGGL(invert_kernel, grid(Np), dim3(256), 0, s, Np, h->new2old.p, old2new.p, (int)no_renum);
    // observations sorted by (landmark, camera)
    VS_TRY(okeys_in.alloc(Ng)); VS_TRY(okeys.alloc(Ng)); VS_TRY(ovals.alloc(Ng)); VS_TRY(gord.alloc(Ng));
    hipLaunchKernelGGL(obs_key_kernel, grid(Ng), dim3(256), 0, s, Ng, Nc, Np, d_cam.p, d_pt.p, old2new.p, okeys_in.p, ovals.p, h->pt_used.p);
    VS_TRY(sort_pairs(tmp, okeys_in.p, okeys.p, ovals.p, gord.p, Ng, 32 + bits_for(Np), s));
    hipLaunchKernelGGL(dup_check_kernel, grid(Ng), dim3(256), 0, s, Ng, okeys.p, sc.p);
    VS_TRY(gptr.alloc((size_t)Np + 1)); VS_TRY(npairs.alloc(Np)); VS_TRY(poff.alloc(Np));
    hipLaunchKernelGGL(landmark_ptr_kernel, grid((int64_t)Np + 1), dim3(256), 0, s, Ng, Np, okeys.p, gptr.p);
    hipLaunchKernelGGL(shard_bounds_kernel, dim3(1), dim3(64), 0, s, Ng, Np, h->rank, h->world, gptr.p, sc.p);
    hipLaunchKernelGGL(pair_count_kernel, grid(Np), dim3(256), 0, s, Np, gptr.p, npairs.p);
    VS_TRY(exclusive_sum(tmp, npairs.p, poff.p, Np, s));
    hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(64), 0, s, Np, poff.p, npairs.p, &sc.p->n_pairs);
    std::vector<int> rank_cam_lo((size_t)h->world, Nc), rank_cam_hi((size_t)h->world, -1);
    DevBuf<int> d_rank_cam;
    if (h->world > 1) {
        VS_TRY(d_rank_cam.alloc(2 * (size_t)h->world));
        hipLaunchKernelGGL(rank_camera_range_kernel, dim3(h->world), dim3(256), 0, s, Ng, Np, Nc, h->world, gptr.p, okeys.p,
                           d_rank_cam.p, d_rank_cam.p + h->world);
    }
    VS_TRY(launch_check("structure kernels (observations)"));
    VS_HIP(hipMemcpyAsync(hsc.p, sc.p, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    if (h->world > 1) {
        VS_HIP(hipMemcpyAsync(rank_cam_lo.data(), d_rank_cam.p, sizeof(int) * (size_t)h->world, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(rank_cam_hi.data(), d_rank_cam.p + h->world, sizeof(int) * (size_t)h->world, hipMemcpyDeviceToHost, s));
    }
    VS_HIP(hipStreamSynchronize(s));
    lap("observation sort");
    {
        const Scalars &q = *hsc.p;
        VS_CHECK(q.bad_cam == INT_MAX, VS_ERR_INVALID, "observation %d: camera index %d out of range [0,%d)", q.bad_cam, cam[q.bad_cam], Nc);
        VS_CHECK(q.bad_pt == INT_MAX, VS_ERR_INVALID, "observation %d: point index %d out of range [0,%d)", q.bad_pt, pt_user[q.bad_pt], Np);
        if (q.dup != INT_MAX) {
            unsigned long long key = 0;
            VS_HIP(hipMemcpy(&key, okeys.p + q.dup, sizeof(key), hipMemcpyDeviceToHost));
            int jn = (int)(key >> 32), ju = jn;
            if (!no_renum) VS_HIP(hipMemcpy(&ju, h->new2old.p + jn, sizeof(int), hipMemcpyDeviceToHost));
            set_error("point %d is observed twice by camera %d", ju, (int)(unsigned)(key & 0xffffffffull));
            return VS_ERR_INVALID;
        }
        VS_CHECK(q.n_pairs >= 0, VS_ERR_UNSUPPORTED, "more than 2^31 camera-pair entries on one GPU");
    }
    const bool renum = !hsc.p->sorted && !no_renum;
    if (renum) {
        if (h->pts_user.n < 3 * (size_t)Np) VS_TRY(h->pts_user.alloc(3 * (size_t)std::max(Np, 1)));
        if (h->perm_stage.n < 9 * (size_t)Np) VS_TRY(h->perm_stage.alloc(9 * (size_t)std::max(Np, 1)));
        h->renum = true;
        if (h->have_pts) {   // the points were set before the edges: bring them into the internal order
            VS_HIP(hipMemcpyAsync(h->pts_user.p, h->pts.p, 3 * (size_t)Np * sizeof(real), hipMemcpyDeviceToDevice, s));
            VS_TRY(rows_to_internal(h, 3, h->pts_user.p, h->pts.p));
        }
    }
    const int pt_lo = hsc.p->pt_lo, pt_hi = hsc.p->pt_hi, lo_pos = hsc.p->lo_pos;
    const int No = hsc.p->hi_pos - hsc.p->lo_pos, P = hsc.p->n_pairs;
    h->pt_lo = pt_lo; h->pt_hi = pt_hi; h->No = No;
    const size_t no1 = (size_t)std::max(No, 1);
    VS_TRY(h->pt_ptr.alloc((size_t)Np + 1)); VS_TRY(h->lcam.alloc(no1)); VS_TRY(h->lpt.alloc(no1)); VS_TRY(h->lobs.alloc(no1));
    VS_TRY(h->luv.alloc(2 * no1)); VS_TRY(h->cam_ptr.alloc((size_t)Nc + 1)); VS_TRY(h->cpt.alloc(no1)); VS_TRY(h->cuv.alloc(2 * no1));
    hipLaunchKernelGGL(local_obs_kernel, grid(No), dim3(256), 0, s, No, lo_pos, okeys.p, gord.p, d_uv.p, h->lcam.p, h->lpt.p,
                       h->lobs.p, h->luv.p, (int)(h->huber_a > 0.0), sc.p);
    hipLaunchKernelGGL(local_ptr_kernel, grid((int64_t)Np + 1), dim3(256), 0, s, Np, No, lo_pos, gptr.p, h->pt_ptr.p);
    // camera-major view: a stable sort of the landmark-major positions by camera
    DevBuf<int> cpos;
    DevBuf<unsigned> ccam;
    VS_TRY(cpos.alloc(no1)); VS_TRY(ccam.alloc(no1));
    if (No > 0) {
        hipLaunchKernelGGL(iota_kernel, grid(No), dim3(256), 0, s, No, iota.p);
        VS_TRY(sort_pairs(tmp, (const unsigned *)h->lcam.p, ccam.p, iota.p, cpos.p, No, bits_for(Nc), s));
        hipLaunchKernelGGL(camera_major_kernel, grid(No), dim3(256), 0, s, No, cpos.p, h->lpt.p, h->luv.p, h->cpt.p, h->cuv.p);
    }
    hipLaunchKernelGGL(camera_ptr_kernel, grid((int64_t)Nc + 1), dim3(256), 0, s, No, Nc, ccam.p, h->cam_ptr.p);
    // camera pairs of every landmark (the global pattern), sorted by (row, column); stable, so the
    // landmarks of a pair stay ascending
    DevBuf<unsigned long long> pkeys_in, pkeys;
    DevBuf<int> pvals_in, pvals, head, loc, bidx, lsum, bptr_dev;
    if (P > 0) {
        VS_TRY(pkeys_in.alloc(P)); VS_TRY(pkeys.alloc(P)); VS_TRY(pvals_in.alloc(P)); VS_TRY(pvals.alloc(P));
        VS_TRY(head.alloc(P)); VS_TRY(loc.alloc(P)); VS_TRY(bidx.alloc(P)); VS_TRY(lsum.alloc(P));
        hipLaunchKernelGGL(pair_gen_kernel, grid(Np), dim3(256), 0, s, Np, gptr.p, poff.p, okeys.p, pkeys_in.p, pvals_in.p);
        VS_TRY(sort_pairs(tmp, pkeys_in.p, pkeys.p, pvals_in.p, pvals.p, P, 32 + bits_for(Nc), s));
        hipLaunchKernelGGL(pair_flags_kernel, grid(P), dim3(256), 0, s, P, pkeys.p, pvals.p, pt_lo, pt_hi, head.p, loc.p);
        VS_TRY(exclusive_sum(tmp, head.p, bidx.p, P, s));
        VS_TRY(exclusive_sum(tmp, loc.p, lsum.p, P, s));
        hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(64), 0, s, P, bidx.p, head.p, &sc.p->nb_off);
        hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(64), 0, s, P, lsum.p, loc.p, &sc.p->n_ent);
    }
    VS_TRY(launch_check("structure kernels (pairs)"));
    VS_HIP(hipMemcpyAsync(hsc.p, sc.p, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    lap("views + pair sort");
    const int nb_off = P > 0 ? hsc.p->nb_off : 0, n_ent = P > 0 ? hsc.p->n_ent : 0;
    h->nb_off = nb_off;
    VS_TRY(h->brow.alloc((size_t)std::max(nb_off, 1))); VS_TRY(h->bcol.alloc((size_t)std::max(nb_off, 1)));
    VS_TRY(bptr_dev.alloc((size_t)nb_off + 1)); VS_TRY(h->bent.alloc((size_t)std::max(n_ent, 1)));
    VS_HIP(hipMemsetAsync(h->brow.p, 0, sizeof(int) * (size_t)std::max(nb_off, 1), s));
    VS_HIP(hipMemsetAsync(h->bcol.p, 0, sizeof(int) * (size_t)std::max(nb_off, 1), s));
    VS_HIP(hipMemsetAsync(bptr_dev.p, 0, sizeof(int) * ((size_t)nb_off + 1), s));
    if (P > 0)
        hipLaunchKernelGGL(pair_scatter_kernel, grid(P), dim3(256), 0, s, P, pkeys.p, pvals.p, head.p, bidx.p, loc.p, lsum.p,
                           h->brow.p, h->bcol.p, bptr_dev.p, h->bent.p);
    VS_TRY(launch_check("pair_scatter_kernel"));
    std::vector<int> cam_ptr((size_t)Nc + 1), bptr((size_t)nb_off + 1, 0);
    h->h_brow.resize((size_t)nb_off); h->h_bcol.resize((size_t)nb_off);
    VS_HIP(hipMemcpyAsync(cam_ptr.data(), h->cam_ptr.p, sizeof(int) * ((size_t)Nc + 1), hipMemcpyDeviceToHost, s));
    VS_HIP(hipMemcpyAsync(bptr.data(), bptr_dev.p, sizeof(int) * ((size_t)nb_off + 1), hipMemcpyDeviceToHost, s));
    if (nb_off > 0) {
        VS_HIP(hipMemcpyAsync(h->h_brow.data(), h->brow.p, sizeof(int) * (size_t)nb_off, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(h->h_bcol.data(), h->bcol.p, sizeof(int) * (size_t)nb_off, hipMemcpyDeviceToHost, s));
    }
    VS_HIP(hipStreamSynchronize(s));
    h->max_pair_gap = 0;
    for (int b = 0; b < nb_off; b++) h->max_pair_gap = std::max(h->max_pair_gap, h->h_bcol[b] - h->h_brow[b]);
    h->hub = HuberObs{ -1, -1, -1, (real)h->huber_a };
    if (h->huber_a > 0.0 && hsc.p->hub_pos >= 0) { h->hub.pos = hsc.p->hub_pos; h->hub.cam = hsc.p->hub_cam; h->hub.pt = hsc.p->hub_pt; }
    lap("blocks read back");
    VS_TRY(finish_segments(h, rank_cam_lo, rank_cam_hi));
    ItemLists lists;
    VS_TRY(make_items(h, cam_ptr.data(), bptr.data(), lists));
    h->lm_cam_span = 0;
    if (!h->stage_cams && getenv("VS_BA_NO_RANGE_STAGING") == nullptr) {
        VS_TRY(h->lm_cam_lo.alloc((size_t)std::max(h->n_lm4_blocks, 1)));
        hipLaunchKernelGGL(tile_span_kernel, dim3(std::max(h->n_lm4_blocks, 1)), dim3(64), 0, s, h->n_lm4_blocks, Np, Nc, kLmPerBlock,
                           h->pt_ptr.p, h->lcam.p, h->lm_cam_lo.p, sc.p);
        VS_TRY(launch_check("tile_span_kernel"));
        VS_HIP(hipMemcpyAsync(hsc.p, sc.p, sizeof(Scalars), hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        if ((size_t)hsc.p->span * sizeof(CamData) <= 32 * 1024) h->lm_cam_span = hsc.p->span;
    }
    VS_TRY(choose_lds_cameras(h));
    VS_TRY(h->ditem_ptr.upload(lists.di_ptr.data(), lists.di_ptr.size(), s));
    VS_TRY(h->oitem_ptr.upload(lists.oi_ptr.data(), lists.oi_ptr.size(), s));
    VS_TRY(h->small_items.upload(lists.small_items.data(), lists.small_items.size(), s));
    VS_TRY(h->items.upload(lists.items.data(), lists.items.size(), s));
    VS_TRY(h->diag_items.upload(lists.diag_items.data(), lists.diag_items.size(), s));
    VS_TRY(alloc_work_buffers(h));
    const real one_r = 1.0;
    VS_HIP(hipMemcpyAsync(h->hub_w.p, &one_r, sizeof(real), hipMemcpyHostToDevice, s));
    VS_HIP(hipStreamSynchronize(s));
    h->busy[0] = h->busy[1] = h->busy[2] = false;
    lap("items + buffers");
    h->have_obs = true;
    h->linearized = false;
    h->have_scale = false; h->prelin = false;
    return VS_OK;
}

// Linearise at parameters (t,rpy,pts on device) and form the undamped-camera reduced system in
// h->red.  Point damping is applied inside (it is local to a landmark); camera damping later.
int linearize(vs_ba *h, const real *t, const real *rpy, const real *pts, PointDamping pd, const real *lam_c = nullptr,
              bool want_vdiag = false)
{
    real *vdiag = want_vdiag ? h->Vdiag.p : nullptr;   // diag(V): only the Jacobi scaling pass reads it
    auto mark = [&](int i) { if (h->prof_ev) (void)hipEventRecord(h->prof_ev[i], h->stream); };
    mark(0);
    // Landmark kernel: all cameras in the workgroup's LDS (few cameras), all cameras in a CU's LDS
    // (persistent workgroups), this workgroup's camera range in LDS, or gathered.  The two
    // all-camera forms build the camera records themselves (no separate pass).  Explicit damping
    // in the caller's landmark numbering costs the persistent form more than range staging plus
    // the camera pass (config 4: 39.6 against 36.1 + 2 us), so that case goes to range staging.
    const bool mapped = pd.lam_p && pd.lam_map;
    const bool use_lds = h->lds_cam_grid > 0 && !(mapped && h->lm_cam_span > 0);
    if (!h->stage_cams && !use_lds) VS_TRY(prep_cameras(h, t, rpy));
    mark(1);
    const size_t lds = cam_lds_bytes(h);
    const int n_tiles = std::max(1, h->tile_hi - h->tile_lo);
    if (h->stage_cams)
        hipLaunchKernelGGL(landmark_kernel<1>, dim3(n_tiles), dim3(kLmThreads), lds, h->stream,
                           h->Np, h->Nc, h->K, (const CamData *)nullptr, t, rpy, h->cams.p, pts, h->pt_ptr.p, h->lcam.p, h->luv.p,
                           h->hub, (const int *)nullptr, 0, pd, h->lmrec.p, vdiag, h->hub_w.p, h->cost_partial.p, h->tile_lo);
    else if (use_lds && mapped)
        hipLaunchKernelGGL(landmark_lds_kernel<true>, dim3(h->lds_cam_grid), dim3(kLmBigThreads),
                           (size_t)h->Nc * kCamLdsReals * sizeof(real), h->stream,
                           h->Np, h->Nc, h->tile_lo, h->tile_hi, h->K, t, rpy, h->cams.p, pts, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub,
                           pd, h->lmrec.p, vdiag, h->hub_w.p, h->cost_partial.p);
    else if (use_lds)
        hipLaunchKernelGGL(landmark_lds_kernel<false>, dim3(h->lds_cam_grid), dim3(kLmBigThreads),
                           (size_t)h->Nc * kCamLdsReals * sizeof(real), h->stream,
                           h->Np, h->Nc, h->tile_lo, h->tile_hi, h->K, t, rpy, h->cams.p, pts, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub,
                           pd, h->lmrec.p, vdiag, h->hub_w.p, h->cost_partial.p);
    else if (h->lm_cam_span > 0)
        hipLaunchKernelGGL(landmark_kernel<2>, dim3(n_tiles), dim3(kLmThreads),
                           (size_t)h->lm_cam_span * sizeof(CamData), h->stream,
                           h->Np, h->Nc, h->K, h->cams.p, (const real *)nullptr, (const real *)nullptr,
                           (CamData *)nullptr, pts, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->lm_cam_lo.p,
                           h->lm_cam_span, pd, h->lmrec.p, vdiag, h->hub_w.p, h->cost_partial.p, h->tile_lo);
    else
        hipLaunchKernelGGL(landmark_kernel<0>, dim3(n_tiles), dim3(kLmThreads), 0, h->stream,
                           h->Np, h->Nc, h->K, h->cams.p, (const real *)nullptr, (const real *)nullptr,
                           (CamData *)nullptr, pts, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, (const int *)nullptr, 0, pd,
                           h->lmrec.p, vdiag, h->hub_w.p, h->cost_partial.p, h->tile_lo);
    VS_TRY(launch_check("landmark_kernel"));
    mark(2);
    {
        // with fixed points only the camera (diagonal) items run
        const int n_items = h->fix_points ? h->n_diag_items : h->n_list_items;
        const BlockItem *list = h->fix_points ? h->diag_items.p : h->items.p;
        if (n_items > 0) {
            if (h->item_waves == 4)
                hipLaunchKernelGGL((blocks_kernel<4, kEpt4>), dim3(n_items), dim3(4 * kWave), 0, h->stream, n_items, list,
                                   h->K, h->cams.p, h->lmrec.p, h->cpt.p, h->cuv.p, h->bent.p, h->hub,
                                   h->hub_w.p, (int)h->fix_points, h->diag_partial.p, h->off_partial.p, h->red.p,
                                   h->Nc, h->nb_off, h->cost_partial.p, h->n_lm4_blocks, lam_c);
            else
                hipLaunchKernelGGL((blocks_kernel<1, kEpt1>), dim3(n_items), dim3(kWave), 0, h->stream, n_items, list,
                                   h->K, h->cams.p, h->lmrec.p, h->cpt.p, h->cuv.p, h->bent.p, h->hub,
                                   h->hub_w.p, (int)h->fix_points, h->diag_partial.p, h->off_partial.p, h->red.p,
                                   h->Nc, h->nb_off, h->cost_partial.p, h->n_lm4_blocks, lam_c);
            VS_TRY(launch_check("blocks_kernel"));
        }
        if (!h->fix_points && h->n_small16 > 0) {
            hipLaunchKernelGGL(blocks_small_kernel<16>, dim3(div_up(h->n_small16, 4)), dim3(kWave), 0, h->stream,
                               h->n_small16, h->small_items.p, h->K, h->cams.p, h->lmrec.p, h->bent.p, h->hub,
                               h->hub_w.p, h->red.p, h->Nc, h->nb_off);
            VS_TRY(launch_check("blocks_small_kernel"));
        }
        if (!h->fix_points && h->n_small32 > 0) {
            hipLaunchKernelGGL(blocks_small_kernel<32>, dim3(div_up(h->n_small32, 2)), dim3(kWave), 0, h->stream,
                               h->n_small32, h->small_items.p + h->n_small16, h->K, h->cams.p, h->lmrec.p, h->bent.p,
                               h->hub, h->hub_w.p, h->red.p, h->Nc, h->nb_off);
            VS_TRY(launch_check("blocks_small_kernel"));
        }
    }
    mark(3);
    mark(4);
    const int nb_fin = h->fix_points ? 0 : h->nb_off;
    if (h->fix_points && h->nb_off > 0)
        VS_HIP(hipMemsetAsync(h->S_off(), 0, (size_t)h->nb_off * 36 * sizeof(real), h->stream));
    // The block kernel has written every block that is exactly one item, and the cost.  The
    // finalize pass is needed for blocks of several items (partials) or none (zero), and for the
    // cost when no item ran at all.
    const int n_items_run = h->n_diag_items + (h->fix_points ? 0 : h->n_off_items);
    const bool need_fin = h->n_multi_diag > 0 || (!h->fix_points && h->n_multi_off > 0) || n_items_run == 0;
    if (need_fin) {
        const int64_t n_fin = (int64_t)h->Nc * kDiagVals + (int64_t)nb_fin * kOffVals;
        hipLaunchKernelGGL(finalize_kernel, dim3(div_up(n_fin, 256) + 1), dim3(256), 0, h->stream, h->Nc, h->nb_off, nb_fin,
                           h->ditem_ptr.p, h->diag_partial.p, h->oitem_ptr.p, h->off_partial.p,
                           n_items_run == 0 ? h->n_lm4_blocks : -1, h->cost_partial.p, h->red.p, lam_c);
        VS_TRY(launch_check("finalize_kernel"));
    }
    mark(5);
    h->linearized = true;
    return VS_OK;
}

int ensure_state(vs_ba *h)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(h->have_cams && h->have_pts && h->have_obs, VS_ERR_STATE,
             "cameras, points and observations must be set first");
    return VS_OK;
}

// Sum of `n` doubles at device pointer `buf` over the landmark shards (in place): the native
// communicator enqueues an RCCL all-reduce on the handle's stream, the callback form synchronises
// the stream and calls out.
int cross_gpu_sum(vs_ba *h, real *buf, int64_t n)
{
    if (h->comm) return comm_allreduce(h->comm, buf, n, false, h->stream);
    if (!h->allreduce) return VS_OK;
    VS_HIP(hipStreamSynchronize(h->stream));
    const int rc = h->allreduce(h->allreduce_user, buf, n);
    VS_CHECK(rc == 0, VS_ERR_HIP, "all-reduce callback failed with code %d", rc);
    return VS_OK;
}

bool is_sharded(const vs_ba *h) { return h->world > 1 && (h->comm || h->allreduce); }

// The shards' partial reduce buffers -> their sum, on every rank.  With a communicator or a gather
// callback each rank sends only the segment its landmarks touch (one all-gather; the pieces are
// added in rank order, so all ranks end with bitwise identical buffers); otherwise the whole buffer
// is all-reduced.  VS_BA_EXCHANGE=allreduce forces the latter (measurement).
int exchange_reduced(vs_ba *h)
{
    if (h->world <= 1) return VS_OK;
    bool gather = (h->comm != nullptr || h->allgather != nullptr) && h->seg_stride > 0;
    if (const char *e = getenv("VS_BA_EXCHANGE")) gather = gather && strcmp(e, "allreduce") != 0;
    if (!gather) return cross_gpu_sum(h, h->red.p, h->red_len);
    hipStream_t s = h->stream;
    const int64_t stride = h->seg_stride;
    if (h->seg_send.n < (size_t)stride) VS_TRY(h->seg_send.alloc((size_t)stride));
    if (h->seg_recv.n < (size_t)(stride * h->world)) VS_TRY(h->seg_recv.alloc((size_t)(stride * h->world)));
    const ShardSeg own = h->segs[(size_t)h->rank];
    hipLaunchKernelGGL(pack_segment_kernel, dim3(div_up(seg_len(own), 256)), dim3(256), 0, s, own, h->Nc, h->nb_off,
                       h->red.p, h->seg_send.p);
    VS_TRY(launch_check("pack_segment_kernel"));
    if (h->comm) {
        VS_TRY(comm_allgather(h->comm, h->seg_send.p, h->seg_recv.p, stride, s));
    } else {
        VS_HIP(hipStreamSynchronize(s));
        const int rc = h->allgather(h->allgather_user, h->seg_send.p, h->seg_recv.p, stride);
        VS_CHECK(rc == 0, VS_ERR_HIP, "all-gather callback failed with code %d", rc);
    }
    hipLaunchKernelGGL(sum_segments_kernel, dim3(div_up(h->red_len, 256)), dim3(256), 0, s, h->world, h->d_segs.p, stride,
                       h->Nc, h->nb_off, h->seg_recv.p, h->red.p);
    return launch_check("sum_segments_kernel");
}

// A shard of a multi-GPU problem cannot take an LM step on its own: its reduced system is a
// partial sum.
int ensure_sum_path(const vs_ba *h)
{
    VS_CHECK(h->world == 1 || h->comm || h->allreduce, VS_ERR_STATE,
             "shard %d of %d has neither a communicator (vs_ba_set_comm) nor an all-reduce callback (vs_ba_set_allreduce)",
             h->rank, h->world);
    return VS_OK;
}

// Maximum over the shards of the double at device pointer `val` (in place).  The callback form
// only sums: every rank writes its value into its own slot of a zeroed array, the array is summed
// and the maximum of the slots is taken.
int cross_gpu_max(vs_ba *h, real *val)
{
    if (!is_sharded(h)) return VS_OK;
    if (h->comm) return comm_allreduce(h->comm, val, 1, true, h->stream);
    if (h->rank_slots.n < (size_t)h->world) VS_TRY(h->rank_slots.alloc((size_t)h->world));
    VS_HIP(hipMemsetAsync(h->rank_slots.p, 0, (size_t)h->world * sizeof(real), h->stream));
    VS_HIP(hipMemcpyAsync(h->rank_slots.p + h->rank, val, sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    VS_TRY(cross_gpu_sum(h, h->rank_slots.p, h->world));
    hipLaunchKernelGGL(max_of_kernel, dim3(1), dim3(64), 0, h->stream, h->world, h->rank_slots.p, val);
    return launch_check("max_of_kernel");
}

struct StepScalars {
    real mcc, xnorm2, snorm2, cost_new;
    // small windows with speculation: the device's own decision and the radius it gave the
    // linearisation that is already enqueued at the candidate
    bool speculated = false, accepted_dev = false;
    real radius_next = 0;
};
// what the LM loop tells a step so that the step's last kernel can take the decision itself
struct StepContext { bool speculate = false; double cost_cur = 0; const vs_lm_options *opt = nullptr; };

// Waits until *word == want: the last kernel of the stream writes it into coherent page-locked
// memory after its results.  A stream synchronisation costs more than the kernels of a small
// window's LM step; should the value not arrive within two seconds the stream is synchronised,
// which also surfaces a fault.
template <typename T>
int wait_for_word(vs_ba *h, const T *word, T want)
{
    const auto t_start = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) {
        __builtin_ia32_pause();
        if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) {
            VS_HIP(hipStreamSynchronize(h->stream));
            VS_CHECK(__atomic_load_n(word, __ATOMIC_ACQUIRE) == want, VS_ERR_HIP, "a kernel's result did not reach host memory");
            break;
        }
    }
    h->busy[0] = h->busy[1] = h->busy[2] = false;    // everything enqueued before that kernel has finished
    return VS_OK;
}

// Undamped pass at the handle's parameters with the Schur terms left out (only the cost, the
// gradient, diag(U) and diag(V) are wanted), and the Jacobi scaling taken from it: the scaling of a
// whole solve comes from the Jacobian at its first iterate.
int unscaled_pass(vs_ba *h)
{
    PointDamping none{ nullptr, nullptr, 1.0, 0.0, 0.0 };
    const bool fp = h->fix_points;
    h->fix_points = true;
    int rc = linearize(h, h->t.p, h->rpy.p, h->pts.p, none, nullptr, true);
    h->fix_points = fp;
    VS_TRY(rc);
    VS_TRY(exchange_reduced(h));
    h->init_has_gmax = false;
    if (!is_sharded(h) && h->Np <= kSmallPrepMaxPoints && h->Nc <= kSmallPrepMaxPoints && getenv("VS_BA_NO_SMALL_SOLVER") == nullptr) {
        // a small window: both scalings and the gradient test of this first pass in one launch
        hipLaunchKernelGGL(small_scales_kernel, dim3(1), dim3(1024), 0, h->stream, 6 * h->Nc, h->udiag(), h->scale_c.p, h->Np,
                           h->Vdiag.p, h->scale_p.p, h->gc(), fp ? 0 : h->Np, h->lmrec.p, h->scalars.p + 4);
        VS_TRY(launch_check("small_scales_kernel"));
        h->init_has_gmax = true;
    } else {
    hipLaunchKernelGGL(jacobi_scale_kernel, dim3(div_up(6 * h->Nc, 256)), dim3(256), 0, h->stream, 6 * h->Nc, h->udiag(), h->scale_c.p);
    hipLaunchKernelGGL(jacobi_scale_kernel, dim3(div_up(3 * (int64_t)h->Np, 256)), dim3(256), 0, h->stream, 3 * h->Np, h->Vdiag.p, h->scale_p.p);
    VS_TRY(launch_check("jacobi_scale_kernel"));
    }
    h->have_scale = true;
    h->prelin = false;
    return VS_OK;
}

// One LM step at the handle's current parameters.  Leaves dcam/dpt and the candidate
// parameters (t2,rpy2,pts2) on the device; fills sc (host).  *numeric_fail set when the reduced
// system is not positive definite.
int lm_step_device(vs_ba *h, double radius, double min_d, double max_d, StepScalars *sc,
                   bool *numeric_fail, bool want_candidate_cost, const StepContext *ctx = nullptr)
{
    hipStream_t s = h->stream;
    const int Nc = h->Nc, Np = h->Np;
    if (!h->have_scale) VS_TRY(unscaled_pass(h));
    if (!(h->prelin && h->prelin_radius == radius)) {
        PointDamping pd{ nullptr, h->scale_p.p, (real)radius, (real)min_d, (real)max_d };
        VS_TRY(linearize(h, h->t.p, h->rpy.p, h->pts.p, pd));
        VS_TRY(exchange_reduced(h));
    }
    h->prelin = false;   // the camera damping below changes the reduce buffer in place
    // a window of a few cameras: the gradient test of the point this linearisation belongs to
    // (scalars[4]), the damping and the dense matrix in one launch instead of five
    const bool small_solve = !h->fix_points && h->nb_off > 0 && 6 * Nc <= kSmallSolveN && h->Np <= kSmallPrepMaxPoints &&
                             getenv("VS_BA_NO_SMALL_SOLVER") == nullptr;
    h->step_has_gmax = small_solve && h->world == 1;
    if (small_solve) {
        const int n = 6 * Nc;
        const int ld = (n + 15) / 16 * 16;
        const size_t elems = (size_t)n * ld;
        if (h->dense.n < elems + kNB) VS_TRY(h->dense.alloc(elems + kNB));
        const size_t linv_elems = (size_t)div_up(n, kNB) * kNB * kNB;
        if (h->linv.n < linv_elems) VS_TRY(h->linv.alloc(linv_elems));
        hipLaunchKernelGGL(small_prepare_kernel, dim3(1), dim3(kSmallSolveThreads), 0, s, Nc, h->nb_off, h->brow.p,
                           h->bcol.p, h->S_diag(), h->S_off(), h->rhs(), h->udiag(), h->scale_c.p, (real)radius, (real)min_d,
                           (real)max_d, h->lam_c.p, h->dcam.p, h->info.p, h->dense.p, ld, (int)(elems + kNB), h->gc(), Np,
                           h->lmrec.p, h->step_has_gmax ? h->scalars.p + 4 : (real *)nullptr);
        VS_TRY(launch_check("small_prepare_kernel"));
        VS_TRY(cholesky_solve(h->dense.p, h->dcam.p, n, ld, n, h->info.p, h->linv.p, s));
    } else {
    hipLaunchKernelGGL(camera_damping_kernel, dim3(div_up(6 * Nc, 256)), dim3(256), 0, s, Nc, h->udiag(),
                       h->scale_c.p, (const real *)nullptr, (real)radius, (real)min_d, (real)max_d, h->lam_c.p, h->S_diag(),
                       h->rhs(), h->dcam.p, h->info.p);
    VS_TRY(launch_check("camera_damping_kernel"));
    if (h->fix_points || h->nb_off == 0) {
        hipLaunchKernelGGL(solve6_kernel, dim3(div_up(Nc, 64)), dim3(64), 0, s, Nc, h->S_diag(), h->rhs(), h->dcam.p, h->info.p);
        VS_TRY(launch_check("solve6_kernel"));
    } else {
        const int n = 6 * Nc;
        // band storage when the camera pairs stay close to the diagonal (keyframes in time
        // order), dense otherwise
        // (at least a panel and a half wide: the one-workgroup band factorisation walks 32-column
        // panels and keeps the next panel's diagonal block inside the window; a wider band than the
        // pattern needs only adds stored zeros)
        const int band = std::max(6 * h->max_pair_gap + 5, kNB + 15);
        bool banded = (int64_t)(band + 1) * 2 <= n;
        if (const char *e = getenv("VS_BA_SOLVER")) banded = banded && strcmp(e, "dense") != 0;   // test knob
        // long chains of cameras: block cyclic reduction, parallel over the chain (bcr_solve.h)
        const BcrPlan plan = banded ? bcr_plan(Nc, h->max_pair_gap) : BcrPlan();
        bool use_bcr = plan.g > 0;
        if (const char *e = getenv("VS_BA_SOLVER")) use_bcr = use_bcr && strcmp(e, "band") != 0 && strcmp(e, "dense") != 0;   // test knob
        if (use_bcr) {
            const size_t need = bcr_workspace_reals(plan);
            if (h->bcr_ws.n < need) VS_TRY(h->bcr_ws.alloc(need));
            VS_TRY(bcr_solve(plan, h->bcr_ws.p, Nc, h->nb_off, h->brow.p, h->bcol.p, h->S_diag(), h->S_off(), h->rhs(),
                             h->dcam.p, h->info.p, s));
        } else {
        const int bw = banded ? band : n;
        // Row stride padded to a multiple of 16 elements: with panels starting at multiples of 32,
        // every 16-element row segment of a 16 x 16 tile is then exactly one aligned 128-byte line
        // (unaligned segments leave partially written lines in L2 whose next read goes to HBM).
        const int64_t ld = ((int64_t)(banded ? band : n) + 15) / 16 * 16;
        const size_t elems = banded ? (size_t)n * (ld + 1) + ld : (size_t)n * ld;
        VS_CHECK(elems * sizeof(real) <= (size_t)16 << 30, VS_ERR_UNSUPPORTED,
                 "reduced system of %d cameras with camera-pair gap %d exceeds this build's solver limit (16 GiB)",
                 Nc, h->max_pair_gap);
        if (h->dense.n < elems + kNB) VS_TRY(h->dense.alloc(elems + kNB));   // padded: see cholesky_solve
        VS_HIP(hipMemsetAsync(h->dense.p, 0, (elems + kNB) * sizeof(real), s));
        real *A = h->dense.p + (banded ? ld : 0);          // band rows reach bw <= ld elements to the left
        const int64_t ne = (int64_t)Nc * 36 + (int64_t)h->nb_off * 36;
        hipLaunchKernelGGL(assemble_lower_kernel, dim3(div_up(ne, 256)), dim3(256), 0, s, Nc, h->nb_off,
                           h->brow.p, h->bcol.p, h->S_diag(), h->S_off(), A, ld);
        VS_TRY(launch_check("assemble_lower_kernel"));
        const size_t linv_elems = (size_t)div_up(n, kNB) * kNB * kNB;
        if (h->linv.n < linv_elems) VS_TRY(h->linv.alloc(linv_elems));
        VS_TRY(cholesky_solve(A, h->dcam.p, n, ld, bw, h->info.p, h->linv.p, s));
        }
    }
    }
    // ... and the rest of the step of such a window (back-substitution, candidate, its cost, the
    // step's sums) in one launch instead of four
    if (small_solve && !is_sharded(h) && Np <= kSmallTailMaxPoints) {
        const bool use_mail = getenv("VS_BA_NO_MAILBOX") == nullptr;
        if (use_mail && !h->mail.p) {
            VS_TRY(h->mail.alloc(16));
            h->mail.p[8] = 0;
            h->mail_seq = 0;
        }
        const unsigned long long seq = ++h->mail_seq;
        const int groups = std::min(kSmallTailMaxGroups, std::max(1, div_up(Np, kSmallTailThreads / 2)));
        if (!h->tail_ticket.p) {
            VS_TRY(h->tail_partial.alloc(4 * kSmallTailMaxGroups));
            VS_TRY(h->tail_ticket.alloc(1));
            VS_HIP(hipMemsetAsync(h->tail_ticket.p, 0, sizeof(int), s));
        }
        // Speculation: the step is usually accepted, and the next step then starts with a
        // linearisation at the candidate.  Enqueued here, behind the tail kernel and before the host
        // has seen the step's numbers, it runs while the host is still on its way back; its damping
        // radius is the one the tail kernel derives on the device.  A step that is not accepted
        // simply linearises again (as it does without speculation, the radius having changed).
        const bool speculate = use_mail && ctx && ctx->speculate && want_candidate_cost;
        TailDecision dec{};
        if (speculate) {
            if (!h->radius_next.p) VS_TRY(h->radius_next.alloc(1));
            dec = TailDecision{ (real)ctx->cost_cur, (real)radius, (real)ctx->opt->min_relative_decrease,
                                (real)ctx->opt->max_trust_region_radius, (real)ctx->opt->parameter_tolerance,
                                (real)ctx->opt->function_tolerance, h->radius_next.p };
        }
        hipLaunchKernelGGL(small_tail_kernel, dim3(groups), dim3(kSmallTailThreads), 0, s, Np, Nc, h->K, h->cams.p, h->t.p, h->rpy.p,
                           h->pts.p, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->hub_w.p, h->lmrec.p, h->dcam.p,
                           h->cam_used.p, h->pt_used.p, h->dpt.p, h->t2.p, h->rpy2.p, h->pts2.p, (int)want_candidate_cost,
                           h->info.p, h->tail_partial.p, h->tail_ticket.p, h->scalars.p,
                           use_mail ? h->mail.p : (unsigned long long *)nullptr, seq, dec);
        VS_TRY(launch_check("small_tail_kernel"));
        if (speculate) {
            PointDamping pd{ nullptr, h->scale_p.p, (real)radius, (real)min_d, (real)max_d };
            pd.radius_dev = h->radius_next.p;
            VS_TRY(linearize(h, h->t2.p, h->rpy2.p, h->pts2.p, pd));
        }
        if (use_mail) {
            VS_TRY(wait_for_word(h, h->mail.p + 8, seq));
            const real *m = reinterpret_cast<const real *>(h->mail.p);
            sc->mcc = m[0]; sc->xnorm2 = m[1]; sc->snorm2 = m[2]; sc->cost_new = m[3];
            sc->speculated = speculate; sc->accepted_dev = m[7] != 0.0; sc->radius_next = m[5];
            *numeric_fail = m[6] != 0.0;
            if (h->step_has_gmax) h->host_scalars.p[8] = m[4];
            return VS_OK;
        }
    } else {
    const size_t lds = cam_lds_bytes(h);
    if (h->stage_cams)
        hipLaunchKernelGGL(backsub_kernel<true>, dim3(h->n_lm_blocks), dim3(kLmThreads), lds, s, Np, Nc, h->K,
                           h->cams.p, h->pts.p, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->hub_w.p,
                           (int)h->fix_points, h->lmrec.p, h->dcam.p, h->dpt.p, h->mcc_partial.p);
    else if (h->lds_cam_grid > 0)
        hipLaunchKernelGGL(backsub_lds_kernel, dim3(std::min(h->lds_cam_grid, div_up(h->n_lm_blocks, kLmBigThreads / kLmThreads))),
                           dim3(kLmBigThreads), (size_t)Nc * kCamLdsReals * sizeof(real), s, Np, Nc, h->n_lm_blocks, h->K,
                           h->cams.p, h->pts.p, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->hub_w.p,
                           (int)h->fix_points, h->lmrec.p, h->dcam.p, h->dpt.p, h->mcc_partial.p);
    else
        hipLaunchKernelGGL(backsub_kernel<false>, dim3(h->n_lm_blocks), dim3(kLmThreads), 0, s, Np, Nc, h->K,
                           h->cams.p, h->pts.p, h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->hub_w.p,
                           (int)h->fix_points, h->lmrec.p, h->dcam.p, h->dpt.p, h->mcc_partial.p);
    VS_TRY(launch_check("backsub_kernel"));
    const int64_t npar = 6 * (int64_t)Nc + 3 * (int64_t)Np;
    const int ncb = div_up(npar, 256);
    if (h->cand_partial.n < (size_t)ncb * 2) VS_TRY(h->cand_partial.alloc((size_t)ncb * 2 + 2));
    hipLaunchKernelGGL(candidate_kernel, dim3(ncb), dim3(256), 0, s, Nc, Np, (int)h->fix_points, h->t.p, h->rpy.p,
                       h->pts.p, h->cam_used.p, h->pt_used.p, (int)(h->rank == 0 || !is_sharded(h)),
                       is_sharded(h) ? h->pt_lo : 0, is_sharded(h) ? h->pt_hi : h->Np, h->dcam.p, h->dpt.p,
                       h->t2.p, h->rpy2.p, h->pts2.p, h->cand_partial.p);
    VS_TRY(launch_check("candidate_kernel"));
    if (want_candidate_cost) {
        if (h->stage_cams)   // few cameras: the workgroups build the candidate's camera records themselves
            hipLaunchKernelGGL(cost_kernel<true>, dim3(h->n_lm_blocks), dim3(kLmThreads), lds, s, Np, Nc, h->K,
                               (const CamData *)nullptr, h->t2.p, h->rpy2.p, (CamData *)nullptr, h->pts2.p, h->pt_ptr.p,
                               h->lcam.p, h->luv.p, h->hub, h->ccost_partial.p);
        else {
            VS_TRY(prep_cameras(h, h->t2.p, h->rpy2.p));
            hipLaunchKernelGGL(cost_kernel<false>, dim3(h->n_lm_blocks), dim3(kLmThreads), 0, s, Np, Nc, h->K,
                               h->cams.p, (const real *)nullptr, (const real *)nullptr, (CamData *)nullptr, h->pts2.p,
                               h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->ccost_partial.p);
        }
        VS_TRY(launch_check("cost_kernel"));
    }
    // model cost change, |x|^2, |d|^2, candidate cost and the solver status: one launch, one copy
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(256), 0, s, h->n_lm_blocks, h->mcc_partial.p, ncb,
                       h->cand_partial.p, want_candidate_cost ? h->n_lm_blocks : 0, h->ccost_partial.p, h->info.p,
                       h->scalars.p);
    VS_TRY(launch_check("step_scalars_kernel"));
    }
    real *host8 = h->host_scalars.p;
    VS_TRY(cross_gpu_sum(h, h->scalars.p, 4));
    VS_HIP(hipMemcpyAsync(host8, h->scalars.p, 8 * sizeof(real), hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    sc->mcc = host8[0]; sc->xnorm2 = host8[1]; sc->snorm2 = host8[2]; sc->cost_new = host8[3];
    if (h->step_has_gmax) h->host_scalars.p[8] = host8[4];
    *numeric_fail = host8[6] != 0.0;
    return VS_OK;
}

// Enqueues max |gradient| at the linearised point (over the shards too) and its copy into
// host_scalars[8]; the value is there once the stream has been synchronised.
int gradient_max_enqueue(vs_ba *h)
{
    const int64_t np = h->fix_points ? 0 : h->Np;
    const int nb = (int)std::min<int64_t>(kGradMaxBlocks, div_up(std::max<int64_t>(6 * (int64_t)h->Nc, np), 256));
    hipLaunchKernelGGL(gradient_max_kernel, dim3(nb), dim3(256), 0, h->stream, (int64_t)6 * h->Nc, h->gc(), np,
                       h->lmrec.p, nb == 1 ? h->scalars.p + 4 : h->gmax_partial.p);
    if (nb > 1) hipLaunchKernelGGL(max_of_kernel, dim3(1), dim3(64), 0, h->stream, nb, h->gmax_partial.p, h->scalars.p + 4);
    VS_TRY(launch_check("gradient_max_kernel"));
    VS_TRY(cross_gpu_max(h, h->scalars.p + 4));   // landmark gradients are local to a shard
    VS_HIP(hipMemcpyAsync(h->host_scalars.p + 8, h->scalars.p + 4, sizeof(real), hipMemcpyDeviceToHost, h->stream));
    return VS_OK;
}

}  // namespace

extern "C" {

void vs_lm_default_options(vs_lm_options *o)
{
    if (!o) return;
    o->max_num_iterations = 100;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->max_consecutive_invalid_steps = 5;
}

int vs_ba_create(vs_ba_handle *out, int32_t Nc, int32_t Np, int64_t No, double fx, double fy,
                 double cx, double cy)
{
    VS_CHECK(out != nullptr, VS_ERR_INVALID, "null handle pointer");
    *out = nullptr;
    VS_CHECK(Nc > 0 && Np > 0 && No >= 0, VS_ERR_INVALID, "sizes must be positive (Nc=%d Np=%d No=%lld)", Nc, Np, (long long)No);
    int ndev = 0;
    VS_HIP(hipGetDeviceCount(&ndev));
    VS_CHECK(ndev > 0, VS_ERR_HIP, "no HIP device available");
    vs_ba *h = new (std::nothrow) vs_ba();
    VS_CHECK(h != nullptr, VS_ERR_NOMEM, "out of host memory");
    h->Nc = Nc; h->Np = Np; h->No_global = No;
    h->K = Intr{ fx, fy, cx, cy };
    int rc = VS_OK;
    auto A = [&](int r) { if (rc == VS_OK) rc = r; };
    A(h->t.alloc(3 * (size_t)Nc)); A(h->rpy.alloc(3 * (size_t)Nc)); A(h->pts.alloc(3 * (size_t)Np));
    A(h->t2.alloc(3 * (size_t)Nc)); A(h->rpy2.alloc(3 * (size_t)Nc)); A(h->pts2.alloc(3 * (size_t)Np));
    A(h->cams.alloc((size_t)Nc));
    A(h->lmrec.alloc((size_t)kRec * Np)); A(h->Vdiag.alloc(3 * (size_t)Np));
    A(h->hub_w.alloc(1)); A(h->scale_c.alloc(6 * (size_t)Nc)); A(h->scale_p.alloc(3 * (size_t)Np));
    A(h->lam_c.alloc(6 * (size_t)Nc)); A(h->dcam.alloc(6 * (size_t)Nc)); A(h->dpt.alloc(3 * (size_t)Np));
    A(h->scalars.alloc(8)); A(h->info.alloc(1)); A(h->linv.alloc(kNB * kNB)); A(h->gmax_partial.alloc(kGradMaxBlocks));
    A(h->host_scalars.alloc(16));
    if (rc != VS_OK) { delete h; return rc; }
    *out = h;
    return VS_OK;
}

int vs_ba_destroy(vs_ba_handle h)
{
    if (h) { (void)hipStreamSynchronize(h->stream); delete h; }
    return VS_OK;
}

int vs_ba_set_stream(vs_ba_handle h, void *stream)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    h->stream = (hipStream_t)stream;
    return VS_OK;
}

// Small host arrays are copied into the handle's page-locked staging and leave from there, so the
// caller's array is free on return without waiting for the stream.
static bool stageable(const vs_ba *h, const void *src)
{
    if ((6 * (size_t)h->Nc + 3 * (size_t)h->Np) * sizeof(real) > ((size_t)512 << 10)) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess) return attr.type == hipMemoryTypeUnregistered;
    (void)hipGetLastError();      // plain host memory is unknown to the runtime
    return true;
}
// memory a plain memcpy may write: unknown to the runtime, or page-locked host memory
static bool host_pointer(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) return attr.type == hipMemoryTypeUnregistered || attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return true;
}
static int claim_stage(vs_ba *h, int region)
{
    if (!h->param_stage.p) VS_TRY(h->param_stage.alloc(6 * (size_t)h->Nc + 3 * (size_t)h->Np));
    if (h->busy[region]) {
        VS_HIP(hipStreamSynchronize(h->stream));
        h->busy[0] = h->busy[1] = h->busy[2] = false;
    }
    return VS_OK;
}

int vs_ba_set_cameras(vs_ba_handle h, const double *t, const double *rpy)
{
    VS_CHECK(h && t && rpy, VS_ERR_INVALID, "null argument");
    const size_t n = 3 * (size_t)h->Nc;
    if (stageable(h, t) && stageable(h, rpy)) {
        VS_TRY(claim_stage(h, 0));
        memcpy(h->param_stage.p, t, n * sizeof(real));
        memcpy(h->param_stage.p + n, rpy, n * sizeof(real));
        VS_HIP(hipMemcpyAsync(h->t.p, h->param_stage.p, n * sizeof(real), hipMemcpyHostToDevice, h->stream));
        VS_HIP(hipMemcpyAsync(h->rpy.p, h->param_stage.p + n, n * sizeof(real), hipMemcpyHostToDevice, h->stream));
        h->busy[0] = true;
    } else {
        VS_HIP(hipMemcpyAsync(h->t.p, t, n * sizeof(real), hipMemcpyDefault, h->stream));
        VS_HIP(hipMemcpyAsync(h->rpy.p, rpy, n * sizeof(real), hipMemcpyDefault, h->stream));
        VS_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_cams = true; h->linearized = false; h->have_scale = false; h->prelin = false;
    h->cams_host_valid = false;
    return VS_OK;
}

int vs_ba_set_points(vs_ba_handle h, const double *xyz)
{
    VS_CHECK(h && xyz, VS_ERR_INVALID, "null argument");
    const size_t n = 3 * (size_t)h->Np;
    const bool staged = stageable(h, xyz);
    const real *src = xyz;
    if (staged) {
        VS_TRY(claim_stage(h, 1));
        memcpy(h->param_stage.p + 6 * (size_t)h->Nc, xyz, n * sizeof(real));
        src = h->param_stage.p + 6 * (size_t)h->Nc;
    }
    if (h->renum) {
        VS_HIP(hipMemcpyAsync(h->pts_user.p, src, n * sizeof(real), hipMemcpyDefault, h->stream));
        VS_TRY(rows_to_internal(h, 3, h->pts_user.p, h->pts.p));
    } else {
        VS_HIP(hipMemcpyAsync(h->pts.p, src, n * sizeof(real), hipMemcpyDefault, h->stream));
    }
    if (staged) h->busy[1] = true;
    else VS_HIP(hipStreamSynchronize(h->stream));
    h->have_pts = true; h->linearized = false; h->have_scale = false; h->prelin = false;
    return VS_OK;
}

int vs_ba_set_observations_sharded(vs_ba_handle h, const int32_t *cam, const int32_t *pt,
                                   const double *uv, int32_t rank, int32_t world)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(h->No_global == 0 || (cam && pt && uv), VS_ERR_INVALID, "null observation arrays");
    VS_CHECK(world >= 1 && rank >= 0 && rank < world, VS_ERR_INVALID, "bad shard %d of %d", rank, world);
    VS_CHECK(h->comm == nullptr || (comm_rank(h->comm) == rank && comm_world(h->comm) == world), VS_ERR_INVALID,
             "shard %d of %d does not match the attached communicator (rank %d of %d)", rank, world,
             comm_rank(h->comm), comm_world(h->comm));
    h->rank = rank; h->world = world;
    // large edge sets are indexed on the device, small ones (a few launches' worth of work) on the
    // host; VS_BA_BUILD=host|device forces one path (tests compare them)
    bool on_device = h->No_global >= 32768;
    if (const char *e = getenv("VS_BA_BUILD")) on_device = strcmp(e, "device") == 0 && h->No_global > 0;
    return on_device ? build_structure_device(h, cam, pt, uv) : build_structure(h, cam, pt, uv);
}

int vs_ba_set_observations(vs_ba_handle h, const int32_t *cam, const int32_t *pt, const double *uv)
{
    return vs_ba_set_observations_sharded(h, cam, pt, uv, 0, 1);
}

int vs_ba_set_fixed(vs_ba_handle h, int32_t fix_points)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    h->fix_points = fix_points != 0;
    h->linearized = false; h->have_scale = false; h->prelin = false;
    return VS_OK;
}

int vs_ba_set_huber_first(vs_ba_handle h, double a)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(!h->have_obs, VS_ERR_STATE, "set the loss before the observations");
    h->huber_a = a > 0.0 ? a : 0.0;
    return VS_OK;
}

int vs_ba_evaluate(vs_ba_handle h, double *cost, double *res, double *Jc, double *Jp)
{
    VS_TRY(ensure_state(h));
    hipStream_t s = h->stream;
    VS_TRY(prep_cameras(h, h->t.p, h->rpy.p));
    const int64_t Ng = h->No_global;
    DevBuf<real> &dres = h->eval_res, &dJc = h->eval_Jc, &dJp = h->eval_Jp;
    if (res && dres.n < 2 * (size_t)Ng) VS_TRY(dres.alloc(2 * (size_t)Ng));
    if (Jc && dJc.n < 12 * (size_t)Ng) VS_TRY(dJc.alloc(12 * (size_t)Ng));
    if (Jp && dJp.n < 6 * (size_t)Ng) VS_TRY(dJp.alloc(6 * (size_t)Ng));
    if (res) { VS_HIP(hipMemsetAsync(dres.p, 0, 2 * (size_t)std::max<int64_t>(Ng, 1) * sizeof(real), s)); }
    if (Jc) { VS_HIP(hipMemsetAsync(dJc.p, 0, 12 * (size_t)std::max<int64_t>(Ng, 1) * sizeof(real), s)); }
    if (Jp) { VS_HIP(hipMemsetAsync(dJp.p, 0, 6 * (size_t)std::max<int64_t>(Ng, 1) * sizeof(real), s)); }
    if ((res || Jc || Jp) && h->No > 0) {
        hipLaunchKernelGGL(obs_jacobian_kernel, dim3(div_up(h->No, 256)), dim3(256), 0, s, h->No, h->K, h->cams.p,
                           h->pts.p, h->lcam.p, h->lpt.p, h->luv.p, h->lobs.p, h->hub,
                           res ? dres.p : nullptr, Jc ? dJc.p : nullptr, Jp ? dJp.p : nullptr);
        VS_TRY(launch_check("obs_jacobian_kernel"));
    }
    if (cost) {
        const size_t lds = cam_lds_bytes(h);
        if (h->stage_cams)
            hipLaunchKernelGGL(cost_kernel<true>, dim3(h->n_lm_blocks), dim3(kLmThreads), lds, s, h->Np, h->Nc, h->K,
                               h->cams.p, (const real *)nullptr, (const real *)nullptr, (CamData *)nullptr, h->pts.p,
                               h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->ccost_partial.p);
        else
            hipLaunchKernelGGL(cost_kernel<false>, dim3(h->n_lm_blocks), dim3(kLmThreads), 0, s, h->Np, h->Nc, h->K,
                               h->cams.p, (const real *)nullptr, (const real *)nullptr, (CamData *)nullptr, h->pts.p,
                               h->pt_ptr.p, h->lcam.p, h->luv.p, h->hub, h->ccost_partial.p);
        VS_TRY(launch_check("cost_kernel"));
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, h->n_lm_blocks, h->ccost_partial.p, h->scalars.p + 5);
        VS_HIP(hipMemcpyAsync(cost, h->scalars.p + 5, sizeof(real), hipMemcpyDefault, s));
    }
    if (res) VS_HIP(hipMemcpyAsync(res, dres.p, 2 * (size_t)Ng * sizeof(real), hipMemcpyDefault, s));
    if (Jc) VS_HIP(hipMemcpyAsync(Jc, dJc.p, 12 * (size_t)Ng * sizeof(real), hipMemcpyDefault, s));
    if (Jp) VS_HIP(hipMemcpyAsync(Jp, dJp.p, 6 * (size_t)Ng * sizeof(real), hipMemcpyDefault, s));
    VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

// Device pointers are used in place; host pointers are uploaded first.
static int resident_or_upload(const double *src, size_t n, DevBuf<real> &stage, hipStream_t s,
                              const real **out, const void **known_device)
{
    *out = nullptr;
    if (!src) return VS_OK;
    if (src == *known_device) { *out = src; return VS_OK; }   // classified by an earlier call
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeDevice) {
        *out = src;
        *known_device = src;
        return VS_OK;
    }
    (void)hipGetLastError();   // an unregistered host pointer reports an error: not a failure
    VS_TRY(stage.upload(src, n, s));
    *out = stage.p;
    return VS_OK;
}

int vs_ba_linearize(vs_ba_handle h, const double *damp_c, const double *damp_p)
{
    VS_TRY(ensure_state(h));
    hipStream_t s = h->stream;
    PointDamping pd{ nullptr, nullptr, 1.0, 0.0, 0.0 };
    const real *lc = nullptr;
    VS_TRY(resident_or_upload(damp_p, 3 * (size_t)h->Np, h->lam_p_user, s, &pd.lam_p, &h->known_dev_dp));
    pd.lam_map = h->renum ? h->new2old.p : nullptr;   // the caller's array is read through the renumbering
    VS_TRY(resident_or_upload(damp_c, 6 * (size_t)h->Nc, h->lam_c_user, s, &lc, &h->known_dev_dc));
    // the camera damping is added as the diagonal blocks are stored (no extra pass); of the shards
    // of a multi-GPU problem only rank 0 adds it, so that the cross-GPU sum carries it once
    return linearize(h, h->t.p, h->rpy.p, h->pts.p, pd, h->rank == 0 ? lc : nullptr);
}

int vs_ba_profile_linearize(vs_ba_handle h, const double *damp_c, const double *damp_p,
                            int32_t iters, float *avg_ms)
{
    VS_TRY(ensure_state(h));
    VS_CHECK(iters > 0 && avg_ms, VS_ERR_INVALID, "bad profiling arguments");
    hipEvent_t ev[6];
    for (auto &e : ev) VS_HIP(hipEventCreate(&e));
    for (int k = 0; k < 6; k++) avg_ms[k] = 0.f;
    int rc = vs_ba_linearize(h, damp_c, damp_p);   // warm-up
    PointDamping pd{ nullptr, nullptr, 1.0, 0.0, 0.0 };
    pd.lam_map = h->renum ? h->new2old.p : nullptr;
    const real *lc = nullptr;
    if (rc == VS_OK) rc = resident_or_upload(damp_p, 3 * (size_t)h->Np, h->lam_p_user, h->stream, &pd.lam_p, &h->known_dev_dp);
    if (rc == VS_OK) rc = resident_or_upload(damp_c, 6 * (size_t)h->Nc, h->lam_c_user, h->stream, &lc, &h->known_dev_dc);
    for (int it = 0; it < iters && rc == VS_OK; it++) {
        h->prof_ev = ev;
        rc = linearize(h, h->t.p, h->rpy.p, h->pts.p, pd, h->rank == 0 ? lc : nullptr);
        h->prof_ev = nullptr;
        if (rc != VS_OK) break;
        if (hipEventSynchronize(ev[5]) != hipSuccess) { set_error("event sync failed"); rc = VS_ERR_HIP; break; }
        for (int k = 0; k < 5; k++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
            avg_ms[k] += ms;
        }
        float tot = 0.f;
        (void)hipEventElapsedTime(&tot, ev[0], ev[5]);
        avg_ms[5] += tot;
    }
    for (int k = 0; k < 6; k++) avg_ms[k] /= (float)iters;
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

int vs_ba_block_count(vs_ba_handle h, int64_t *n_blocks)
{
    VS_CHECK(h && n_blocks, VS_ERR_INVALID, "null argument");
    VS_CHECK(h->have_obs, VS_ERR_STATE, "observations not set");
    *n_blocks = (int64_t)h->Nc + h->nb_off;
    return VS_OK;
}

int vs_ba_block_index(vs_ba_handle h, int32_t *block_row, int32_t *block_col)
{
    VS_CHECK(h && block_row && block_col, VS_ERR_INVALID, "null argument");
    VS_CHECK(h->have_obs, VS_ERR_STATE, "observations not set");
    for (int c = 0; c < h->Nc; c++) { block_row[c] = c; block_col[c] = c; }
    for (int b = 0; b < h->nb_off; b++) { block_row[h->Nc + b] = h->h_brow[b]; block_col[h->Nc + b] = h->h_bcol[b]; }
    return VS_OK;
}

int vs_ba_get_reduced_system(vs_ba_handle h, double *S_blocks, double *rhs, double *cost)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(h->linearized, VS_ERR_STATE, "call vs_ba_linearize first");
    hipStream_t s = h->stream;
    if (S_blocks) VS_HIP(hipMemcpyAsync(S_blocks, h->red.p, ((size_t)h->Nc + h->nb_off) * 36 * sizeof(real), hipMemcpyDefault, s));
    if (rhs) VS_HIP(hipMemcpyAsync(rhs, h->rhs(), 6 * (size_t)h->Nc * sizeof(real), hipMemcpyDefault, s));
    if (cost) VS_HIP(hipMemcpyAsync(cost, h->cost(), sizeof(real), hipMemcpyDefault, s));
    VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

int vs_ba_get_landmark_blocks(vs_ba_handle h, double *Vinv6, double *gp)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(h->linearized, VS_ERR_STATE, "call vs_ba_linearize first");
    hipStream_t s = h->stream;
    const size_t pitch = kRec * sizeof(real);
    if (h->renum) {   // caller's landmark order through the staging buffer: [Vinv 6 Np | g_p 3 Np]
        const size_t np = (size_t)h->Np;
        if (Vinv6) {
            VS_TRY(rows_to_caller(h, 6, h->lmrec.p, kRec, 3, h->perm_stage.p));
            VS_HIP(hipMemcpyAsync(Vinv6, h->perm_stage.p, 6 * np * sizeof(real), hipMemcpyDefault, s));
        }
        if (gp) {
            VS_TRY(rows_to_caller(h, 3, h->lmrec.p, kRec, 9, h->perm_stage.p + 6 * np));
            VS_HIP(hipMemcpyAsync(gp, h->perm_stage.p + 6 * np, 3 * np * sizeof(real), hipMemcpyDefault, s));
        }
    } else {
        if (Vinv6) VS_HIP(hipMemcpy2DAsync(Vinv6, 6 * sizeof(real), h->lmrec.p + 3, pitch, 6 * sizeof(real), (size_t)h->Np, hipMemcpyDefault, s));
        if (gp) VS_HIP(hipMemcpy2DAsync(gp, 3 * sizeof(real), h->lmrec.p + 9, pitch, 3 * sizeof(real), (size_t)h->Np, hipMemcpyDefault, s));
    }
    VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

int vs_ba_reduce_buffer(vs_ba_handle h, void **device_ptr, int64_t *n_doubles)
{
    VS_CHECK(h && device_ptr && n_doubles, VS_ERR_INVALID, "null argument");
    VS_CHECK(h->have_obs, VS_ERR_STATE, "observations not set");
    *device_ptr = h->red.p;
    *n_doubles = h->red_len;
    return VS_OK;
}

int vs_ba_set_allreduce(vs_ba_handle h, vs_allreduce_fn fn, void *user)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    h->allreduce = fn; h->allreduce_user = user;
    return VS_OK;
}

int vs_ba_set_allgather(vs_ba_handle h, vs_allgather_fn fn, void *user)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    h->allgather = fn; h->allgather_user = user;
    return VS_OK;
}

int vs_ba_exchange_reduced(vs_ba_handle h)
{
    VS_TRY(ensure_state(h));
    VS_CHECK(h->linearized, VS_ERR_STATE, "nothing linearised yet");
    VS_TRY(ensure_sum_path(h));
    return exchange_reduced(h);
}

int vs_ba_exchange_size(vs_ba_handle h, int64_t *send_doubles, int64_t *total_doubles)
{
    VS_CHECK(h && h->have_obs, VS_ERR_STATE, "observations not set");
    if (send_doubles) *send_doubles = h->world > 1 ? seg_len(h->segs[(size_t)h->rank]) : 0;
    if (total_doubles) *total_doubles = h->world > 1 ? h->seg_stride * h->world : 0;
    return VS_OK;
}

int vs_ba_set_comm(vs_ba_handle h, vs_comm_handle c)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(c == nullptr || !h->have_obs || (comm_rank(c) == h->rank && comm_world(c) == h->world), VS_ERR_INVALID,
             "communicator is rank %d of %d but the handle holds shard %d of %d", c ? comm_rank(c) : 0,
             c ? comm_world(c) : 0, h->rank, h->world);
    h->comm = c;
    return VS_OK;
}

int vs_ba_lm_step(vs_ba_handle h, double radius, double *dcam, double *dpt, double *mcc)
{
    VS_TRY(ensure_state(h));
    VS_TRY(ensure_sum_path(h));
    VS_CHECK(radius > 0.0, VS_ERR_INVALID, "radius must be positive");
    vs_lm_options o;
    vs_lm_default_options(&o);
    StepScalars sc{};
    bool fail = false;
    VS_TRY(lm_step_device(h, radius, o.min_lm_diagonal, o.max_lm_diagonal, &sc, &fail, false));
    VS_CHECK(!fail, VS_ERR_NUMERIC, "reduced camera system is not positive definite");
    hipStream_t s = h->stream;
    if (dcam) VS_HIP(hipMemcpyAsync(dcam, h->dcam.p, 6 * (size_t)h->Nc * sizeof(real), hipMemcpyDefault, s));
    if (dpt) {
        const real *src = h->dpt.p;
        if (h->renum) { VS_TRY(rows_to_caller(h, 3, h->dpt.p, 3, 0, h->perm_stage.p)); src = h->perm_stage.p; }
        VS_HIP(hipMemcpyAsync(dpt, src, 3 * (size_t)h->Np * sizeof(real), hipMemcpyDefault, s));
    }
    VS_HIP(hipStreamSynchronize(s));
    if (mcc) *mcc = sc.mcc;
    return VS_OK;
}

int vs_ba_solve(vs_ba_handle h, const vs_lm_options *opts, vs_lm_summary *summary)
{
    VS_TRY(ensure_state(h));
    VS_TRY(ensure_sum_path(h));
    vs_lm_options o;
    if (opts) o = *opts; else vs_lm_default_options(&o);
    hipStream_t s = h->stream;
    // Motion-only BA of a small window (the reference's live mode): the whole loop in one launch
    if (h->fix_points && h->world == 1 && h->Nc <= kSmallMaxCams && h->No > 0 && h->No <= kSmallMaxObs &&
        getenv("VS_BA_NO_SMALL_SOLVER") == nullptr) {
        if (!h->small_host.p) {
            VS_TRY(h->small_host.alloc(1));
            h->small_host.p->seq = 0;
            h->small_seq = 0;
        }
        const int seq = ++h->small_seq;
        const bool poll = getenv("VS_BA_NO_MAILBOX") == nullptr;
        if (poll && !h->cams_host.p) VS_TRY(h->cams_host.alloc(6 * (size_t)kSmallMaxCams));
        hipLaunchKernelGGL(motion_only_solve_kernel, dim3(1), dim3(kSmallThreads), 0, s, h->Nc, h->K, h->t.p, h->rpy.p,
                           h->pts.p, h->cam_ptr.p, h->cpt.p, h->cuv.p, h->cam_used.p, h->hub, o, h->scale_c.p, h->small_host.p,
                           poll ? h->cams_host.p : (real *)nullptr, seq);
        VS_TRY(launch_check("motion_only_solve_kernel"));
        if (poll) {
            VS_TRY(wait_for_word(h, &h->small_host.p->seq, seq));
            h->cams_host_valid = true;
        } else {
            VS_HIP(hipStreamSynchronize(s));
            h->busy[0] = h->busy[1] = h->busy[2] = false;
        }
        h->linearized = false; h->prelin = false; h->have_scale = true;
        if (summary) {
            const SmallOut &q = *h->small_host.p;
            summary->iterations = q.iterations; summary->successful_steps = q.successful_steps;
            summary->termination = q.termination; summary->initial_cost = q.initial_cost;
            summary->final_cost = q.final_cost; summary->final_radius = q.final_radius;
        }
        return VS_OK;
    }
    h->cams_host_valid = false;
    double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
    int it = 0, term = 1, nsucc = 0, invalid = 0;
    double cost = 0.0, cost0 = 0.0;
    // initial cost and gradient (unscaled, undamped); the same pass gives the Jacobi scaling
    VS_TRY(unscaled_pass(h));
    VS_HIP(hipMemcpyAsync(h->host_scalars.p + 9, h->cost(), sizeof(real), hipMemcpyDeviceToHost, s));
    // the gradient test of the starting point rides on the same synchronisation
    if (h->No_global != 0) {
        if (h->init_has_gmax)
            VS_HIP(hipMemcpyAsync(h->host_scalars.p + 8, h->scalars.p + 4, sizeof(real), hipMemcpyDeviceToHost, s));
        else VS_TRY(gradient_max_enqueue(h));
    }
    VS_HIP(hipStreamSynchronize(s));
    cost0 = cost = h->host_scalars.p[9];
    if (h->No_global == 0) term = 0;
    else {
        const real gmax = h->host_scalars.p[8];
        if (gmax <= o.gradient_tolerance) term = 0;
        else {
            // The gradient test of an accepted point is not waited for: its result arrives with the
            // scalars of the next step, and a step taken from a point that had already converged is
            // dropped (one host synchronisation per iteration instead of two).
            bool gmax_pending = false, gmax_enqueued = false;
            const bool small_window = !h->fix_points && h->nb_off > 0 && 6 * h->Nc <= kSmallSolveN && h->world == 1 &&
                                      h->Np <= kSmallPrepMaxPoints && getenv("VS_BA_NO_SMALL_SOLVER") == nullptr;
            const bool speculation = getenv("VS_BA_NO_SPECULATION") == nullptr;
            while (true) {
                if (it >= o.max_num_iterations || radius < o.min_trust_region_radius) {
                    term = it >= o.max_num_iterations ? 1 : 0;
                    if (gmax_pending) {
                        if (!gmax_enqueued) VS_TRY(gradient_max_enqueue(h));
                        VS_HIP(hipStreamSynchronize(s));
                        if (h->host_scalars.p[8] <= o.gradient_tolerance) term = 0;
                    }
                    break;
                }
                it++;
                StepScalars sc{};
                bool fail = false;
                StepContext ctx{ small_window && speculation, cost, &o };
                VS_TRY(lm_step_device(h, radius, o.min_lm_diagonal, o.max_lm_diagonal, &sc, &fail, true, &ctx));
                if (gmax_pending) {
                    gmax_pending = false;
                    if (h->host_scalars.p[8] <= o.gradient_tolerance) { it--; term = 0; break; }
                }
                if (fail || !(sc.mcc > 0.0)) {
                    if (++invalid >= o.max_consecutive_invalid_steps) { term = 2; break; }
                    radius *= 0.5;
                    continue;
                }
                invalid = 0;
                const double xn = std::sqrt(sc.xnorm2), sn = std::sqrt(sc.snorm2);
                if (sn <= o.parameter_tolerance * (xn + o.parameter_tolerance)) { term = 0; break; }
                const double change = cost - sc.cost_new;
                if (std::fabs(change) <= o.function_tolerance * cost) { term = 0; break; }
                const double rho = change / sc.mcc;
                if (rho > o.min_relative_decrease) {
                    std::swap(h->t.p, h->t2.p); std::swap(h->rpy.p, h->rpy2.p);
                    if (!h->fix_points) std::swap(h->pts.p, h->pts2.p);
                    cost = sc.cost_new; nsucc++;
                    decrease_factor = 2.0;
                    if (sc.speculated && sc.accepted_dev) {
                        // the linearisation at this point is already enqueued, with the radius the
                        // device derived from the same numbers
                        radius = sc.radius_next;
                    } else {
                    double f = 1.0 - std::pow(2.0 * rho - 1.0, 3.0);
                    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
                    radius = std::min(radius / f, o.max_trust_region_radius);
                    // gradient at the accepted point, from the pass the next step starts from (the
                    // gradient does not depend on the damping)
                    PointDamping pd{ nullptr, h->scale_p.p, (real)radius, (real)o.min_lm_diagonal, (real)o.max_lm_diagonal };
                    VS_TRY(linearize(h, h->t.p, h->rpy.p, h->pts.p, pd));
                    VS_TRY(exchange_reduced(h));
                    }
                    h->prelin = true;
                    h->prelin_radius = radius;
                    // (a small window's next step computes the gradient test itself)
                    gmax_enqueued = !small_window;
                    if (gmax_enqueued) VS_TRY(gradient_max_enqueue(h));
                    gmax_pending = true;
                } else {
                    radius /= decrease_factor;
                    decrease_factor *= 2.0;
                }
            }
        }
    }
    h->linearized = false;
    h->prelin = false;
    if (is_sharded(h) && !h->fix_points) {
        // every rank has updated only its own landmarks: zero the others and sum the shards
        hipLaunchKernelGGL(mask_points_kernel, dim3(div_up(3 * (int64_t)h->Np, 256)), dim3(256), 0, s, h->Np, h->pt_lo,
                           h->pt_hi, h->pts.p);
        VS_TRY(launch_check("mask_points_kernel"));
        VS_TRY(cross_gpu_sum(h, h->pts.p, 3 * (int64_t)h->Np));
        VS_HIP(hipStreamSynchronize(s));
    }
    if (summary) {
        summary->iterations = it; summary->successful_steps = nsucc; summary->termination = term;
        summary->initial_cost = cost0; summary->final_cost = cost; summary->final_radius = radius;
    }
    return VS_OK;
}

int vs_ba_get_cameras(vs_ba_handle h, double *t, double *rpy)
{
    VS_CHECK(h != nullptr, VS_ERR_INVALID, "null handle");
    VS_CHECK(h->have_cams, VS_ERR_STATE, "cameras not set");
    if (h->cams_host_valid) {        // the last solve left them in host memory
        if ((!t || host_pointer(t)) && (!rpy || host_pointer(rpy))) {
            if (t) memcpy(t, h->cams_host.p, 3 * (size_t)h->Nc * sizeof(real));
            if (rpy) memcpy(rpy, h->cams_host.p + 3 * (size_t)h->Nc, 3 * (size_t)h->Nc * sizeof(real));
            return VS_OK;
        }
    }
    if (t) VS_HIP(hipMemcpyAsync(t, h->t.p, 3 * (size_t)h->Nc * sizeof(real), hipMemcpyDefault, h->stream));
    if (rpy) VS_HIP(hipMemcpyAsync(rpy, h->rpy.p, 3 * (size_t)h->Nc * sizeof(real), hipMemcpyDefault, h->stream));
    VS_HIP(hipStreamSynchronize(h->stream));
    return VS_OK;
}

int vs_ba_get_points(vs_ba_handle h, double *xyz)
{
    VS_CHECK(h && xyz, VS_ERR_INVALID, "null argument");
    VS_CHECK(h->have_pts, VS_ERR_STATE, "points not set");
    const real *src = h->pts.p;
    if (h->renum) {
        VS_TRY(rows_to_caller(h, 3, h->pts.p, 3, 0, h->pts_user.p));
        src = h->pts_user.p;
    }
    VS_HIP(hipMemcpyAsync(xyz, src, 3 * (size_t)h->Np * sizeof(real), hipMemcpyDefault, h->stream));
    VS_HIP(hipStreamSynchronize(h->stream));
    return VS_OK;
}

}  // extern "C"
