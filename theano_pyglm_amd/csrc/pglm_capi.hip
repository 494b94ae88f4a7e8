// C ABI (include/pyglm_hip.h) over the gfx950 kernels in pglm_kernels.hip.h.
// Host side: context, device buffers, spike-event (CSR) index, launch plans, timing.
#include "pglm_kernels.hip.h"
#include "../../include/pyglm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(PGL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct pgl_context {
    int N = 0, B = 0, R = 0, nlin = 0, device = 0;
    int64_t nT = 0;
    double dt = 0;
    int Dstim = 0, Kimp = 0, Ktot = 0, nT16 = 0;
    int numCU = 256;
    hipStream_t stream = nullptr;        // stream every call of this handle is ordered on
    hipStream_t own_stream = nullptr;    // the handle's own stream (stream == own_stream unless pgl_set_stream)
    hipStream_t aux_stream = nullptr;    // side stream: reduction of the first G half beside pass 2
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    static constexpr int NEV = 256;      // ring of per-launch event sets {start, fused begin, fused end, end}
    hipEvent_t evr[NEV][4] = {};
    hipEvent_t* ev = evr[0];             // event set of the most recent pgl_ll_grad call
    long long ev_launches = 0;           // launches recorded since the last timing reset
    bool have_spikes = false, have_basis = false;
    int64_t nnz = 0;
    DevBuf S, ST, spk, wlo, whi, phi, fstim;
    std::vector<int> h_ptr;              // host copy of the event-list row pointers (N+1)
    std::vector<int2> h_ev;              // host copy of the event lists (window tables are rebuilt
                                         // when the basis support changes)
    int Rk = 0;                          // taps the kernels use: R minus trailing all-zero basis rows
    DevBuf theta, Weff, ll, grad, Wfrag, bias, Gpart, llpart, gbpart, Xbuf;
    // resident feature tiles (k_build_fimg): a few image sets side by side -- a masked MAP line search
    // alternates between post-block widths whose kernels want different column paddings (e.g. k_fused6 with
    // 10 or 12 k-tiles at C2), and rebuilding 0.4 GB of images per call would cost more than the call
    struct ImgSlot {
        DevBuf buf;
        int key = 0, tile0 = 0, ntiles = 0;   // key = ktl << 8 | kth (kth = 0: one-part images); 0 = empty / stale
        unsigned long long stamp = 0;
    };
    static constexpr int NIMG = 8;          // (a wide population keeps one image set per column slice)
    ImgSlot imgs[NIMG];
    int img_cur = 0;
    unsigned long long img_clock = 0;
    double* pin_out = nullptr;           // pinned host buffer for small results (PGL_KMAX doubles)
    DevBuf IimpT, Inet, Istim, tmpA, tmpB, tmpC, wsmall, part, outK, lam, wcol, thetan;
    // separable (rank-1) stimulus: theta rows are [bias, w_t(Bt), w_x(Bx), w_imp]; Dstim = Bt + Bx
    bool sep = false;
    int sepBt = 0, sepBx = 0, sepRt = 0;
    int64_t sepT = 0;
    double sep_dt_stim = 0;
    DevBuf zf, zfT, sbt, Yf, Qb, Qf, spart;
    // frame-rate form of the separable stimulus (dt_stim an integer multiple q of dt, J = ceil(Rt / q) + 2 <= 8 frame
    // values per bin): coefficient table C[q (M + 1)][sepJ][sepBTp], see k_sepf_fwd
    bool sepf = false;
    int sepq = 0, sepM = 0, sepJ = 0, sepBTp = 0;
    int sepNH = 0, sepGs = 0;            // A-fragment table of the fused forward (k_fused7<.., 2>): head tiles, log2 gcd(q, 16)
    bool sepA_ok = false;
    DevBuf sepC, sepA, sepAT, sepD, YfT, Hb, wpart, QvT;
    int opt_sepf = 0;                    // dev option 94: 2 = never take the frame-rate path, 3 = stimulus current through the slab (k_sepf_fwd)
    const int* cur_pidx = nullptr;       // post-neuron list of the evaluation being enqueued (device)
    int gibbs_npost = -1;
    double gibbs_bias = 0;
    // batched column Gibbs (pgl_gibbs_prepare_all / _ll_cols / _update_cols)
    DevBuf GX, gtheta, gargs, gpart, gout, ghs, gfs;
    DevBuf staA;                         // pgl_sta with A_out == NULL: the averages (sta_n, sta_L, sta_D) stay here for
    int sta_n = 0, sta_L = 0, sta_D = 0; // pgl_leading_singular_pairs(A == NULL)
    int gx_xs = 0;                       // row stride of GX (16 * post tiles); 0 = not prepared
    int64_t gx_t_lo = 0, gx_t_hi = 0;    // time range GX was prepared for
    unsigned char* pin_args = nullptr;   // pinned staging of the per-call column arguments / results
    size_t pin_args_cap = 0;
    int opt_f32 = 0, opt_nchunks = 0, opt_dbg = 0, opt_kernel = 0, opt_ptw = 0, opt_gibbs = 0, opt_finw = 0, opt_sb6 = 0, opt_epi64 = 0, opt_timing = 1, opt_slice_cols = 0, opt_hlp = 0, opt_pbmajor = 0;
    int opt_img32 = 0;                   // PGL_OPT_FEATURE_F32 = 2: f32 resident blocks for the narrow-shard kernel (k_fused8<.., 1>)
    long long call_no = 0;               // evaluations enqueued since the last pgl_set_option(PGL_OPT_TIMING)
    int64_t t_lo = 0, t_hi = 0;          // evaluated time range [t_lo, t_hi) (pgl_set_time_range)
    bool timing_valid = false;
    // lock-step optimiser: the pinned flag buffer of pgl_bfgs_step_dev and its device address (looked up once);
    // history size (hk_bound * P doubles per row) up to which a whole iteration is ONE row kernel (PGL_OPT_BFGS_MERGE)
    double* flags_host[4] = {nullptr, nullptr, nullptr, nullptr};
    double* flags_dev[4] = {nullptr, nullptr, nullptr, nullptr};
    int flags_next = 0;
    int opt_bfgs_merge = 65536;
};

static int ensure(DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return PGL_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return PGL_OK;
}
static void release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}
static int find_img(const pgl_context* h, int key, int tile0, int ntiles)
{
    for (int i = 0; i < pgl_context::NIMG; ++i)
        if (h->imgs[i].key == key && h->imgs[i].tile0 == tile0 && h->imgs[i].ntiles == ntiles && h->imgs[i].buf.p)
            return i;
    return -1;
}
static void invalidate_images(pgl_context* h)
{
    for (int i = 0; i < pgl_context::NIMG; ++i) h->imgs[i].key = 0;
}
// can a new image set of `want` bytes be placed (in a free slot, or over the least recently used one)?
static bool img_room(const pgl_context* h, size_t want)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true;
    size_t reclaim = 0;
    int lru = 0;
    for (int i = 1; i < pgl_context::NIMG; ++i)
        if (h->imgs[i].stamp < h->imgs[lru].stamp) lru = i;
    for (int i = 0; i < pgl_context::NIMG; ++i)
        if (h->imgs[i].key == 0) reclaim = std::max(reclaim, h->imgs[i].buf.cap);
    reclaim = std::max(reclaim, h->imgs[lru].buf.cap);
    return want <= reclaim || want - reclaim <= free_b / 10 * 9;
}
#define ENSURE(buf, bytes)                       \
    do {                                         \
        int rc_ = ensure(buf, bytes);            \
        if (rc_ != PGL_OK) return rc_;           \
    } while (0)

struct Plan {
    int npost, nPT, wpb, nPB, KT, KS, rsf, RP, nTiles, nChunks, tilesPerChunk, blocks, threads;
    int tile0;
    int version, PTW, KTW, KSPLIT, cap; // version 2/3: K split over KSPLIT waves per post tile
    int ktl, kth;                       // version 5: k-tiles of the L / H column parts
    int hlp = 0;                        // version 5: 1 = idle waves of a 5- or 6-tile block help (k_fused5<.., HLP = 1>)
    int pb_major = 0;                   // version 5, wide: post-block-major grid of one chunk per CU and post block
    int mt;                             // version 6: 16-bin tiles per step
    int nw6;                            // version 6: waves per workgroup (8, or 4 with two workgroups per CU)
    int sb6 = 0;                        // version 6: 1 = one image buffer per workgroup (k_fused6 DB = 0), 2 = per-wave block
                                        // rings on block-form images (k_fused8)
    int img32 = 0;                      // ... whose blocks are stored as f32 (PGL_OPT_FEATURE_F32 = 2)
    int nw7, wg7;                       // version 7: waves per workgroup (1, 2, 4), workgroups per CU
    size_t lds7x = 0;                   // version 7: extra LDS of the separable-stimulus forms
    size_t lds;
    bool f32;
};

// A launch covers a slice of the feature columns: presynaptic neurons [np0, np0+Ns) and dense
// stimulus columns [ds0, ds0+Ds).  One slice = everything when N <= 128 and N*B + Dstim <= 640
// (the fused path); otherwise the 3-phase path runs forward / backward launches per slice.
struct Slice {
    int np0, Ns, ds0, Ds;
};

static std::vector<Slice> make_slices(const pgl_context* h, bool balanced = false)
{
    std::vector<Slice> out;
    int maxNs = std::min(128, (h->opt_slice_cols > 0 ? h->opt_slice_cols : 640) / h->B);
    // balanced: slices of equal width (N = 160: 80 + 80, not 128 + 32) -- every slice then has a long enough feature row for
    // the two-pass kernel on resident tiles (the wide-population path).  The K-split kernel of the other sliced paths pads
    // its rows to 5 / 10 / 20 / 40 k-tiles and is better off with full 640-column slices and a short last one.
    if (balanced && h->N > maxNs) maxNs = (h->N + (h->N + maxNs - 1) / maxNs - 1) / ((h->N + maxNs - 1) / maxNs);
    int ds_left = h->sep ? 0 : h->Dstim, ds0 = 0;      // a separable stimulus is not a set of feature columns
    for (int np0 = 0; np0 < h->N; np0 += maxNs) {
        Slice sl{np0, std::min(maxNs, h->N - np0), 0, 0};
        if (np0 + sl.Ns >= h->N && ds_left > 0 && sl.Ns * h->B + ds_left <= 640) {
            sl.ds0 = ds0; sl.Ds = ds_left; ds0 += ds_left; ds_left = 0;     // stimulus rides along
        }
        out.push_back(sl);
    }
    while (ds_left > 0) {
        const int d = std::min(640, ds_left);
        out.push_back(Slice{0, 0, ds0, d});
        ds0 += d; ds_left -= d;
    }
    return out;
}

static const int kKTW[] = {1, 2, 3, 5, 7, 10, 20};
static const int kKTH[] = {1, 2, 3, 5, 7, 10, 13, 16};         // k-tiles per half, two-pass kernel (on the fly)
// resident-tile kernel: (L, H) k-tile pairs; pass 1 (forward + L columns of G) gets the smaller share
#ifndef PGL_SPLIT_L
#define PGL_SPLIT_L 18       // L / H k-tiles of the 40-k-tile (K = 640) split; measured: 18/22 (see docs/NOTEBOOK.md §4.1)
#endif
static const int kKTP[][2] = {{1, 1}, {2, 2}, {3, 3}, {5, 5}, {7, 7}, {9, 11}, {12, 14}, {14, 18},
                              {PGL_SPLIT_L, 40 - PGL_SPLIT_L}};
static bool pick_pair(int need, int& ktl, int& kth)
{
    for (const auto& pr : kKTP)
        if (pr[0] + pr[1] >= need) {
            ktl = pr[0];
            kth = pr[1];
            return true;
        }
    return false;
}
static size_t img_pair_bytes(int ktl, int kth) { return (size_t)pgl_img_bytes(ktl) + pgl_img_bytes(kth); }
// one-part images (k_fused6 / 7), padded rows or the block form of k_fused8: slot key and bytes per tile
// (blk: 0 padded rows, 1 blocks of 2 KB, 2 the same blocks stored as f32)
static int img_key6(int kt, int blk) { return (blk ? 0x8000 : 0) | (blk == 2 ? 0x4000 : 0) | kt << 8; }
static size_t img_bytes6(int kt, int blk) { return blk ? (size_t)kt * (blk == 2 ? 1024 : 2048) : (size_t)pgl_img_bytes(kt); }
static constexpr int kRing8 = 8;            // k_fused8<5, kRing8>: blocks per wave
static size_t lds_fused8(int slots) { return (size_t)8 * slots * 2048 + (size_t)(8 * 256 + 256 + 32 + 8 * 48) * 8; }

static int fused6_wg_per_cu(const Plan& pl);

static int make_plan(const pgl_context* h, int n_lo, int n_hi, const Slice& sl, Plan& pl,
                     bool single_slice = true, bool force7 = false, bool wide = false)
{
    const bool hlp_ok = (single_slice || wide) && !force7;      // (the slab-input form of a separable stimulus has no helper variant)
    if (n_lo < 0 || n_hi > h->N || n_lo >= n_hi) return fail(PGL_ERR_ARG, "bad neuron range");
    pl.npost = n_hi - n_lo;
    pl.nPT = (pl.npost + 15) / 16;
    const int ktot_s = sl.Ns * h->B + sl.Ds;
    const int need = std::max(1, (ktot_s + 15) / 16);
    pl.f32 = h->opt_f32 != 0;
    // version 2: f64 features, 8 waves (2 per SIMD), one workgroup per CU
    // version 3: the same kernel with f32 features / basis taps (PGL_OPT_FEATURE_F32)
    // version 1: the 4-wave kernel of the first round (PGL_OPT_KERNEL = 1)
    // version 4: the two-pass kernel (one workgroup = 8 post tiles, no K split; PGL_OPT_KERNEL 0 = auto
    //            for >= 5 post tiles, 3 = force, 2 = force version 2); f64 features, one slice only
    pl.version = pl.f32 ? 3 : 2;
    if (pl.version == 2 && single_slice && need >= 2) {
        if (h->opt_kernel == 3) pl.version = 4;
        // two-pass kernel on resident tiles from 5 post tiles on; from 3 when the feature row is too long for
        // the resident K-split kernel (measured at K = 640: 64 neurons 2.05 ms against 2.33 ms of k_fused2; a 48-neuron
        // list of a lock-step sweep at C3: 2.36 ms on k_fused2)
        // (force7 with a feature row too long for k_fused7 (> 16 k-tiles: its G no longer fits the registers beside the
        // epilogue) -- e.g. a short neuron list of a wide separable-stimulus population: the slab-input form of the two-pass
        // kernel, whatever the number of post tiles)
        else if (h->opt_kernel == 4 || (h->opt_kernel == 0 && (pl.nPT >= 5 || (pl.nPT >= 3 && need > 20))) ||
                 (force7 && need > 16 && (h->opt_kernel == 0 || h->opt_kernel == 7)))
            pl.version = 5;
    }
    // wide: one column slice of a wide population on the resident-tile two-pass kernel (select_plans has checked the
    // row lengths and the memory)
    if (wide) pl.version = 5;
    pl.tile0 = (int)(h->t_lo / 16);
    pl.nTiles = (int)((h->t_hi + 15) / 16) - pl.tile0;
    pl.ktl = pl.kth = 0;
    if (pl.version == 5 && !pick_pair(need, pl.ktl, pl.kth)) pl.version = 4;
    if (pl.version == 5 && !wide && h->opt_kernel == 0 && find_img(h, pl.ktl << 8 | pl.kth, pl.tile0, pl.nTiles) < 0) {
        // resident feature tiles need nTiles * (L + H image bytes) of HBM (3.1 GB at C3); in auto mode
        // fall back to on-the-fly generation (version 4) when the device cannot spare them
        if (!img_room(h, (size_t)pl.nTiles * img_pair_bytes(pl.ktl, pl.kth))) pl.version = 4;
    }
    // the in-kernel-feature two-pass kernel carries at most 16 k-tiles per half in its registers (k_fused3<20, ..> spilled
    // 22 VGPRs): rows of 33-40 k-tiles go to the K-split kernel, which is as fast there (2.20 against 2.16 ms at N = 128)
    if (pl.version == 4 && (need + 1) / 2 > 16) pl.version = 2;
    pl.RP = h->Rk + 32;
    if (pl.version == 3) {
        while (pl.RP % 64 != 8) ++pl.RP;  // f32 table rows one 32-byte span apart (mod 256 B)
    } else {
        // bank spread of the per-basis table rows for ds_read_b128: the row-interleaved items of
        // gen_items want the rows of b = 0..3 four 16-byte slots (64 B) apart
        while (pl.RP % 32 != 8) ++pl.RP;
    }
    pl.cap = PGL_CAP;
    if (pl.version == 5) {
        pl.PTW = 8; pl.KSPLIT = 1; pl.KTW = pl.kth; pl.KT = pl.ktl + pl.kth; pl.wpb = 8;
        pl.nPB = (pl.nPT + 7) / 8;
    } else if (pl.version == 4) {
        const int needh = (need + 1) / 2;
        int kth = 0;
        for (int k : kKTH)
            if (k >= needh) {
                kth = k;
                break;
            }
        if (kth == 0) return fail(PGL_ERR_UNSUPPORTED, "slice exceeds 640 feature columns");
        pl.PTW = 8; pl.KSPLIT = 1; pl.KTW = kth; pl.KT = 2 * kth; pl.wpb = 8;
        pl.nPB = (pl.nPT + 7) / 8;
    } else {
        const int nw = 8;
        const int maxptw = 4;
        pl.PTW = (pl.nPT >= 3) ? 4 : pl.nPT;
        pl.PTW = std::min(pl.PTW, maxptw);
        if (h->opt_ptw == 1 || h->opt_ptw == 2 || h->opt_ptw == 4) pl.PTW = std::min(h->opt_ptw, pl.PTW);
        pl.KSPLIT = nw / pl.PTW;
        const int needw = (need + pl.KSPLIT - 1) / pl.KSPLIT;
        pl.KTW = 0;
        for (int k : kKTW)
            if (k >= needw) {
                pl.KTW = k;
                break;
            }
        if (pl.KTW == 0 || pl.KTW * pl.KSPLIT > 40)
            return fail(PGL_ERR_UNSUPPORTED,
                        "slice of " + std::to_string(ktot_s) + " feature columns exceeds 640");
        pl.KT = pl.KTW * pl.KSPLIT;
        pl.wpb = nw;
        pl.nPB = (pl.nPT + pl.PTW - 1) / pl.PTW;
        // version 6: the K-split scheme on resident feature tiles (k_fused6) when two step buffers of
        // whole-row images fit the LDS: short feature rows (C1, C2, C5).  Post blocks of one or two tiles
        // run as 4-wave workgroups, two per CU (less padding of K, barrier waits overlap).
        pl.mt = 0;
        pl.nw6 = 8;
        if (pl.version == 2 && single_slice && (h->opt_kernel == 0 || h->opt_kernel == 6) && h->opt_ptw == 0) {
            int ptw6 = pl.PTW, nw6 = 8, ktw6 = pl.KTW;
            if (pl.nPT <= 2) {
                nw6 = 4;
                ptw6 = pl.nPT;
                const int needw6 = (need + nw6 / ptw6 - 1) / (nw6 / ptw6);
                ktw6 = 0;
                for (int k : kKTW)
                    if (k >= needw6 && k <= 10) {
                        ktw6 = k;
                        break;
                    }
                if (ktw6 == 0) { nw6 = 8; ptw6 = pl.PTW; ktw6 = pl.KTW; }
            }
            const int kt6 = ktw6 * (nw6 / ptw6);
            int mt6 = 0;
            for (int mt = (nw6 == 4 ? 1 : 2); mt >= 1 && mt6 == 0; --mt) {
                const size_t lds6 = (size_t)2 * mt * pgl_img_bytes(kt6) + (size_t)mt * nw6 * 2048 + 256 + (size_t)nw6 * 384;
                const size_t cap = (nw6 == 4) ? 80 * 1024 : 160 * 1024;      // two 4-wave workgroups per CU
                if (lds6 <= cap && (mt == 1 || pl.nTiles >= 4)) mt6 = mt;
            }
            if (mt6 == 0 && nw6 == 4) {                                      // does not fit twice: 8-wave form
                nw6 = 8; ptw6 = pl.PTW; ktw6 = pl.KTW;
                const int kt8 = ktw6 * (8 / ptw6);
                for (int mt = 2; mt >= 1 && mt6 == 0; --mt) {
                    const size_t lds6 = (size_t)2 * mt * pgl_img_bytes(kt8) + (size_t)mt * 8 * 2048 + 256 + (size_t)8 * 384;
                    if (lds6 <= 160 * 1024 && (mt == 1 || pl.nTiles >= 4)) mt6 = mt;
                }
            }
            pl.sb6 = 0;
            if (mt6 == 0 && pl.nPT <= 2 && h->opt_sb6 != 2) {
                // the row does not fit twice (K = 640: 81 KB per tile): two post tiles -> 8-wave form with ONE image buffer
                // (<10,2>: 4-way K split); one post tile -> 8-way K split with a private block ring per wave (k_fused8<5, 8>)
                nw6 = 8; ptw6 = pl.nPT;
                const int needw8 = (need + 8 / ptw6 - 1) / (8 / ptw6);
                ktw6 = (ptw6 == 1) ? 5 : 10;
                const size_t lds1 = (size_t)pgl_img_bytes(ktw6 * (8 / ptw6)) + (size_t)8 * 2048 + 256 + (size_t)8 * 384;
                if (needw8 <= ktw6 && needw8 > ktw6 / 2 && lds1 <= 160 * 1024) {
                    mt6 = 1;
                    // one post tile: every wave streams its own K slice through a private block ring (k_fused8) -- the
                    // HBM stream never stops for the fragment read-out (0.615 against 0.665 ms for a 16-neuron shard of C3)
                    pl.sb6 = (ptw6 == 1) ? 2 : 1;
                }
            }
            if (mt6 > 0) {
                const int ktall = ktw6 * (nw6 / ptw6);
                bool ok = true;
                pl.img32 = (pl.sb6 == 2 && h->opt_img32) ? 1 : 0;
                const int blk6 = (pl.sb6 == 2) ? 1 + pl.img32 : 0;
                if (h->opt_kernel == 0 && find_img(h, img_key6(ktall, blk6), pl.tile0, pl.nTiles) < 0)
                    ok = img_room(h, (size_t)pl.nTiles * img_bytes6(ktall, blk6));
                if (!ok) pl.sb6 = pl.img32 = 0;
                if (ok) {
                    pl.version = 6;
                    pl.mt = mt6; pl.nw6 = nw6; pl.PTW = ptw6; pl.KTW = ktw6; pl.KSPLIT = nw6 / ptw6;
                    pl.KT = ktall; pl.wpb = nw6;
                    pl.nPB = (pl.nPT + pl.PTW - 1) / pl.PTW;
                }
            }
        }
    }
    // version 7: no K split at all -- one wave per post tile carries the whole feature row (<= 20 k-tiles)
    // through forward, epilogue and backward; small workgroups, several per CU (k_fused7)
    static const int kKT7[] = {1, 2, 3, 5, 7, 10, 12, 13, 16};
    pl.nw7 = 0;
    pl.wg7 = 1;
    // (measured, tools/small_shape_scan.py / config_table.py: 3-4 post tiles 46 TFLOP/s against 39 of the
    // K-split kernel at C5; with 1-2 post tiles only 2-6 waves fit a CU and the 4-wave K-split form wins -- except one
    // post tile of 4-5 k-tiles (N = 16 at B = 5: 0.065 against 0.083 ms, tools/shape_sweep.py).  Rows of 17-20 k-tiles
    // (N = 52..64 at B = 5) stay with the K-split kernel: 160 registers of G beside the epilogue spill (9-108 VGPRs
    // by variant) and k_fused6 is the faster one there anyway, 0.49 against 0.57 ms at N = 64)
    if ((pl.version == 2 || pl.version == 6) && !pl.f32 && single_slice && pl.nPT <= 4 && need <= 16 &&
        ((h->opt_kernel == 0 && (pl.nPT >= 3 || (pl.nPT == 1 && need >= 4 && need <= 5))) || h->opt_kernel == 7 || force7) &&
        h->opt_ptw == 0) {
        int kt7 = 0;
        for (int k : kKT7)
            if (k >= need) {
                kt7 = k;
                break;
            }
        const int nw7 = (pl.nPT >= 3 || force7) ? 4 : pl.nPT;       // force7: the slab-input form exists for 4 waves only
        // (force7 = separable stimulus: + the per-wave accumulators of the fused stimulus backward, k_fused7<.., 3>)
        const size_t lds7 = (size_t)2 * pgl_img_bytes(kt7) + 256 + (size_t)nw7 * 192 * 8 + (force7 ? (size_t)nw7 * 320 * 8 : 0);
        bool ok = kt7 > 0 && lds7 <= 160 * 1024;
        if (ok && h->opt_kernel == 0 && find_img(h, kt7 << 8, pl.tile0, pl.nTiles) < 0)
            ok = img_room(h, (size_t)pl.nTiles * pgl_img_bytes(kt7));
        if (ok) {
            pl.version = 7;
            pl.nw7 = nw7;
            pl.wg7 = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds7, (size_t)(8 / nw7)));
            // one-wave workgroups: five to seven per CU put two waves on some SIMDs and one on the others -- the kernel ends
            // with the doubly loaded SIMDs while the others idle (N = 16: exits spread over 28 .. 67 us); one wave per SIMD
            // and longer chunks: 0.0876 -> 0.0828 ms per evaluation (tools/shape_sweep.py, chunk-count scan of round 6)
            if (nw7 == 1 && pl.wg7 > 4 && pl.wg7 < 8) pl.wg7 = 4;
            pl.PTW = nw7; pl.KSPLIT = 1; pl.KTW = kt7; pl.KT = kt7; pl.wpb = nw7;
            pl.nPB = (pl.nPT + nw7 - 1) / nw7;
            pl.mt = 0;
            pl.lds7x = force7 ? (size_t)nw7 * 320 * 8 : 0;
        }
    }
    pl.KS = pl.KT * 4;
    const int kpad = pl.KT * 16;
    pl.rsf = pl.f32 ? kpad + 4 : kpad + 2;
    int wgPerCU = (pl.version == 7) ? pl.wg7 : 1;
    if (pl.version == 6) {
        // as many workgroups per CU as registers and LDS allow (4-wave form at C2: three)
        pl.lds = (size_t)(pl.sb6 ? 1 : 2) * pl.mt * pgl_img_bytes(pl.KT) + (size_t)pl.mt * pl.nw6 * 2048 + 256 + (size_t)pl.nw6 * 384;
        if (pl.sb6 == 2) pl.lds = lds_fused8(pl.img32 ? 5 : kRing8);
        wgPerCU = (pl.sb6 == 2) ? 1 : fused6_wg_per_cu(pl);
    }
    int target = h->opt_nchunks > 0 ? h->opt_nchunks : std::max(1, wgPerCU * h->numCU / pl.nPB);
    // a wide population whose last post block holds one to six tiles (N = 144, 160, 192, 200, 320 ..): with the post blocks of a
    // chunk side by side, half the CUs (a third, ..) carry the light blocks and idle behind them (N = 160: 0.45 of the
    // peak).  One chunk per CU and post block, post-block-major: the dispatcher hands every CU a full block first and a
    // light one behind it -- balanced whatever the cost ratio (dev option 91 = 1: the chunk-major grid)
    pl.pb_major = (wide && pl.version == 5 && pl.nPB > 1 && pl.nPT % 8 >= 1 && pl.nPT % 8 <= 6 && h->opt_nchunks == 0 &&
                   h->opt_pbmajor != 1) ? 1 : 0;
    if (pl.pb_major) target = h->numCU;
    target = std::min(target, pl.nTiles);
    if (h->opt_nchunks == 0 && wgPerCU > 1) {
        // short recordings: a chunk keeps >= 8 tiles as long as every CU still gets a workgroup (per-chunk
        // prologue, partial write-out and the reduction over chunks are paid per chunk)
        // (one-wave workgroups -- a single post tile on k_fused7 -- have a light prologue and fill a SIMD each: chunks from
        //  four tiles on, every SIMD a wave; N = 16, T = 60 s: 0.047 -> 0.042 ms per evaluation, T = 20 s: 0.035 -> 0.033)
        const bool one_wave = pl.version == 7 && pl.nw7 == 1;
        const int floor_t = std::min(std::max(1, (one_wave ? 4 : 1) * h->numCU / pl.nPB), pl.nTiles);
        target = std::min(target, std::max(floor_t, pl.nTiles / (one_wave ? 4 : 8)));
    }
    pl.tilesPerChunk = (pl.nTiles + target - 1) / target;
    pl.nChunks = (pl.nTiles + pl.tilesPerChunk - 1) / pl.tilesPerChunk;
    pl.blocks = pl.nChunks * pl.nPB;
    pl.threads = 64 * pl.wpb;
    const size_t esz = pl.f32 ? 4 : 8;
    size_t off = ((size_t)16 * pl.rsf * esz + 15) & ~(size_t)15;
    if (pl.version == 5) {
        pl.lds = (size_t)2 * pgl_img_bytes(pl.ktl) + pgl_img_bytes(pl.kth) + 256 + 8 * 192 * 8;   // + per-wave spike scratch
        // the last block leaves waves without a post tile: they help (rows from 10 k-tiles on; dev option 92 = 1: never)
        const int nb = pl.nPT % 8;
        // (measured, r06_shape_sweep*.md / r06_shard_steps.txt: five or six tiles +7 .. 12 %; a light block of one or two tiles
        //  at the end of a wide population +4 .. 5 %; three tiles of a single slice (a 48-neuron range of C3) +5 %, but -2 % as the
        //  last block of a wide population, whose full blocks pay for the helper form; four tiles: the helpers share their
        //  tile's SIMD, +-0)
        const bool nb_ok = nb == 5 || nb == 6 || (wide ? (nb == 1 || nb == 2) : nb == 3);
        pl.hlp = (hlp_ok && nb_ok && pl.ktl >= 5 && h->opt_hlp != 1) ? 1 : 0;
        if (pl.hlp) pl.lds += 4 * 256 * 8;                                                         // + the helpers' partial currents
        if (pl.lds > 160 * 1024) return fail(PGL_ERR_UNSUPPORTED, "LDS budget exceeded");
        return PGL_OK;
    }
    if (pl.version == 7) {
        pl.lds = (size_t)2 * pgl_img_bytes(pl.KT) + 256 + (size_t)pl.nw7 * 192 * 8 + pl.lds7x;
        return PGL_OK;
    }
    if (pl.version == 6) {
        pl.lds = (size_t)(pl.sb6 ? 1 : 2) * pl.mt * pgl_img_bytes(pl.KT) + (size_t)pl.mt * pl.nw6 * 2048 + 256 + (size_t)pl.nw6 * 384;
        if (pl.sb6 == 2) pl.lds = lds_fused8(pl.img32 ? 5 : kRing8);
        // chunks are whole steps of mt tiles
        pl.tilesPerChunk = (pl.tilesPerChunk + pl.mt - 1) / pl.mt * pl.mt;
        pl.nChunks = (pl.nTiles + pl.tilesPerChunk - 1) / pl.tilesPerChunk;
        pl.blocks = pl.nChunks * pl.nPB;
        return PGL_OK;
    }
    if (pl.version == 4) {
        const int c0 = pl.KTW * 16;
        const int rsfh = c0 + ((c0 % 32 == 0) ? 16 : 32);
        off = std::max(off, (((size_t)2 * 16 * rsfh * 8) + 15) & ~(size_t)15);
        off += (((size_t)2 * h->B * pl.RP * 8) + 15) & ~(size_t)15;
        off += (size_t)sl.Ns * pl.cap * 8;
        off += 2 * ((((size_t)2 * sl.Ns * 4) + 15) & ~(size_t)15);
        off += (((size_t)sl.Ns * 4) + 15) & ~(size_t)15;
        off += 256;
        pl.lds = off;
        if (pl.lds > 160 * 1024) return fail(PGL_ERR_UNSUPPORTED, "LDS budget exceeded");
        return PGL_OK;
    }
    off += (((size_t)2 * h->B * pl.RP * esz) + 15) & ~(size_t)15;
    off += (size_t)sl.Ns * pl.cap * 8;
    off += 2 * ((((size_t)2 * sl.Ns * 4) + 15) & ~(size_t)15);
    off += (((size_t)sl.Ns * 4) + 15) & ~(size_t)15;                       // ring-valid flags
    off += (size_t)pl.wpb * 256 * 8 + (size_t)pl.PTW * 256 * 8 + 256;
    pl.lds = off;
    if (pl.lds > 160 * 1024) return fail(PGL_ERR_UNSUPPORTED, "LDS budget exceeded");
    return PGL_OK;
}

// Dry run of the dispatch (pgl_plan_kernels): when g_dry is set the launch_* templates record the name of the kernel
// instantiation they would launch (as the code object's demangled symbol reads) and launch nothing.  The recorded set
// over a grid of shapes is the set of instantiations the dispatcher can reach: tests/test_capi_symbols.py holds every
// one of them to zero bytes of scratch, tools/reachable_kernels.py diffs it against the built library.
static thread_local std::vector<std::string>* g_dry = nullptr;
static bool dry_record(const char* fam, std::initializer_list<int> args, const char* tail = nullptr)
{
    if (!g_dry) return false;
    std::string n = std::string(fam) + "<";
    bool first = true;
    for (int a : args) {
        if (!first) n += ", ";
        n += std::to_string(a);
        first = false;
    }
    if (tail) n += std::string(", ") + tail;
    n += ">";
    g_dry->push_back(n);
    return true;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) costs several microseconds of host time: it is issued once per
// kernel instantiation and device (and again only for a larger size), not on every launch -- the small configurations
// are bound by the host's submission rate (tools/step_bench.py)
template <typename K>
static hipError_t ensure_dyn_lds(K kern, size_t bytes)
{
    static std::map<std::pair<const void*, int>, size_t> have;      // (kernel, device) -> size already granted
    static std::mutex mu;                                           // ctypes drops the GIL: one handle per thread is legal
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& h = have[std::make_pair(reinterpret_cast<const void*>(kern), dev)];
    if (bytes <= h) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e == hipSuccess) h = bytes;
    return e;
}

template <int KTW, int PTW, int NW, int CAP, typename FT>
static hipError_t launch_fused2_t(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    if (dry_record("k_fused2", {KTW, PTW, NW, CAP}, sizeof(FT) == 4 ? "float" : "double")) return hipSuccess;
    auto kern = k_fused2<KTW, PTW, NW, CAP, FT>;
    hipError_t e = ensure_dyn_lds(kern, pl.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(NW * 64), pl.lds, s, fp);
    return hipGetLastError();
}

template <int PTW, int NW, int CAP, typename FT>
static hipError_t launch_fused2_k(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    constexpr int KSPLIT = NW / PTW;
    switch (pl.KTW) {
    case 1: return launch_fused2_t<1, PTW, NW, CAP, FT>(pl, fp, s);
    case 2: return launch_fused2_t<2, PTW, NW, CAP, FT>(pl, fp, s);
    case 3: return launch_fused2_t<3, PTW, NW, CAP, FT>(pl, fp, s);
    case 5: return launch_fused2_t<5, PTW, NW, CAP, FT>(pl, fp, s);
    case 7: if constexpr (7 * KSPLIT <= 40) return launch_fused2_t<7, PTW, NW, CAP, FT>(pl, fp, s); break;
    case 10: if constexpr (10 * KSPLIT <= 40) return launch_fused2_t<10, PTW, NW, CAP, FT>(pl, fp, s); break;
    case 20: if constexpr (20 * KSPLIT <= 40) return launch_fused2_t<20, PTW, NW, CAP, FT>(pl, fp, s); break;
    }
    return hipErrorInvalidValue;
}

template <int KTH>
static hipError_t launch_fused3_t(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    if (g_dry) {
        dry_record("k_fused3", {KTH, PGL_CAP, 1});
        if (fp.want_grad) dry_record("k_fused3", {KTH, PGL_CAP, 2});
        return hipSuccess;
    }
    auto k1 = k_fused3<KTH, PGL_CAP, 1>;
    auto k2 = k_fused3<KTH, PGL_CAP, 2>;
    hipError_t e = ensure_dyn_lds(k1, pl.lds);
    if (e == hipSuccess) e = ensure_dyn_lds(k2, pl.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k1, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
    e = hipGetLastError();
    if (e != hipSuccess || !fp.want_grad) return e;
    hipLaunchKernelGGL(k2, dim3(pl.blocks), dim3(512), pl.lds, s, fp);   // second pass: other half of G
    return hipGetLastError();
}

static hipError_t launch_fused3(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    switch (pl.KTW) {
    case 1: return launch_fused3_t<1>(pl, fp, s);
    case 2: return launch_fused3_t<2>(pl, fp, s);
    case 3: return launch_fused3_t<3>(pl, fp, s);
    case 5: return launch_fused3_t<5>(pl, fp, s);
    case 7: return launch_fused3_t<7>(pl, fp, s);
    case 10: return launch_fused3_t<10>(pl, fp, s);
    case 13: return launch_fused3_t<13>(pl, fp, s);
    case 16: return launch_fused3_t<16>(pl, fp, s);
    }
    return hipErrorInvalidValue;
}

// passes: 1, 2, or 0 = both back to back
template <int KTL, int KTH, int XIN = 0, int HLP = 0>
static hipError_t launch_fused5_t(const Plan& pl, const FusedParams& fp, hipStream_t s, int pass)
{
    if (g_dry) {
        if (pass != 2) dry_record("k_fused5", {KTL, KTH, 1, XIN, 0, HLP});
        if (pass != 1 && fp.want_grad) dry_record("k_fused5", {KTL, KTH, 2, 0, 0, HLP});
        return hipSuccess;
    }
    auto k1 = k_fused5<KTL, KTH, 1, XIN, 0, HLP>;
    auto k2 = k_fused5<KTL, KTH, 2, 0, 0, HLP>;
    const size_t lds2 = (size_t)2 * pgl_img_bytes(KTH) + 256;
    hipError_t e = ensure_dyn_lds(k1, pl.lds);
    if (e == hipSuccess) e = ensure_dyn_lds(k2, lds2);
    if (e != hipSuccess) return e;
    if (pass != 2) {
        hipLaunchKernelGGL(k1, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (pass != 1 && fp.want_grad) {
        hipLaunchKernelGGL(k2, dim3(pl.blocks), dim3(512), lds2, s, fp);
        e = hipGetLastError();
    }
    return e;
}

static hipError_t launch_fused5(const Plan& pl, const FusedParams& fp, hipStream_t s, int pass = 0)
{
    if (pl.hlp) {                                // blocks of five or six post tiles: the idle waves help (make_plan)
        switch (pl.ktl << 8 | pl.kth) {
        case 5 << 8 | 5: return launch_fused5_t<5, 5, 0, 1>(pl, fp, s, pass);
        case 7 << 8 | 7: return launch_fused5_t<7, 7, 0, 1>(pl, fp, s, pass);
        case 9 << 8 | 11: return launch_fused5_t<9, 11, 0, 1>(pl, fp, s, pass);
        case 12 << 8 | 14: return launch_fused5_t<12, 14, 0, 1>(pl, fp, s, pass);
        case 14 << 8 | 18: return launch_fused5_t<14, 18, 0, 1>(pl, fp, s, pass);
        case PGL_SPLIT_L << 8 | (40 - PGL_SPLIT_L): return launch_fused5_t<PGL_SPLIT_L, 40 - PGL_SPLIT_L, 0, 1>(pl, fp, s, pass);
        }
        return hipErrorInvalidValue;
    }
    switch (pl.ktl << 8 | pl.kth) {
    case 1 << 8 | 1: return launch_fused5_t<1, 1>(pl, fp, s, pass);
    case 2 << 8 | 2: return launch_fused5_t<2, 2>(pl, fp, s, pass);
    case 3 << 8 | 3: return launch_fused5_t<3, 3>(pl, fp, s, pass);
    case 5 << 8 | 5: return launch_fused5_t<5, 5>(pl, fp, s, pass);
    case 7 << 8 | 7: return launch_fused5_t<7, 7>(pl, fp, s, pass);
    case 9 << 8 | 11: return launch_fused5_t<9, 11>(pl, fp, s, pass);
    case 12 << 8 | 14: return launch_fused5_t<12, 14>(pl, fp, s, pass);
    case 14 << 8 | 18: return launch_fused5_t<14, 18>(pl, fp, s, pass);
    case PGL_SPLIT_L << 8 | (40 - PGL_SPLIT_L): return launch_fused5_t<PGL_SPLIT_L, 40 - PGL_SPLIT_L>(pl, fp, s, pass);
    }
    return hipErrorInvalidValue;
}

// slab-input form of pass 1 (separable stimulus at the frame rate, 65 .. 128 neurons: at least 5 post tiles of >= 2 bases)
static hipError_t launch_fused5_xin(const Plan& pl, const FusedParams& fp, hipStream_t s, int pass = 0)
{
    switch (pl.ktl << 8 | pl.kth) {
    case 5 << 8 | 5: return launch_fused5_t<5, 5, 1>(pl, fp, s, pass);
    case 7 << 8 | 7: return launch_fused5_t<7, 7, 1>(pl, fp, s, pass);
    case 9 << 8 | 11: return launch_fused5_t<9, 11, 1>(pl, fp, s, pass);
    case 12 << 8 | 14: return launch_fused5_t<12, 14, 1>(pl, fp, s, pass);
    case 14 << 8 | 18: return launch_fused5_t<14, 18, 1>(pl, fp, s, pass);
    case PGL_SPLIT_L << 8 | (40 - PGL_SPLIT_L): return launch_fused5_t<PGL_SPLIT_L, 40 - PGL_SPLIT_L, 1>(pl, fp, s, pass);
    }
    return hipErrorInvalidValue;
}

// column slices of a wide population (N > 128 or more than 640 feature columns) on resident tiles.  mode 0: forward only,
// the slice's partial currents written to the slab; 1: forward only, added to the slab; 2: the last slice -- pass 1 from the
// slab (epilogue, residuals out, G of its L columns); 3: pass 2 on the H part; 4: pass 2 on the L part (the gradient of the
// L columns of a slice whose pass 1 was forward only)
template <int KTL, int KTH, int HLP = 0>
static hipError_t launch_fused5_wide_t(const Plan& pl, const FusedParams& fp, hipStream_t s, int mode)
{
    const size_t lds2h = (size_t)2 * pgl_img_bytes(KTH) + 256, lds2l = (size_t)2 * pgl_img_bytes(KTL) + 256;
    if (g_dry) {
        if (mode <= 1) dry_record("k_fused5", {KTL, KTH, 1, mode == 0 ? 2 : 3, 0, HLP});
        else if (mode == 2) dry_record("k_fused5", {KTL, KTH, 1, 1, 0, HLP});
        else dry_record("k_fused5", {KTL, KTH, 2, 0, mode == 4 ? 1 : 0, HLP});
        return hipSuccess;
    }
    hipError_t e = hipSuccess;
    if (mode == 0) {
        auto k = k_fused5<KTL, KTH, 1, 2, 0, HLP>;
        if ((e = ensure_dyn_lds(k, pl.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
    } else if (mode == 1) {
        auto k = k_fused5<KTL, KTH, 1, 3, 0, HLP>;
        if ((e = ensure_dyn_lds(k, pl.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
    } else if (mode == 2) {
        auto k = k_fused5<KTL, KTH, 1, 1, 0, HLP>;
        if ((e = ensure_dyn_lds(k, pl.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
    } else if (mode == 3) {
        auto k = k_fused5<KTL, KTH, 2, 0, 0, HLP>;
        if ((e = ensure_dyn_lds(k, lds2h)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(512), lds2h, s, fp);
    } else {
        auto k = k_fused5<KTL, KTH, 2, 0, 1, HLP>;
        if ((e = ensure_dyn_lds(k, lds2l)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(512), lds2l, s, fp);
    }
    return hipGetLastError();
}
static hipError_t launch_fused5_wide(const Plan& pl, const FusedParams& fp, hipStream_t s, int mode)
{
    if (pl.hlp) {                                // the last post block holds one to six tiles: its idle waves help (make_plan)
        switch (pl.ktl << 8 | pl.kth) {
        case 5 << 8 | 5: return launch_fused5_wide_t<5, 5, 1>(pl, fp, s, mode);
        case 7 << 8 | 7: return launch_fused5_wide_t<7, 7, 1>(pl, fp, s, mode);
        case 9 << 8 | 11: return launch_fused5_wide_t<9, 11, 1>(pl, fp, s, mode);
        case 12 << 8 | 14: return launch_fused5_wide_t<12, 14, 1>(pl, fp, s, mode);
        case 14 << 8 | 18: return launch_fused5_wide_t<14, 18, 1>(pl, fp, s, mode);
        case PGL_SPLIT_L << 8 | (40 - PGL_SPLIT_L): return launch_fused5_wide_t<PGL_SPLIT_L, 40 - PGL_SPLIT_L, 1>(pl, fp, s, mode);
        }
        return hipErrorInvalidValue;
    }
    switch (pl.ktl << 8 | pl.kth) {
    case 5 << 8 | 5: return launch_fused5_wide_t<5, 5>(pl, fp, s, mode);
    case 7 << 8 | 7: return launch_fused5_wide_t<7, 7>(pl, fp, s, mode);
    case 9 << 8 | 11: return launch_fused5_wide_t<9, 11>(pl, fp, s, mode);
    case 12 << 8 | 14: return launch_fused5_wide_t<12, 14>(pl, fp, s, mode);
    case 14 << 8 | 18: return launch_fused5_wide_t<14, 18>(pl, fp, s, mode);
    case PGL_SPLIT_L << 8 | (40 - PGL_SPLIT_L): return launch_fused5_wide_t<PGL_SPLIT_L, 40 - PGL_SPLIT_L>(pl, fp, s, mode);
    }
    return hipErrorInvalidValue;
}

// Instantiations of the switch below that no plan of make_plan selects (tools/reachable_kernels.py: dry run of the
// dispatch over a grid of shapes, with and without forcing options) are not built; tests/test_capi_symbols.py fails when a
// reachable instantiation is missing from the library, so a change of make_plan shows up here.
constexpr bool fused6_built(int KTW, int PTW, int MT, int NW)
{
    if (NW == 4) return MT == 1 && ((PTW == 1 && KTW <= 3) || (PTW == 2 && KTW <= 7));
    if (MT == 2) return PTW == 4 && KTW >= 2 && KTW <= 7;
    // (four post tiles, one tile per step: the long-row form, and every row length on recordings of fewer than four tiles)
    return (PTW == 1 && (KTW == 2 || KTW == 3)) || (PTW == 2 && (KTW == 5 || KTW == 7)) || (PTW == 4 && KTW >= 2);
}
constexpr bool fused7_built(int KT, int NWV, int XIO)
{
    return !(XIO == 0 && ((KT == 1 && NWV >= 2) || (KT == 2 && NWV == 4)));
}

// occ != nullptr: no launch, *occ = workgroups of this instantiation a CU holds with pl.lds bytes of LDS
template <int KTW, int PTW, int MT, int NW, int DB = 1>
static hipError_t launch_fused6_t(const Plan& pl, const FusedParams& fp, hipStream_t s, int* occ)
{
    constexpr size_t need = (size_t)(DB ? 2 : 1) * MT * pgl_img_bytes(KTW * (NW / PTW)) + (size_t)MT * NW * 2048 + 256 + (size_t)NW * 384;
    if constexpr (need <= 160 * 1024 && KTW * 4 <= 40 && (DB == 0 || fused6_built(KTW, PTW, MT, NW))) {
        auto kern = k_fused6<KTW, PTW, MT, NW, DB>;
        if (!occ && dry_record("k_fused6", {KTW, PTW, MT, NW, DB})) return hipSuccess;
        if (occ && g_dry) return hipErrorInvalidValue;           // (dry run: no device to ask; the caller's default holds)
        hipError_t e = ensure_dyn_lds(kern, pl.lds);
        if (e != hipSuccess) return e;
        if (occ) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, kern, NW * 64, pl.lds);
        hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(NW * 64), pl.lds, s, fp);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int PTW, int MT, int NW>
static hipError_t launch_fused6_k(const Plan& pl, const FusedParams& fp, hipStream_t s, int* occ)
{
    switch (pl.KTW) {
    case 1: return launch_fused6_t<1, PTW, MT, NW>(pl, fp, s, occ);
    case 2: return launch_fused6_t<2, PTW, MT, NW>(pl, fp, s, occ);
    case 3: return launch_fused6_t<3, PTW, MT, NW>(pl, fp, s, occ);
    case 5: return launch_fused6_t<5, PTW, MT, NW>(pl, fp, s, occ);
    case 7: return launch_fused6_t<7, PTW, MT, NW>(pl, fp, s, occ);
    case 10: return launch_fused6_t<10, PTW, MT, NW>(pl, fp, s, occ);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_fused6(const Plan& pl, const FusedParams& fp, hipStream_t s, int* occ = nullptr)
{
    if (pl.nw6 == 4) {
        switch (pl.PTW * 4 + pl.mt) {
        case 1 * 4 + 1: return launch_fused6_k<1, 1, 4>(pl, fp, s, occ);
        case 2 * 4 + 1: return launch_fused6_k<2, 1, 4>(pl, fp, s, occ);
        }
        return hipErrorInvalidValue;
    }
    if (pl.sb6 == 2) {                             // one post tile of a 25 .. 40 k-tile row: per-wave block rings
        if (pl.mt != 1 || pl.PTW != 1 || pl.KTW != 5 || occ) return hipErrorInvalidValue;
        if (dry_record("k_fused8", {5, kRing8, pl.img32})) return hipSuccess;
        if (pl.img32) {
            auto kern = k_fused8<5, kRing8, 1>;
            hipError_t e = ensure_dyn_lds(kern, pl.lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
            return hipGetLastError();
        }
        auto kern = k_fused8<5, kRing8, 0>;
        hipError_t e = ensure_dyn_lds(kern, pl.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(512), pl.lds, s, fp);
        return hipGetLastError();
    }
    if (pl.sb6) {                                  // one image buffer: 640-column rows for two post tiles
        if (pl.mt == 1 && pl.PTW == 2 && pl.KTW == 10) return launch_fused6_t<10, 2, 1, 8, 0>(pl, fp, s, occ);
        return hipErrorInvalidValue;
    }
    switch (pl.PTW * 4 + pl.mt) {
    case 1 * 4 + 1: return launch_fused6_k<1, 1, 8>(pl, fp, s, occ);
    case 1 * 4 + 2: return launch_fused6_k<1, 2, 8>(pl, fp, s, occ);
    case 2 * 4 + 1: return launch_fused6_k<2, 1, 8>(pl, fp, s, occ);
    case 2 * 4 + 2: return launch_fused6_k<2, 2, 8>(pl, fp, s, occ);
    case 4 * 4 + 1: return launch_fused6_k<4, 1, 8>(pl, fp, s, occ);
    case 4 * 4 + 2: return launch_fused6_k<4, 2, 8>(pl, fp, s, occ);
    }
    return hipErrorInvalidValue;
}

// workgroups per CU of the k_fused6 instantiation a plan selects (registers and LDS), cached per shape
static int fused6_wg_per_cu(const Plan& pl)
{
    // occupancy is a property of the kernel and the architecture (every device of a node is the same gfx950 part)
    static int cache[2][11][5][3][9];          // [sb6][KTW][PTW][mt][nw6]; 0 = not asked yet
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int& c = cache[pl.sb6 ? 1 : 0][pl.KTW][pl.PTW][pl.mt][pl.nw6];
    if (c == 0) {
        int occ = 0;
        FusedParams fp{};
        if (launch_fused6(pl, fp, nullptr, &occ) != hipSuccess || occ < 1) occ = (pl.nw6 == 4) ? 2 : 1;
        c = occ;
    }
    return c;
}

// workgroups of k_gibbs_rate_cols a CU holds at `lds` bytes of dynamic LDS (occupancy query, cached per size)
static int gibbs_rate_wg_per_cu(size_t lds)
{
    static std::map<size_t, int> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int& c = cache[lds];
    if (c == 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gibbs_rate_cols, 256, lds) != hipSuccess || occ < 1)
            occ = 3;
        c = occ;
    }
    return c;
}

template <int KT, int NWV, int XIO = 0>
static hipError_t launch_fused7_t(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    if constexpr (fused7_built(KT, NWV, XIO)) {
        if (dry_record("k_fused7", {KT, NWV, XIO})) return hipSuccess;
        auto kern = k_fused7<KT, NWV, XIO>;
        hipError_t e = ensure_dyn_lds(kern, pl.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(NWV * 64), pl.lds, s, fp);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int NWV>
static hipError_t launch_fused7_k(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    switch (pl.KT) {
    case 1: return launch_fused7_t<1, NWV>(pl, fp, s);
    case 2: return launch_fused7_t<2, NWV>(pl, fp, s);
    case 3: return launch_fused7_t<3, NWV>(pl, fp, s);
    case 5: return launch_fused7_t<5, NWV>(pl, fp, s);
    case 7: return launch_fused7_t<7, NWV>(pl, fp, s);
    case 10: return launch_fused7_t<10, NWV>(pl, fp, s);
    case 12: return launch_fused7_t<12, NWV>(pl, fp, s);
    case 13: return launch_fused7_t<13, NWV>(pl, fp, s);
    case 16: return launch_fused7_t<16, NWV>(pl, fp, s);
    }
    return hipErrorInvalidValue;
}

// slab-input form (separable stimulus at the frame rate): 4-wave workgroups only
static hipError_t launch_fused7_xio(const Plan& pl, const FusedParams& fp, hipStream_t s, int xio = 1)
{
    if (pl.nw7 != 4) return hipErrorInvalidValue;
    if (xio == 3) {                              // ... and its backward inside the kernel as well (no residual slab)
        switch (pl.KT) {
        case 1: return launch_fused7_t<1, 4, 3>(pl, fp, s);
        case 2: return launch_fused7_t<2, 4, 3>(pl, fp, s);
        case 3: return launch_fused7_t<3, 4, 3>(pl, fp, s);
        case 5: return launch_fused7_t<5, 4, 3>(pl, fp, s);
        case 7: return launch_fused7_t<7, 4, 3>(pl, fp, s);
        case 10: return launch_fused7_t<10, 4, 3>(pl, fp, s);
        case 12: return launch_fused7_t<12, 4, 3>(pl, fp, s);
        }
        return hipErrorInvalidValue;
    }
    if (xio == 2) {                              // stimulus current inside the forward contraction
        switch (pl.KT) {
        case 1: return launch_fused7_t<1, 4, 2>(pl, fp, s);
        case 2: return launch_fused7_t<2, 4, 2>(pl, fp, s);
        case 3: return launch_fused7_t<3, 4, 2>(pl, fp, s);
        case 5: return launch_fused7_t<5, 4, 2>(pl, fp, s);
        case 7: return launch_fused7_t<7, 4, 2>(pl, fp, s);
        case 10: return launch_fused7_t<10, 4, 2>(pl, fp, s);
        case 12: return launch_fused7_t<12, 4, 2>(pl, fp, s);
        case 13: return launch_fused7_t<13, 4, 2>(pl, fp, s);
        }
        return hipErrorInvalidValue;
    }
    switch (pl.KT) {
    case 1: return launch_fused7_t<1, 4, 1>(pl, fp, s);
    case 2: return launch_fused7_t<2, 4, 1>(pl, fp, s);
    case 3: return launch_fused7_t<3, 4, 1>(pl, fp, s);
    case 5: return launch_fused7_t<5, 4, 1>(pl, fp, s);
    case 7: return launch_fused7_t<7, 4, 1>(pl, fp, s);
    case 10: return launch_fused7_t<10, 4, 1>(pl, fp, s);
    case 12: return launch_fused7_t<12, 4, 1>(pl, fp, s);
    case 13: return launch_fused7_t<13, 4, 1>(pl, fp, s);
    case 16: return launch_fused7_t<16, 4, 1>(pl, fp, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_fused7(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    switch (pl.nw7) {
    case 1: return launch_fused7_k<1>(pl, fp, s);
    case 2: return launch_fused7_k<2>(pl, fp, s);
    case 4: return launch_fused7_k<4>(pl, fp, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_fused2(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    if (pl.version == 7) return launch_fused7(pl, fp, s);
    if (pl.version == 6) return launch_fused6(pl, fp, s);
    if (pl.version == 5) return launch_fused5(pl, fp, s);
    if (pl.version == 4) return launch_fused3(pl, fp, s);
    if (pl.version == 3) {
        switch (pl.PTW) {
        case 1: return launch_fused2_k<1, 8, PGL_CAP, float>(pl, fp, s);
        case 2: return launch_fused2_k<2, 8, PGL_CAP, float>(pl, fp, s);
        case 4: return launch_fused2_k<4, 8, PGL_CAP, float>(pl, fp, s);
        }
        return hipErrorInvalidValue;
    }
    switch (pl.PTW) {
    case 1: return launch_fused2_k<1, 8, PGL_CAP, double>(pl, fp, s);
    case 2: return launch_fused2_k<2, 8, PGL_CAP, double>(pl, fp, s);
    case 4: return launch_fused2_k<4, 8, PGL_CAP, double>(pl, fp, s);
    }
    return hipErrorInvalidValue;
}

// separable stimulus at the frame rate: which = 0 forward, 1 backward, 2 finish
template <int J, int BT>
static void launch_sepf(int which, const SepfParams& sp, hipStream_t s);
// (the thin GEMMs of the separable stimulus: k_gemm_kc when the k-contiguous operands are 16-byte aligned, else k_gemm_mfma)
static inline bool gemm_aligned(const void* p, long long ld) { return ((uintptr_t)p % 16) == 0 && (ld % 2) == 0; }

template <int J, int BT>
static void launch_sepf(int which, const SepfParams& sp, hipStream_t s)
{
    const long long nF = sp.F1 - sp.F0 + 1;
    const int nG = (sp.nPT + 3) / 4;
    dim3 grid((unsigned)((nF + 3) / 4), (unsigned)nG);
    if (which == 0) hipLaunchKernelGGL((k_sepf_fwd<J, BT>), grid, dim3(256), 0, s, sp);
    else if (which == 1) hipLaunchKernelGGL((k_sepf_bwd<J, BT>), grid, dim3(256), 0, s, sp);
    else {
        const int nA = (int)((sp.Tstim + 15) / 16) * nG;
        hipLaunchKernelGGL((k_sepf_finish<J, BT>), dim3(nA + nG * BT), dim3(1024), 0, s, sp, nA, nG);
    }
}

extern "C" {

const char* pgl_last_error(void) { return g_err.c_str(); }
int pgl_version(void) { return 100; }

int pgl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pgl_create(int N, int64_t nT, int B, int R, int nlin, double dt, int device, pgl_handle* out)
{
    if (!out) return fail(PGL_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (N <= 0 || nT <= 0 || B <= 0 || R <= 0) return fail(PGL_ERR_ARG, "N, nT, B, R must be positive");
    if (B > PGL_MAXB) return fail(PGL_ERR_UNSUPPORTED, "B > 8 basis functions");
    if (nlin != PGL_NLIN_EXP && nlin != PGL_NLIN_EXPLINEAR) return fail(PGL_ERR_ARG, "unknown nonlinearity");
    if (nT > (int64_t)1 << 30) return fail(PGL_ERR_UNSUPPORTED, "nT > 2^30 bins");
    if (!(dt > 0)) return fail(PGL_ERR_ARG, "dt must be positive");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(PGL_ERR_HIP, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(PGL_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    pgl_context* h = new pgl_context();
    h->N = N; h->nT = nT; h->B = B; h->R = R; h->Rk = R; h->nlin = nlin; h->dt = dt; h->device = device;
    h->Kimp = N * B; h->Ktot = h->Kimp; h->nT16 = (int)((nT + 15) / 16);
    h->t_lo = 0; h->t_hi = nT;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->numCU = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(PGL_ERR_HIP, hipGetErrorString(e)); }
    h->stream = h->own_stream;
    {
        // the side stream reduces the first G half beside pass 2 of the two-pass kernel: highest priority, so that its
        // blocks take every wave slot pass 2 leaves free and the reduction is over before pass 2 is (at default
        // priority it finished ~15 us after a 1/8-recording pass 2 and delayed the second reduction)
        int prio_lo = 0, prio_hi = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_hi = 0;
        e = hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) { pgl_destroy(h); return fail(PGL_ERR_HIP, hipGetErrorString(e)); }
    for (int s = 0; s < pgl_context::NEV; ++s)
        for (int i = 0; i < 4; ++i) {
            e = hipEventCreate(&h->evr[s][i]);
            if (e != hipSuccess) { pgl_destroy(h); return fail(PGL_ERR_HIP, hipGetErrorString(e)); }
        }
    *out = h;
    return PGL_OK;
}

int pgl_destroy(pgl_handle h)
{
    if (!h) return PGL_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    DevBuf* bufs[] = {&h->S, &h->ST, &h->spk, &h->wlo, &h->whi, &h->phi, &h->fstim, &h->theta,
                      &h->Weff, &h->ll, &h->grad, &h->Wfrag, &h->bias, &h->Gpart, &h->llpart,
                      &h->gbpart, &h->Xbuf, &h->imgs[0].buf, &h->imgs[1].buf, &h->imgs[2].buf, &h->imgs[3].buf, &h->imgs[4].buf, &h->imgs[5].buf, &h->imgs[6].buf, &h->imgs[7].buf, &h->IimpT, &h->Inet, &h->Istim, &h->tmpA, &h->tmpB, &h->tmpC,
                      &h->wsmall, &h->part, &h->outK, &h->lam, &h->wcol, &h->thetan, &h->GX, &h->gtheta,
                      &h->gargs, &h->gpart, &h->gout, &h->ghs, &h->gfs, &h->zf, &h->zfT, &h->sbt, &h->Yf, &h->Qb, &h->Qf,
                      &h->spart, &h->sepC, &h->sepA, &h->sepAT, &h->sepD, &h->YfT, &h->Hb, &h->wpart, &h->QvT, &h->staA};
    for (DevBuf* b : bufs) release(*b);
    for (int s = 0; s < pgl_context::NEV; ++s)
        for (int i = 0; i < 4; ++i)
            if (h->evr[s][i]) (void)hipEventDestroy(h->evr[s][i]);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_args) (void)hipHostFree(h->pin_args);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return PGL_OK;
}

int pgl_set_time_range(pgl_handle h, int64_t t_lo, int64_t t_hi)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    if (t_lo < 0 || t_hi > h->nT || t_lo >= t_hi) return fail(PGL_ERR_ARG, "bad time range");
    if (t_lo % 16 != 0) return fail(PGL_ERR_ARG, "t_lo must be a multiple of 16 bins");
    h->t_lo = t_lo;
    h->t_hi = t_hi;
    return PGL_OK;
}

int pgl_set_option(pgl_handle h, int option, int value)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    switch (option) {
    case PGL_OPT_FEATURE_F32: h->opt_f32 = (value == 1) ? 1 : 0; h->opt_img32 = (value == 2) ? 1 : 0; return PGL_OK;
    case 99: h->opt_dbg = value; return PGL_OK;
    case 98: h->opt_ptw = value; return PGL_OK;
    case 97: if (value < 0 || value > 16) return fail(PGL_ERR_ARG, "finalize waves: 0 (auto) .. 16"); h->opt_finw = value; return PGL_OK;
    case 91: h->opt_pbmajor = value; return PGL_OK;          // dev: 1 = chunk-major grid for wide populations (A/B)
    case 92: h->opt_hlp = value; return PGL_OK;              // dev: 1 = no helper waves in the two-pass kernel (A/B)
    case 93: h->opt_slice_cols = value; return PGL_OK;      // dev: feature columns per slice of the 3-phase path (0 = 640)
    case 94: h->opt_sepf = value; return PGL_OK;            // dev: 2 = separable stimulus always by the tap-rate kernels, 3 = stimulus current through the slab, 4 = residual slab + k_sepf_bwd (no fused backward)
    case 95: h->opt_sb6 = value; return PGL_OK;              // dev: 2 = never the one-buffer / block-ring forms
    case PGL_OPT_KERNEL: h->opt_kernel = value; return PGL_OK;
    case PGL_OPT_GIBBS_KERNEL: h->opt_gibbs = value; return PGL_OK;
    case PGL_OPT_EPI_F64: h->opt_epi64 = value ? 1 : 0; return PGL_OK;
    case PGL_OPT_TIMING: if (value < 0) return fail(PGL_ERR_ARG, "timing interval < 0"); h->opt_timing = value; h->call_no = 0; return PGL_OK;
    case PGL_OPT_BFGS_MERGE: if (value < 0) return fail(PGL_ERR_ARG, "merge threshold < 0"); h->opt_bfgs_merge = value; return PGL_OK;
    case PGL_OPT_NCHUNKS: if (value < 0) return fail(PGL_ERR_ARG, "nchunks < 0"); h->opt_nchunks = value; return PGL_OK;
    }
    return fail(PGL_ERR_ARG, "unknown option");
}

// per 16-row tile event windows of every neuron: lo = first event with s >= 16*tile - Rk,
// hi = first event with s >= 16*tile + 15 (events are sorted by time within a neuron)
static int upload_windows(pgl_handle h)
{
    const int N = h->N, nT16 = h->nT16;
    const std::vector<int>& ptr = h->h_ptr;
    const std::vector<int2>& ev = h->h_ev;
    std::vector<int> wlo((size_t)nT16 * N), whi((size_t)nT16 * N);
    for (int n = 0; n < N; ++n) {
        int lo = ptr[n], hi = ptr[n];
        const int end = ptr[n + 1];
        for (int tile = 0; tile < nT16; ++tile) {
            const int64_t klo = (int64_t)16 * tile - h->Rk;
            const int64_t khi = (int64_t)16 * tile + 15;
            while (lo < end && ev[(size_t)lo].x < klo) ++lo;
            while (hi < end && ev[(size_t)hi].x < khi) ++hi;
            wlo[(size_t)tile * N + n] = lo;
            whi[(size_t)tile * N + n] = hi;
        }
    }
    ENSURE(h->wlo, wlo.size() * 4);
    ENSURE(h->whi, whi.size() * 4);
    HIPCHK(hipMemcpyAsync(h->wlo.p, wlo.data(), wlo.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->whi.p, whi.data(), whi.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));          // the host vectors die with this frame
    return PGL_OK;
}

// Build the spike-event index from the dense count matrix and upload everything.
static int upload_spikes(pgl_handle h, const uint8_t* S)
{
    HIPCHK(hipSetDevice(h->device));
    const int N = h->N;
    const int64_t nT = h->nT;
    // two passes over the nT x N counts (77 MB at C3), each on up to 16 host threads over slabs of time: per-slab event
    // counts of every neuron, a prefix over (neuron, slab) = every slab's write position in a neuron's list, then the
    // slabs fill their own pieces -- the lists stay time-sorted (single-threaded: 0.15 s of a 0.19 s upload at C3)
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, std::max(1u, std::thread::hardware_concurrency())),
                                                               nT / 65536));
    std::vector<int64_t> t0s(nth + 1);
    for (int i = 0; i <= nth; ++i) t0s[i] = nT * i / nth;
    std::vector<std::vector<int>> slab((size_t)nth, std::vector<int>((size_t)N, 0));
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int i = 1; i < nth; ++i) th.emplace_back(fn, i);
        fn(0);
        for (auto& t : th) t.join();
    };
    run([&](const int i) {
        int* c = slab[(size_t)i].data();
        for (int64_t t = t0s[i]; t < t0s[i + 1]; ++t) {
            const uint8_t* row = S + t * N;
            for (int n = 0; n < N; ++n) c[n] += row[n] != 0;
        }
    });
    std::vector<int> cnt(N + 1, 0);
    for (int n = 0; n < N; ++n) {
        int tot = cnt[n];
        for (int i = 0; i < nth; ++i) {                    // slab[i][n] becomes the slab's first position in neuron n's list
            const int c = slab[(size_t)i][(size_t)n];
            slab[(size_t)i][(size_t)n] = tot;
            tot += c;
        }
        cnt[n + 1] = tot;
    }
    const int64_t nnz = cnt[N];
    std::vector<int2> ev((size_t)std::max<int64_t>(nnz, 1));
    run([&](const int i) {
        int* cur = slab[(size_t)i].data();
        for (int64_t t = t0s[i]; t < t0s[i + 1]; ++t) {
            const uint8_t* row = S + t * N;
            for (int n = 0; n < N; ++n)
                if (row[n]) ev[(size_t)cur[n]++] = make_int2((int)t, (int)row[n]);
        }
    });
    h->h_ptr = cnt;
    h->h_ev.swap(ev);
    const std::vector<int2>& evr = h->h_ev;
    // S is padded with zero rows up to the 16-row tile grid: the fused kernels read the counts of whole
    // tiles through incremented pointers, without clamping the row index
    const size_t s_rows = (size_t)h->nT16 * 16;
    ENSURE(h->S, s_rows * N);
    ENSURE(h->ST, (size_t)nT * N);
    ENSURE(h->spk, evr.size() * sizeof(int2));
    HIPCHK(hipMemsetAsync((uint8_t*)h->S.p + (size_t)nT * N, 0, (s_rows - (size_t)nT) * N, h->stream));
    HIPCHK(hipMemcpyAsync(h->S.p, S, (size_t)nT * N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->spk.p, evr.data(), evr.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    {
        const int rc = upload_windows(h);
        if (rc) return rc;
    }
    dim3 grid((unsigned)((nT + 63) / 64), (unsigned)((N + 63) / 64));
    hipLaunchKernelGGL(k_transpose_u8, grid, dim3(256), 0, h->stream, (const uint8_t*)h->S.p,
                       (uint8_t*)h->ST.p, (long long)nT, N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->nnz = nnz;
    h->have_spikes = true;
    h->gibbs_npost = -1;
    invalidate_images(h);
    h->gx_xs = 0;
    return PGL_OK;
}

int pgl_set_spikes_u8(pgl_handle h, const uint8_t* S)
{
    if (!h || !S) return fail(PGL_ERR_ARG, "null argument");
    return upload_spikes(h, S);
}

int pgl_set_spikes_f64(pgl_handle h, const double* S)
{
    if (!h || !S) return fail(PGL_ERR_ARG, "null argument");
    const size_t n = (size_t)h->nT * h->N;
    std::vector<uint8_t> u(n);
    for (size_t i = 0; i < n; ++i) {
        const double v = S[i];
        if (!(v >= 0.0 && v <= 255.0) || v != std::floor(v))
            return fail(PGL_ERR_ARG, "spike counts must be integers in 0..255");
        u[i] = (uint8_t)v;
    }
    return upload_spikes(h, u.data());
}

int pgl_set_basis(pgl_handle h, const double* ibasis)
{
    if (!h || !ibasis) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    // Trailing taps that are zero in every basis column contribute nothing (the raised-cosine
    // bases of the reference end before dt_max: basis.py:56-106 leaves the last ~10 % of the
    // 100-point grid at zero): the kernels run with Rk <= R taps and correspondingly shorter
    // event windows -- same sums, fewer events per tile.
    int Rk = 1;
    for (int d = 0; d < h->R; ++d)
        for (int b = 0; b < h->B; ++b)
            if (ibasis[(size_t)d * h->B + b] != 0.0) Rk = d + 1;
    std::vector<double> ph((size_t)h->B * Rk);
    for (int d = 0; d < Rk; ++d)
        for (int b = 0; b < h->B; ++b) ph[(size_t)b * Rk + d] = ibasis[(size_t)d * h->B + b];
    ENSURE(h->phi, ph.size() * 8);
    HIPCHK(hipMemcpyAsync(h->phi.p, ph.data(), ph.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (Rk != h->Rk) {
        h->Rk = Rk;
        if (h->have_spikes) {
            const int rc = upload_windows(h);
            if (rc) return rc;
        }
    }
    h->have_basis = true;
    h->gibbs_npost = -1;
    invalidate_images(h);
    h->gx_xs = 0;
    return PGL_OK;
}

int pgl_set_stim_features(pgl_handle h, const double* fstim, int Dstim)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    if (Dstim < 0 || (Dstim > 0 && !fstim)) return fail(PGL_ERR_ARG, "bad stimulus features");
    HIPCHK(hipSetDevice(h->device));
    h->Dstim = Dstim;
    h->Ktot = h->Kimp + Dstim;
    h->sep = false;
    h->gibbs_npost = -1;
    invalidate_images(h);
    h->gx_xs = 0;
    if (Dstim > 0) {
        const size_t bytes = (size_t)h->nT * Dstim * 8;
        ENSURE(h->fstim, bytes);
        HIPCHK(hipMemcpyAsync(h->fstim.p, fstim, bytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return PGL_OK;
}

int pgl_set_stimulus(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim,
                     const double* basis_x, int Bx, const double* basis_t, int Rt, int Bt, int layout)
{
    if (!h || !stim || !basis_t) return fail(PGL_ERR_ARG, "null argument");
    if (Tstim <= 0 || D <= 0 || Bx <= 0 || Rt <= 0 || Bt <= 0 || !(dt_stim > 0) || (layout != 0 && layout != 1))
        return fail(PGL_ERR_ARG, "bad stimulus description");
    if (!basis_x && Bx != D) return fail(PGL_ERR_ARG, "identity spatial basis needs Bx == D");
    const int Dstim = Bx * Bt;
    if ((size_t)(Rt + 256 + Rt * Bt) * 8 > 64 * 1024) return fail(PGL_ERR_UNSUPPORTED, "temporal basis too long");
    HIPCHK(hipSetDevice(h->device));
    DevBuf dstim, dbx, dbt, dzx;
    auto cleanup = [&]() { release(dstim); release(dbx); release(dbt); release(dzx); };
    int rc = ensure(dstim, (size_t)Tstim * D * 8);
    if (!rc) rc = ensure(dbt, (size_t)Rt * Bt * 8);
    if (!rc) rc = ensure(dzx, (size_t)h->nT * Bx * 8);
    if (!rc && basis_x) rc = ensure(dbx, (size_t)D * Bx * 8);
    if (!rc) rc = ensure(h->fstim, (size_t)h->nT * Dstim * 8);
    if (rc) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(dstim.p, stim, (size_t)Tstim * D * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dbt.p, basis_t, (size_t)Rt * Bt * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && basis_x)
        e = hipMemcpyAsync(dbx.p, basis_x, (size_t)D * Bx * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        const long long total = (long long)h->nT * Bx;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 65535);
        hipLaunchKernelGGL(k_stim_project, dim3(blocks), dim3(256), 0, h->stream, (const double*)dstim.p,
                           (long long)Tstim, D, dt_stim, h->dt, basis_x ? (const double*)dbx.p : nullptr,
                           Bx, (double*)dzx.p, (long long)h->nT);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        dim3 grid((unsigned)((h->nT + 255) / 256), (unsigned)Bx);
        hipLaunchKernelGGL(k_stim_conv, grid, dim3(256), (size_t)(Rt + 256 + Rt * Bt) * 8, h->stream,
                           (const double*)dzx.p, (const double*)dbt.p, Rt, Bt, Bx, layout,
                           (double*)h->fstim.p, (long long)h->nT);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pgl_set_stimulus: ") + hipGetErrorString(e));
    h->Dstim = Dstim;
    h->Ktot = h->Kimp + Dstim;
    h->sep = false;
    h->gibbs_npost = -1;
    invalidate_images(h);
    h->gx_xs = 0;
    return PGL_OK;
}

static int launch_gemm_nt(pgl_handle h, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                          int M, int N, int K)
{
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, h->stream, A, lda, B, ldb, C, ldc, M, N, K);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

// Frame-rate coefficient table of the separable stimulus (k_sepf_fwd): for bin t = q F + o, base = max(F - M, 0),
//   C[t][j][bt] = sum_{tau = 1 .. min(Rt, t)} basis_t[tau-1][bt] * (weight of frame base + j in np.interp at bin t - tau)
// for t < q (M + 1); later bins repeat the rows q M + o.  np.interp between frames f = s / q and f + 1 at bin s
// weighs them (1 - a, a) with a = (s % q) / q (bkgd.py:303-340 interpolates on the dt grid; the clamp behind the
// last frame is an index clamp in the kernels).  Only when dt_stim is an integer multiple of dt and J <= 8, Bt <= 4.
static int build_frame_table(pgl_handle h, const double* basis_t, int Rt, int Bt, double dt_stim)
{
    h->sepf = false;
    const double ratio = dt_stim / h->dt;
    const long long q = llround(ratio);
    if (q < 1 || std::fabs(ratio - (double)q) > 1e-9 * ratio) return PGL_OK;
    const long long M = (Rt + q - 1) / q, J = M + 2;
    if (J > 8 || Bt > 4) return PGL_OK;
    const int Jp = (J <= 5 && Bt <= 3) ? 5 : 8, BTp = (J <= 5 && Bt <= 3) ? 3 : 4;
    const long long rows = q * (M + 1);
    if (rows * Jp * BTp > (1ll << 24)) return PGL_OK;
    std::vector<double> C((size_t)rows * Jp * BTp, 0.0);
    for (long long t = 0; t < rows; ++t) {
        const long long F = t / q, base = std::max<long long>(F - M, 0);
        double* row = C.data() + (size_t)t * Jp * BTp;
        for (long long tau = 1; tau <= std::min<long long>(Rt, t); ++tau) {
            const long long sb = t - tau, f = sb / q;
            const double a = (double)(sb % q) / (double)q;
            for (int bt = 0; bt < Bt; ++bt) {
                const double b = basis_t[(size_t)(tau - 1) * Bt + bt];
                row[(f - base) * BTp + bt] += b * (1.0 - a);
                if (a != 0.0) row[(f + 1 - base) * BTp + bt] += b * a;
            }
        }
    }
    ENSURE(h->sepC, C.size() * 8);
    HIPCHK(hipMemcpyAsync(h->sepC.p, C.data(), C.size() * 8, hipMemcpyHostToDevice, h->stream));
    // A fragments of the fused forward (k_fused7<.., XIO = 2>): for a 16-bin tile that starts at bin t0 (frame F0, base =
    // max(F0 - M, 0)) the coefficient of bin t0 + i on column (j', bt) of [w_t (x) z][base + j'] is
    // C[row(t0 + i)][j' - (base(F_i) - base)][bt]; it depends on t0 only through t0 mod q once F0 >= M (phases of
    // gcd(q, 16) bins), the tiles before that get their own entries.  MFMA A layout: lane l holds row l & 15, column
    // 4 s + (l >> 4) of k-step s.  (J + 1) Bt <= 18 columns = five k-steps; q >= 16 so that a tile spans two frames at most.
    h->sepA_ok = false;
    if (Jp == 5 && BTp == 3 && q >= 16 && (J + 1) * Bt <= 18) {
        int gs = 0;
        while (gs < 4 && (q % (2 << gs)) == 0) ++gs;                   // gcd(q, 16) = 1 << gs
        const long long g = 1ll << gs, NH = (q * M + 15) / 16, NP = q / g;
        std::vector<double> A((size_t)(NH + NP) * 5 * 64, 0.0);
        for (long long e = 0; e < NH + NP; ++e) {
            // a representative first bin: the tile itself in the head, else the first tile behind the head with this phase
            long long t0 = 16 * e;
            if (e >= NH) {
                const long long o0 = (e - NH) * g;
                t0 = -1;
                for (long long c = NH; c < NH + 2 * q; ++c)
                    if ((16 * c) % q == o0) { t0 = 16 * c; break; }
                if (t0 < 0) continue;                                   // (phase never occurs)
            }
            const long long F0 = t0 / q, base0 = std::max<long long>(F0 - M, 0);
            for (int i = 0; i < 16; ++i) {
                const long long t = t0 + i, F = t / q, o = t - F * q, dlt = std::max<long long>(F - M, 0) - base0;
                const double* crow = C.data() + (size_t)((std::min(F, M)) * q + o) * Jp * BTp;
                for (long long j = 0; j < J; ++j)
                    for (int bt = 0; bt < Bt; ++bt) {
                        const long long k = (j + dlt) * 3 + bt;          // column (j', bt), j' = j + dlt <= J
                        A[((size_t)e * 5 + k / 4) * 64 + (k % 4) * 16 + i] = crow[j * BTp + bt];
                    }
            }
        }
        // ... and the A^T fragments of the fused backward (k_fused7<.., 3>): D[(j', bt)][n] += sum_i A[i][(j', bt)] r[i][n] as
        // eight MFMAs (two accumulator tiles of 16 columns x four k-steps of four bins); lane l of fragment 4 mt + ks holds
        // column 16 mt + (l & 15) of bin 4 ks + (l >> 4)
        std::vector<double> AT((size_t)(NH + NP) * 8 * 64, 0.0);
        for (long long e = 0; e < NH + NP; ++e)
            for (int k = 0; k < 18; ++k)
                for (int i = 0; i < 16; ++i) {
                    const double v = A[((size_t)e * 5 + k / 4) * 64 + (k % 4) * 16 + i];
                    AT[((size_t)e * 8 + (k / 16) * 4 + i / 4) * 64 + (i % 4) * 16 + (k % 16)] = v;
                }
        ENSURE(h->sepA, A.size() * 8);
        HIPCHK(hipMemcpyAsync(h->sepA.p, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->stream));
        ENSURE(h->sepAT, AT.size() * 8);
        HIPCHK(hipMemcpyAsync(h->sepAT.p, AT.data(), AT.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->sepNH = (int)NH; h->sepGs = gs;
        h->sepA_ok = true;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->sepq = (int)q; h->sepM = (int)M; h->sepJ = Jp; h->sepBTp = BTp;
    h->sepf = true;
    return PGL_OK;
}

int pgl_set_stimulus_separable(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim,
                               const double* basis_x, int Bx, const double* basis_t, int Rt, int Bt)
{
    if (!h || !stim || !basis_t) return fail(PGL_ERR_ARG, "null argument");
    if (Tstim <= 0 || D <= 0 || Bx <= 0 || Rt <= 0 || Bt <= 0 || !(dt_stim > 0))
        return fail(PGL_ERR_ARG, "bad stimulus description");
    if (!basis_x && Bx != D) return fail(PGL_ERR_ARG, "identity spatial basis needs Bx == D");
    if ((size_t)(2 * Rt + 2 * PGL_SEP_TB) * 8 > 64 * 1024) return fail(PGL_ERR_UNSUPPORTED, "temporal basis too long");
    HIPCHK(hipSetDevice(h->device));
    ENSURE(h->zf, (size_t)Tstim * Bx * 8);
    ENSURE(h->zfT, (size_t)Tstim * Bx * 8);
    ENSURE(h->sbt, (size_t)Rt * Bt * 8);
    HIPCHK(hipMemcpyAsync(h->sbt.p, basis_t, (size_t)Rt * Bt * 8, hipMemcpyHostToDevice, h->stream));
    // z = stim . basis_x at the stimulus frame rate (Tstim, Bx) and its transpose, both by the NT GEMM
    std::vector<double> bxT((size_t)Bx * D, 0.0);
    for (int d = 0; d < D; ++d)
        for (int b = 0; b < Bx; ++b) bxT[(size_t)b * D + d] = basis_x ? basis_x[(size_t)d * Bx + b] : (d == b ? 1.0 : 0.0);
    DevBuf dstim, dbxT;
    int rc = ensure(dstim, (size_t)Tstim * D * 8);
    if (!rc) rc = ensure(dbxT, (size_t)Bx * D * 8);
    if (rc) { release(dstim); release(dbxT); return rc; }
    hipError_t e = hipMemcpyAsync(dstim.p, stim, (size_t)Tstim * D * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dbxT.p, bxT.data(), (size_t)Bx * D * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        rc = launch_gemm_nt(h, (const double*)dstim.p, D, (const double*)dbxT.p, D, (double*)h->zf.p, Bx,
                            (int)Tstim, Bx, D);
        if (!rc) rc = launch_gemm_nt(h, (const double*)dbxT.p, D, (const double*)dstim.p, D, (double*)h->zfT.p,
                                     (int)Tstim, Bx, (int)Tstim, D);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    release(dstim);
    release(dbxT);
    if (rc) return rc;
    if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pgl_set_stimulus_separable: ") + hipGetErrorString(e));
    h->sep = true;
    h->sepBt = Bt; h->sepBx = Bx; h->sepRt = Rt; h->sepT = Tstim; h->sep_dt_stim = dt_stim;
    {
        int rc2 = build_frame_table(h, basis_t, Rt, Bt, dt_stim);
        if (rc2) return rc2;
    }
    h->Dstim = Bt + Bx;                       // layout of a theta row; NOT feature columns of the fused kernels
    h->Ktot = h->Kimp;
    h->gibbs_npost = -1;
    invalidate_images(h);
    h->gx_xs = 0;
    return PGL_OK;
}

// separable stimulus, forward: X[t][j] += I_stim[t, n_lo + j] for the rows of d_theta (npost rows of P)
static int sep_forward(pgl_handle h, const double* d_theta, int npost, double* X, int xs, int64_t t_lo, int64_t t_hi,
                       SepParams& sp)
{
    const int P = 1 + h->Dstim + h->Kimp;
    ENSURE(h->Yf, (size_t)npost * h->sepT * 8);
    int rc = launch_gemm_nt(h, d_theta + 1 + h->sepBt, P, (const double*)h->zf.p, h->sepBx, (double*)h->Yf.p,
                            (int)h->sepT, npost, (int)h->sepT, h->sepBx);
    if (rc) return rc;
    sp.Yf = (const double*)h->Yf.p; sp.basis_t = (const double*)h->sbt.p; sp.theta = d_theta;
    sp.P = P; sp.Bt = h->sepBt; sp.Rt = h->sepRt; sp.npost = npost; sp.xs = xs;
    sp.Tstim = h->sepT; sp.nT = h->nT; sp.t_lo = t_lo; sp.t_hi = t_hi; sp.dt = h->dt; sp.dt_stim = h->sep_dt_stim;
    sp.X = X;
    const int nblk = (int)((t_hi - t_lo + PGL_SEP_TB - 1) / PGL_SEP_TB);
    hipLaunchKernelGGL(k_sep_conv_fwd, dim3(nblk, npost), dim3(256), (size_t)(2 * h->sepRt + PGL_SEP_TB) * 8,
                       h->stream, sp);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

// separable stimulus, backward: X holds r = d ll / d x on [t_lo, t_hi) (zero behind t_hi); writes the w_t and
// w_x columns of d_grad
static int sep_backward(pgl_handle h, const SepParams& sp, double* d_grad)
{
    const int npost = sp.npost, Rt = h->sepRt;
    const int nblk = (int)((sp.t_hi - sp.t_lo + PGL_SEP_TB - 1) / PGL_SEP_TB);
    ENSURE(h->spart, (size_t)nblk * npost * Rt * 8);
    hipLaunchKernelGGL(k_sep_corr, dim3(nblk, npost), dim3(256), (size_t)(Rt + 2 * PGL_SEP_TB) * 8, h->stream, sp,
                       (double*)h->spart.p);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_sep_wt_grad, dim3(npost), dim3(64), 0, h->stream, sp, (const double*)h->spart.p, nblk,
                       d_grad);
    HIPCHK(hipGetLastError());
    const long long s_lo = std::max<long long>(0, sp.t_lo - Rt), nS = sp.t_hi - s_lo;
    ENSURE(h->Qb, (size_t)npost * nS * 8);
    ENSURE(h->Qf, (size_t)npost * h->sepT * 8);
    const int nblk_s = (int)((nS + PGL_SEP_TB - 1) / PGL_SEP_TB);
    hipLaunchKernelGGL(k_sep_conv_bwd, dim3(nblk_s, npost), dim3(256), (size_t)(2 * Rt + PGL_SEP_TB) * 8, h->stream,
                       sp, (double*)h->Qb.p, s_lo, nS);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_sep_interp_T, dim3((unsigned)((h->sepT + 255) / 256), npost), dim3(256), 0, h->stream, sp,
                       (const double*)h->Qb.p, s_lo, nS, (double*)h->Qf.p);
    HIPCHK(hipGetLastError());
    return launch_gemm_nt(h, (const double*)h->Qf.p, (int)h->sepT, (const double*)h->zfT.p, (int)h->sepT,
                          d_grad + 1 + h->sepBt, sp.P, npost, h->sepBx, (int)h->sepT);
}

static int launch_gemm_mfma(pgl_handle h, const double* A, long long sam, long long sak, const double* B,
                            long long sbn, long long sbk, double* C, long long scm, long long scn, int M, int N, int K)
{
    // operands whose m / n index is the contiguous one load whole 128-byte lines per 16-lane group; a k-contiguous
    // operand costs a line per lane (pick A for it: one tile per wave against NT of B)
    const int mb = (M + 15) / 16;
    if ((long long)mb * ((N + 63) / 64) >= 128) {       // 16 x 64 per wave; else 16 x 16 (more workgroups)
        hipLaunchKernelGGL(k_gemm_mfma<4>, dim3(mb, (N + 63) / 64), dim3(512), 0, h->stream, A, sam, sak, B, sbn, sbk,
                           C, scm, scn, M, N, K);
    } else {
        hipLaunchKernelGGL(k_gemm_mfma<1>, dim3(mb, (N + 15) / 16), dim3(512), 0, h->stream, A, sam, sak, B, sbn, sbk,
                           C, scm, scn, M, N, K);
    }
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

static int launch_sepf_any(pgl_handle h, int which, const SepfParams& sp)
{
    if (h->sepJ == 5 && h->sepBTp == 3) launch_sepf<5, 3>(which, sp, h->stream);
    else launch_sepf<8, 4>(which, sp, h->stream);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

// separable stimulus at the frame rate, forward: the slab Xbuf[tile - tile0][post tile][r][lane] = I_stim of the
// npost rows of d_theta on the plan's tile range
static int sepf_forward(pgl_handle h, const Plan& pl, const double* d_theta, SepfParams& sp, bool fused_fwd)
{
    const int P = 1 + h->Dstim + h->Kimp, ldy = pl.nPT * 16;
    ENSURE(h->YfT, (size_t)h->sepT * ldy * 8);
    // z_n[f] = (stim . basis_x)[f, :] . w_x[n]  ->  YfT[f][n]
    int rc = PGL_OK;
    const double* wx = d_theta + 1 + h->sepBt;
    if (gemm_aligned(h->zf.p, h->sepBx) && gemm_aligned(wx, P)) {      // m = f, n = n, k = x (even): both operands k-contiguous
        hipLaunchKernelGGL((k_gemm_kc<4, 1, 8>), dim3((unsigned)((h->sepT + 15) / 16), (unsigned)((pl.npost + 63) / 64)), dim3(512),
                           0, h->stream, (const double*)h->zf.p, (long long)h->sepBx, wx, (long long)P, (double*)h->YfT.p,
                           (long long)ldy, 1LL, (int)h->sepT, pl.npost, h->sepBx);
        HIPCHK(hipGetLastError());
    } else {                                                           // m = n, n = f, k = x
        rc = launch_gemm_mfma(h, wx, P, 1, (const double*)h->zfT.p, 1, h->sepT, (double*)h->YfT.p, 1, ldy, pl.npost,
                              (int)h->sepT, h->sepBx);
    }
    if (rc) return rc;
    sp.Ctab = (const double*)h->sepC.p; sp.YfT = (const double*)h->YfT.p; sp.theta = d_theta;
    sp.X = (double*)h->Xbuf.p; sp.Hb = nullptr; sp.wpart = nullptr; sp.QvT = nullptr; sp.grad = nullptr;
    sp.D = nullptr; sp.B0 = sp.B1 = 0; sp.SL = 0; sp.tilesPerChunk = 0;
    sp.P = P; sp.Bt = h->sepBt; sp.M = h->sepM; sp.q = h->sepq; sp.npost = pl.npost; sp.nPT = pl.nPT; sp.ldy = ldy;
    sp.tile0 = pl.tile0; sp.nTiles = pl.nTiles; sp.Tstim = h->sepT;
    const long long tb = (long long)pl.tile0 * 16, te = tb + (long long)pl.nTiles * 16;
    sp.F0 = tb / h->sepq; sp.F1 = (te - 1) / h->sepq;
    if (fused_fwd) return PGL_OK;               // the stimulus current is five k-steps of the fused kernel (k_fused7<.., 2>)
    return launch_sepf_any(h, 0, sp);
}

// backward: the slab holds r = d ll / d x (fused_bwd: the fused kernel left the per-base pieces sp.D instead); writes the
// w_x columns of d_grad and the w_t columns -- fused_bwd: their block partials sp.wpart, nwt of them, which the trailing
// blocks of k_finalize sum
static int sepf_backward(pgl_handle h, SepfParams& sp, double* d_grad, bool fused_bwd = false, int* nwt = nullptr)
{
    const long long nF = sp.F1 - sp.F0 + 1;
    ENSURE(h->QvT, (size_t)h->sepT * sp.ldy * 8);
    int rc = PGL_OK;
    if (fused_bwd) {
        const int nblk = (int)((h->sepT + 15) / 16);
        ENSURE(h->wpart, (size_t)nblk * 3 * sp.ldy * 8);
        sp.wpart = (double*)h->wpart.p; sp.QvT = (double*)h->QvT.p; sp.grad = d_grad;
        hipLaunchKernelGGL(k_sepf_finish_d, dim3((unsigned)nblk), dim3(1024), 0, h->stream, sp);
        HIPCHK(hipGetLastError());
        if (nwt) *nwt = nblk;
    } else {
        ENSURE(h->Hb, (size_t)nF * h->sepJ * sp.ldy * 8);
        ENSURE(h->wpart, (size_t)nF * h->sepBTp * sp.ldy * 8);
        sp.Hb = (double*)h->Hb.p; sp.wpart = (double*)h->wpart.p; sp.QvT = (double*)h->QvT.p; sp.grad = d_grad;
        rc = launch_sepf_any(h, 1, sp);
        if (!rc) rc = launch_sepf_any(h, 2, sp);
    }
    if (rc) return rc;
    // d ll / d w_x[n][x] = sum_f (stim . basis_x)[f][x] QvT[f][n]
    if (gemm_aligned(h->zfT.p, h->sepT)) {                             // m = x (rows of zfT: f contiguous), n = n, k = f
        hipLaunchKernelGGL((k_gemm_kc<1, 0, 16>), dim3((unsigned)((h->sepBx + 15) / 16), (unsigned)((sp.npost + 15) / 16)), dim3(512),
                           0, h->stream, (const double*)h->zfT.p, (long long)h->sepT, (const double*)h->QvT.p, (long long)sp.ldy,
                           d_grad + 1 + h->sepBt, 1LL, (long long)sp.P, h->sepBx, sp.npost, (int)h->sepT);
        HIPCHK(hipGetLastError());
        return PGL_OK;
    }
    return launch_gemm_mfma(h, (const double*)h->zf.p, 1, h->sepBx, (const double*)h->QvT.p, 1, sp.ldy,
                            d_grad + 1 + h->sepBt, 1, sp.P, h->sepBx, sp.npost, (int)h->sepT);
}

int pgl_sta(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim, int L,
            const int* neurons, int n_sel, double* A_out)
{
    if (!h || !stim) return fail(PGL_ERR_ARG, "null argument");
    if (!h->have_spikes) return fail(PGL_ERR_STATE, "pgl_set_spikes_* has not been called");
    if (Tstim <= 0 || D <= 0 || L <= 0 || !(dt_stim > 0)) return fail(PGL_ERR_ARG, "bad stimulus description");
    const int nSel = neurons ? n_sel : h->N;
    if (nSel <= 0 || nSel > 65535) return fail(PGL_ERR_ARG, "bad neuron selection");
    std::vector<int> eoff((size_t)2 * nSel);
    for (int i = 0; i < nSel; ++i) {
        const int n = neurons ? neurons[i] : i;
        if (n < 0 || n >= h->N) return fail(PGL_ERR_ARG, "neuron index out of range");
        eoff[2 * i] = h->h_ptr[n];
        eoff[2 * i + 1] = h->h_ptr[n + 1];
    }
    const long long LD = (long long)L * D;
    const long long xb = (LD + 255) / 256;
    if (xb > 0x7fffffffLL) return fail(PGL_ERR_UNSUPPORTED, "L*D too large");
    // enough blocks to fill the chip when the output is small: split each neuron's events
    int echunks = (int)std::min<long long>(64, std::max<long long>(1, 4096 / (xb * nSel)));
    HIPCHK(hipSetDevice(h->device));
    DevBuf dstim, distim, deoff, dpart, dA;
    auto cleanup = [&]() { release(dstim); release(distim); release(deoff); release(dpart); release(dA); };
    // A_out == NULL: the averages stay on the device for pgl_leading_singular_pairs (157 MB for 64 neurons x 300 lags x 1024
    // pixels would otherwise cross PCIe twice)
    auto keep = [&]() {
        release(h->staA);
        h->staA = dA;
        dA = DevBuf();
        h->sta_n = nSel; h->sta_L = L; h->sta_D = D;
    };
    // frame-rate form (k_sta_weights + one thin GEMM): dt_stim an integer multiple of dt, an even number of frames (the
    // GEMM's 16-byte loads), a weight matrix of at most 2 GB; dev option 94 = 7: the bin-rate form
    const double ratio = dt_stim / h->dt;
    const long long q = llround(ratio);
    const bool frame_rate = q >= 1 && std::fabs(ratio - (double)q) <= 1e-9 * ratio && (Tstim % 2) == 0 && Tstim >= 2 &&
                            (long long)nSel * L * Tstim <= (1ll << 28) && h->opt_sepf != 7;
    int rc = ensure(dstim, (size_t)Tstim * D * 8);
    if (!rc && !frame_rate) rc = ensure(distim, (size_t)h->nT * D * 8);
    if (!rc) rc = ensure(deoff, eoff.size() * sizeof(int));
    if (!rc && !frame_rate) rc = ensure(dpart, (size_t)echunks * nSel * LD * 8);
    if (!rc) rc = ensure(dA, (size_t)nSel * LD * 8);
    if (rc) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(dstim.p, stim, (size_t)Tstim * D * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(deoff.p, eoff.data(), eoff.size() * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (frame_rate && e == hipSuccess) {
        DevBuf dW, dscale;
        std::vector<double> scale((size_t)nSel);
        for (int i = 0; i < nSel; ++i) {
            double cnt = 0.0;
            for (int j = eoff[2 * i]; j < eoff[2 * i + 1]; ++j) cnt += (double)h->h_ev[j].y;
            scale[i] = (h->dt / dt_stim) / cnt;                   // 0 spikes: inf -> NaN rows, like the reference's 0 / 0
        }
        rc = ensure(dW, (size_t)nSel * L * Tstim * 8);
        if (!rc) rc = ensure(dscale, (size_t)nSel * 8);
        if (rc) { release(dW); release(dscale); cleanup(); return rc; }
        e = hipMemcpyAsync(dscale.p, scale.data(), (size_t)nSel * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_sta_weights, dim3((unsigned)((Tstim + 255) / 256), (unsigned)L, (unsigned)nSel), dim3(256), 0,
                               h->stream, (const int2*)h->spk.p, (const int*)deoff.p, (const double*)dscale.p, L,
                               (long long)Tstim, (int)q, (double*)dW.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {                                    // A (nSel L, D) = Wt (nSel L, Tstim) . stim (Tstim, D)
            const int Mrows = nSel * L;
            hipLaunchKernelGGL((k_gemm_kc<1, 0, 16>), dim3((unsigned)((Mrows + 15) / 16), (unsigned)((D + 15) / 16)), dim3(512), 0,
                               h->stream, (const double*)dW.p, (long long)Tstim, (const double*)dstim.p, (long long)D,
                               (double*)dA.p, (long long)D, 1LL, Mrows, D, (int)Tstim);
            e = hipGetLastError();
        }
        if (e == hipSuccess && A_out)
            e = hipMemcpyAsync(A_out, dA.p, (size_t)nSel * LD * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        release(dW); release(dscale);
        if (e == hipSuccess && !A_out) keep();
        cleanup();
        if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pgl_sta: ") + hipGetErrorString(e));
        return PGL_OK;
    }
    if (e == hipSuccess) {
        const long long total = (long long)h->nT * D;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 65535);
        hipLaunchKernelGGL(k_stim_project, dim3(blocks), dim3(256), 0, h->stream, (const double*)dstim.p,
                           (long long)Tstim, D, dt_stim, h->dt, (const double*)nullptr, D,
                           (double*)distim.p, (long long)h->nT);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sta, dim3((unsigned)xb, (unsigned)nSel, (unsigned)echunks), dim3(256), 0,
                           h->stream, (const int2*)h->spk.p, (const int*)deoff.p, (const double*)distim.p,
                           D, L, (double*)dpart.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sta_finish, dim3((unsigned)xb, (unsigned)nSel), dim3(256), 0, h->stream,
                           (const double*)dpart.p, (const int2*)h->spk.p, (const int*)deoff.p, LD, echunks,
                           h->dt / dt_stim, (double*)dA.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess && A_out)
        e = hipMemcpyAsync(A_out, dA.p, (size_t)nSel * LD * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && !A_out) keep();
    cleanup();
    if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pgl_sta: ") + hipGetErrorString(e));
    return PGL_OK;
}

int pgl_get_stim_features(pgl_handle h, double* fstim_out)
{
    if (h && h->sep) return fail(PGL_ERR_STATE, "a separable stimulus has no dense feature matrix");
    if (!h || !fstim_out) return fail(PGL_ERR_ARG, "null argument");
    if (h->Dstim <= 0) return fail(PGL_ERR_STATE, "no stimulus features on the device");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(fstim_out, h->fstim.p, (size_t)h->nT * h->Dstim * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

static int check_ready(pgl_handle h)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    if (!h->have_spikes) return fail(PGL_ERR_STATE, "pgl_set_spikes_* has not been called");
    if (!h->have_basis) return fail(PGL_ERR_STATE, "pgl_set_basis has not been called");
    return PGL_OK;
}

static void fill_params(pgl_handle h, const Plan& pl, const Slice& sl, int n_lo, bool want_grad,
                        int mode, FusedParams& fp)
{
    fp.nT = h->nT; fp.N = sl.Ns; fp.B = h->B; fp.R = h->Rk; fp.nlin = h->nlin;
    fp.Dstim = sl.Ds; fp.Kimp = sl.Ns * h->B; fp.Ktot = fp.Kimp + sl.Ds; fp.dt = h->dt;
    fp.Nall = h->N; fp.np0 = sl.np0; fp.DsAll = h->Dstim; fp.ds0 = sl.ds0;
    fp.mode = mode; fp.Xbuf = (double*)h->Xbuf.p; fp.xstride = pl.nPT * 16;
    fp.sepA = nullptr; fp.sepZ = nullptr; fp.sepTheta = nullptr; fp.sepT = 0;
    fp.sepLdy = fp.sepQ = fp.sepM = fp.sepNH = fp.sepG = fp.sepBt = 0;
    fp.sepAT = nullptr; fp.sepD = nullptr; fp.sepB0 = 0; fp.sepSL = 0;
    fp.spk = (const int2*)h->spk.p; fp.wlo = (const int*)h->wlo.p; fp.whi = (const int*)h->whi.p;
    fp.S = (const uint8_t*)h->S.p; fp.fstim = (const double*)h->fstim.p; fp.phi = (const double*)h->phi.p;
    fp.Wfrag = (const double*)h->Wfrag.p; fp.bias = (const double*)h->bias.p;
    fp.n_lo = n_lo; fp.npost = pl.npost; fp.nPT = pl.nPT;
    fp.nT16 = h->nT16; fp.tilesPerChunk = pl.tilesPerChunk; fp.nChunks = pl.nChunks; fp.nTiles = pl.nTiles;
    fp.pb_major = pl.pb_major;
    fp.rsf = pl.rsf;
    fp.RP = pl.RP;
    fp.Gpart = (double*)h->Gpart.p; fp.llpart = (double*)h->llpart.p; fp.gbpart = (double*)h->gbpart.p;
    fp.tile0 = pl.tile0;
    fp.t_hi = h->t_hi;
    fp.want_grad = want_grad ? 1 : 0;
    fp.dbg = h->opt_dbg;
    fp.Fimg = (const unsigned char*)h->imgs[h->img_cur].buf.p;
    fp.img_tile0 = h->imgs[h->img_cur].tile0;
    fp.pidx = h->cur_pidx;
    fp.theta = nullptr; fp.Weff = nullptr; fp.P = 1 + h->Dstim + h->Kimp;
    fp.epi64 = h->opt_epi64 ? 2 : 0;
}

// kernels with register-resident Wmat fragments read theta / Weff themselves (no k_prep_w launch)
static bool plan_reads_theta(const Plan& pl)
{
    return pl.version == 6 || (pl.version == 7 && pl.KS <= PGL_WREG_MAX);
}

static int launch_prep(pgl_handle h, const Plan& pl, const Slice& sl, int n_lo, const double* d_theta,
                       const double* d_Weff)
{
    ENSURE(h->Wfrag, (size_t)pl.nPT * pl.KS * 64 * 8);
    ENSURE(h->bias, (size_t)pl.nPT * 16 * 8);
    const long long total = (long long)pl.nPT * pl.KS * 64;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(k_prep_w, dim3(blocks), dim3(256), 0, h->stream, d_theta, d_Weff,
                       (double*)h->Wfrag.p, (double*)h->bias.p, sl.Ns, h->B, sl.Ds, sl.Ns * h->B,
                       sl.Ns * h->B + sl.Ds, pl.KS, n_lo, pl.npost, pl.nPT, 1,
                       h->N, sl.np0, h->Dstim, sl.ds0, h->cur_pidx);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

static int launch_finalize_grad(pgl_handle h, const Plan& pl, const Slice& sl, int n_lo,
                                const double* d_Weff, double* d_ll, double* d_grad, bool with_ll = false,
                                int kt0 = 0, int nkt = -1, hipStream_t stream = nullptr,
                                const double* wtpart = nullptr, int nwt = 0, int ldy = 0)
{
    if (nkt < 0) nkt = pl.KT;
    if (!stream) stream = h->stream;
    const long long nfrag = (long long)pl.nPT * nkt * 256;
    // one block per 64-element fragment, its waves share the chunks (>= 32 chunks = 16 KB per wave, at most 8 waves)
    // (measured round 3, tools/step_bench.py + rocprofv3 timeline: the reduction of a SMALL population -- 20 blocks at
    //  C1 -- takes 20 us whatever the number of waves per fragment or loads in flight per lane (8 -> 16 of either):
    //  it is not the latency chain of its loads; 1024-thread blocks cost the 1/8-recording shard of C3 +25 us)
    int nwf = h->opt_finw > 0 ? h->opt_finw : std::max(4, std::min(8, pl.nChunks / 32));
    if (with_ll) nwf = std::max(nwf, 4);                  // the ll blocks reduce with 256 threads (pgl_reduce_ll)
    int blocks = (int)((nfrag + 63) / 64);
    if (with_ll) blocks += pl.npost;                      // trailing blocks (one per neuron) reduce ll and d ll / d bias
    if (with_ll && wtpart) blocks += h->sepBt;            // ... and the w_t partials of the fused stimulus backward
    hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(64 * nwf), 0, stream, (const double*)h->Gpart.p,
                       (const double*)h->llpart.p, (const double*)h->gbpart.p, d_Weff, d_ll, d_grad,
                       sl.Ns, h->B, sl.Ds, sl.Ns * h->B, sl.Ns * h->B + sl.Ds, pl.KT, n_lo, pl.npost,
                       pl.nPT, pl.nChunks, h->N, sl.np0, h->Dstim, sl.ds0, with_ll ? pl.KSPLIT : 0, kt0,
                       nkt, h->cur_pidx, wtpart, nwt, ldy, h->sepBt);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

static hipError_t launch_any(const Plan& pl, const FusedParams& fp, hipStream_t s)
{
    return launch_fused2(pl, fp, s);
}

// Resident feature tiles (k_build_fimg) of the evaluated time range [tile0, tile0 + ntiles) for the
// column split (ktl, kth): built on first use and after every change of spikes / basis / stimulus
// features / time range.  A time-sharded rank (pgl_set_time_range) therefore builds and keeps only its
// own 1/G of the recording.
static int ensure_feature_images(pgl_handle h, int ktl, int kth, int tile0, int ntiles, const Slice* sl = nullptr,
                                 int slice_no = 0, int blk = 0)
{
    // kth == 0: one image per tile holding all ktl k-tiles (k_fused6; blk: as 2 KB blocks, k_fused8); else the L / H pair
    // of k_fused5.  sl: the images of ONE column slice of a wide population (key carries the slice number)
    const int key = (sl ? (slice_no + 1) << 16 : 0) | (blk ? img_key6(ktl, blk) : ktl << 8 | kth);
    int slot = find_img(h, key, tile0, ntiles);
    if (slot >= 0) {
        h->img_cur = slot;
        h->imgs[slot].stamp = ++h->img_clock;
        return PGL_OK;
    }
    // an empty / stale slot first, else the least recently used one
    slot = 0;
    for (int i = 0; i < pgl_context::NIMG; ++i)
        if (h->imgs[i].key == 0) { slot = i; break; }
        else if (h->imgs[i].stamp < h->imgs[slot].stamp) slot = i;
    pgl_context::ImgSlot& im = h->imgs[slot];
    im.key = 0;
    const size_t bytes = (size_t)ntiles * (kth ? img_pair_bytes(ktl, kth) : img_bytes6(ktl, blk));
    if (bytes > im.buf.cap) {
        // make room before allocating: the other slots' stale buffers go first
        for (int i = 0; i < pgl_context::NIMG; ++i)
            if (i != slot && h->imgs[i].key == 0) release(h->imgs[i].buf);
    }
    ENSURE(im.buf, bytes);
    dim3 grid((unsigned)ntiles, kth ? 2 : 1);
    hipLaunchKernelGGL(k_build_fimg, grid, dim3(256), (size_t)h->B * h->Rk * 8, h->stream,
                       (const int2*)h->spk.p, (const int*)h->wlo.p, (const int*)h->whi.p,
                       (const double*)h->phi.p, (const double*)h->fstim.p, (long long)h->nT, sl ? sl->Ns : h->N, h->B,
                       h->Rk, sl ? sl->Ds : (h->sep ? 0 : h->Dstim), ktl, kth, tile0, (unsigned char*)im.buf.p,
                       h->N, sl ? sl->np0 : 0, h->Dstim, sl ? sl->ds0 : 0, blk);
    HIPCHK(hipGetLastError());
    im.key = key;
    im.tile0 = tile0;
    im.ntiles = ntiles;
    im.stamp = ++h->img_clock;
    h->img_cur = slot;
    return PGL_OK;
}

// The path an evaluation of neurons [n_lo, n_hi) (or of a list of n_hi - n_lo neurons) takes and its launch plans:
//   sepf   -- separable stimulus at the frame rate: impulse columns on resident tiles (k_fused7 with the slab-input form up to
//             four post tiles, the two-pass kernel from five on or when the feature row is too long for k_fused7), the
//             stimulus current / its gradients by k_sepf_*; neuron lists are fine there (the stimulus kernels work on the
//             listed rows, the fused kernel maps rows to neurons).  A kernel forced by PGL_OPT_KERNEL other than 7 / 4 keeps
//             the stimulus on the 3-phase path it asks for;
//   sliced -- the 3-phase path (more than one slice of feature columns, or a separable stimulus by the tap-rate kernels).
// One function for enqueue_ll_grad, pgl_info and the dry run of the dispatch (pgl_plan_kernels).
static int select_plans(const pgl_context* h, int n_lo, int n_hi, std::vector<Slice>& slices, std::vector<Plan>& plans,
                        bool& sepf, bool& sliced, bool* wide_out = nullptr)
{
    slices = make_slices(h, wide_out != nullptr);
    plans.assign(slices.size(), Plan());
    if (wide_out) *wide_out = false;
    // wide -- more than one slice of feature columns (N > 128 or more than 640 columns) with every slice on the resident-tile
    //         two-pass kernel: forward-only passes of the first slices add their currents in the slab, the last slice runs
    //         pass 1 from the slab, then the pass-2 kernels take the gradients of all column parts from the residuals.
    //         Needs rows of 7 .. 40 k-tiles in every slice (equal-width slices see to that from B = 2 on), no separable
    //         stimulus, f64 features, and the memory for one image set per slice; else the in-kernel-feature path below.
    if (wide_out && slices.size() > 1 && slices.size() <= (size_t)pgl_context::NIMG && !h->sep && !h->opt_f32 &&
        (h->opt_kernel == 0 || h->opt_kernel == 4)) {
        bool ok = true;
        size_t bytes = 0;
        for (const Slice& sl : slices) {
            const int need = (sl.Ns * h->B + sl.Ds + 15) / 16;
            int ktl = 0, kth = 0;
            if (sl.Ns <= 0 || need < 7 || need > 40 || !pick_pair(need, ktl, kth)) { ok = false; break; }
            const int tile0 = (int)(h->t_lo / 16), nTiles = (int)((h->t_hi + 15) / 16) - tile0;
            bytes += (size_t)nTiles * img_pair_bytes(ktl, kth);
        }
        if (ok && h->opt_kernel == 0) {
            // the image sets that are not resident yet must fit (with the residual slab) into 90 % of the free memory
            size_t free_b = 0, total_b = 0, have = 0;
            for (int i = 0; i < pgl_context::NIMG; ++i)
                if (h->imgs[i].key >> 16) have += h->imgs[i].buf.cap;
            if (bytes > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes - have > free_b / 10 * 9) ok = false;
        }
        if (ok) {
            for (size_t i = 0; i < slices.size(); ++i) {
                int rc = make_plan(h, n_lo, n_hi, slices[i], plans[i], false, false, true);
                if (rc) return rc;
                if (plans[i].version != 5) ok = false;
            }
        }
        if (ok) {
            sepf = false;
            sliced = true;
            *wide_out = true;
            return PGL_OK;
        }
    }
    if (wide_out) {
        slices = make_slices(h);
        plans.assign(slices.size(), Plan());
    }
    sepf = h->sep && h->sepf && h->opt_sepf != 2 && slices.size() == 1 && !h->opt_f32 &&
           (h->opt_kernel == 0 || h->opt_kernel == 7 || h->opt_kernel == 4);
    if (sepf) {
        int rc = make_plan(h, n_lo, n_hi, slices[0], plans[0], true, true);
        if (rc) return rc;
        sepf = (plans[0].version == 7 && plans[0].nw7 == 4) || (plans[0].version == 5 && plans[0].ktl >= 5);
    }
    for (size_t i = 0; i < slices.size() && !sepf; ++i) {
        int rc = make_plan(h, n_lo, n_hi, slices[i], plans[i], slices.size() == 1 && !h->sep);
        if (rc) return rc;
    }
    sliced = !sepf && (slices.size() > 1 || h->sep);     // else the separable stimulus rides on the 3-phase path
    return PGL_OK;
}

// Enqueue one evaluation on the handle's stream.  All pointers are device pointers.
//  * one slice (N <= 128 and N*B + Dstim <= 640): prep + fused kernel + finalize;
//  * otherwise the 3-phase path: per slice a forward-only launch accumulating the currents in
//    Xbuf (nT x 16*nPT doubles), one elementwise pass Xbuf -> (ll, r), per slice a backward-only
//    launch.  Same kernels, the F tile is simply generated twice per slice.
static int enqueue_ll_grad(pgl_handle h, int n_lo, int n_hi, const double* d_theta,
                           const double* d_Weff, double* d_ll, double* d_grad)
{
    std::vector<Slice> slices;
    std::vector<Plan> plans;
    bool sepf = false, sliced = false, wide = false;
    {
        int rc = select_plans(h, n_lo, n_hi, slices, plans, sepf, sliced, &wide);
        if (rc) return rc;
    }
    size_t maxG = 0, maxLL = 0;
    for (const Plan& pl : plans) {
        maxG = std::max(maxG, (size_t)pl.nChunks * pl.nPT * pl.KT * 256 * 8);
        maxLL = std::max(maxLL, (size_t)pl.nChunks * pl.nPT * pl.KSPLIT * 64 * 8);
    }
    ENSURE(h->llpart, maxLL);
    ENSURE(h->gbpart, maxLL);
    if (d_grad) ENSURE(h->Gpart, maxG);
    const int P = 1 + h->Dstim + h->Kimp;

    // HIP events of this evaluation (pgl_last_timing / pgl_timing_summary): every PGL_OPT_TIMING-th call only -- an
    // event between two kernels of a stream costs ~6 us of GPU idle time (the next dispatch waits for the marker:
    // rocprofv3 timeline, pass 1 -> pass 2 without an event between them start back to back)
    const bool rec = h->opt_timing > 0 && (h->call_no++ % h->opt_timing) == 0;
    if (rec) {
        h->ev = h->evr[h->ev_launches % pgl_context::NEV];
        HIPCHK(hipEventRecord(h->ev[0], h->stream));
    }
    if (sepf) {
        const Plan& pl = plans[0];
        const bool direct = plan_reads_theta(pl);
        int rc = direct ? PGL_OK : launch_prep(h, pl, slices[0], n_lo, d_theta, d_Weff);
        if (rc) return rc;
        ENSURE(h->Xbuf, (size_t)pl.nTiles * pl.nPT * 256 * 8);
        rc = (pl.version == 5) ? ensure_feature_images(h, pl.ktl, pl.kth, pl.tile0, pl.nTiles)
                               : ensure_feature_images(h, pl.KT, 0, pl.tile0, pl.nTiles, nullptr, 0, (pl.version == 6 && pl.sb6 == 2) ? 1 + pl.img32 : 0);
        if (rc) return rc;
        if (rec) HIPCHK(hipEventRecord(h->ev[1], h->stream));
        SepfParams sp;
        // up to four post tiles with the (5, 3) table: the stimulus current rides in the forward contraction
        // (beyond 13 k-tiles the five extra k-steps do not fit the registers: k_fused7<16, 4, 2> spills -- slab form there)
        const bool fused_fwd = pl.version == 7 && pl.KT <= 13 && h->sepA_ok && h->opt_sepf != 3;
        // ... and its backward too (k_fused7<.., 3>: no residual slab, no k_sepf_bwd) up to 12 k-tiles (registers);
        // dev option 94 = 4 keeps the slab form
        const bool fused_bwd = fused_fwd && d_grad && pl.KT <= 12 && h->opt_sepf != 4;
        rc = sepf_forward(h, pl, d_theta, sp, fused_fwd);
        if (rc) return rc;
        if (fused_bwd) {
            // pieces of the stimulus backward: one per frame base and chunk that holds tiles of it
            const long long q = h->sepq, M = h->sepM;
            const long long tE = (long long)pl.tile0 + pl.nTiles - 1;
            sp.B0 = std::max<long long>(((long long)pl.tile0 * 16) / q - M, 0);
            sp.B1 = std::max<long long>((tE * 16) / q - M, 0);
            const long long span = ((M + 1) * q + 15) / 16 + 1;               // tiles of base 0, the longest
            sp.SL = (int)(span / pl.tilesPerChunk) + 2;
            sp.tilesPerChunk = pl.tilesPerChunk;
            ENSURE(h->sepD, (size_t)(sp.B1 - sp.B0 + 1) * sp.SL * pl.nPT * 320 * 8);
            sp.D = (const double*)h->sepD.p;
        }
        FusedParams fp;
        fill_params(h, pl, slices[0], n_lo, d_grad != nullptr, 0, fp);
        if (direct) {
            fp.theta = d_theta;
            fp.Weff = d_Weff;
        }
        if (fused_fwd) {
            fp.sepA = (const double*)h->sepA.p; fp.sepZ = sp.YfT; fp.sepTheta = d_theta; fp.sepT = h->sepT;
            fp.sepLdy = sp.ldy; fp.sepQ = h->sepq; fp.sepM = h->sepM; fp.sepNH = h->sepNH; fp.sepG = h->sepGs;
            fp.sepBt = h->sepBt;
        }
        if (fused_bwd) {
            fp.sepAT = (const double*)h->sepAT.p; fp.sepD = (double*)h->sepD.p; fp.sepB0 = sp.B0; fp.sepSL = sp.SL;
        }
        hipError_t e = hipSuccess;
        if (pl.version == 5) {                   // pass 1 (slab in, residuals out) and, for the gradient, pass 2
            e = launch_fused5_xin(pl, fp, h->stream, 1);
            if (e == hipSuccess && d_grad) e = launch_fused5_xin(pl, fp, h->stream, 2);
        } else {
            e = launch_fused7_xio(pl, fp, h->stream, fused_bwd ? 3 : (fused_fwd ? 2 : 1));
        }
        if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("fused launch: ") + hipGetErrorString(e));
        if (d_grad) {
            int nwt = 0;
            rc = sepf_backward(h, sp, d_grad, fused_bwd, &nwt);
            if (rc) return rc;
            if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
            rc = launch_finalize_grad(h, pl, slices[0], n_lo, d_Weff, d_ll, d_grad, true, 0, -1, nullptr,
                                      fused_bwd ? (const double*)h->wpart.p : nullptr, nwt, sp.ldy);
            if (rc) return rc;
        } else {
            if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
            hipLaunchKernelGGL(k_finalize_ll, dim3(pl.npost), dim3(256), 0, h->stream,
                               (const double*)h->llpart.p, (const double*)h->gbpart.p, d_ll, d_grad, P,
                               pl.npost, pl.nPT, pl.nChunks, pl.KSPLIT);
            HIPCHK(hipGetLastError());
        }
    } else if (!sliced) {
        const Plan& pl = plans[0];
        const bool direct = plan_reads_theta(pl);
        int rc = direct ? PGL_OK : launch_prep(h, pl, slices[0], n_lo, d_theta, d_Weff);
        if (rc) return rc;
        if ((pl.version == 4 || pl.version == 5) && d_grad) {   // residual slab of the two-pass kernels
            const size_t need = (size_t)pl.nTiles * pl.nPT * 256 * 8;
            const bool fresh = need > h->Xbuf.cap || !h->Xbuf.p;
            ENSURE(h->Xbuf, need);
            // first touch of a fresh allocation costs ~8 % of an evaluation: pay it here, once
            if (fresh) HIPCHK(hipMemsetAsync(h->Xbuf.p, 0, need, h->stream));
        }
        if (pl.version == 5) {
            rc = ensure_feature_images(h, pl.ktl, pl.kth, pl.tile0, pl.nTiles);
            if (rc) return rc;
        }
        if (pl.version == 6 || pl.version == 7) {
            rc = ensure_feature_images(h, pl.KT, 0, pl.tile0, pl.nTiles, nullptr, 0, (pl.version == 6 && pl.sb6 == 2) ? 1 + pl.img32 : 0);
            if (rc) return rc;
        }
        FusedParams fp;
        fill_params(h, pl, slices[0], n_lo, d_grad != nullptr, 0, fp);
        if (direct) {
            fp.theta = d_theta;
            fp.Weff = d_Weff;
        }
        if (rec) HIPCHK(hipEventRecord(h->ev[1], h->stream));
        if (pl.version == 5 && d_grad) {
            // pass 1 | pass 2 | one reduction of all per-chunk partials.  (Until round 3 the first G half was reduced on a
            // side stream "beside" pass 2: the dispatch timeline shows that reduction finishing ~14 us AFTER pass 2 --
            // whatever the stream priority -- and holding up the second one, 42 us in all against 30 us for a single
            // streaming pass over both halves; at full size the difference is below the noise.)
            hipError_t e = launch_fused5(pl, fp, h->stream, 1);
            if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pass 1 launch: ") + hipGetErrorString(e));
            e = launch_fused5(pl, fp, h->stream, 2);
            if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("pass 2 launch: ") + hipGetErrorString(e));
            if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
            rc = launch_finalize_grad(h, pl, slices[0], n_lo, d_Weff, d_ll, d_grad, true);
            if (rc) return rc;
            if (rec) HIPCHK(hipEventRecord(h->ev[3], h->stream));
            h->timing_valid = rec;
            if (rec) ++h->ev_launches;
            return PGL_OK;
        }
        hipError_t e = launch_any(pl, fp, h->stream);
        if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("fused launch: ") + hipGetErrorString(e));
        if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
        if (d_grad) {                       // one launch: G reduction + (trailing blocks) ll reduction
            rc = launch_finalize_grad(h, pl, slices[0], n_lo, d_Weff, d_ll, d_grad, true);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_finalize_ll, dim3(pl.npost), dim3(256), 0, h->stream,
                               (const double*)h->llpart.p, (const double*)h->gbpart.p, d_ll, d_grad, P,
                               pl.npost, pl.nPT, pl.nChunks, pl.KSPLIT);
            HIPCHK(hipGetLastError());
        }
    } else if (wide) {
        // a wide population on resident tiles: one image set, one set of Wmat fragments and up to three launches of the two-pass
        // kernel per column slice; currents and residuals travel in the slab
        const Plan& p0 = plans[0];
        const size_t S = slices.size();
        ENSURE(h->Xbuf, (size_t)p0.nTiles * p0.nPT * 256 * 8);
        std::vector<int> slot(S);
        for (size_t i = 0; i < S; ++i) {
            int rc = ensure_feature_images(h, plans[i].ktl, plans[i].kth, p0.tile0, p0.nTiles, &slices[i], (int)i);
            if (rc) return rc;
            slot[i] = h->img_cur;
        }
        if (rec) HIPCHK(hipEventRecord(h->ev[1], h->stream));
        auto params = [&](const size_t i, FusedParams& fp) {
            h->img_cur = slot[i];
            fill_params(h, plans[i], slices[i], n_lo, d_grad != nullptr, 0, fp);
        };
        for (size_t i = 0; i < S; ++i) {                       // forward: the slices add up in the slab; the last one closes
            int rc = launch_prep(h, plans[i], slices[i], n_lo, d_theta, d_Weff);
            if (rc) return rc;
            FusedParams fp;
            params(i, fp);
            const bool last = i + 1 == S;
            hipError_t e = launch_fused5_wide(plans[i], fp, h->stream, last ? 2 : (i == 0 ? 0 : 1));
            if (e == hipSuccess && last && d_grad) e = launch_fused5_wide(plans[i], fp, h->stream, 3);
            if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("wide launch: ") + hipGetErrorString(e));
        }
        if (d_grad) {
            int rc = launch_finalize_grad(h, plans[S - 1], slices[S - 1], n_lo, d_Weff, d_ll, d_grad, true);
            if (rc) return rc;
            for (size_t i = 0; i + 1 < S; ++i) {               // the gradients of the earlier slices from the residuals
                FusedParams fp;
                params(i, fp);
                hipError_t e = launch_fused5_wide(plans[i], fp, h->stream, 4);
                if (e == hipSuccess) e = launch_fused5_wide(plans[i], fp, h->stream, 3);
                if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("wide backward launch: ") + hipGetErrorString(e));
                rc = launch_finalize_grad(h, plans[i], slices[i], n_lo, d_Weff, d_ll, d_grad);
                if (rc) return rc;
            }
        } else {
            const Plan& pl = plans[S - 1];
            hipLaunchKernelGGL(k_finalize_ll, dim3(pl.npost), dim3(256), 0, h->stream,
                               (const double*)h->llpart.p, (const double*)h->gbpart.p, d_ll, d_grad, P,
                               pl.npost, pl.nPT, pl.nChunks, pl.KSPLIT);
            HIPCHK(hipGetLastError());
        }
        if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
    } else {
        const Plan& p0 = plans[0];
        const int xs = p0.nPT * 16;
        const long long row0 = (long long)p0.tile0 * 16;
        const long long row1 = std::min<long long>(h->nT, (long long)(p0.tile0 + p0.nTiles) * 16);
        ENSURE(h->Xbuf, (size_t)h->nT * xs * 8);
        HIPCHK(hipMemsetAsync((double*)h->Xbuf.p + row0 * xs, 0, (size_t)(row1 - row0) * xs * 8, h->stream));
        if (rec) HIPCHK(hipEventRecord(h->ev[1], h->stream));
        SepParams sp;
        if (h->sep) {                                          // phase 0: X = I_stim (separable stimulus)
            if (h->cur_pidx) return fail(PGL_ERR_UNSUPPORTED, "neuron lists with a separable stimulus");
            int rc = sep_forward(h, d_theta, p0.npost, (double*)h->Xbuf.p, xs, h->t_lo, h->t_hi, sp);
            if (rc) return rc;
        }
        for (size_t i = 0; i < slices.size(); ++i) {          // phase 1: X += F_s . W_s
            int rc = launch_prep(h, plans[i], slices[i], n_lo, d_theta, d_Weff);
            if (rc) return rc;
            FusedParams fp;
            fill_params(h, plans[i], slices[i], n_lo, false, 1, fp);
            hipError_t e = launch_any(plans[i], fp, h->stream);
            if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("forward launch: ") + hipGetErrorString(e));
        }
        {                                                      // phase 2: X -> (ll, r)
            const int rows = 512;
            const long long nrows = h->t_hi - h->t_lo;
            const int nblk = (int)((nrows + rows - 1) / rows);
            ENSURE(h->tmpA, (size_t)nblk * p0.npost * 8);
            ENSURE(h->tmpB, (size_t)nblk * p0.npost * 8);
            dim3 grid((unsigned)nblk, (unsigned)((p0.npost + 255) / 256));
            hipLaunchKernelGGL(k_rows_epilogue, grid, dim3(256), 0, h->stream, (double*)h->Xbuf.p, xs,
                               (const double*)h->bias.p, (const uint8_t*)h->S.p, h->N, n_lo, p0.npost,
                               (long long)h->t_lo, (long long)h->t_hi, rows, h->nlin, h->dt,
                               (double*)h->tmpA.p, (double*)h->tmpB.p, h->cur_pidx);
            HIPCHK(hipGetLastError());
            if (row1 > h->t_hi) {
                hipLaunchKernelGGL(k_rows_zero, dim3(64), dim3(256), 0, h->stream, (double*)h->Xbuf.p, xs,
                                   (long long)h->t_hi, row1);
                HIPCHK(hipGetLastError());
            }
            hipLaunchKernelGGL(k_rows_reduce, dim3(p0.npost), dim3(64), 0, h->stream,
                               (const double*)h->tmpA.p, (const double*)h->tmpB.p, nblk, p0.npost, P,
                               d_ll, d_grad);
            HIPCHK(hipGetLastError());
        }
        if (d_grad) {                                          // phase 3: G_s += F_s^T . r
            for (size_t i = 0; i < slices.size(); ++i) {
                int rc = launch_prep(h, plans[i], slices[i], n_lo, d_theta, d_Weff);   // only geometry/bias
                if (rc) return rc;
                FusedParams fp;
                fill_params(h, plans[i], slices[i], n_lo, true, 2, fp);
                hipError_t e = launch_any(plans[i], fp, h->stream);
                if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("backward launch: ") + hipGetErrorString(e));
                rc = launch_finalize_grad(h, plans[i], slices[i], n_lo, d_Weff, d_ll, d_grad);
                if (rc) return rc;
            }
            if (h->sep) {
                int rc = sep_backward(h, sp, d_grad);
                if (rc) return rc;
            }
        }
        if (rec) HIPCHK(hipEventRecord(h->ev[2], h->stream));
    }
    if (rec) HIPCHK(hipEventRecord(h->ev[3], h->stream));
    h->timing_valid = rec;
    if (rec) ++h->ev_launches;
    return PGL_OK;
}

int pgl_ll_grad_dev(pgl_handle h, int n_lo, int n_hi, const double* d_theta, const double* d_Weff,
                    double* d_ll, double* d_grad)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!d_theta || !d_Weff || !d_ll) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (n_lo < 0 || n_hi > h->N || n_lo >= n_hi) return fail(PGL_ERR_ARG, "bad neuron range");
    return enqueue_ll_grad(h, n_lo, n_hi, d_theta, d_Weff, d_ll, d_grad);
}

int pgl_ll_grad_list_dev(pgl_handle h, const int* d_idx, int count, const double* d_theta,
                         const double* d_Weff, double* d_ll, double* d_grad)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!d_idx || !d_theta || !d_Weff || !d_ll) return fail(PGL_ERR_ARG, "null argument");
    if (count <= 0 || count > h->N) return fail(PGL_ERR_ARG, "bad neuron count");
    HIPCHK(hipSetDevice(h->device));
    h->cur_pidx = d_idx;
    rc = enqueue_ll_grad(h, 0, count, d_theta, d_Weff, d_ll, d_grad);
    h->cur_pidx = nullptr;
    return rc;
}

// ---- lock-step BFGS bookkeeping kernels (inference/batched_bfgs.py) ----------------------------------------------
long long pgl_bfgs_state_doubles(int M, int P) { return (long long)pgl_bfgs_doubles(M, P); }

int pgl_bfgs_trial_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, double* d_Xt)
{
    if (!h || !d_state || !d_Xt || M <= 0 || P <= 0 || L <= 0 || L > M) return fail(PGL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_bfgs_trial, dim3(L), dim3(256), 0, h->stream, pgl_bfgs_view(d_state, M, P), d_rows, d_Xt);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_bfgs_objective_dev(pgl_handle h, int L, int P, const double* d_Xt, double* d_ll_f, double* d_grad_g,
                           int prior_kind, double mu_b, double sg_b, double stim_sigma, double mu, double sigma,
                           double lam)
{
    if (!h || !d_Xt || !d_ll_f || !d_grad_g || L <= 0) return fail(PGL_ERR_ARG, "bad argument");
    // (separable stimulus: the stimulus block of a row is [w_t, w_x], both under N(0, stim_sigma) -- bkgd.py:223-224 with mu = 0)
    if (P != 1 + h->Dstim + h->Kimp) return fail(PGL_ERR_ARG, "rows must be theta rows [bias, w_stim, w_ir]");
    if (prior_kind != 0 && prior_kind != 1) return fail(PGL_ERR_ARG, "prior kind: 0 Gaussian, 1 group lasso");
    HIPCHK(hipSetDevice(h->device));
    BfgsPrior q;
    q.N = h->N; q.B = h->B; q.Dstim = h->Dstim; q.kind = prior_kind;
    q.mu_b = mu_b; q.sg_b = sg_b; q.stim_sigma = stim_sigma; q.mu = mu; q.sigma = sigma; q.lam = lam;
    hipLaunchKernelGGL(k_bfgs_objective, dim3(L), dim3(256), 0, h->stream, P, d_Xt, d_ll_f, d_grad_g, q);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_bfgs_init_dev(pgl_handle h, double* d_state, int M, int P, double gtol)
{
    if (!h || !d_state || M <= 0 || P <= 0) return fail(PGL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_bfgs_init, dim3(M), dim3(256), 0, h->stream, pgl_bfgs_view(d_state, M, P), gtol);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

static BfgsStepArgs bfgs_step_args(int phases)
{
    BfgsStepArgs a;
    memset(&a, 0, sizeof(a));
    a.phases = phases;
    a.max_trials = 100;
    return a;
}

int pgl_bfgs_linesearch_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_Xt,
                            const double* d_f, const double* d_g, int max_trials)
{
    if (!h || !d_state || !d_Xt || !d_f || !d_g || M <= 0 || P <= 0 || L <= 0 || L > M || max_trials <= 0)
        return fail(PGL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    BfgsStepArgs a = bfgs_step_args(PGL_STEP_LS);            // f, g arrive final: no prior phase
    a.rows = d_rows; a.Xt = d_Xt; a.ft = const_cast<double*>(d_f); a.gt = const_cast<double*>(d_g);
    a.max_trials = max_trials;
    hipLaunchKernelGGL(k_bfgs_step<256>, dim3(L), dim3(256), 0, h->stream, pgl_bfgs_view(d_state, M, P), a);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_bfgs_hmul_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, double* d_H, int ld)
{
    if (!h || !d_state || !d_H || M <= 0 || P <= 0 || L <= 0 || L > M) return fail(PGL_ERR_ARG, "bad argument");
    if (ld < P || (ld & 1) || (reinterpret_cast<uintptr_t>(d_H) & 15))
        return fail(PGL_ERR_ARG, "H: leading dimension even and >= P, base 16-byte aligned");
    HIPCHK(hipSetDevice(h->device));
    const unsigned bx = (unsigned)((P + 4 * PGL_HM_ROWS - 1) / (4 * PGL_HM_ROWS));
    hipLaunchKernelGGL(k_bfgs_hmul, dim3(bx, (unsigned)L), dim3(256), 0, h->stream, pgl_bfgs_view(d_state, M, P), d_rows,
                       d_H, ld);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_bfgs_hmul_hist_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_hist,
                           const double* d_coef, int Kmax, double* d_ab)
{
    if (!h || !d_state || !d_hist || !d_coef || !d_ab || M <= 0 || P <= 0 || L <= 0 || L > M || Kmax <= 0)
        return fail(PGL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    const BfgsView v = pgl_bfgs_view(d_state, M, P);
    hipLaunchKernelGGL(k_bfgs_hdots, dim3((unsigned)((Kmax + 3) / 4), (unsigned)L), dim3(256), 0, h->stream, v, d_rows, d_hist,
                       d_coef, Kmax, d_ab);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bfgs_hcomb, dim3((unsigned)((P + 63) / 64), (unsigned)L), dim3(512), 0, h->stream, v, d_rows, d_hist, Kmax,
                       (const double*)d_ab);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_bfgs_update_dev(pgl_handle h, double* d_state, int M, int P, double gtol, int maxiter, int init_scaling, double* d_hist,
                        double* d_coef, int Kmax)
{
    if (!h || !d_state || M <= 0 || P <= 0) return fail(PGL_ERR_ARG, "bad argument");
    if ((d_hist != nullptr) != (d_coef != nullptr) || (d_hist && Kmax < maxiter))
        return fail(PGL_ERR_ARG, "history buffers: both or none, room for maxiter updates");
    HIPCHK(hipSetDevice(h->device));
    BfgsStepArgs a = bfgs_step_args(PGL_STEP_UPDATE);        // every row of the shard, no trial points
    a.gtol = gtol; a.maxiter = maxiter; a.init_scaling = init_scaling; a.Wh = d_hist; a.cs = d_coef; a.Kmax = Kmax;
    hipLaunchKernelGGL(k_bfgs_step<256>, dim3(M), dim3(256), 0, h->stream, pgl_bfgs_view(d_state, M, P), a);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

// One whole iteration behind an evaluation (see k_bfgs_step): in ONE launch while the update history of a row is short
// (hk_bound * P numbers, PGL_OPT_BFGS_MERGE), else as line search | k_bfgs_hdots | k_bfgs_hcomb | update (dense inverse
// Hessians: line search | k_bfgs_hmul | update).
int pgl_bfgs_step_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_Xt, double* d_ll_f,
                      double* d_grad_g, int prior_kind, double mu_b, double sg_b, double stim_sigma, double mu, double sigma,
                      double lam, int max_trials, double gtol, int maxiter, int init_scaling, double* d_hist, double* d_coef,
                      int Kmax, double* d_ab, int hk_bound, double* d_H, int ld, const int* d_pos_next, double* d_Xt_next,
                      double* flags_out)
{
    if (!h || !d_state || !d_Xt || !d_ll_f || !d_grad_g || M <= 0 || P <= 0 || L <= 0 || L > M || max_trials <= 0)
        return fail(PGL_ERR_ARG, "bad argument");
    if (prior_kind > 1) return fail(PGL_ERR_ARG, "prior kind: 0 Gaussian, 1 group lasso, < 0: f and g arrive final");
    if (prior_kind >= 0 && P != 1 + h->Dstim + h->Kimp) return fail(PGL_ERR_ARG, "rows must be theta rows [bias, w_stim, w_ir]");
    if ((d_hist != nullptr) != (d_coef != nullptr) || (d_hist && (Kmax < maxiter || !d_ab)))
        return fail(PGL_ERR_ARG, "history buffers: all or none, room for maxiter updates");
    if ((d_hist != nullptr) == (d_H != nullptr)) return fail(PGL_ERR_ARG, "either the update history or dense inverse Hessians");
    if (d_H && (ld < P || (ld & 1) || (reinterpret_cast<uintptr_t>(d_H) & 15)))
        return fail(PGL_ERR_ARG, "H: leading dimension even and >= P, base 16-byte aligned");
    HIPCHK(hipSetDevice(h->device));
    double* flags_dev = nullptr;
    if (flags_out) {                                         // pinned host memory: the address the device uses for it
        for (int i = 0; i < 4 && !flags_dev; ++i)
            if (h->flags_host[i] == flags_out) flags_dev = h->flags_dev[i];
        if (!flags_dev) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, flags_out, 0) != hipSuccess || !dp) {
                (void)hipGetLastError();
                return fail(PGL_ERR_ARG, "flags_out must be pinned (page-locked, device-mapped) host memory");
            }
            flags_dev = (double*)dp;
            h->flags_host[h->flags_next] = flags_out;
            h->flags_dev[h->flags_next] = flags_dev;
            h->flags_next = (h->flags_next + 1) & 3;
        }
    }
    const BfgsView v = pgl_bfgs_view(d_state, M, P);
    BfgsStepArgs a = bfgs_step_args(0);
    a.rows = d_rows; a.Xt = d_Xt; a.ft = d_ll_f; a.gt = d_grad_g;
    a.have_prior = prior_kind >= 0 ? 1 : 0;
    a.q.N = h->N; a.q.B = h->B; a.q.Dstim = h->Dstim; a.q.kind = prior_kind;
    a.q.mu_b = mu_b; a.q.sg_b = sg_b; a.q.stim_sigma = stim_sigma; a.q.mu = mu; a.q.sigma = sigma; a.q.lam = lam;
    a.max_trials = max_trials; a.gtol = gtol; a.maxiter = maxiter; a.init_scaling = init_scaling;
    a.Wh = d_hist; a.cs = d_coef; a.Kmax = Kmax; a.ab = d_ab;
    a.pos_next = d_pos_next; a.Xt_next = d_Xt_next; a.flags_out = flags_dev;
    const bool merged = d_hist && (long long)std::max(hk_bound, 0) * P <= (long long)h->opt_bfgs_merge;
    if (merged) {
        a.phases = PGL_STEP_LS | PGL_STEP_HIST | PGL_STEP_UPDATE;
        hipLaunchKernelGGL(k_bfgs_step<1024>, dim3(L), dim3(1024), 0, h->stream, v, a);
        HIPCHK(hipGetLastError());
        return PGL_OK;
    }
    a.phases = PGL_STEP_LS;
    hipLaunchKernelGGL(k_bfgs_step<256>, dim3(L), dim3(256), 0, h->stream, v, a);
    HIPCHK(hipGetLastError());
    if (d_hist) {
        const int Kb = std::min(Kmax, std::max(hk_bound, 1));       // no row holds more than hk_bound updates
        hipLaunchKernelGGL(k_bfgs_hdots, dim3((unsigned)((Kb + 3) / 4), (unsigned)L), dim3(256), 0, h->stream, v, d_rows,
                           (const double*)d_hist, (const double*)d_coef, Kmax, d_ab);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_bfgs_hcomb, dim3((unsigned)((P + 63) / 64), (unsigned)L), dim3(512), 0, h->stream, v, d_rows,
                           (const double*)d_hist, Kmax, (const double*)d_ab);
        HIPCHK(hipGetLastError());
    } else {
        const unsigned bx = (unsigned)((P + 4 * PGL_HM_ROWS - 1) / (4 * PGL_HM_ROWS));
        hipLaunchKernelGGL(k_bfgs_hmul, dim3(bx, (unsigned)L), dim3(256), 0, h->stream, v, d_rows, d_H, ld);
        HIPCHK(hipGetLastError());
    }
    a.phases = PGL_STEP_UPDATE;
    hipLaunchKernelGGL(k_bfgs_step<256>, dim3(L), dim3(256), 0, h->stream, v, a);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_sync(pgl_handle h)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

int pgl_ll_grad(pgl_handle h, int n_lo, int n_hi, const double* theta, const double* Weff,
                double* ll_out, double* grad_out)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!theta || !Weff || !ll_out) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (n_lo < 0 || n_hi > h->N || n_lo >= n_hi) return fail(PGL_ERR_ARG, "bad neuron range");
    const size_t P = 1 + (size_t)h->Dstim + h->Kimp;
    const size_t np = (size_t)(n_hi - n_lo);
    ENSURE(h->theta, np * P * 8);
    ENSURE(h->Weff, (size_t)h->N * h->N * 8);
    ENSURE(h->ll, np * 8);
    if (grad_out) ENSURE(h->grad, np * P * 8);
    HIPCHK(hipMemcpyAsync(h->theta.p, theta, np * P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->Weff.p, Weff, (size_t)h->N * h->N * 8, hipMemcpyHostToDevice, h->stream));
    rc = enqueue_ll_grad(h, n_lo, n_hi, (const double*)h->theta.p, (const double*)h->Weff.p,
                         (double*)h->ll.p, grad_out ? (double*)h->grad.p : nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ll_out, h->ll.p, np * 8, hipMemcpyDeviceToHost, h->stream));
    if (grad_out)
        HIPCHK(hipMemcpyAsync(grad_out, h->grad.p, np * P * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

int pgl_last_timing(pgl_handle h, double* fused_ms, double* total_ms)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    if (!h->timing_valid) return fail(PGL_ERR_STATE, "the last pgl_ll_grad call recorded no events (none yet, or PGL_OPT_TIMING skipped it)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev[3]));
    float a = 0, b = 0;
    HIPCHK(hipEventElapsedTime(&a, h->ev[1], h->ev[2]));
    HIPCHK(hipEventElapsedTime(&b, h->ev[0], h->ev[3]));
    if (fused_ms) *fused_ms = a;
    if (total_ms) *total_ms = b;
    return PGL_OK;
}

int pgl_timing_summary(pgl_handle h, int reset, int* n_launches, double* mean_fused_ms, double* mean_total_ms)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const long long n = std::min<long long>(h->ev_launches, pgl_context::NEV);
    double sa = 0, sb = 0;
    for (long long i = 0; i < n; ++i) {
        hipEvent_t* ev = h->evr[(h->ev_launches - 1 - i) % pgl_context::NEV];
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, ev[1], ev[2]));
        HIPCHK(hipEventElapsedTime(&b, ev[0], ev[3]));
        sa += a;
        sb += b;
    }
    if (n_launches) *n_launches = (int)n;
    if (mean_fused_ms) *mean_fused_ms = n ? sa / n : 0.0;
    if (mean_total_ms) *mean_total_ms = n ? sb / n : 0.0;
    if (reset) h->ev_launches = 0;
    return PGL_OK;
}

int pgl_set_stream(pgl_handle h, void* stream)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));          // drain work queued on the previous stream
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
    h->ev_launches = 0;
    h->timing_valid = false;
    return PGL_OK;
}

// Dry run of the dispatch, no device needed: the fused kernel instantiations (names as in the code object) that an
// ll(+grad) evaluation of `count` neurons starting at n_lo -- a range, or a list -- of a population of this shape would
// launch, one per line in `out`.  stim: 0 none / dense stimulus columns (Dstim of them), 1 separable stimulus by the
// tap-rate kernels, 2 separable at the frame rate with the stimulus current inside the fused forward where that form
// exists, 3 at the frame rate through the slab.  path: 0 ll+grad, 1 ll only, 2 the forward launches of pgl_gibbs_prepare_all.
int pgl_plan_kernels(int N, int B, int R, int Dstim, long long nT, int stim, int n_lo, int count, int path, int opt_kernel,
                     int opt_f32, char* out, int cap)
{
    if (!out || cap <= 0) return fail(PGL_ERR_ARG, "null argument");
    if (N <= 0 || B <= 0 || B > PGL_MAXB || R <= 0 || nT <= 0 || Dstim < 0 || count <= 0 || n_lo < 0 || n_lo + count > N)
        return fail(PGL_ERR_ARG, "bad shape");
    pgl_context c;
    c.N = N; c.B = B; c.R = R; c.Rk = R; c.nT = nT; c.Dstim = Dstim; c.Kimp = N * B;
    c.nT16 = (int)((nT + 15) / 16); c.t_lo = 0; c.t_hi = nT; c.numCU = 256;
    c.sep = stim >= 1; c.sepf = stim >= 2; c.sepA_ok = stim == 2; c.opt_sepf = 0;
    c.Ktot = c.Kimp + (c.sep ? 0 : Dstim);
    c.opt_kernel = opt_kernel; c.opt_f32 = (opt_f32 == 1) ? 1 : 0; c.opt_img32 = (opt_f32 == 2) ? 1 : 0;
    std::vector<std::string> names;
    g_dry = &names;
    int rc = PGL_OK;
    hipError_t e = hipSuccess;
    FusedParams fp{};
    std::vector<Slice> slices = make_slices(&c);
    std::vector<Plan> plans;
    if (path == 2) {
        plans.assign(slices.size(), Plan());
        for (size_t i = 0; i < slices.size() && rc == PGL_OK; ++i) {
            rc = make_plan(&c, 0, N, slices[i], plans[i], false);
            if (rc == PGL_OK) e = launch_any(plans[i], fp, nullptr);
        }
    } else {
        const bool grad = path == 0;
        fp.want_grad = grad;
        bool sepf = false, sliced = false, wide = false;
        rc = select_plans(&c, n_lo, n_lo + count, slices, plans, sepf, sliced, &wide);
        if (rc == PGL_OK) {
            const Plan& pl = plans[0];
            if (wide) {
                const size_t S = slices.size();
                for (size_t i = 0; i < S && e == hipSuccess; ++i) {
                    e = launch_fused5_wide(plans[i], fp, nullptr, i + 1 == S ? 2 : (i == 0 ? 0 : 1));
                    if (e == hipSuccess && i + 1 == S && grad) e = launch_fused5_wide(plans[i], fp, nullptr, 3);
                }
                for (size_t i = 0; i + 1 < S && e == hipSuccess && grad; ++i) {
                    e = launch_fused5_wide(plans[i], fp, nullptr, 4);
                    if (e == hipSuccess) e = launch_fused5_wide(plans[i], fp, nullptr, 3);
                }
            } else if (sepf) {
                if (pl.version == 5) {
                    e = launch_fused5_xin(pl, fp, nullptr, 1);
                    if (e == hipSuccess && grad) e = launch_fused5_xin(pl, fp, nullptr, 2);
                } else {
                    {
                        const bool ff = pl.KT <= 13 && c.sepA_ok && c.opt_sepf != 3;
                        e = launch_fused7_xio(pl, fp, nullptr, (ff && grad && pl.KT <= 12) ? 3 : (ff ? 2 : 1));
                    }
                }
            } else if (!sliced) {
                if (pl.version == 5 && grad) {
                    e = launch_fused5(pl, fp, nullptr, 1);
                    if (e == hipSuccess) e = launch_fused5(pl, fp, nullptr, 2);
                } else {
                    e = launch_any(pl, fp, nullptr);
                }
            } else {
                for (size_t i = 0; i < slices.size() && e == hipSuccess; ++i) e = launch_any(plans[i], fp, nullptr);
            }
        }
    }
    g_dry = nullptr;
    if (rc) return rc;
    if (e != hipSuccess) return fail(PGL_ERR_UNSUPPORTED, "no kernel instantiation for this plan");
    std::string all;
    for (const std::string& n : names) all += n + "\n";
    if ((int)all.size() + 1 > cap) return fail(PGL_ERR_ARG, "output buffer too small");
    std::memcpy(out, all.c_str(), all.size() + 1);
    return PGL_OK;
}

int pgl_info(pgl_handle h, int n_lo, int n_hi, double* info, int n_info)
{
    if (!h || !info) return fail(PGL_ERR_ARG, "null argument");
    std::vector<Slice> slices;
    // the stimulus path an evaluation would take: 0 none / dense feature columns, 1 separable by the tap-rate kernels on
    // the 3-phase path, 2 separable at the frame rate (k_sepf_*, impulse columns on resident tiles)
    std::vector<Plan> plans;
    bool sepf = false, sliced = false, wide = false;
    int rc = select_plans(h, n_lo, n_hi, slices, plans, sepf, sliced, &wide);
    if (rc) return rc;
    const int stim_path = sepf ? 2 : (h->sep ? 1 : 0);
    const Plan& pl = plans[0];
    const double P = 1.0 + h->Dstim + h->Kimp;
    double v[13];
    v[12] = stim_path;
    v[9] = pl.version;                       // 1 4-wave, 2 K-split, 3 K-split f32, 4 two-pass, 5 two-pass on resident feature tiles
    v[10] = (pl.version == 5) ? (double)pl.nTiles * (double)img_pair_bytes(pl.ktl, pl.kth)
            : (pl.version == 6 || pl.version == 7) ? (double)pl.nTiles * (double)img_bytes6(pl.KT, (pl.version == 6 && pl.sb6 == 2) ? 1 + pl.img32 : 0) : 0.0;   // resident feature bytes
    // HBM bytes the hot kernels stream per evaluation beyond the algorithmic ones (feature tiles read in
    // pass 1 and the H part again in pass 2, residual slab written and read)
    v[11] = (pl.version == 5) ? v[10] + (double)pl.nTiles * pgl_img_bytes(pl.kth) + 2.0 * (double)pl.nTiles * pl.nPT * 2048.0
            : (pl.version == 4) ? 2.0 * (double)pl.nTiles * pl.nPT * 2048.0
            : (pl.version == 6 || pl.version == 7) ? v[10] * pl.nPB : 0.0;
    if (wide) {                              // every column slice has its image set; each post block streams all of them
        v[10] = v[11] = 0.0;
        for (const Plan& q : plans) {
            const double b = (double)q.nTiles * (double)img_pair_bytes(q.ktl, q.kth);
            v[10] += b;
            v[11] += 2.0 * b * q.nPB;
        }
        v[11] += 2.0 * (double)plans.size() * (double)pl.nTiles * pl.nPT * 2048.0;
    }
    v[0] = pl.blocks; v[1] = pl.threads; v[2] = pl.nChunks; v[3] = pl.KT; v[4] = (double)pl.lds;
    v[5] = 16;
    const double nrows = (double)(h->t_hi - h->t_lo);
    v[6] = 4.0 * nrows * (double)h->Ktot * (double)pl.npost;
    // nT*N*1 (u8 counts) + nT*Dstim*8 + params in + (ll+grad) out
    v[7] = nrows * h->N + (h->sep ? (double)h->sepT * h->sepBx * 8.0 : nrows * h->Dstim * 8.0) + 8.0 * pl.npost * P +
           8.0 * pl.npost * (1.0 + P);          // separable: the projected stimulus at its frame rate
    v[8] = (double)h->nnz;
    for (int i = 0; i < n_info && i < 13; ++i) info[i] = v[i];
    return PGL_OK;
}

int pgl_features(pgl_handle h, double* fS_out)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!fS_out) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->nT * h->Kimp * 8;
    ENSURE(h->tmpA, bytes);
    hipLaunchKernelGGL(k_features, dim3(h->nT16), dim3(256), (size_t)h->B * h->Rk * 8, h->stream,
                       (const int2*)h->spk.p, (const int*)h->wlo.p, (const int*)h->whi.p,
                       (const double*)h->phi.p, (double*)h->tmpA.p, (long long)h->nT, h->N, h->B, h->Rk);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(fS_out, h->tmpA.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

// I_impT (N,nT) on device for impulse weights d_w (N,B)
static int enqueue_impulse_T(pgl_handle h, const double* d_w)
{
    ENSURE(h->IimpT, (size_t)h->N * h->nT * 8);
    const int blocks = (int)((h->nT + 63) / 64);
    const size_t lds = ((size_t)h->B * h->Rk + (size_t)h->N * h->B) * 8;
    hipLaunchKernelGGL(k_impulse_T, dim3(blocks), dim3(256), lds, h->stream, (const int2*)h->spk.p,
                       (const int*)h->wlo.p, (const int*)h->whi.p, (const double*)h->phi.p, d_w,
                       (double*)h->IimpT.p, (long long)h->nT, h->nT16, h->N, h->B, h->Rk);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

int pgl_impulse_currents(pgl_handle h, const double* w, double* I_imp_out)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!w || !I_imp_out) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    ENSURE(h->wsmall, (size_t)h->Kimp * 8);
    HIPCHK(hipMemcpyAsync(h->wsmall.p, w, (size_t)h->Kimp * 8, hipMemcpyHostToDevice, h->stream));
    rc = enqueue_impulse_T(h, (const double*)h->wsmall.p);
    if (rc) return rc;
    h->gibbs_npost = -1;
    std::vector<double> T((size_t)h->N * h->nT);
    HIPCHK(hipMemcpyAsync(T.data(), h->IimpT.p, T.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int n = 0; n < h->N; ++n)
        for (int64_t t = 0; t < h->nT; ++t) I_imp_out[(size_t)t * h->N + n] = T[(size_t)n * h->nT + t];
    return PGL_OK;
}

int pgl_gibbs_prepare(pgl_handle h, int n_post, const double* theta_n, const double* Weff_col)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!theta_n || !Weff_col) return fail(PGL_ERR_ARG, "null argument");
    if (n_post < 0 || n_post >= h->N) return fail(PGL_ERR_ARG, "n_post out of range");
    HIPCHK(hipSetDevice(h->device));
    const size_t P = 1 + (size_t)h->Dstim + h->Kimp;
    ENSURE(h->thetan, P * 8);
    ENSURE(h->wcol, (size_t)h->N * 8);
    ENSURE(h->Inet, (size_t)h->nT * 8);
    ENSURE(h->Istim, (size_t)h->nT * 8);
    HIPCHK(hipMemcpyAsync(h->thetan.p, theta_n, P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->wcol.p, Weff_col, (size_t)h->N * 8, hipMemcpyHostToDevice, h->stream));
    const double* d_th = (const double*)h->thetan.p;
    rc = enqueue_impulse_T(h, d_th + 1 + h->Dstim);
    if (rc) return rc;
    hipLaunchKernelGGL(k_inet, dim3(1024), dim3(256), 0, h->stream, (const double*)h->IimpT.p,
                       (const double*)h->wcol.p, (const double*)h->fstim.p, d_th + 1,
                       (double*)h->Inet.p, (double*)h->Istim.p, (long long)h->nT, h->N, h->sep ? 0 : h->Dstim);
    HIPCHK(hipGetLastError());
    if (h->sep) {                                 // I_stim of this neuron by the separable path
        SepParams sp;
        rc = sep_forward(h, d_th, 1, (double*)h->Istim.p, 1, 0, h->nT, sp);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gibbs_npost = n_post;
    h->gibbs_bias = theta_n[0];
    return PGL_OK;
}

static int run_ll_current(pgl_handle h, const double* d_base, const double* d_stim,
                          const double* d_col, int n_post, double bias, double aw_cur,
                          const double* w, int K, double* ll_out)
{
    // blocks of the streaming part; the spike-bin part writes one more partial row
    // the sums run over the evaluated time range [t_lo, t_hi) (pgl_set_time_range)
    const int nblocks = (int)std::min<int64_t>(1024, (h->t_hi - h->t_lo + 255) / 256);
    const int2* evb = h->h_ev.data();
    const auto before = [](const int2& e, int64_t t) { return (int64_t)e.x < t; };
    const int e_lo = (int)(std::lower_bound(evb + h->h_ptr[n_post], evb + h->h_ptr[n_post + 1], h->t_lo, before) - evb);
    const int e_hi = (int)(std::lower_bound(evb + h->h_ptr[n_post], evb + h->h_ptr[n_post + 1], h->t_hi, before) - evb);
    const int sblocks = std::max(1, std::min(256, (e_hi - e_lo + 255) / 256));
    ENSURE(h->part, (size_t)(nblocks + sblocks) * PGL_KMAX * 8);
    ENSURE(h->outK, PGL_KMAX * 8);
    if (!h->pin_out) HIPCHK(hipHostMalloc((void**)&h->pin_out, PGL_KMAX * 8, hipHostMallocDefault));
    for (int k0 = 0; k0 < K; k0 += PGL_KMAX) {
        const int kk = std::min(PGL_KMAX, K - k0);
        PglWeights wv;
        for (int k = 0; k < PGL_KMAX; ++k) wv.w[k] = (k < kk) ? w[k0 + k] : 0.0;
        hipLaunchKernelGGL(k_ll_current, dim3(nblocks), dim3(256), 0, h->stream, d_base, d_stim,
                           d_col, bias, aw_cur, wv, kk, h->nlin, h->dt,
                           (long long)h->t_lo, (long long)h->t_hi, (double*)h->part.p);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_ll_current_spikes, dim3(sblocks), dim3(256), 0, h->stream,
                           (const int2*)h->spk.p, e_lo, e_hi, d_base, d_stim, d_col, bias, aw_cur,
                           wv, kk, h->nlin,
                           (double*)h->part.p + (size_t)nblocks * PGL_KMAX);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_reduce_parts, dim3(kk), dim3(64), 0, h->stream, (const double*)h->part.p,
                           nblocks + sblocks, kk, (double*)h->outK.p);
        HIPCHK(hipGetLastError());
        // results through a pinned buffer: a pageable destination makes the copy a staged, slower one
        // (a single launch with a last-ticket reduction into host memory measured the same 56 us:
        // its agent-scope fences cost what the two launches do)
        HIPCHK(hipMemcpyAsync(h->pin_out, h->outK.p, (size_t)kk * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        std::memcpy(ll_out + k0, h->pin_out, (size_t)kk * 8);
    }
    return PGL_OK;
}

int pgl_gibbs_ll(pgl_handle h, int n_pre, double aw_cur, const double* w, int K, double* ll_out)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->gibbs_npost < 0) return fail(PGL_ERR_STATE, "pgl_gibbs_prepare has not been called");
    if (!w || !ll_out || K <= 0) return fail(PGL_ERR_ARG, "bad argument");
    if (n_pre < 0 || n_pre >= h->N) return fail(PGL_ERR_ARG, "n_pre out of range");
    HIPCHK(hipSetDevice(h->device));
    const double* col = (const double*)h->IimpT.p + (size_t)n_pre * h->nT;
    return run_ll_current(h, (const double*)h->Inet.p, (const double*)h->Istim.p, col,
                          h->gibbs_npost, h->gibbs_bias, aw_cur, w, K, ll_out);
}

int pgl_gibbs_update(pgl_handle h, int n_pre, double delta)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->gibbs_npost < 0) return fail(PGL_ERR_STATE, "pgl_gibbs_prepare has not been called");
    if (n_pre < 0 || n_pre >= h->N) return fail(PGL_ERR_ARG, "n_pre out of range");
    HIPCHK(hipSetDevice(h->device));
    const double* col = (const double*)h->IimpT.p + (size_t)n_pre * h->nT;
    hipLaunchKernelGGL(k_axpy, dim3(1024), dim3(256), 0, h->stream, (double*)h->Inet.p, col, delta,
                       (long long)h->nT);
    HIPCHK(hipGetLastError());
    return PGL_OK;
}

// ---- batched column Gibbs: all post-synaptic columns on the device at once ----------------------
int pgl_gibbs_prepare_all(pgl_handle h, const double* theta, const double* Weff)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!theta || !Weff) return fail(PGL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    const size_t P = 1 + (size_t)h->Dstim + h->Kimp;
    const int N = h->N;
    ENSURE(h->gtheta, (size_t)N * P * 8);
    ENSURE(h->Weff, (size_t)N * N * 8);
    HIPCHK(hipMemcpyAsync(h->gtheta.p, theta, (size_t)N * P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->Weff.p, Weff, (size_t)N * N * 8, hipMemcpyHostToDevice, h->stream));
    // the caller's (pageable) theta / Weff may be temporaries that die when this call returns: the uploads
    // must have left host memory by then, whatever the runtime does with pageable sources
    HIPCHK(hipStreamSynchronize(h->stream));
    // forward-only launches of the K-split kernel: GX[t][n] = sum_slices F_s . W_s (stimulus columns
    // included), the same phase 1 as the sliced ll+grad path
    const std::vector<Slice> slices = make_slices(h);
    std::vector<Plan> plans(slices.size());
    size_t maxLL = 0;
    for (size_t i = 0; i < slices.size(); ++i) {
        rc = make_plan(h, 0, N, slices[i], plans[i], false);
        if (rc) return rc;
        maxLL = std::max(maxLL, (size_t)plans[i].nChunks * plans[i].nPT * plans[i].KSPLIT * 64 * 8);
    }
    ENSURE(h->llpart, maxLL);
    ENSURE(h->gbpart, maxLL);
    const int xs = plans[0].nPT * 16;
    ENSURE(h->GX, (size_t)h->nT * xs * 8);
    const long long row0 = (long long)plans[0].tile0 * 16;
    const long long row1 = std::min<long long>(h->nT, (long long)(plans[0].tile0 + plans[0].nTiles) * 16);
    HIPCHK(hipMemsetAsync((double*)h->GX.p + row0 * xs, 0, (size_t)(row1 - row0) * xs * 8, h->stream));
    if (h->sep) {
        SepParams sp;
        rc = sep_forward(h, (const double*)h->gtheta.p, N, (double*)h->GX.p, xs, h->t_lo, h->t_hi, sp);
        if (rc) return rc;
    }
    for (size_t i = 0; i < slices.size(); ++i) {
        rc = launch_prep(h, plans[i], slices[i], 0, (const double*)h->gtheta.p, (const double*)h->Weff.p);
        if (rc) return rc;
        FusedParams fp;
        fill_params(h, plans[i], slices[i], 0, false, 1, fp);
        fp.Xbuf = (double*)h->GX.p;
        fp.xstride = xs;
        hipError_t e = launch_any(plans[i], fp, h->stream);
        if (e != hipSuccess) return fail(PGL_ERR_HIP, std::string("forward launch: ") + hipGetErrorString(e));
    }
    h->gx_xs = xs;
    h->gx_t_lo = h->t_lo;
    h->gx_t_hi = h->t_hi;
    return PGL_OK;
}

// stage the column arguments in one pinned block, one H2D copy: [cols | pre | aw | w]
static int stage_cols(pgl_handle h, int ncols, const int* n_post, const int* n_pre, const double* aw,
                      const double* w, int nw, GibbsColsParams& gp, size_t extra_out_bytes,
                      bool with_events = false, int* max_events = nullptr)
{
    if (h->gx_xs == 0) return fail(PGL_ERR_STATE, "pgl_gibbs_prepare_all has not been called");
    if (h->gx_t_lo != h->t_lo || h->gx_t_hi != h->t_hi)
        return fail(PGL_ERR_STATE, "time range changed since pgl_gibbs_prepare_all");
    if (ncols <= 0 || !n_post || !n_pre || !w) return fail(PGL_ERR_ARG, "bad argument");
    for (int c = 0; c < ncols; ++c)
        if (n_post[c] < 0 || n_post[c] >= h->N || n_pre[c] < 0 || n_pre[c] >= h->N)
            return fail(PGL_ERR_ARG, "neuron index out of range");
    const size_t o_pre = (size_t)ncols * 4, o_aw = ((o_pre + (size_t)ncols * 4 + 7) / 8) * 8;
    const size_t o_w = o_aw + (size_t)ncols * 8, o_elo = o_w + (size_t)ncols * nw * 8;
    const size_t o_ehi = o_elo + (size_t)ncols * 4;
    const size_t bytes = with_events ? ((o_ehi + (size_t)ncols * 4 + 7) / 8) * 8 : o_elo;
    const size_t need = std::max(bytes, extra_out_bytes);
    if (need > h->pin_args_cap) {
        if (h->pin_args) (void)hipHostFree(h->pin_args);
        h->pin_args = nullptr;
        h->pin_args_cap = 0;
        HIPCHK(hipHostMalloc((void**)&h->pin_args, need * 2, hipHostMallocDefault));
        h->pin_args_cap = need * 2;
    }
    ENSURE(h->gargs, bytes);
    std::memcpy(h->pin_args, n_post, (size_t)ncols * 4);
    std::memcpy(h->pin_args + o_pre, n_pre, (size_t)ncols * 4);
    for (int c = 0; c < ncols; ++c) reinterpret_cast<double*>(h->pin_args + o_aw)[c] = aw ? aw[c] : 0.0;
    std::memcpy(h->pin_args + o_w, w, (size_t)ncols * nw * 8);
    if (with_events) {
        // events of every listed post-synaptic neuron inside the evaluated time range
        const int2* evb = h->h_ev.data();
        const auto before = [](const int2& e, int64_t t) { return (int64_t)e.x < t; };
        int mx = 0;
        for (int c = 0; c < ncols; ++c) {
            const int2* b = evb + h->h_ptr[n_post[c]];
            const int2* e = evb + h->h_ptr[n_post[c] + 1];
            const int lo = (int)(std::lower_bound(b, e, h->t_lo, before) - evb);
            const int hi = (int)(std::lower_bound(b, e, h->t_hi, before) - evb);
            reinterpret_cast<int*>(h->pin_args + o_elo)[c] = lo;
            reinterpret_cast<int*>(h->pin_args + o_ehi)[c] = hi;
            mx = std::max(mx, hi - lo);
        }
        if (max_events) *max_events = mx;
    }
    HIPCHK(hipMemcpyAsync(h->gargs.p, h->pin_args, bytes, hipMemcpyHostToDevice, h->stream));
    const unsigned char* d = (const unsigned char*)h->gargs.p;
    gp.GX = (const double*)h->GX.p; gp.xs = h->gx_xs; gp.S = (const uint8_t*)h->S.p;
    gp.N = h->N; gp.B = h->B; gp.R = h->Rk; gp.P = 1 + h->Dstim + h->Kimp; gp.woff = 1 + h->Dstim;
    gp.spk = (const int2*)h->spk.p; gp.wlo = (const int*)h->wlo.p; gp.whi = (const int*)h->whi.p;
    gp.phi = (const double*)h->phi.p; gp.theta = (const double*)h->gtheta.p;
    gp.cols = (const int*)d; gp.pre = (const int*)(d + o_pre); gp.aw = (const double*)(d + o_aw);
    gp.w = (const double*)(d + o_w);
    gp.ncols = ncols; gp.K = nw; gp.nlin = h->nlin; gp.dt = h->dt;
    gp.t_lo = h->t_lo; gp.t_hi = h->t_hi;
    gp.CP = std::max(1, std::min(ncols, 256 / std::max(1, nw)));   // columns per workgroup (ll) / per sweep (update)
    gp.part = nullptr;
    gp.nsplit = 1;
    gp.elo = with_events ? (const int*)(d + o_elo) : nullptr;
    gp.ehi = with_events ? (const int*)(d + o_ehi) : nullptr;
    gp.partS = nullptr;
    gp.nloop = 1;
    gp.hs = nullptr;
    gp.fs = nullptr; gp.fs_stride = 0; gp.hs_region = 0;
    gp.dbg = 0;
    return PGL_OK;
}

int pgl_gibbs_ll_cols(pgl_handle h, int ncols, const int* n_post, const int* n_pre,
                      const double* aw_cur, const double* w, int K, double* ll_out)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (K <= 0 || K > PGL_KMAX || !ll_out) return fail(PGL_ERR_ARG, "K must be in 1..16");
    HIPCHK(hipSetDevice(h->device));
    GibbsColsParams gp;
    if (h->nlin == PGL_NLIN_EXPLINEAR && h->opt_gibbs != 1 && h->Rk + PGL_GRB + 32 < 65535) {   // (16-bit event counts per window)
        // regime-split path: rate terms by k_gibbs_rate_cols (single precision for the log1p term where
        // |x| >= 12, compacted f64 elsewhere), spike terms from the event lists
        int max_ev = 0;
        rc = stage_cols(h, ncols, n_post, n_pre, aw_cur, w, K, gp, (size_t)ncols * K * 8, true, &max_ev);
        if (rc) return rc;
        gp.CP = std::min(ncols, 8);
        gp.nsplit = (gp.CP >= 3) ? 1 : (gp.CP == 2) ? 2 : ((PGL_GRB / 64) % 4 == 0 ? 4 : 3);   // (PGL_GRB / 64) % nsplit == 0
        const int ygroups = (ncols + gp.CP - 1) / gp.CP;
        const long long nrows = h->t_hi - h->t_lo;
        const long long nsub = (nrows + PGL_GRB - 1) / PGL_GRB;
        // every column with the same presynaptic neuron (a sweep step): its filtered spike train once per launch
        bool same_pre = ncols >= 2 && !(h->opt_dbg & 0x1000);
        for (int c = 1; c < ncols && same_pre; ++c) same_pre = n_pre[c] == n_pre[0];
        gp.hs_region = gp.CP * h->Rk;
        if (same_pre) gp.hs_region = std::max(gp.hs_region, h->B * (PGL_GRB + 2) + gp.CP * 8);
        const size_t lds = ((size_t)gp.hs_region + (size_t)gp.CP * (PGL_GRB + 2) + (size_t)gp.CP * PGL_KMAX +
                            (size_t)4 * PGL_GQ + (size_t)PGL_SPT_N + (size_t)gp.CP) * 8 +     // (band queue, softplus-tail table, max |w|)
                           (size_t)gp.CP * PGL_GECAP_R * 8 + (size_t)gp.CP * PGL_GNL * 6 + 16;
        auto rate_kernel = k_gibbs_rate_cols;
        {
            hipError_t e = ensure_dyn_lds(rate_kernel, lds);
            if (e != hipSuccess) return fail(PGL_ERR_HIP, hipGetErrorString(e));
        }
        // sub-blocks per workgroup: short loops.  The columns differ in how many of their elements fall into the f64
        // band, so workgroups differ in run time and the dispatcher's dynamic placement has to even that out: at C4
        // (37 504 sub-block units on 1 024 slots -- occupancy query: four workgroups per CU at <= 40 KB of LDS and 128
        // VGPRs) loops of 3..10 sub-blocks run within 1 % of each other (1.14 ms per launch), 13 = three well-filled
        // rounds 1.19, 16 1.25, 37 = ONE round filled to 99 % 1.23 ms (tools/r4/gibbs_nloop_scan.py).  So: at least six
        // rounds of workgroups, loops of at most eight sub-blocks (the per-workgroup setup is amortised from ~4).
        {
            const long long slots = (long long)gibbs_rate_wg_per_cu(lds) * h->numCU;
            gp.nloop = (int)std::max(1LL, std::min(8LL, nsub * ygroups / (6 * slots)));
        }
        if ((h->opt_dbg >> 8) & 0xf) gp.nloop = std::min(PGL_GNL, (h->opt_dbg >> 8) & 0xf);      // tests: force the sub-block loop
        const int nblk = (int)((nsub + gp.nloop - 1) / gp.nloop);
        const int sblk = std::max(1, (max_ev + 255) / 256);
        ENSURE(h->gpart, (size_t)(nblk + sblk) * ncols * PGL_KMAX * 8);
        ENSURE(h->gout, (size_t)ncols * K * 8);
        ENSURE(h->ghs, (size_t)ncols * h->Rk * 8);
        gp.part = (double*)h->gpart.p;
        gp.partS = gp.part + (size_t)nblk * ncols * PGL_KMAX;
        gp.hs = (double*)h->ghs.p;
        gp.dbg = h->opt_dbg & 0xff;
        if (same_pre) {
            gp.fs_stride = (nrows + 63) / 64 * 64;
            ENSURE(h->gfs, (size_t)h->B * gp.fs_stride * 8);
            gp.fs = (const double*)h->gfs.p;
        }
        hipLaunchKernelGGL(k_gibbs_cols_setup, dim3((ncols * h->Rk + 255) / 256), dim3(256), 0, h->stream, gp);
        HIPCHK(hipGetLastError());
        if (same_pre) {
            hipLaunchKernelGGL(k_gibbs_pre_features, dim3((unsigned)((nrows + 255) / 256)), dim3(256), (size_t)h->B * h->Rk * 8,
                               h->stream, gp, n_pre[0], (double*)h->gfs.p);
            HIPCHK(hipGetLastError());
        }
        // rate and spike workgroups in ONE launch, the spike workgroups (short chains of dependent loads: 87 us as a launch
        // of their own) dispatched last: they fill the slots the rate workgroups free at the end.  (On a side stream beside
        // the rate kernel -- two event hops -- they cost more than in line: 1.365 against 1.295 ms, round 3.)
        gp.nblkR = nblk; gp.nygR = ygroups; gp.nblkS = sblk;
        hipLaunchKernelGGL(rate_kernel, dim3((unsigned)(nblk * ygroups + sblk * ncols)), dim3(256), lds, h->stream, gp);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_gibbs_reduce_cols2, dim3(ncols, K), dim3(64), 0, h->stream, (const double*)gp.part, nblk,
                           (const double*)gp.partS, sblk, ncols, K, h->dt, (double*)h->pin_args);
        HIPCHK(hipGetLastError());
        // the results land in the pinned host block directly (its argument bytes were consumed by the upload
        // that precedes the kernels on the stream): no device-to-host copy call
        HIPCHK(hipStreamSynchronize(h->stream));
        std::memcpy(ll_out, h->pin_args, (size_t)ncols * K * 8);
        return PGL_OK;
    }
    rc = stage_cols(h, ncols, n_post, n_pre, aw_cur, w, K, gp, (size_t)ncols * K * 8);
    if (rc) return rc;
    // enough blocks for a few per CU, at most 384 bins each (the block's presynaptic events -- window of
    // rows + R bins, ~11 at 20 Hz -- are staged in LDS, PGL_GECAP per column), whole sub-blocks
    const int ygroups = (ncols + gp.CP - 1) / gp.CP;
    const long long nrows = h->t_hi - h->t_lo;
    int gtb = 32;
    while (gtb * gp.CP < 256) gtb += 32;                 // narrow launches: longer sub-blocks keep 256 threads busy
    gp.gtb = gtb;
    long long rows = (nrows * ygroups + 4 * h->numCU - 1) / (4LL * h->numCU);
    rows = std::max<long long>(gtb, std::min<long long>(rows, 384));
    rows = (rows + gtb - 1) / gtb * gtb;
    gp.rows = (int)rows;
    const int nblk = (int)((nrows + rows - 1) / rows);
    ENSURE(h->gpart, (size_t)nblk * ncols * PGL_KMAX * 8);
    ENSURE(h->gout, (size_t)ncols * K * 8);
    gp.part = (double*)h->gpart.p;
    const size_t lds = ((size_t)h->B * h->Rk + (size_t)3 * gtb * gp.CP) * 8 + (size_t)gp.CP * PGL_GECAP * 8 +
                       (size_t)gp.CP * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gibbs_ll_cols),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(PGL_ERR_HIP, hipGetErrorString(e));
    hipLaunchKernelGGL(k_gibbs_ll_cols, dim3(nblk, ygroups), dim3(256), lds, h->stream, gp);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_gibbs_reduce_cols, dim3(ncols), dim3(64), 0, h->stream, (const double*)h->gpart.p,
                       nblk, ncols, K, (double*)h->gout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->pin_args, h->gout.p, (size_t)ncols * K * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::memcpy(ll_out, h->pin_args, (size_t)ncols * K * 8);
    return PGL_OK;
}

int pgl_gibbs_update_cols(pgl_handle h, int ncols, const int* n_post, const int* n_pre, const double* delta)
{
    int rc = check_ready(h);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    for (int a = 0; a < ncols; ++a)                       // two deltas for one post neuron would race
        for (int b = a + 1; b < ncols; ++b)
            if (n_post && n_post[a] == n_post[b]) return fail(PGL_ERR_ARG, "duplicate post-synaptic column");
    GibbsColsParams gp;
    rc = stage_cols(h, ncols, n_post, n_pre, nullptr, delta, 1, gp, 0);
    if (rc) return rc;
    const int ygroups = (ncols + gp.CP - 1) / gp.CP;
    const long long nrows = h->t_hi - h->t_lo;
    gp.rows = 1024;
    const int nblk = (int)((nrows + gp.rows - 1) / gp.rows);
    const size_t lds = (size_t)h->B * h->Rk * 8;
    hipLaunchKernelGGL(k_gibbs_update_cols, dim3(nblk, ygroups), dim3(256), lds, h->stream, gp, (double*)h->GX.p);
    HIPCHK(hipGetLastError());
    // the column arguments were staged in the handle's pinned block: the next call refills it, so the
    // asynchronous upload must have happened before this one returns
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

int pgl_gibbs_currents(pgl_handle h, int n_post, double* x_out)
{
    if (!h || !x_out) return fail(PGL_ERR_ARG, "null argument");
    if (h->gx_xs == 0) return fail(PGL_ERR_STATE, "pgl_gibbs_prepare_all has not been called");
    if (n_post < 0 || n_post >= h->N) return fail(PGL_ERR_ARG, "n_post out of range");
    HIPCHK(hipSetDevice(h->device));
    const long long nrows = h->gx_t_hi - h->gx_t_lo;
    HIPCHK(hipMemcpy2DAsync(x_out, 8, (const double*)h->GX.p + h->gx_t_lo * h->gx_xs + n_post,
                            (size_t)h->gx_xs * 8, 8, (size_t)nrows, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

int pgl_ll_from_current(pgl_handle h, int n_post, double I_bias, const double* I_stim,
                        const double* I_other, const double* I_col, const double* w, int K,
                        double* ll_out)
{
    if (!h) return fail(PGL_ERR_ARG, "null handle");
    if (!h->have_spikes) return fail(PGL_ERR_STATE, "pgl_set_spikes_* has not been called");
    if (!I_other || !I_col || !w || !ll_out || K <= 0) return fail(PGL_ERR_ARG, "bad argument");
    if (n_post < 0 || n_post >= h->N) return fail(PGL_ERR_ARG, "n_post out of range");
    HIPCHK(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->nT * 8;
    ENSURE(h->tmpA, bytes);
    ENSURE(h->tmpB, bytes);
    HIPCHK(hipMemcpyAsync(h->tmpA.p, I_other, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->tmpB.p, I_col, bytes, hipMemcpyHostToDevice, h->stream));
    const double* d_stim = nullptr;
    if (I_stim) {
        ENSURE(h->tmpC, bytes);
        HIPCHK(hipMemcpyAsync(h->tmpC.p, I_stim, bytes, hipMemcpyHostToDevice, h->stream));
        d_stim = (const double*)h->tmpC.p;
    }
    return run_ll_current(h, (const double*)h->tmpA.p, d_stim, (const double*)h->tmpB.p, n_post,
                          I_bias, 0.0, w, K, ll_out);
}

namespace {
struct UniformStream {
    const double* buf;
    int64_t n, pos;
    uint64_t state;
    double next()
    {
        if (buf && pos < n) return buf[pos++];
        uint64_t z = (state += 0x9e3779b97f4a7c15ULL);        // splitmix64
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z ^= z >> 31;
        return ((z >> 11) + 0.5) * (1.0 / 9007199254740992.0); // (0,1)
    }
};
inline double host_nlin(double x, int nlin)
{
    return nlin == 1 ? std::fmax(x, 0.0) + std::log1p(std::exp(-std::fabs(x))) : std::exp(x);
}
}  // namespace

int pgl_simulate(int N, int64_t nT, int R, int nlin, double dt, double* X, const double* AW,
                 const double* uniforms, int64_t n_uniforms, uint64_t seed, double* S,
                 int64_t* n_exceptions_out)
{
    if (N <= 0 || nT <= 0 || R <= 0 || !X || !AW || !S) return fail(PGL_ERR_ARG, "bad argument");
    if (nlin != PGL_NLIN_EXP && nlin != PGL_NLIN_EXPLINEAR) return fail(PGL_ERR_ARG, "unknown nonlinearity");
    UniformStream u{uniforms, uniforms ? n_uniforms : 0, 0, seed};
    std::vector<double> acc(N, 0.0), thr(N);
    std::vector<char> spk(N);
    for (int n = 0; n < N; ++n) thr[n] = -std::log(u.next());             // population.py:293
    std::memset(S, 0, sizeof(double) * (size_t)nT * N);
    int64_t n_exc = 0;
    for (int64_t t = 0; t < nT; ++t) {
        double* Xt = X + (size_t)t * N;
        double* St = S + (size_t)t * N;
        int n_spk = 0;
        for (int n = 0; n < N; ++n) {
            acc[n] += host_nlin(Xt[n], nlin) * dt;                       // population.py:317-318
            spk[n] = acc[n] > thr[n];
            if (spk[n]) { St[n] += 1.0; ++n_spk; }
        }
        const int64_t t_imp = std::min<int64_t>(nT - t - 1, R);          // population.py:326
        while (n_spk > 0) {
            bool capped = false;
            for (int n = 0; n < N; ++n) capped = capped || (St[n] >= 10.0);
            if (capped) { ++n_exc; break; }                              // population.py:345-349
            for (int np = 0; np < N; ++np) {
                if (!spk[np]) continue;
                const double* aw = AW + (size_t)np * R * N;
                for (int64_t tau = 0; tau < t_imp; ++tau) {              // population.py:351-353
                    double* xr = X + (size_t)(t + 1 + tau) * N;
                    const double* ar = aw + (size_t)tau * N;
                    for (int n = 0; n < N; ++n) xr[n] += ar[n];
                }
            }
            for (int n = 0; n < N; ++n) {                                // population.py:355-360
                if (spk[n]) acc[n] -= thr[n];
                if (acc[n] < 0) acc[n] = 0;
            }
            for (int n = 0; n < N; ++n)
                if (spk[n]) thr[n] = -std::log(u.next());
            n_spk = 0;
            for (int n = 0; n < N; ++n) {
                spk[n] = acc[n] > thr[n];
                if (spk[n]) { St[n] += 1.0; ++n_spk; }
            }
        }
    }
    if (n_exceptions_out) *n_exceptions_out = n_exc;
    return PGL_OK;
}

// Leading singular pairs of a batch of (L x D) matrices (k_lsp_*, pglm_stim.hip.h).  Host arrays; everything in between on
// the device with the library's own kernels.
int pgl_leading_singular_pairs(pgl_handle h, const double* A, int n, int L, int D, double* U, double* sigma, double* V)
{
    if (!h || !U || !sigma || !V || n <= 0 || L <= 0 || D <= 0) return fail(PGL_ERR_ARG, "bad argument");
    if (!A && (!h->staA.p || h->sta_n != n || h->sta_L != L || h->sta_D != D))
        return fail(PGL_ERR_STATE, "A == NULL: no spike-triggered averages of this shape on the device (pgl_sta with A_out == NULL)");
    HIPCHK(hipSetDevice(h->device));
    const bool wide = L <= D;                              // Gram matrix of the smaller side
    const int m = wide ? L : D, big = wide ? D : L;
    const size_t na = (size_t)n * L * D, ng = (size_t)n * m * m;
    DevBuf dA, dG0, dG1, dx, dy, dn;
    auto done = [&](int rc) {
        release(dA); release(dG0); release(dG1); release(dx); release(dy); release(dn);
        return rc;
    };
    if (A) {
        if (ensure(dA, na * 8)) return done(PGL_ERR_HIP);
        if (hipMemcpyAsync(dA.p, A, na * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) return done(fail(PGL_ERR_HIP, "upload"));
    } else {                                               // the averages pgl_sta left on the device; consumed by this call
        dA = h->staA;
        h->staA = DevBuf();
        h->sta_n = h->sta_L = h->sta_D = 0;
    }
    if (ensure(dG0, ng * 8) || ensure(dG1, ng * 8) || ensure(dx, (size_t)n * m * 8) ||
        ensure(dy, (size_t)n * big * 8) || ensure(dn, (size_t)n * 8))
        return done(PGL_ERR_HIP);
    const double* a = (const double*)dA.p;
    const long long sa = (long long)L * D, sg = (long long)m * m;
    auto gemm = [&](const double* Am, long long sam, long long sak, long long sAb, const double* Bm, long long sbn, long long sbk,
                    long long sBb, double* C, long long scm, long long scn, long long sCb, int M, int N, int K) {
        if (N >= 64) {
            hipLaunchKernelGGL(k_gemm_mfma<4>, dim3((M + 15) / 16, (N + 63) / 64, n), dim3(512), 0, h->stream, Am, sam, sak, Bm, sbn,
                               sbk, C, scm, scn, M, N, K, sAb, sBb, sCb);
        } else {
            hipLaunchKernelGGL(k_gemm_mfma<1>, dim3((M + 15) / 16, (N + 15) / 16, n), dim3(512), 0, h->stream, Am, sam, sak, Bm, sbn,
                               sbk, C, scm, scn, M, N, K, sAb, sBb, sCb);
        }
    };
    // G = A A^T (wide) or A^T A (tall); element (i, j) of the small side
    if (wide) gemm(a, D, 1, sa, a, D, 1, sa, (double*)dG0.p, m, 1, sg, m, m, D);
    else gemm(a, 1, D, sa, a, 1, D, sa, (double*)dG0.p, m, 1, sg, m, m, L);
    double* g0 = (double*)dG0.p;
    double* g1 = (double*)dG1.p;
    hipLaunchKernelGGL(k_lsp_trace_scale, dim3(n), dim3(1024), 0, h->stream, g0, m, sg);
    for (int s = 0; s < 16; ++s) {                         // G <- G^2 / trace(G^2)  (G symmetric: G G^T): G^(2^16) in the end
        gemm(g0, m, 1, sg, g0, m, 1, sg, g1, m, 1, sg, m, m, m);
        hipLaunchKernelGGL(k_lsp_trace_scale, dim3(n), dim3(1024), 0, h->stream, g1, m, sg);
        std::swap(g0, g1);
    }
    double* x = (double*)dx.p;                             // (n, m): on the small side
    double* y = (double*)dy.p;                             // (n, big)
    hipLaunchKernelGGL(k_lsp_pick, dim3(n), dim3(1024), 0, h->stream, (const double*)g0, m, sg, x);
    // y = A^T x (wide: D-vector) or A x (tall: L-vector); x back from y
    auto to_big = [&]() {
        if (wide) gemm(a, 1, D, sa, x, 0, 1, m, y, 1, 0, big, D, 1, L);       // y[d] = sum_l A[l][d] x[l]
        else gemm(a, D, 1, sa, x, 0, 1, m, y, 1, 0, big, L, 1, D);            // y[l] = sum_d A[l][d] x[d]
    };
    auto to_small = [&]() {
        if (wide) gemm(a, D, 1, sa, y, 0, 1, big, x, 1, 0, m, L, 1, D);       // x[l] = sum_d A[l][d] y[d]
        else gemm(a, 1, D, sa, y, 0, 1, big, x, 1, 0, m, D, 1, L);            // x[d] = sum_l A[l][d] y[l]
    };
    for (int it = 0; it < 2; ++it) {
        to_big();
        hipLaunchKernelGGL(k_lsp_normalize, dim3(n), dim3(1024), 0, h->stream, y, big, (double*)nullptr);
        to_small();
        hipLaunchKernelGGL(k_lsp_normalize, dim3(n), dim3(1024), 0, h->stream, x, m, (double*)nullptr);
    }
    to_big();                                              // |A^T x| (wide) or |A x| (tall) = sigma_0
    hipLaunchKernelGGL(k_lsp_normalize, dim3(n), dim3(1024), 0, h->stream, y, big, (double*)dn.p);
    if (hipGetLastError() != hipSuccess) return done(fail(PGL_ERR_HIP, "leading singular pairs: launch failed"));
    double* Uo = wide ? x : y;
    double* Vo = wide ? y : x;
    bool ok = hipMemcpyAsync(U, Uo, (size_t)n * L * 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
              hipMemcpyAsync(V, Vo, (size_t)n * D * 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
              hipMemcpyAsync(sigma, dn.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
              hipStreamSynchronize(h->stream) == hipSuccess;
    if (!ok) return done(fail(PGL_ERR_HIP, "leading singular pairs: copy failed"));
    // sign convention: the component of u_0 of largest magnitude is positive (the pair's sign is LAPACK's business in the
    // reference; the rank-1 filter u_0 v_0^T does not depend on it)
    for (int b = 0; b < n; ++b) {
        double* u = U + (size_t)b * L;
        double* v = V + (size_t)b * D;
        int j = 0;
        for (int i = 1; i < L; ++i)
            if (std::fabs(u[i]) > std::fabs(u[j])) j = i;
        if (u[j] < 0.0) {
            for (int i = 0; i < L; ++i) u[i] = -u[i];
            for (int i = 0; i < D; ++i) v[i] = -v[i];
        }
    }
    return done(PGL_OK);
}

int pgl_state(pgl_handle h, int n, const double* theta_n, const double* Weff_col, double* lam_out,
              double* I_net_out, double* I_stim_out)
{
    int rc = pgl_gibbs_prepare(h, n, theta_n, Weff_col);
    if (rc) return rc;
    const size_t bytes = (size_t)h->nT * 8;
    if (lam_out) {
        ENSURE(h->lam, bytes);
        hipLaunchKernelGGL(k_lam, dim3(1024), dim3(256), 0, h->stream, (const double*)h->Inet.p,
                           (const double*)h->Istim.p, theta_n[0], h->nlin, (double*)h->lam.p,
                           (long long)h->nT);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(lam_out, h->lam.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    if (I_net_out) HIPCHK(hipMemcpyAsync(I_net_out, h->Inet.p, bytes, hipMemcpyDeviceToHost, h->stream));
    if (I_stim_out) HIPCHK(hipMemcpyAsync(I_stim_out, h->Istim.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PGL_OK;
}

}  // extern "C"

#ifdef PGL_PROF
// dev builds only (tools/phase_profile.py): copy out the per-wave phase cycle sums of k_fused5 / 6 / 7
extern "C" int pgl_debug_prof(long long* out, int n_ll)
{
    const size_t bytes = std::min((size_t)n_ll * 8, sizeof(long long) * 2 * 4096 * 8 * 12);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pgl_prof), bytes, 0, hipMemcpyDeviceToHost));
    return PGL_OK;
}
extern "C" int pgl_debug_prof_ts(long long* out, int n_ll)
{
    const size_t bytes = std::min((size_t)n_ll * 8, sizeof(long long) * 4096 * 5);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pgl_prof_ts), bytes, 0, hipMemcpyDeviceToHost));
    return PGL_OK;
}
#endif
