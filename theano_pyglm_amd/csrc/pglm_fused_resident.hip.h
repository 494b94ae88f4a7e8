// fused ll+grad kernels on resident feature tiles: k_build_fimg, k_fused5 / 6 / 8 / 7
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// Resident feature tiles.  The basis-convolved spike history fS does not depend on the
// parameters: the reference builds it once per data set (LinearBasisImpulses.preprocess_data,
// impulse.py:114-130, data['fS']) and so does this path -- k_build_fimg writes, for every 16-bin
// time tile, the two half-width F tiles (columns [0,C0) and [C0,2*C0), stimulus columns
// included) as ready-made LDS images.  k_fused5 then streams them with LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no ds_write) instead of regenerating F from the spike
// events in every evaluation: 3.1 GB of HBM reads per C3 evaluation (0.4 ms of HBM time, hidden
// under 2.5 ms of f64 MFMA) replace ~0.7 ms of LDS/VALU-bound generation.
//   image = [16 rows][RSH doubles], RSH = C0 + 2 (bank spread of the forward A reads), padded to
//   a multiple of 1 KiB (one DMA instruction moves 64 lanes x 16 B, lane-linear).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double conv_one(const int2* __restrict__ spk, int lo, int hi, int tg,
                                           int R, const double* __restrict__ phi_b)
{
    double a = 0.0;
    for (int j = lo; j < hi; ++j) {
        const int2 e = spk[j];
        const int d = tg - e.x - 1;
        if (d >= 0 && d < R) a = fma((double)e.y, phi_b[d], a);
    }
    return a;
}

__host__ __device__ constexpr int pgl_img_rsh(int kt) { return kt * 16 + 2; }
// Row order of an image: time row i = 4q + r of the tile is stored at physical row 2q + 8 (r & 1) + (r >> 1).
// With the row stride = 2 (mod 32) doubles both A-fragment patterns of ds_read_b64 (32-lane groups, 64 banks) are
// then conflict-free: forward -- lanes (time i = 0..15) x (two consecutive columns) -- sees the 16 rows 2 double-banks
// apart whatever their order; backward -- lanes (two time rows 4q + {0,1} or 4q + {2,3}) x (16 consecutive columns) --
// needs the two rows 16 double-banks apart, i.e. 8 physical rows.  (Stored in time order the backward reads of lanes
// with r = 0 and r = 1 overlap in 14 of 16 banks: SQ_LDS_BANK_CONFLICT = one cycle per backward read.)
__host__ __device__ constexpr int pgl_img_row(int i) { return 2 * (i >> 2) + ((i & 1) << 3) + ((i >> 1) & 1); }
// backward: lane group grp reads time row 4q + grp at k-step q = physical row pgl_img_brow(grp) + 2q
__host__ __device__ constexpr int pgl_img_brow(int grp) { return ((grp & 1) << 3) + (grp >> 1); }
// time row stored at physical row p
__host__ __device__ constexpr int pgl_img_row_inv(int p) { return 4 * ((p & 7) >> 1) + ((p >> 3) & 1) + 2 * (p & 1); }
__host__ __device__ constexpr int pgl_img_bytes(int kt) { return ((16 * pgl_img_rsh(kt) * 8 + 1023) / 1024) * 1024; }

// grid = (nT16, 2); block = 256.  One block builds one image: part 0 = the first ktl k-tiles of
// feature columns ("L"), part 1 = the following kth k-tiles ("H"); a tile's L and H images are adjacent.
__global__ __launch_bounds__(256) void k_build_fimg(const int2* __restrict__ spk,
                                                    const int* __restrict__ wlo,
                                                    const int* __restrict__ whi,
                                                    const double* __restrict__ phi,
                                                    const double* __restrict__ fstim, long long nT,
                                                    int N, int B, int R, int Dstim, int ktl, int kth,
                                                    int tile0, unsigned char* __restrict__ Fimg,
                                                    int Nall, int np0, int DsAll, int ds0, int blk = 0)
{
    // N presynaptic neurons from np0 on and Dstim stimulus columns from ds0 on: the whole feature row, or one column slice of
    // a wide population (Nall / DsAll = the strides of the window tables and of fstim)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < B * R; i += blockDim.x) phiS[i] = phi[i];
    __syncthreads();
    const int tile = tile0 + blockIdx.x, part = blockIdx.y;
    if (blk) {
        // block form (k_fused8): ktl blocks of [16 bins][16 columns] per tile, columns XOR-swizzled (pgl_blk_off), no padding
        double* dstb = reinterpret_cast<double*>(Fimg + (size_t)blockIdx.x * ktl * 2048);
        const int Kimpb = N * B;
        for (int i = threadIdx.x; i < ktl * 256; i += blockDim.x) {
            const int kb = i >> 8, t = (i >> 4) & 15, pc = i & 15;
            const int col = kb * 16 + (pc ^ (2 * (t >> 1)));
            const long long tg = (long long)tile * 16 + t;
            double v = 0.0;
            if (col < Kimpb) {
                const int np = col / B, b = col - np * B;
                v = conv_one(spk, wlo[(size_t)tile * Nall + np0 + np], whi[(size_t)tile * Nall + np0 + np], (int)tg, R, phiS + b * R);
            } else if (col < Kimpb + Dstim) {
                v = (tg < nT) ? fstim[tg * DsAll + ds0 + (col - Kimpb)] : 0.0;
            }
            if (blk == 2) {
                // f32 blocks (k_fused8<.., F32 = 1>): lane l of the reading wave holds the doubles 2l, 2l + 1, 128 + 2l,
                // 128 + 2l + 1 of the block as one float4
                float* dstf = reinterpret_cast<float*>(Fimg + (size_t)blockIdx.x * ktl * 1024) + kb * 256;
                const int e = i & 255, hi2 = e >> 7, l = (e & 127) >> 1;
                dstf[4 * l + 2 * hi2 + (e & 1)] = (float)v;
            } else {
                dstb[i] = v;
            }
        }
        return;
    }
    const int kt = part ? kth : ktl;
    const int rsh = pgl_img_rsh(kt), cw = kt * 16, cbeg = part ? ktl * 16 : 0, Kimp = N * B;
    const size_t imgl = (size_t)pgl_img_bytes(ktl), imgh = (gridDim.y > 1) ? (size_t)pgl_img_bytes(kth) : 0;
    double* dst = reinterpret_cast<double*>(Fimg + (size_t)blockIdx.x * (imgl + imgh) + (part ? imgl : 0));
    const int nel = (int)((part ? imgh : imgl) / 8);
    for (int i = threadIdx.x; i < nel; i += blockDim.x) {
        const int tp = i / rsh, c = i - tp * rsh;        // physical row tp of the image holds time row t (pgl_img_row)
        const int t = pgl_img_row_inv(tp & 15);
        double v = 0.0;
        if (tp < 16 && c < cw) {
            const int col = cbeg + c;
            const long long tg = (long long)tile * 16 + t;
            if (col < Kimp) {
                const int np = col / B, b = col - np * B;
                v = conv_one(spk, wlo[(size_t)tile * Nall + np0 + np], whi[(size_t)tile * Nall + np0 + np], (int)tg, R, phiS + b * R);
            } else if (col < Kimp + Dstim) {
                v = (tg < nT) ? fstim[tg * DsAll + ds0 + (col - Kimp)] : 0.0;
            }
        }
        dst[i] = v;
    }
}

// whole image of KT k-tiles by LDS-DMA: 1 KiB pieces, piece c by wave c % 8
template <int KT>
__device__ __forceinline__ void pgl_dma_half(const unsigned char* __restrict__ gimg, unsigned char* lds_dst,
                                             const int wave, const int lane)
{
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    constexpr int NCH = pgl_img_bytes(KT) / 1024;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += 8) {
        const int c = c0 + wave;
        if (c < NCH)
            __builtin_amdgcn_global_load_lds((gvoid*)(gimg + (size_t)c * 1024 + lane * 16),
                                             (lvoid*)(lds_dst + (size_t)c * 1024), 16, 0, 0);
    }
}

// whole image of KT k-tiles by LDS-DMA over NWV waves (1 KiB pieces, piece c by wave c % NWV)
template <int KT, int NWV>
__device__ __forceinline__ void pgl_dma_img(const unsigned char* __restrict__ gimg, unsigned char* lds_dst,
                                            const int wave, const int lane)
{
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    constexpr int NCH = pgl_img_bytes(KT) / 1024;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += NWV) {
        const int c = c0 + wave;
        if (c < NCH)
            __builtin_amdgcn_global_load_lds((gvoid*)(gimg + (size_t)c * 1024 + lane * 16),
                                             (lvoid*)(lds_dst + (size_t)c * 1024), 16, 0, 0);
    }
}

// one round of an image DMA: round j moves the 1 KiB pieces 8j .. 8j+7, one per wave
template <int KT>
__device__ __forceinline__ void pgl_dma_round(const unsigned char* __restrict__ gimg, unsigned char* lds_dst,
                                              const int round, const int wave, const int lane)
{
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    constexpr int NCH = pgl_img_bytes(KT) / 1024;
    const int c = round * 8 + wave;
    if (c < NCH) {
        // wave-uniform piece address in SGPRs + the lane's 16 bytes: saddr form, no 64-bit VALU arithmetic
        const unsigned char* gs = gimg + (size_t)c * 1024;
        asm volatile("" : "+s"(gs));
        __builtin_amdgcn_global_load_lds((gvoid*)(gs + lane * 16), (lvoid*)(lds_dst + (size_t)c * 1024), 16, 0, 0);
    }
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 5: the two-pass structure of k_fused3 on resident feature tiles.
// The feature columns are cut into an "L" part of KTL k-tiles and an "H" part of KTH (KTL <= KTH:
// pass 1 also carries the forward ring, the accumulators and the epilogue in its 256 registers, so
// it keeps the smaller share of G).
//   pass 1, per tile: [L_i | H_i in LDS] forward over both parts | barrier | epilogue on the
//           accumulator registers (both waves of a SIMD side by side) | barrier | r to HBM | backward
//           for the L columns from L_i, the DMA of L_{i+1} (third buffer) and H_{i+1} (over H_i) issued
//           between its MFMAs | wait for the DMA | barrier.
//   pass 2, per tile: wait for H_i | barrier | backward for the H columns with r read back, the DMA of
//           H_{i+1} (other buffer) issued between its MFMAs.
// No event windows, no basis tables, no staging: the waves only issue DMA, LDS reads and MFMAs.
// What the phase timeline (tools/phase_profile.py) taught:
//   * one LDS-DMA piece costs ~32 cycles of the CU's address path and blocks the issuing wave: a
//     burst of 84 pieces behind a barrier idles the MFMA pipes for ~2.7k cycles per tile.  Issued
//     between the backward MFMAs, waves 0-3 and their SIMD partners 4-7 half a period apart, a SIMD
//     always has one wave feeding the pipe.  (Not in the forward loop: a DMA in flight sits in front
//     of the Wmat ring loads in the in-order vmcnt queue.)
//   * MFMA arbitration between the two waves of a SIMD goes by priority, then age: at equal priority
//     the older wave leaves every loop thousands of cycles early and its partner, alone, cannot keep
//     the pipe full.  Waves 4-7 lead the first half of every loop (s_setprio), waves 0-3 the second.
//   * f64 VALU work beside a partner's back-to-back MFMAs gets one issue slot per 64-cycle MFMA:
//     the two epilogues of a SIMD run side by side between two barriers, four elements per lane in
//     a fixed instruction order (pgl_rate4).
// ---------------------------------------------------------------------------
// XIN = 1 (pass 1): the currents start from the slab p.Xbuf[tile - tile0][post tile][r][lane] -- the stimulus current of a
// separable stimulus (k_sepf_fwd) -- which pass 1 then overwrites with the residuals as always
// HLP = 1: blocks of ONE to SIX post tiles leave waves without a tile of their own.  Five or six (N = 65 .. 96) leave three or two of the
// eight waves without a tile of their own, and two SIMDs with two tiles each: the idle waves take over part of the work of
// the tiles of a doubly loaded SIMD -- in pass 1 the second half of a tile's forward k-steps (the partial currents reach
// the tile's own wave through LDS, in front of the barrier that closes the forward phase anyway), in pass 2 the second half
// of its k-tiles (own G registers, own partials).  Five tiles: waves 5 / 6 help tiles 0 / 4 (both on SIMD 0); six tiles:
// waves 6 / 7 help tiles 0 / 1.  Per tile step the busiest SIMD then carries 1.5 forward passes instead of 2
// (five tiles: 1 + the two L backward passes).  Tiles without a helper compute exactly what HLP = 0 computes.
template <int KTL, int KTH, int PASS, int XIN = 0, int PART = 0, int HLP = 0>
__global__ __launch_bounds__(512, 2) void k_fused5(const FusedParams p)
{
    static_assert(!HLP || (KTL >= 2 && KTH >= 2), "helper waves: at least two k-tiles per column part");
    constexpr int TT = 16, NW = 8;
    constexpr int KT_ALL = KTL + KTH;
    constexpr int KS_ALL = 4 * KT_ALL;
    constexpr int KSL = 4 * KTL;                 // k-steps of the L part
    constexpr int RSL = pgl_img_rsh(KTL), RSH = pgl_img_rsh(KTH);
    constexpr int IMGL = pgl_img_bytes(KTL), IMGH = pgl_img_bytes(KTH);
    // pass 2 works on the H part of the column split (PART = 0) or, for the earlier column slices of a wide population whose
    // forward-only pass 1 left their L columns without a gradient, on the L part (PART = 1)
    constexpr int KTP = PART ? KTL : KTH;        // k-tiles of the part pass 2 walks
    constexpr int IMGP = PART ? pgl_img_bytes(KTL) : pgl_img_bytes(KTH);
    constexpr size_t OFFP = PART ? 0 : (size_t)pgl_img_bytes(KTL);   // its offset inside a tile's image pair
    constexpr int KTG = (PASS == 1) ? KTL : KTP; // k-tiles of G this pass accumulates
    constexpr bool FWO = (PASS == 1) && (XIN >= 2);   // forward only: raw currents to the slab (XIN = 3: added to what is there)
    constexpr bool XRD = (XIN == 1) || (XIN == 3);    // the currents start from the slab
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PGL_PROF_ENTRY

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nPB = (p.nPT + NW - 1) / NW;
    const int pb = p.pb_major ? (int)blockIdx.x / p.nChunks : (int)blockIdx.x % nPB;
    const int chunk = p.pb_major ? (int)blockIdx.x % p.nChunks : (int)blockIdx.x / nPB;
    const int pt = pb * NW + wave;
    const bool active = pt < p.nPT;
    // role of the wave: 0 its own tile in full (or none), 1 its own tile with a helper, 2 helper of tile wpt (slot hslot of
    // the exchange area)
    int role = 0, wpt = pt, hslot = 0;
    if constexpr (HLP != 0) {
        const int nb = (p.nPT - pb * NW < NW) ? p.nPT - pb * NW : NW;
        if (nb == 5) {
            role = (wave == 0 || wave == 4) ? 1 : ((wave == 5 || wave == 6) ? 2 : 0);
            wpt = pb * NW + ((wave == 5) ? 0 : ((wave == 6) ? 4 : wave));
            hslot = (wave == 4 || wave == 6) ? 1 : 0;
        } else if (nb == 6) {
            role = (wave <= 1) ? 1 : ((wave >= 6) ? 2 : 0);
            wpt = pb * NW + ((wave >= 6) ? wave - 6 : wave);
            hslot = (wave == 1 || wave == 7) ? 1 : 0;
        } else if (nb <= 4) {
            // a light block (one to four tiles: 33 .. 64 neurons against a long row, the last block of a wide population):
            // every tile has a helper, on a SIMD without a tile of its own where there is one
            //   4 tiles: waves 4 .. 7 help tiles 0 .. 3;  3: waves 3, 7, 5 help tiles 0, 1, 2;  2: waves 2, 3;  1: wave 1
            int ht = -1;
            if (nb == 4) ht = (wave >= 4) ? wave - 4 : -1;
            else if (nb == 3) ht = (wave == 3) ? 0 : ((wave == 7) ? 1 : ((wave == 5) ? 2 : -1));
            else if (nb == 2) ht = (wave == 2 || wave == 3) ? wave - 2 : -1;
            else ht = (wave == 1) ? 0 : -1;
            role = (wave < nb) ? 1 : ((ht >= 0) ? 2 : 0);
            wpt = pb * NW + ((ht >= 0) ? ht : wave);
            hslot = (ht >= 0) ? ht : ((wave < nb) ? wave : 0);
        }
        role = __builtin_amdgcn_readfirstlane(role);
        wpt = __builtin_amdgcn_readfirstlane(wpt);
        hslot = __builtin_amdgcn_readfirstlane(hslot);
    }
    const bool helper = (HLP != 0) && role == 2;
    const bool works = active || helper;          // the wave runs MFMAs (on tile wpt)
    // forward k-steps / pass-2 k-tiles that stay with a helped tile's own wave
    constexpr int QS = ((4 * (KTL + KTH)) / 2 + 3) / 4 * 4;
    constexpr int KTPH = ((PART ? KTL : KTH) + 1) / 2;

    // pass 1: L buffers at 0 and IMGL, H buffer behind them; pass 2: two H buffers
    unsigned char* buf0 = smem;
    unsigned char* buf2 = smem + IMGL;           // pass 1 only
    unsigned char* buf1 = smem + ((PASS == 1) ? 2 * IMGL : IMGP);
    double* Cs = reinterpret_cast<double*>(smem + ((PASS == 1) ? 2 * IMGL + IMGH : 2 * IMGP));
    if (tid < 32) Cs[tid] = PGL_C[tid];


    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);

    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;
    double* const rslab = p.Xbuf + ((size_t)(works ? wpt : 0)) * 256 + lane;
    const size_t rstride = (size_t)p.nPT * 256;
    // images are indexed relative to the first tile they were built for (p.img_tile0)
    const unsigned char* __restrict__ fimg = p.Fimg - (size_t)p.img_tile0 * (IMGL + IMGH);
    constexpr size_t IMGS = (size_t)IMGL + IMGH;

    // backward over one image of KTG k-tiles.  The DMA rounds of up to two images of the next tile
    // (NR0 rounds g0 -> l0, then NR1 rounds g1 -> l1) go out between the MFMAs.
    constexpr int NRL = (IMGL / 1024 + 7) / 8, NRH = (IMGH / 1024 + 7) / 8;
    constexpr int NR0 = (PASS == 1 || PART) ? NRL : NRH, NR1 = (PASS == 1) ? NRH : 0;
    constexpr int RSG = (PASS == 1 || PART) ? RSL : RSH;
    // (k0c, nkc, krt: the k-tiles [K0 + krt, K0 + krt + NK) of the image, accumulated in G[0 .. NK) -- all KTG of them, or
    //  the share of a helped tile's own wave / of its helper: ONE instantiation for both, the helper's at a runtime offset)
    auto bwd_part = [&](auto k0c, auto nkc, auto& G, const int krt, const unsigned char* Fb, const double (&rq)[4],
                        const unsigned char* g0, unsigned char* l0, const unsigned char* g1, unsigned char* l1,
                        const bool dma) {
        constexpr int K0 = decltype(k0c)::value, NK = decltype(nkc)::value;
        const double* fb = reinterpret_cast<const double*>(Fb) + pgl_img_brow(grp) * RSG + col + 16 * (K0 + krt);
        constexpr int NS = 4 * NK;
        constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
        constexpr int NRT = NR0 + NR1;
        constexpr int DSFULL = NS / NRT;
        constexpr int DSCAP = (PGL_DS1 > 0) ? PGL_DS1 : NS;
        constexpr int DSTEP = (NS >= 2 * NRT) ? ((DSFULL < DSCAP) ? DSFULL : DSCAP) : 0;   // MFMAs between rounds
        double ar[PD];
#pragma unroll
        for (int s = 0; s < PD; ++s) ar[s] = pgl_lds_f64(fb + (2 * (s / NK)) * RSG + 16 * (s % NK));
        auto round = [&](const int j) {
            if (j < NR0) {
                pgl_dma_round<(PASS == 1 || PART) ? KTL : KTH>(g0, l0, j, wave, lane);
            } else {
                pgl_dma_round<KTH>(g1, l1, j - NR0, wave, lane);
            }
        };
        if (DSTEP == 0 && dma) {
#pragma unroll
            for (int j = 0; j < NRT; ++j) round(j);
        }
        const int phase = (wave < 4) ? ((DSTEP > 1) ? DSTEP / 2 - 1 : 0) : DSTEP - 1;
        if (PGL_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (PGL_PRIO && s == NS / 2 && wave >= 4) __builtin_amdgcn_s_setprio(0);
            const double a = ar[s % PD];
            if (s + PD < NS) ar[s % PD] = pgl_lds_f64(fb + (2 * ((s + PD) / NK)) * RSG + 16 * ((s + PD) % NK));
            G[s % NK] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rq[s / NK], G[s % NK], 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            if (DSTEP > 0) {
                constexpr int DS = (DSTEP > 0) ? DSTEP : 1;
                const int j = s / DS;
                const int ph = s % DS;
                if ((ph == DS - 1 || (DS > 1 && ph == DS / 2 - 1)) && j < NRT) {
                    if (ph == phase && dma) round(j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    if constexpr (PASS == 1) {
        d4_t G[KTL];
#pragma unroll
        for (int kt = 0; kt < KTL; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
        double ll_acc = 0.0, gb_acc = 0.0;
        // padding lanes (neurons >= npost) get a benign current: they must not push their wave out of
        // the epilogue's series regime; nothing they produce is ever read
        const double bias_l = valid_n ? p.bias[nloc] : (p.nlin == 1 ? 30.0 : 0.0);
        const double* __restrict__ wrow = p.Wfrag + (size_t)(works ? wpt : 0) * KS_ALL * 64;
        double* const wscratch = reinterpret_cast<double*>(smem + 2 * IMGL + IMGH + 256) + wave * 192;   // spike compaction
        // HLP: partial currents of the (up to four) helpers, [4][4][64], behind the spike scratch
        double* const Xh = reinterpret_cast<double*>(smem + 2 * IMGL + IMGH + 256) + NW * 192 + hslot * 256 + lane;
        // prologue: L and H of the first tile
        if (tile_beg < tile_end) {
            pgl_dma_half<KTL>(fimg + (size_t)tile_beg * IMGS, buf0, wave, lane);
            pgl_dma_half<KTH>(fimg + (size_t)tile_beg * IMGS + IMGL, buf1, wave, lane);
        }
        // spike counts of this lane's four elements (rows grp + 4r of neuron nglob): requested one tile
        // ahead -- issued at the head of a tile these byte loads (HBM misses) would sit in front of the
        // Wmat ring in the in-order vmcnt queue and delay every forward pass
        unsigned scb[4] = {0u, 0u, 0u, 0u}, scn[4] = {0u, 0u, 0u, 0u};
        // S is zero-padded to whole tiles (upload_spikes): one pointer per lane, advanced by a tile per
        // request; the four rows of a lane are 4 * Nall bytes apart
        const uint8_t* cnt_ptr = p.S + ((size_t)tile_beg * TT + grp) * p.Nall + nglob;
        const int cnt_r1 = 4 * p.Nall, cnt_tile = TT * p.Nall;
        auto load_counts = [&](unsigned (&dst)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = cnt_ptr[r * cnt_r1];
            cnt_ptr += cnt_tile;
        };
        if (tile_beg < tile_end) load_counts(scn);
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): the DMAs have landed
        __syncthreads();
        PGL_PROF_DECL
        for (int tile = tile_beg; tile < tile_end; ++tile) {
            const int t0 = tile * TT;
            const int par = (tile - tile_beg) & 1;
            const unsigned char* Lb = par ? buf2 : buf0;     // L alternates buf0 / buf2, H lives in buf1
            unsigned char* Ln = par ? buf0 : buf2;
            const bool more = tile + 1 < tile_end;
#pragma unroll
            for (int r = 0; r < 4; ++r) scb[r] = scn[r];
            // ---- forward over both parts ----
            d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
            d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
            // forward k-steps [Q0, Q1) of tile wpt: all of them, or the share of a helped tile's own wave / of its helper
            auto forward = [&](auto q0c, auto q1c) {
                constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value, NQ = Q1 - Q0;
                constexpr int PW2 = (NQ / 2 < PGL_PW / 2) ? NQ / 2 : PGL_PW / 2;
                static_assert(Q0 % 2 == 0 && NQ % 2 == 0 && NQ >= 2, "whole fragment pairs");
                const double* faL = reinterpret_cast<const double*>(Lb) + pgl_img_row(col) * RSL + grp;
                const double* faH = reinterpret_cast<const double*>(buf1) + pgl_img_row(col) * RSH + grp;
                const double* wr_s = wrow;
                asm volatile("" : "+s"(wr_s));
                constexpr int PA = 4;
                pgl_d2 wr[PW2];
                double ar[PA];
                auto afrag = [&](const int s) -> double {
                    return (s < KSL) ? pgl_lds_f64(faL + 4 * s) : pgl_lds_f64(faH + 4 * (s - KSL));
                };
                // scalar bases of the Wmat fragment stream, one per 4 KB (four pairs of k-steps)
                pgl_glb_cd2p wr_base[KS_ALL / 8 + 1];
#pragma unroll
                for (int b4 = Q0 / 8; b4 < (Q1 + 7) / 8; ++b4) {
                    const double* bs = wr_s + (size_t)b4 * 512;
                    asm volatile("" : "+s"(bs));
                    wr_base[b4] = (pgl_glb_cd2p)bs;
                }
#pragma unroll
                for (int q = 0; q < PW2; ++q) {
                    const int pair = Q0 / 2 + q;
                    wr[q] = wr_base[pair / 4][(pair % 4) * 64 + lane];
                }
#pragma unroll
                for (int q = 0; q < PA; ++q) ar[q] = afrag(Q0 + q);
                if (PGL_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int q = Q0; q < Q1; ++q) {
                    if (PGL_PRIO && q == Q0 + NQ / 2 && wave >= 4) __builtin_amdgcn_s_setprio(0);
                    const double a = ar[(q - Q0) % PA];
                    const double b = (q & 1) ? wr[((q - Q0) / 2) % PW2].y : wr[((q - Q0) / 2) % PW2].x;
                    if (q + PA < Q1) ar[(q - Q0) % PA] = afrag(q + PA);
                    if ((q & 1) && ((q - Q0) / 2 + PW2 < NQ / 2)) {
                        // scalar base + lane offset + small immediate: the base moves on in SGPRs every four
                        // fragment pairs (4 KB), no 64-bit VALU address arithmetic
                        const int pair = q / 2 + PW2;      // compile-time (unrolled)
                        wr[((q - Q0) / 2) % PW2] = wr_base[pair / 4][(pair % 4) * 64 + lane];
                    }
                    if (q & 1)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                    else
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                    if (((q - Q0) & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (works && !PGL_DBG(8)) {
                if constexpr (HLP != 0) {
                    if (role == 1) forward(std::integral_constant<int, 0>{}, std::integral_constant<int, QS>{});
                    else if (role == 2) forward(std::integral_constant<int, QS>{}, std::integral_constant<int, KS_ALL>{});
                    else forward(std::integral_constant<int, 0>{}, std::integral_constant<int, KS_ALL>{});
                } else {
                    forward(std::integral_constant<int, 0>{}, std::integral_constant<int, KS_ALL>{});
                }
            }
            if constexpr (HLP != 0) {
                if (helper) {                              // the partial currents, in front of the barrier below
#pragma unroll
                    for (int r = 0; r < 4; ++r) Xh[r * 64] = acc0[r] + acc1[r];
                }
            }
            PGL_PROF_MARK(0);
            const bool do_bwd = !FWO && active && p.want_grad && !PGL_DBG(16);
            double xin[XRD ? 4 : 1];
            if constexpr (XRD) {                          // requested in front of the barrier: its wait hides the latency
                const double* xs_ = rslab + (size_t)(tile - p.tile0) * rstride;
#pragma unroll
                for (int r = 0; r < 4; ++r) xin[r] = xs_[r * 64];
            }
            // every wave is done with H_i (buf1): the next tile's DMA may overwrite it.  The barrier also lines
            // the waves up for the epilogue.
            __syncthreads();
            PGL_PROF_MARK(2);
            double xh[(HLP != 0) ? 4 : 1];                // HLP: the helper's share of a helped tile's currents (else 0)
            if constexpr (HLP != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[r] = (role == 1) ? Xh[r * 64] : 0.0;
            }
            // ---- epilogue on the accumulator registers ----
            double rr[4];
            if constexpr (FWO) {
                // forward only (an earlier column slice of a wide population): the partial currents go to the slab
                if (active) {
                    double* rs = rslab + (size_t)(tile - p.tile0) * rstride;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double x = acc0[r] + acc1[r];
                        if constexpr (HLP != 0) x += xh[r];
                        if constexpr (XRD) x += xin[r];
                        rs[r * 64] = x;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = 0.0;
            } else if (active) {
                bool done = false;
                if (PGL_ENE == 4 && !PGL_DBG(4) && (long long)t0 + TT <= p.t_hi) {
                    // whole tile inside the evaluated range: four elements at a time, fixed order
                    double xs[4], term4 = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (HLP != 0) xs[r] = bias_l + ((acc0[r] + acc1[r]) + xh[r]);
                        else xs[r] = bias_l + (acc0[r] + acc1[r]);
                    }
                    if constexpr (XRD) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xs[r] += xin[r];
                    }
                    const double* cg = PGL_C;
                    asm volatile("" : "+s"(cg));           // keeps the scalar loads inside the tile loop
                    done = pgl_rate4(xs, scb, p.nlin | p.epi64, p.dt, (pgl_k_cdp)cg, wscratch, lane, term4, rr PGL_PROF_PASS);
                    if (done) {
                        ll_acc += term4;                    // lanes of padding neurons are never read back
#pragma unroll
                        for (int r = 0; r < 4; ++r) gb_acc += rr[r];
                    }
                }
                constexpr int ENE = (PGL_ENE == 4) ? 2 : PGL_ENE;
                if (!done) {
#pragma unroll
                    for (int h2 = 0; h2 < 4 / ENE; ++h2) {
                        double xe[ENE], se[ENE], terme[ENE], rese[ENE];
                        bool vte[ENE];
#pragma unroll
                        for (int e = 0; e < ENE; ++e) {
                            const int r = ENE * h2 + e;
                            if constexpr (HLP != 0) xe[e] = bias_l + ((acc0[r] + acc1[r]) + xh[r]);
                            else xe[e] = bias_l + (acc0[r] + acc1[r]);
                            if constexpr (XRD) xe[e] += xin[r];
                            se[e] = (double)scb[r];
                            const long long tg = (long long)t0 + grp + 4 * r;
                            vte[e] = valid_n && (tg < p.t_hi);
                        }
                        if PGL_DBG(4) {
#pragma unroll
                            for (int e = 0; e < ENE; ++e) {
                                terme[e] = xe[e] * se[e];
                                rese[e] = xe[e] - se[e];
                            }
                        } else {
                            pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                            asm volatile("" : "+v"(Cl));
                            pgl_rate_terms_n<ENE>(xe, se, p.nlin | p.epi64, p.dt, terme, rese, Cl);
                        }
#pragma unroll
                        for (int e = 0; e < ENE; ++e) {
                            const double res = vte[e] ? rese[e] : 0.0;
                            rr[ENE * h2 + e] = res;
                            ll_acc += vte[e] ? terme[e] : 0.0;
                            gb_acc += res;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = 0.0;
            }
            PGL_PROF_MARK(1);
            if (PGL_EBAR) __syncthreads();     // both epilogues of a SIMD end before any backward MFMA
            if (more) load_counts(scn);                   // retired by the closing vmcnt(0) of this tile
            if (!do_bwd && more) {
                pgl_dma_half<KTL>(fimg + (size_t)(tile + 1) * IMGS, Ln, wave, lane);
                pgl_dma_half<KTH>(fimg + (size_t)(tile + 1) * IMGS + IMGL, buf1, wave, lane);
            }
            PGL_PROF_MARK(3);
            if (do_bwd) {                                 // L_{i+1}, H_{i+1} go out between the MFMAs
                double* rs = rslab + (size_t)(tile - p.tile0) * rstride;
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[r * 64] = rr[r];
                bwd_part(std::integral_constant<int, 0>{}, std::integral_constant<int, KTL>{}, G, 0, Lb, rr,
                         fimg + (size_t)(tile + 1) * IMGS, Ln, fimg + (size_t)(tile + 1) * IMGS + IMGL, buf1, more);
            }
            PGL_PROF_MARK(4);
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): L_{i+1}, H_{i+1} landed, r stored
            PGL_PROF_MARK(5);
            __syncthreads();
            PGL_PROF_MARK(6);
        }
        PGL_PROF_STORE(1);
        if (active && !FWO) {
            pgl_store_ll(p, chunk, pt, 0, 1, lane, ll_acc, gb_acc);
        }
        if (active && p.want_grad && !FWO) {
            double* gp = pgl_gpart(p.Gpart, pt, KT_ALL, 0, p.nChunks, chunk, lane);
            const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
            for (int kt = 0; kt < KTL; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
            }
        }
    } else {
        // =============================== pass 2 ===============================
        // nkc k-tiles of the part from k-tile krt on, their G in registers: the whole part (NK = KTP), or -- HLP, a helped
        // tile -- its first KTPH k-tiles (the tile's own wave) / its last KTPH (the helper; with an odd KTP the middle
        // k-tile is done twice and the helper's copy, kskip = 1, dropped at the write-out).  One tile loop per form: the
        // accumulators of the two forms are never alive together.
        auto pass2 = [&](auto nkc, const int krt, const int kskip) {
            constexpr int NK = decltype(nkc)::value;
            d4_t G[NK];
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
            double rv[4] = {0.0, 0.0, 0.0, 0.0}, rn[4] = {0.0, 0.0, 0.0, 0.0};
            if (tile_beg < tile_end) {
                pgl_dma_half<KTP>(fimg + (size_t)tile_beg * IMGS + OFFP, buf0, wave, lane);
                if (works) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rn[r] = rslab[(size_t)(tile_beg - p.tile0) * rstride + r * 64];
                }
            }
            PGL_PROF_DECL
            for (int tile = tile_beg; tile < tile_end; ++tile) {
                const int par = (tile - tile_beg) & 1;
                const unsigned char* Hb = par ? buf1 : buf0;
                unsigned char* Hn = par ? buf0 : buf1;
                __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): H_i and r_i are here
                PGL_PROF_MARK(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[r] = rn[r];
                __syncthreads();                              // ... for every wave; H_{i-1}'s buffer is free
                PGL_PROF_MARK(1);
                const bool more = tile + 1 < tile_end;
                const bool do_bwd = works && !PGL_DBG(16);
                if (more) {
                    if (!do_bwd) pgl_dma_half<KTP>(fimg + (size_t)(tile + 1) * IMGS + OFFP, Hn, wave, lane);
                    if (works) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) rn[r] = rslab[(size_t)(tile + 1 - p.tile0) * rstride + r * 64];
                    }
                }
                PGL_PROF_MARK(2);
                if (do_bwd)
                    bwd_part(std::integral_constant<int, 0>{}, nkc, G, krt, Hb, rv, fimg + (size_t)(tile + 1) * IMGS + OFFP, Hn,
                             nullptr, nullptr, more);
                PGL_PROF_MARK(3);
            }
            PGL_PROF_STORE(2);
            if (works) {
                double* gp = pgl_gpart(p.Gpart, wpt, KT_ALL, (PART ? 0 : KTL) + krt, p.nChunks, chunk, lane);
                const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    if (HLP == 0 || kt >= kskip) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
                    }
                }
            }
        };
        if constexpr (HLP != 0) {
            if (role != 0) pass2(std::integral_constant<int, KTPH>{}, (role == 2) ? KTP - KTPH : 0, (role == 2) ? 2 * KTPH - KTP : 0);
            else pass2(std::integral_constant<int, KTP>{}, 0, 0);
        } else {
            pass2(std::integral_constant<int, KTP>{}, 0, 0);
        }
    }
    PGL_PROF_EXIT;
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 6: the K-split scheme of k_fused2 (PTW post tiles x KSPLIT slices of
// the feature columns over the 8 waves of a workgroup, ONE pass, G in registers) on RESIDENT feature
// tiles -- for the populations whose whole feature row is short (N*B + Dstim <= ~320 columns: C1, C2,
// C5, masked subsets of them), where k_fused5's one-wave-per-post-tile layout would leave most waves
// idle and k_fused2 spends more time regenerating features than multiplying them.
//   image of one 16-bin tile = [16][RS] f64, RS = 16*KT_ALL + 2, padded to 1 KiB (k_build_fimg, one part)
//   a STEP covers MT consecutive tiles (MT = 2 when the images are small): the three workgroup barriers
//   of the scheme (images landed | partial currents exchanged | residuals exchanged) are paid once per
//   step; the images of step i+1 arrive by LDS-DMA in the other buffer while step i computes.
// Workgroups of NW = 8 waves, or NW = 4 waves (two workgroups per CU: one group's barrier waits are
// filled by the other's MFMAs) for post blocks of one or two tiles.
// Same partial layout as k_fused2 (k_finalize / k_finalize_ll reduce it).
// ---------------------------------------------------------------------------
template <int KTW, int PTW, int MT, int NW, int DB = 1>
__global__ __launch_bounds__(NW * 64, 2) void k_fused6(const FusedParams p)
{
    constexpr int TT = 16;
    constexpr int KSPLIT = NW / PTW;
    constexpr int KSW = KTW * 4;
    constexpr int KT_ALL = KTW * KSPLIT;
    constexpr int KS_ALL = KSW * KSPLIT;
    constexpr int RS = pgl_img_rsh(KT_ALL);
    constexpr int IMG = pgl_img_bytes(KT_ALL);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PGL_PROF_ENTRY

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptl = wave % PTW;
    const int ksl = wave / PTW;
    const int nPB = (p.nPT + PTW - 1) / PTW;
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
    const int pt = pb * PTW + ptl;
    const bool active = pt < p.nPT;

    // DB = 0: ONE image buffer per workgroup, for rows too long to hold twice (a 16-bin tile of 640 columns is 81 KB:
    // the narrow post blocks of a wide population) -- the next step's images are requested behind a fourth barrier,
    // when the backward loop has read the current ones
    unsigned char* bufs = smem;                                          // [DB ? 2 : 1][MT][IMG]
    double* Xp = reinterpret_cast<double*>(smem + (size_t)(DB ? 2 : 1) * MT * IMG); // [MT][NW][4][64] partial currents
    // residuals [MT][PTW][4][64]: they take the place of the k-slice-0 partials -- element (m, ptl, r, lane) of both
    // is read (partial) and then written (residual) by the one wave that owns register r in the epilogue, and the
    // next step's partials are only written behind the "landed" barrier, when every wave has read its residuals
    double* Rb = Xp;
    constexpr int RBS = NW;                                              // residual tile stride in 2 KB slots
    double* Cs = Xp + (size_t)MT * NW * 256;                             // [32] math constants
    double* const wscratch = Cs + 32 + wave * 48;                         // per wave: spike compaction of the epilogue (CAP = 16:
                                                                          // 3 x 16 doubles; LDS is what limits the workgroups per CU)
    if (tid < 32) Cs[tid] = PGL_C[tid];

    d4_t G[KTW];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);
    // padding lanes (neurons >= npost) get a benign current: they must not push their wave out of the epilogue's
    // fast regime; nothing they produce is ever read
    const double bias_l = valid_n ? (p.theta ? p.theta[(size_t)nloc * p.P] : p.bias[nloc]) : (p.nlin == 1 ? 30.0 : 0.0);
    const double* __restrict__ wrow =
        p.Wfrag + ((size_t)(active ? pt : 0) * KS_ALL + (size_t)ksl * KSW) * 64;
    const int kcol0 = ksl * KTW * 16;
    // epilogue ownership as in k_fused2: the 256 elements of a post tile are split over its KSPLIT waves
    constexpr int EPW = (KSPLIT >= 4) ? 1 : 4 / KSPLIT;
    int er[EPW];
#pragma unroll
    for (int e = 0; e < EPW; ++e) er[e] = (KSPLIT == 8) ? (ksl >> 1) : (KSPLIT == 4) ? ksl : ksl * EPW + e;
    const bool emine = (KSPLIT == 8) ? ((lane >> 5) == (ksl & 1)) : true;

    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;
    const unsigned char* __restrict__ fimg = p.Fimg - (size_t)p.img_tile0 * IMG;

    auto dma_step = [&](const int tile0s, unsigned char* dst) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (tile0s + m < tile_end)
                pgl_dma_img<KT_ALL, NW>(fimg + (size_t)(tile0s + m) * IMG, dst + (size_t)m * IMG, wave, lane);
    };
    if (tile_beg < tile_end) dma_step(tile_beg, bufs);
    // this wave's slice of Wmat stays in registers for the whole chunk (KSW <= 40 fragments: the forward
    // loop of a short slice would otherwise wait for its first L2 loads in every tile)
    double wreg[KSW];
#pragma unroll
    for (int s = 0; s < KSW; ++s) wreg[s] = 0.0;
    if (active && p.theta) {
        pgl_wfrag_direct<KSW>(p, ksl * KSW, grp, nloc, nglob, valid_n, wreg);
    } else if (active) {
        const pgl_d2* wr2 = reinterpret_cast<const pgl_d2*>(wrow);
#pragma unroll
        for (int s2 = 0; s2 < KSW / 2; ++s2) {
            const pgl_d2 v = wr2[s2 * 64 + lane];
            wreg[2 * s2] = v.x;
            wreg[2 * s2 + 1] = v.y;
        }
    }

    // post-synaptic counts of the elements this wave owns in the epilogue, requested one step ahead (a
    // short step -- two tiles of a 160-column row are 1 500 MFMA cycles -- is over before an HBM miss returns)
    // (S is zero-padded to whole tiles, upload_spikes: one pointer per owned accumulator register, advanced by a step
    //  per request -- no 64-bit multiplies in the tile loop; a last partial step reads at most MT - 1 tiles past the
    //  chunk, inside the padded array as long as the tile exists, hence the clamp on the tile index only)
    unsigned scn[MT * EPW];
    const uint8_t* cptr[EPW];
    const int last_tile = p.nT16 - 1;
#pragma unroll
    for (int e = 0; e < EPW; ++e)
        cptr[e] = p.S + ((size_t)(tile_beg < last_tile ? tile_beg : last_tile) * TT + grp + 4 * er[e]) * p.Nall + nglob;
    const size_t ctile = (size_t)TT * p.Nall;
    auto load_counts = [&](const int tile0s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const size_t moff = (tile0s + m <= last_tile) ? (size_t)m * ctile : 0;
#pragma unroll
            for (int e = 0; e < EPW; ++e) scn[m * EPW + e] = cptr[e][moff];
        }
#pragma unroll
        for (int e = 0; e < EPW; ++e) cptr[e] += (size_t)MT * ctile;
    };
    load_counts(tile_beg);
    // One image buffer and a short K slice per wave (DB = 0, KTW <= 5: one post tile of a 640-column row): the wave also
    // fetches its BACKWARD fragments of the tile while the image is there (20 doubles), so that the buffer is free from
    // the "partials" barrier on and the next tile's image travels during the epilogue and the backward MFMAs
    constexpr bool BREG = (DB == 0 && MT == 1 && KTW <= 5);
    double fbreg[BREG ? 4 * KTW : 1], fareg[BREG ? KSW : 1];
    int par = 0;
    PGL_PROF_DECL
    for (int tile = tile_beg; tile < tile_end; tile += MT, par ^= 1) {
        const unsigned char* cur = bufs + (size_t)(DB ? par : 0) * MT * IMG;
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's pieces of the step landed
        PGL_PROF_MARK(0);
        __syncthreads();                                  // ... everybody's; the other buffer is free
        PGL_PROF_MARK(1);
        double sc[MT * EPW];
        unsigned scu[MT * EPW];
#pragma unroll
        for (int i = 0; i < MT * EPW; ++i) {
            scu[i] = scn[i];
            sc[i] = (double)scn[i];
        }
        // the images of the next step: PWV pieces of 1 KiB per wave, one every DS forward MFMAs (PGL_DMA_IL), the
        // whole burst up front for waves without MFMA work
        const bool more = tile + MT < tile_end;
        unsigned char* const nxt = bufs + (size_t)(DB ? (par ^ 1) : 0) * MT * IMG;
        constexpr int NCH = IMG / 1024, PPI = (NCH + NW - 1) / NW, PWV = MT * PPI, NMF = MT * KSW;
        constexpr int DS = (DB && PGL_DMA_IL && NMF >= PWV) ? NMF / PWV : 0;
        auto piece = [&](const int j) {
            typedef __attribute__((address_space(1))) void gvoid;
            typedef __attribute__((address_space(3))) void lvoid;
            const int m = j / PPI, cc = (j % PPI) * NW + wave;
            if (cc < NCH && tile + MT + m < tile_end) {
                const unsigned char* gs = fimg + (size_t)(tile + MT + m) * IMG + (size_t)cc * 1024;
                asm volatile("" : "+s"(gs));
                __builtin_amdgcn_global_load_lds((gvoid*)(gs + lane * 16), (lvoid*)(nxt + (size_t)m * IMG + (size_t)cc * 1024),
                                                 16, 0, 0);
            }
        };
        if (more) {
            load_counts(tile + MT);
            if (DB && (DS == 0 || !active)) {
#pragma unroll
                for (int j = 0; j < PWV; ++j) piece(j);
            }
        }
        if constexpr (BREG) {
            // all the fragments of the tile this wave will need, forward and backward, into registers; then the buffer
            // is free and the next image travels during the whole tile
            if (active) {
                const double* fa = reinterpret_cast<const double*>(cur) + pgl_img_row(col) * RS + kcol0 + grp;
#pragma unroll
                for (int s = 0; s < KSW; ++s) fareg[s] = pgl_lds_f64(fa + 4 * s);
                if (p.want_grad) {
                    const double* fb = reinterpret_cast<const double*>(cur) + pgl_img_brow(grp) * RS + kcol0 + col;
#pragma unroll
                    for (int s = 0; s < 4 * KTW; ++s) fbreg[s] = pgl_lds_f64(fb + (2 * (s / KTW)) * RS + 16 * (s % KTW));
                }
            }
            __syncthreads();                              // every wave holds its fragments
            if (more) {
#pragma unroll
                for (int j = 0; j < PWV; ++j) piece(j);
            }
        }
        PGL_PROF_MARK(2);
        // ---- forward over this wave's K slice, tile by tile ----
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
            d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
            if (active && tile + m < tile_end) {
                const double* fa = reinterpret_cast<const double*>(cur + (size_t)m * IMG) + pgl_img_row(col) * RS + kcol0 + grp;
                constexpr int PA = (KSW < 4) ? KSW : 4;
                double ar[PA];
                if constexpr (!BREG) {
#pragma unroll
                    for (int s = 0; s < PA; ++s) ar[s] = pgl_lds_f64(fa + 4 * s);
                }
#pragma unroll
                for (int s = 0; s < KSW; ++s) {
                    const double a = BREG ? fareg[BREG ? s : 0] : ar[s % PA];
                    if (!BREG && s + PA < KSW) ar[s % PA] = pgl_lds_f64(fa + 4 * (s + PA));
                    if (s & 1)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wreg[s], acc1, 0, 0, 0);
                    else
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wreg[s], acc0, 0, 0, 0);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    if (DS > 0) {
                        constexpr int DSS = (DS > 0) ? DS : 1;
                        const int q1 = m * KSW + s + 1;
                        if (q1 % DSS == 0 && q1 / DSS <= PWV) {
                            if (more) piece(q1 / DSS - 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            double* xw = Xp + ((size_t)m * NW + wave) * 256 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[r * 64] = acc0[r] + acc1[r];
        }
        PGL_PROF_MARK(3);
        __syncthreads();
        PGL_PROF_MARK(4);
        // ---- epilogue: sum of the KSPLIT partials + bias -> ll terms, residuals; the elements of all MT
        // tiles go through the rate chains together (independent chains interleave) ----
        if (PGL_EPI_PRIO) __builtin_amdgcn_s_setprio(PGL_EPI_PRIO);
        if (active) {
            double xe[MT * EPW], rese[MT * EPW], terme[MT * EPW];
            bool vte[MT * EPW];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < EPW; ++e) {
                    const int r = er[e];
                    double x = bias_l;
                    if (emine) {                       // (the other half-wave's elements may already hold residuals)
#pragma unroll
                        for (int k2 = 0; k2 < KSPLIT; ++k2)
                            x += Xp[((size_t)m * NW + ptl + PTW * k2) * 256 + r * 64 + lane];
                    }
                    const long long tg = (long long)(tile + m) * TT + grp + 4 * r;
                    vte[m * EPW + e] = valid_n && (tg < p.t_hi) && emine && (tile + m < tile_end);
                    xe[m * EPW + e] = x;
                }
            }
            // whole step inside the evaluated range and every lane owns its elements: the fixed-order epilogue with
            // the spike terms compacted (log lam and 1/lam once per step for the ~2 % of elements with a spike instead
            // of for every element of every wave that holds one: ~90 f64 instructions per tile and wave at C2)
            bool done = false;
            if (KSPLIT <= 4 && tile + MT <= tile_end && (long long)(tile + MT) * TT <= p.t_hi && !PGL_DBG(4)) {
                const double* cg = PGL_C;
                asm volatile("" : "+s"(cg));               // keeps the scalar loads inside the tile loop
                double termx = 0.0;
#ifdef PGL_PROF
                long long pgl_prof_dummy_acc[12] = {0};
                long long pgl_prof_dummy_t = 0;
#endif
                done = pgl_rate_fx<MT * EPW, 16>(xe, scu, p.nlin | p.epi64, p.dt, (pgl_k_cdp)cg, wscratch, lane, termx,
                                                 rese PGL_PROF_DUMMY);
                if (done) {
                    ll_acc += termx;                       // lanes of padding neurons are never read back
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int e = 0; e < EPW; ++e) {
                            gb_acc += rese[m * EPW + e];
                            Rb[((size_t)m * RBS + ptl) * 256 + er[e] * 64 + lane] = rese[m * EPW + e];
                        }
                    }
                }
            }
            if (!done) {
                pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                asm volatile("" : "+v"(Cl));
                pgl_rate_terms_n<MT * EPW>(xe, sc, p.nlin | p.epi64, p.dt, terme, rese, Cl);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int e = 0; e < EPW; ++e) {
                        const double res = vte[m * EPW + e] ? rese[m * EPW + e] : 0.0;
                        ll_acc += vte[m * EPW + e] ? terme[m * EPW + e] : 0.0;
                        gb_acc += res;
                        if (emine) Rb[((size_t)m * RBS + ptl) * 256 + er[e] * 64 + lane] = res;
                    }
                }
            }
        }
        PGL_PROF_MARK(5);
        if (PGL_EPI_PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        PGL_PROF_MARK(6);
        // ---- backward on this wave's K slice ----
        if (active && p.want_grad) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (tile + m >= tile_end) break;
                double rr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = Rb[((size_t)m * RBS + ptl) * 256 + r * 64 + lane];
                const double* fb = reinterpret_cast<const double*>(cur + (size_t)m * IMG) + pgl_img_brow(grp) * RS + kcol0 + col;
                constexpr int NS = 4 * KTW;
                if constexpr (BREG) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        G[s % KTW] = __builtin_amdgcn_mfma_f64_16x16x4f64(fbreg[s], rr[s / KTW], G[s % KTW], 0, 0, 0);
                        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
                double ar[PD];
#pragma unroll
                for (int s = 0; s < PD; ++s) ar[s] = pgl_lds_f64(fb + (2 * (s / KTW)) * RS + 16 * (s % KTW));
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double a = ar[s % PD];
                    if (s + PD < NS) ar[s % PD] = pgl_lds_f64(fb + (2 * ((s + PD) / KTW)) * RS + 16 * ((s + PD) % KTW));
                    G[s % KTW] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KTW], G[s % KTW], 0, 0, 0);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
        }
        PGL_PROF_MARK(7);
        if (!DB && !BREG && more) {
            __syncthreads();                              // every wave has read the images of this step
#pragma unroll
            for (int j = 0; j < PWV; ++j) piece(j);
        }
    }
    PGL_PROF_STORE(1);

    if (active) {
        pgl_store_ll(p, chunk, pt, ksl, KSPLIT, lane, ll_acc, gb_acc);
        if (p.want_grad) {
            double* gp = pgl_gpart(p.Gpart, pt, KT_ALL, ksl * KTW, p.nChunks, chunk, lane);
            const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
            }
        }
    }
    PGL_PROF_EXIT;
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 8: ONE post tile (a shard of <= 16 neurons: north star's neuron split at 8 GPUs) against
// a long feature row (25 .. 40 k-tiles), K split over the 8 waves of the one workgroup a CU holds -- the HBM-bound corner:
// 3.07 GB of resident features per C3 evaluation for 1/8 of the MFMA work.  k_fused6<5,1,1,8,0> has one image buffer there
// (two 81 KB images do not fit the LDS), so the stream stops while the fragments are read out of it.  Here every wave
// owns its K slice of the image END TO END: the images are stored as 2 KB blocks (one k-tile x 16 bins, XOR-swizzled
// columns: conflict-free for both MFMA operand patterns without padding -- pgl_blk_off), a wave requests ITS blocks into a
// private ring of RING blocks (LDS-DMA), waits for them with s_waitcnt alone -- no workgroup barrier on the data path --,
// copies the forward and backward fragments of the tile to registers and hands the slots straight back to the DMA: RING - KTW
// blocks per wave (48 KB per CU) are always in flight.  Two barriers per tile remain (partial currents, residuals).
// Same partial layout as k_fused6.
// ---------------------------------------------------------------------------
// element (time row t, column c) of a 16 x 16 block, in doubles
__host__ __device__ constexpr int pgl_blk_off(int t, int c) { return t * 16 + (c ^ (2 * (t >> 1))); }

#ifndef PGL_F8_ABL
#define PGL_F8_ABL 0
#endif
#define F8A(bit) ((PGL_F8_ABL & (bit)) != 0)
// F32 = 1 (PGL_OPT_FEATURE_F32 = 2, opt-in): the resident blocks are stored as f32 -- HALF the HBM stream of this HBM-bound
// corner -- and every arithmetic operation stays f64: a wave loads a 1 KB block with one 16-byte load per lane into a
// register queue two tiles ahead (the compiler counts vmcnt), converts it and writes the f64 block (same swizzled layout,
// two conflict-free ds_write_b128) into the LDS slot the block's predecessor has just been read out of; the slots are the
// KTW blocks of one tile.  Only the STORED feature is rounded (2^-24 relative).
typedef float pgl_f4 __attribute__((ext_vector_type(4)));
template <int KTW, int RING, int F32 = 0>
__global__ __launch_bounds__(512, 1) void k_fused8(const FusedParams p)
{
    constexpr int TT = 16, NW = 8, KSPLIT = 8;
    constexpr int KSW = KTW * 4, KT_ALL = KTW * KSPLIT;
    constexpr int BLK = 2048;                               // a block in LDS (f64)
    constexpr int GBLK = F32 ? 1024 : 2048;                 // ... and in HBM
    constexpr int NSLOT = F32 ? KTW : RING;                 // LDS slots per wave
    constexpr size_t IMG = (size_t)KT_ALL * GBLK;
    static_assert(RING > KTW && RING <= 2 * KTW && 2 * (RING - KTW) < 16, "ring: more than a tile, waitcnt immediate");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ksl = wave;
    const int pt = blockIdx.x % p.nPT;
    const int chunk = blockIdx.x / p.nPT;
    unsigned char* const ring = smem + (size_t)wave * NSLOT * BLK;           // this wave's blocks
    double* Xp = reinterpret_cast<double*>(smem + (size_t)NW * NSLOT * BLK); // [NW][4][64] partial currents
    double* Rb = Xp + NW * 256;                                             // [4][64] residuals
    double* Cs = Rb + 256;                                                  // [32] math constants
    double* const wscratch = Cs + 32 + wave * 48;                           // per wave: spike compaction of the epilogue
    if (tid < 32) Cs[tid] = PGL_C[tid];

    d4_t G[KTW];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = nloc < p.npost;
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);
    const double bias_l = valid_n ? (p.theta ? p.theta[(size_t)nloc * p.P] : p.bias[nloc]) : (p.nlin == 1 ? 30.0 : 0.0);
    // epilogue: waves 0-3 (one per SIMD) own one accumulator register of the tile each, all 64 lanes -- the fixed-order
    // fast path of the other kernels (pgl_rate_fx: 9 instructions per element in standard_glm's regime, spike terms
    // compacted) instead of eight half-empty waves on the general path: the rate chain of ONE element per lane is
    // latency-bound, ~2 000 cycles between the two barriers of every tile with no MFMA to hide behind
    const int er = wave & 3;
    const bool emine = wave < 4;

    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;
    const int total = (tile_end > tile_beg) ? (tile_end - tile_beg) * KTW : 0;       // blocks of this wave in the chunk
    const unsigned char* __restrict__ fimg = p.Fimg - (size_t)p.img_tile0 * IMG + (size_t)ksl * KTW * GBLK;

    // this wave's slice of Wmat stays in registers for the whole chunk
    double wreg[KSW];
#pragma unroll
    for (int s = 0; s < KSW; ++s) wreg[s] = 0.0;
    if (p.theta) {
        pgl_wfrag_direct<KSW>(p, ksl * KSW, grp, nloc, nglob, valid_n, wreg);
    } else {
        const pgl_d2* wr2 = reinterpret_cast<const pgl_d2*>(p.Wfrag + ((size_t)pt * (KSW * KSPLIT) + (size_t)ksl * KSW) * 64);
#pragma unroll
        for (int s2 = 0; s2 < KSW / 2; ++s2) {
            const pgl_d2 v = wr2[s2 * 64 + lane];
            wreg[2 * s2] = v.x;
            wreg[2 * s2 + 1] = v.y;
        }
    }
    // post-synaptic counts of the element this wave owns in the epilogue, requested one tile ahead
    const int last_tile = p.nT16 - 1;
    const uint8_t* cptr = p.S + ((size_t)(tile_beg < last_tile ? tile_beg : last_tile) * TT + grp + 4 * er) * p.Nall + nglob;
    const size_t ctile = (size_t)TT * p.Nall;
    unsigned scn = cptr[0];

    // block n of the wave (tile tile_beg + n / KTW, k-tile n % KTW of its slice) lives in ring slot n % RING
    // (behind the chunk's last block the requests go on, to its last block again: the number of loads in flight behind a
    //  tile's blocks is then the same in every iteration -- one s_waitcnt immediate, and the compiler's own waits for
    //  the spike counts stay partial; 16 KB per wave and chunk of extra traffic)
    int n_issued = 0, is_tile = tile_beg, is_kt = 0, is_slot = 0;
    auto issue_half = [&](const int half) {               // half 0 / 1 of the next block; the block advances behind half 1
        const unsigned char* gs = fimg + (size_t)is_tile * IMG + (size_t)is_kt * BLK + half * 1024;
        asm volatile("" : "+s"(gs));
        unsigned char* dst = ring + is_slot * BLK + half * 1024;
        if (!(F8A(16) && n_issued >= RING)) __builtin_amdgcn_global_load_lds((gvoid*)(gs + lane * 16), (lvoid*)dst, 16, 0, 0);
        if (half) {
            ++n_issued;
            if (n_issued < total) {
                if (++is_kt == KTW) { is_kt = 0; ++is_tile; }
            }
            if (++is_slot == RING) is_slot = 0;
        }
    };
    auto issue = [&](const int count) {
#pragma unroll
        for (int j = 0; j < count; ++j) {
            issue_half(0);
            issue_half(1);
        }
    };
    if (total == 0) return;                               // (never: a chunk has a tile)
    // F32: the register queue -- set (li + 1) & 1 holds the blocks of the chunk's tile li + 1 while tile li computes
    pgl_f4 qreg[F32 ? 2 * KTW : 1];
    const int ntl = tile_end - tile_beg;
    auto gload = [&](const int li, const int kt) -> pgl_f4 {      // block kt of tile li (behind the chunk: its last tile again)
        const int lt = (li < ntl) ? li : ntl - 1;
        const unsigned char* gs = fimg + (size_t)(tile_beg + lt) * IMG + (size_t)kt * GBLK;
        asm volatile("" : "+s"(gs));
        return *reinterpret_cast<const pgl_f4*>(gs + lane * 16);
    };
    // lane l carries the doubles 2l, 2l + 1, 128 + 2l, 128 + 2l + 1 of the block (k_build_fimg, blk = 2)
    auto lds_put = [&](const int kt, const pgl_f4 v) {
        typedef __attribute__((address_space(3))) pgl_d2 ld2;
        volatile ld2* dst = (volatile ld2*)(ring + kt * BLK + lane * 16);
        pgl_d2 lo, hi;
        lo.x = (double)v.x; lo.y = (double)v.y; hi.x = (double)v.z; hi.y = (double)v.w;
        dst[0] = lo;
        dst[64] = hi;
    };
    if constexpr (F32 != 0) {
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) qreg[kt] = gload(0, kt);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) qreg[KTW + kt] = gload(1, kt);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) lds_put(kt, qreg[kt]);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) qreg[kt] = gload(2, kt);
    } else {
        issue(RING);
    }
    // workgroup barrier for LDS traffic only: __syncthreads() also waits for vmcnt(0) -- the blocks in flight
    // (timing ablation, -DPGL_ABLATE builds only: 1 no MFMAs, 2 no rate epilogue, 4 no barriers, 8 no fragment reads)
    auto lds_barrier = [&] { if (!F8A(4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    lds_barrier();                                        // the constants

    // fragment offsets inside a block (bytes): forward -- time row col, columns 4 ks + grp; backward -- time row 4 q + grp,
    // column col
    int offa[4], offb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        offa[q] = pgl_blk_off(col, 4 * q + grp) * 8;
        offb[q] = pgl_blk_off(4 * q + grp, col) * 8;
    }
    double fareg[KSW], fbreg[KSW];
    int slot0 = 0;                                        // ring slot of the tile's first block
    auto tile_step = [&](auto parc, const int tile, const int li) {
        constexpr int PAR = decltype(parc)::value;           // F32: parity of li (selects the register set)
        // the KTW blocks of this tile have landed when at most the RING - KTW blocks requested behind them are in flight
        // (loads return in order)
        if constexpr (F32 == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * (RING - KTW)));
        const unsigned scu = scn;
        cptr += (tile + 1 < tile_end && tile + 1 <= last_tile) ? ctile : 0;
        if (!F8A(32)) scn = cptr[0];                      // (requested BEFORE the blocks below: it is back before them)
        // k-tile by k-tile: the forward and backward fragments of block kt + 1 are read into registers while the four forward
        // MFMAs of block kt run; behind them block kt's slot goes back to the DMA.  A request waits for room in the CU's miss
        // queue and holds its wave meanwhile -- the SIMD's other wave has MFMAs to run then (all five requests in front of
        // the MFMAs: 0.662 ms; one per four MFMAs: 0.619; spread over the backward loop as well: 0.644)
        auto read_frags = [&](const int kt) {
            int sl = slot0 + kt;
            sl = (sl >= RING) ? sl - RING : sl;
            if constexpr (F32 != 0) sl = kt;
            const unsigned char* blk = ring + sl * BLK;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (F8A(8)) { fareg[4 * kt + q] = 1.0; fbreg[q * KTW + kt] = 1.0; continue; }
                fareg[4 * kt + q] = pgl_lds_f64(reinterpret_cast<const double*>(blk + offa[q]));
                fbreg[q * KTW + kt] = pgl_lds_f64(reinterpret_cast<const double*>(blk + offb[q]));
            }
        };
        read_frags(0);
        // ---- forward over this wave's K slice ----
        d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
        d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            if (kt + 1 < KTW) read_frags(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * kt + q;
                if (F8A(1)) { acc0[0] += fareg[s]; continue; }
                if (s & 1)
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fareg[s], wreg[s], acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fareg[s], wreg[s], acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the fragments of blocks kt and kt + 1 are in registers
            if constexpr (F32 != 0) {
                // slot kt is free: the next tile's block kt goes in (loaded two tiles ago), its register takes the request
                // for the block three tiles ahead
                constexpr int QI = ((PAR + 1) & 1) * KTW;
                lds_put(kt, qreg[QI + kt]);
                qreg[QI + kt] = gload(li + 3, kt);
            } else {
                issue(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        slot0 += KTW;
        slot0 = (slot0 >= RING) ? slot0 - RING : slot0;
        double* xw = Xp + (size_t)wave * 256 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) xw[r * 64] = acc0[r] + acc1[r];
        lds_barrier();
        // ---- epilogue: sum of the eight partials + bias -> ll term, residual, for the elements this wave owns ----
        if (emine) {
            double xe[1], sc[1], rese[1], terme[1];
            unsigned scv[1];
            double x = bias_l;
#pragma unroll
            for (int k2 = 0; k2 < KSPLIT; ++k2) x += Xp[(size_t)k2 * 256 + er * 64 + lane];
            xe[0] = x;
            scv[0] = scu;
            sc[0] = (double)scu;
            bool done = false;
            if ((long long)(tile + 1) * TT <= p.t_hi && !F8A(2)) {     // whole tile inside the evaluated range
                const double* cg = PGL_C;
                asm volatile("" : "+s"(cg));               // keeps the scalar loads inside the tile loop
                double termx = 0.0;
#ifdef PGL_PROF
                long long pgl_prof_dummy_acc[12] = {0};
                long long pgl_prof_dummy_t = 0;
#endif
                done = pgl_rate_fx<1, 16>(xe, scv, p.nlin | p.epi64, p.dt, (pgl_k_cdp)cg, wscratch, lane, termx, rese PGL_PROF_DUMMY);
                if (done) {
                    ll_acc += termx;                       // lanes of padding neurons are never read back
                    gb_acc += rese[0];
                    Rb[er * 64 + lane] = rese[0];
                }
            }
            if (!done) {
                const long long tg = (long long)tile * TT + grp + 4 * er;
                const bool vt = valid_n && (tg < p.t_hi);
                pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                asm volatile("" : "+v"(Cl));
                if (F8A(2)) { terme[0] = xe[0]; rese[0] = sc[0]; }
                else pgl_rate_terms_n<1>(xe, sc, p.nlin | p.epi64, p.dt, terme, rese, Cl);
                const double res = vt ? rese[0] : 0.0;
                ll_acc += vt ? terme[0] : 0.0;
                gb_acc += res;
                Rb[er * 64 + lane] = res;
            }
        }
        lds_barrier();
        // ---- backward on this wave's K slice ----
        if (p.want_grad) {
            double rr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rr[r] = Rb[r * 64 + lane];
#pragma unroll
            for (int s = 0; s < KSW; ++s) {
                if (F8A(1)) { G[s % KTW][0] += fbreg[s] * rr[s / KTW]; continue; }
                G[s % KTW] = __builtin_amdgcn_mfma_f64_16x16x4f64(fbreg[s], rr[s / KTW], G[s % KTW], 0, 0, 0);
            }
        }
    };
    if constexpr (F32 != 0) {
        for (int tile = tile_beg, li = 0; tile < tile_end; tile += 2, li += 2) {
            tile_step(std::integral_constant<int, 0>{}, tile, li);
            if (tile + 1 < tile_end) tile_step(std::integral_constant<int, 1>{}, tile + 1, li + 1);
        }
    } else {
        for (int tile = tile_beg, li = 0; tile < tile_end; ++tile, ++li) tile_step(std::integral_constant<int, 0>{}, tile, li);
    }

    pgl_store_ll(p, chunk, pt, ksl, KSPLIT, lane, ll_acc, gb_acc);
    if (p.want_grad) {
        double* gp = pgl_gpart(p.Gpart, pt, KT_ALL, ksl * KTW, p.nChunks, chunk, lane);
        const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
        }
    }
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 7: one pass on resident feature tiles WITHOUT a K split, for short
// feature rows (KT <= 20 k-tiles, i.e. the whole G of a post tile fits a wave's registers) and few post
// tiles: a workgroup is NWV waves = NWV post tiles on the same 16-bin tile; every wave runs the whole
// chain forward (all K) -> rate epilogue on its own accumulator registers (four elements per lane,
// pgl_rate4) -> backward (all K) by itself.  No partial currents or residuals travel through LDS, so the
// only workgroup barrier per tile is "the image has landed"; several small workgroups share a CU
// (NWV = 4: two, NWV = 2: three), each on its own time chunk, and fill one another's waits.
//   image of a tile = [16][RS] f64 as for k_fused6 (one part), double-buffered per workgroup.
// Partials as k_fused5 (KSPLIT = 1).
// ---------------------------------------------------------------------------
// XIO = 1: the currents start from the slab p.Xbuf[tile - tile0][post tile][r][lane] (the stimulus current of a separable
// stimulus, k_sepf_fwd) and the residuals r = d ll / d x are written back to the same slab (for k_sepf_bwd)
// XIO = 2: the stimulus current is part of the forward contraction instead (no k_sepf_fwd launch, no slab read): with
// F0 the frame of the tile's first bin and base = max(F0 - M, 0), bin i of the tile (frame F0 or F0 + 1) has
//   I_stim[i][n] = sum_{j' <= J, bt} A[i][(j', bt)] * w_t[n][bt] z_n[min(base + j', Tstim - 1)],
//   A[i][(j', bt)] = C[row(t0 + i)][j' - (base(F_i) - base)][bt]
// -- (J + 1) Bt = 18 columns = five k-steps whose A fragments depend on the tile only through t0 mod q (head tiles
// apart) and come from a table (p.sepA), the B fragments are five gathers of z and a multiply; residuals out as XIO = 1
template <int KT, int NWV, int XIO = 0>
__global__ __launch_bounds__(NWV * 64, 2) void k_fused7(const FusedParams p)
{
    constexpr int TT = 16;
    constexpr int KS = 4 * KT;
    constexpr int RS = pgl_img_rsh(KT);
    constexpr int IMG = pgl_img_bytes(KT);
    constexpr bool WREG = (KS <= PGL_WREG_MAX);  // the wave's Wmat fragments stay in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PGL_PROF_ENTRY

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nPB = (p.nPT + NWV - 1) / NWV;
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
    const int pt = pb * NWV + wave;
    const bool active = pt < p.nPT;

    unsigned char* bufs = smem;                                           // [2][IMG]
    double* Cs = reinterpret_cast<double*>(smem + (size_t)2 * IMG);       // [32] math constants
    double* const wscratch_base = Cs + 32;
    double* const wscratch = wscratch_base + wave * 192;                  // spike compaction scratch
    if (tid < 32) Cs[tid] = PGL_C[tid];

    d4_t G[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);
    // padding lanes get a benign current: they must not push their wave out of the epilogue's series regime
    const double bias_l = valid_n ? (p.theta ? p.theta[(size_t)nloc * p.P] : p.bias[nloc]) : (p.nlin == 1 ? 30.0 : 0.0);
    const double* __restrict__ wrow = p.Wfrag + (size_t)(active ? pt : 0) * KS * 64;

    // forward k-steps that hold feature columns: the row is padded to whole 16-column tiles (the backward MFMA's M), the
    // forward k-step is 4 columns wide -- up to three all-zero steps at the end of the row are skipped (195 columns at C5:
    // 49 of 52)
    const int ksf = (p.Ktot + 3) >> 2;
    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;
    const unsigned char* __restrict__ fimg = p.Fimg - (size_t)p.img_tile0 * IMG;

    double wreg[WREG ? KS : 1];
    if (WREG && p.theta) {
        pgl_wfrag_direct<(WREG ? KS : 1)>(p, 0, grp, nloc, nglob, valid_n, wreg);
    } else if (WREG) {
        const pgl_d2* wr2 = reinterpret_cast<const pgl_d2*>(wrow);
#pragma unroll
        for (int s2 = 0; s2 < (WREG ? KS / 2 : 0); ++s2) {
            const pgl_d2 v = wr2[s2 * 64 + lane];
            wreg[2 * s2] = v.x;
            wreg[2 * s2 + 1] = v.y;
        }
    }
    unsigned scb[4] = {0u, 0u, 0u, 0u}, scn[4] = {0u, 0u, 0u, 0u};
    double xib[XIO == 1 ? 4 : 1], xin[XIO == 1 ? 4 : 1];
    double* const xslab = XIO ? p.Xbuf + (size_t)(active ? pt : 0) * 256 + lane : nullptr;
    const size_t xstride = (size_t)p.nPT * 256;
    // XIO = 2: the lane's five B entries are (j', bt) = divmod(4 s + grp, 3) of its neuron
    constexpr int SF = (XIO >= 2) ? 5 : 1;
    double sfa[SF], sfz[SF];
    int sepF = 0, sepO = 0, sepPh = 0;
    long long sepBase = 0;
    constexpr bool SBWC = (XIO == 3);                                   // stimulus backward inside this kernel
    const bool SBW = SBWC && p.want_grad;
    // its accumulators (five values per lane) live in LDS between the tiles: in registers they would be alive across
    // the rate epilogue, where this kernel has none to spare (22-30 VGPRs spilled at 12-13 k-tiles)
    double* const Dl = wscratch_base + NWV * 192 + wave * 320 + lane;
    if constexpr (XIO >= 2) {
#pragma unroll
        for (int s = 0; s < SF; ++s) sfa[s] = sfz[s] = 0.0;
        if (SBW) {
#pragma unroll
            for (int r = 0; r < 5; ++r) Dl[r * 64] = 0.0;
        }
    }
    auto load_counts = [&](const int tile, unsigned (&dst)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long tg = (long long)tile * TT + grp + 4 * r;
            const long long tc = (tg < p.nT) ? tg : (p.nT - 1);
            dst[r] = p.S[tc * p.Nall + nglob];
        }
        if constexpr (XIO == 1) {
            const double* xs_ = xslab + (size_t)(tile - p.tile0) * xstride;
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[r] = xs_[r * 64];
        }
    };
    // XIO = 2: A fragments and z of a tile are requested when the tile starts and used behind its K loop
    auto load_stim = [&](const int tile) {
        if constexpr (XIO >= 2) {
            // frame and offset of the tile's first bin: one division per chunk, then increments (q >= 16)
            if (tile == tile_beg) {
                const int tb = tile * TT;
                sepF = tb / p.sepQ;
                sepO = tb - sepF * p.sepQ;
            } else {
                sepO += TT;
                if (sepO >= p.sepQ) {
                    sepO -= p.sepQ;
                    ++sepF;
                }
            }
            const int ph = (tile < p.sepNH) ? tile : p.sepNH + (sepO >> p.sepG);
            const long long base = (sepF > p.sepM) ? sepF - p.sepM : 0;
            sepPh = ph;
            sepBase = base;
            const double* ap = p.sepA + (size_t)ph * (SF * 64) + lane;
            const double* zp = p.sepZ + (valid_n ? nloc : 0);
            const double* wp = p.sepTheta + (size_t)(valid_n ? nloc : 0) * p.P + 1;
            // w_t of the lane's neuron: read again per tile (L1) rather than held across the epilogue
            const double w0 = wp[0], w1 = wp[(p.sepBt > 1) ? 1 : 0], w2 = wp[(p.sepBt > 2) ? 2 : 0];
#pragma unroll
            for (int s = 0; s < SF; ++s) {
                // the lane's B entry of k-step s: (j', bt) = divmod(4 s + grp, 3)
                const int k = 4 * s + grp, j = k / 3, bt = k - 3 * j;
                long long f = base + j;
                f = (f < p.sepT) ? f : p.sepT - 1;
                sfa[s] = ap[s * 64];
                const double z = zp[(size_t)f * p.sepLdy];
                const double w = (bt == 0) ? w0 : ((bt == 1) ? w1 : w2);
                sfz[s] = (valid_n && k < 18 && bt < p.sepBt) ? z * w : 0.0;
            }
        }
    };
    if (tile_beg < tile_end) {
        pgl_dma_img<KT, NWV>(fimg + (size_t)tile_beg * IMG, bufs, wave, lane);
        load_counts(tile_beg, scn);
    }

    int par = 0;
    PGL_PROF_DECL
    for (int tile = tile_beg; tile < tile_end; ++tile, par ^= 1) {
        const int t0 = tile * TT;
        const unsigned char* cur = bufs + (size_t)par * IMG;
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's pieces landed, counts are here
        PGL_PROF_MARK(0);
        __syncthreads();                                  // ... every wave's; the other buffer is free
        PGL_PROF_MARK(1);
#pragma unroll
        for (int r = 0; r < 4; ++r) scb[r] = scn[r];
        if constexpr (XIO == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xib[r] = xin[r];
        }
        load_stim(tile);
        // the image of the next tile: PWV pieces of 1 KiB per wave, issued between the MFMAs (PGL_DMA_IL) of the
        // forward loop when the Wmat fragments live in registers -- with the streamed Wmat ring a DMA in flight
        // would sit in front of the ring loads in the in-order vmcnt queue -- else of the backward loop; waves
        // without that loop send the burst up front
        const bool more = tile + 1 < tile_end;
        unsigned char* const nxt = bufs + (size_t)(par ^ 1) * IMG;
        constexpr int NCH = IMG / 1024, PWV = (NCH + NWV - 1) / NWV;
        constexpr int DSF = (PGL_DMA_IL && WREG && KS >= PWV) ? KS / PWV : 0;                 // forward-loop spacing
        constexpr int DSB = (PGL_DMA_IL && !WREG && 4 * KT >= 2 * PWV) ? (2 * KT) / PWV : 0;   // backward: first half
        auto piece = [&](const int j) {
            typedef __attribute__((address_space(1))) void gvoid;
            typedef __attribute__((address_space(3))) void lvoid;
            const int cc = j * NWV + wave;
            if (cc < NCH) {
                const unsigned char* gs = fimg + (size_t)(tile + 1) * IMG + (size_t)cc * 1024;
                asm volatile("" : "+s"(gs));
                __builtin_amdgcn_global_load_lds((gvoid*)(gs + lane * 16), (lvoid*)(nxt + (size_t)cc * 1024), 16, 0, 0);
            }
        };
        const bool il_bwd = DSB > 0 && active && p.want_grad;
        if (more) {
            load_counts(tile + 1, scn);
            if (!active || (DSF == 0 && !il_bwd)) {
#pragma unroll
                for (int j = 0; j < PWV; ++j) piece(j);
            }
        }
        PGL_PROF_MARK(2);
        if (!active) continue;
        // ---- forward over all K ----
        d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
        d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
        {
            const double* fa = reinterpret_cast<const double*>(cur) + pgl_img_row(col) * RS + grp;
            constexpr int PA = 4;
            double ar[PA];
#pragma unroll
            for (int s = 0; s < PA; ++s) ar[s] = pgl_lds_f64(fa + 4 * s);
            if constexpr (WREG) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ar[s % PA];
                    if (s + PA < KS) ar[s % PA] = pgl_lds_f64(fa + 4 * (s + PA));
                    if (s < KS - 3 || s < ksf) {
                        if (s & 1)
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wreg[s], acc1, 0, 0, 0);
                        else
                            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wreg[s], acc0, 0, 0, 0);
                    }
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    if (DSF > 0) {
                        constexpr int DSS = (DSF > 0) ? DSF : 1;
                        if ((s + 1) % DSS == 0 && (s + 1) / DSS <= PWV) {
                            if (more) piece((s + 1) / DSS - 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
                const double* wr_s = wrow;
                asm volatile("" : "+s"(wr_s));
                constexpr int PW2 = (KS / 2 < PGL_PW / 2) ? KS / 2 : PGL_PW / 2;
                const pgl_glb_cd2p wr2 = (pgl_glb_cd2p)wr_s;
                pgl_d2 wr[PW2];
#pragma unroll
                for (int s = 0; s < PW2; ++s) wr[s] = wr2[s * 64 + lane];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ar[s % PA];
                    const double b = (s & 1) ? wr[(s / 2) % PW2].y : wr[(s / 2) % PW2].x;
                    if (s + PA < KS) ar[s % PA] = pgl_lds_f64(fa + 4 * (s + PA));
                    if ((s & 1) && (s / 2 + PW2 < KS / 2)) wr[(s / 2) % PW2] = wr2[(s / 2 + PW2) * 64 + lane];
                    if (s < KS - 3 || s < ksf) {
                        if (s & 1)
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                        else
                            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                    }
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (XIO >= 2) {                         // the stimulus current: five more k-steps
#pragma unroll
            for (int s = 0; s < SF; ++s) {
                if (s & 1)
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sfa[s], sfz[s], acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sfa[s], sfz[s], acc0, 0, 0, 0);
            }
        }
        PGL_PROF_MARK(3);
        // ---- epilogue on the accumulator registers ----
        if (PGL_EPI_PRIO) __builtin_amdgcn_s_setprio(PGL_EPI_PRIO);
        double rr[4];
        {
            bool done = false;
            if ((long long)t0 + TT <= p.t_hi) {
                double xs[4], term4 = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[r] = bias_l + (acc0[r] + acc1[r]);
                if constexpr (XIO == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[r] += xib[r];
                }
                const double* cg = PGL_C;
                asm volatile("" : "+s"(cg));               // keeps the scalar loads inside the tile loop
                done = pgl_rate4(xs, scb, p.nlin | p.epi64, p.dt, (pgl_k_cdp)cg, wscratch, lane, term4, rr PGL_PROF_PASS);
                if (done) {
                    ll_acc += valid_n ? term4 : 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rr[r] = valid_n ? rr[r] : 0.0;
                        gb_acc += rr[r];
                    }
                }
            }
            if (!done) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    double xe[2], se[2], terme[2], rese[2];
                    bool vte[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * h2 + e;
                        xe[e] = bias_l + (acc0[r] + acc1[r]);
                        if constexpr (XIO == 1) xe[e] += xib[r];
                        se[e] = (double)scb[r];
                        const long long tg = (long long)t0 + grp + 4 * r;
                        vte[e] = valid_n && (tg < p.t_hi);
                    }
                    pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                    asm volatile("" : "+v"(Cl));
                    pgl_rate_terms_n<2>(xe, se, p.nlin | p.epi64, p.dt, terme, rese, Cl);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double res = vte[e] ? rese[e] : 0.0;
                        rr[2 * h2 + e] = res;
                        ll_acc += vte[e] ? terme[e] : 0.0;
                        gb_acc += res;
                    }
                }
            }
        }
        PGL_PROF_MARK(4);
        if (PGL_EPI_PRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (XIO != 0) {
            if (p.want_grad && !SBW) {
                double* xs_ = xslab + (size_t)(tile - p.tile0) * xstride;
#pragma unroll
                for (int r = 0; r < 4; ++r) xs_[r * 64] = rr[r];
            }
        }
        // stimulus backward: the A^T fragments of this tile's phase, in flight behind the backward MFMAs below
        // (columns 0..15 requested here, 16..17 -- the second accumulator tile -- twelve MFMAs before the end of the
        //  backward loop: all eight at once did not fit the registers of the 12- and 13-k-tile forms)
        double sbt[SBWC ? 8 : 1];
        const double* const atp = SBWC ? p.sepAT + (size_t)sepPh * (8 * 64) + lane : nullptr;
        if constexpr (SBWC) {
            if (SBW) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sbt[s] = atp[s * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- backward over all K ----
        if (p.want_grad) {
            const double* fb = reinterpret_cast<const double*>(cur) + pgl_img_brow(grp) * RS + col;
            constexpr int NS = 4 * KT;
            constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = pgl_lds_f64(fb + (2 * (s / KT)) * RS + 16 * (s % KT));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS) ar[s % PD] = pgl_lds_f64(fb + (2 * ((s + PD) / KT)) * RS + 16 * ((s + PD) % KT));
                G[s % KT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KT], G[s % KT], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                if (DSB > 0) {
                    constexpr int DSS = (DSB > 0) ? DSB : 1;
                    if ((s + 1) % DSS == 0 && (s + 1) / DSS <= PWV) {
                        if (more) piece((s + 1) / DSS - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (SBWC) {
                    if (s == ((NS > 12) ? NS - 12 : 0) && SBW) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 4; u < 8; ++u) sbt[u] = atp[u * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (SBWC) {
                if (SBW) {
                    d4_t Dst[2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dst[0][r] = Dl[r * 64];
                    Dst[1] = (d4_t){Dl[4 * 64], 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < 8; ++s)
                        Dst[s >> 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sbt[s], rr[s & 3], Dst[s >> 2], 0, 0, 0);
                    // the next tile has another frame base (or the chunk ends): write the piece out
                    bool flush = !more;
                    if (more) {
                        const int o2 = sepO + TT;
                        const long long F2 = sepF + ((o2 >= p.sepQ) ? 1 : 0);
                        flush = ((F2 > p.sepM) ? F2 - p.sepM : 0) != sepBase;
                    }
                    if (flush) {
                        const long long tf = pgl_sepd_first_tile(sepBase, p.sepM, p.sepQ, p.tile0);
                        const int slot = chunk - (int)((tf - p.tile0) / p.tilesPerChunk);
                        double* dp = p.sepD + ((((size_t)(sepBase - p.sepB0) * p.sepSL + slot) * p.nPT + pt) * 5) * 64 + lane;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dp[r * 64] = Dst[0][r];
                        dp[4 * 64] = Dst[1][0];
                        Dst[0] = (d4_t){0.0, 0.0, 0.0, 0.0};
                        Dst[1] = (d4_t){0.0, 0.0, 0.0, 0.0};
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dl[r * 64] = Dst[0][r];
                    Dl[4 * 64] = Dst[1][0];
                }
            }
        }
        PGL_PROF_MARK(5);
    }
    PGL_PROF_STORE(1);

    if (active) {
        pgl_store_ll(p, chunk, pt, 0, 1, lane, ll_acc, gb_acc);
        if (p.want_grad) {
            double* gp = pgl_gpart(p.Gpart, pt, KT, 0, p.nChunks, chunk, lane);
            const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
            }
        }
    }
    PGL_PROF_EXIT;
}
