// fused ll+grad kernels with in-kernel feature generation: gen_items, k_fused2, k_fused3
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// Feature generation (on-the-fly kernels): work item = (feature column, block of ROWS rows)
// so that all 512 threads are busy (640 columns x 4 row blocks = 5 items per thread at C3)
// and a wave spans few presynaptic neurons (less trip-count divergence).  Events are
// staged in LDS already decoded for this tile: {byte offset of row 0's 16-tap slice in the
// even/odd basis table, count as float} -- the even/odd select and lag arithmetic are done
// once per event at staging time instead of once per (event, column, row block).
// ---------------------------------------------------------------------------
template <int ESZ>
__device__ __forceinline__ int2 pgl_decode_event(const int2 e, const int t0, const int oddoff)
{
    const int base = t0 - e.x - 1 + 16;                       // >= 1, < R + 16
    const int off = (base & 1) ? (oddoff + (base - 1) * ESZ) : base * ESZ;
    return make_int2(off, __float_as_int((float)e.y));
}

typedef const __attribute__((address_space(3))) double* pgl_lds_cdp;
typedef const __attribute__((address_space(1))) pgl_d2* pgl_glb_cd2p;

template <typename T> struct pgl_vec2;
template <> struct pgl_vec2<double> { typedef double2 type; };
template <> struct pgl_vec2<float> { typedef float2 type; };

template <int BB, int CAP, typename FT>
__device__ __forceinline__ void gen_items(FT* __restrict__ Fs, const int rsf,
                                          const unsigned char* __restrict__ phiBytes,
                                          const int RP, const int2* __restrict__ s_dec,
                                          const int* __restrict__ s_lo,
                                          const int* __restrict__ s_cnt,
                                          const int2* __restrict__ spk, const int t0, const int B,
                                          const int Kimp, const int tid, const int nthr,
                                          const int shift, const int item_lo = 0)
{
    // Staged events live in a per-neuron ring of CAP slots (slot = event index mod CAP) and are
    // decoded once, relative to the first tile of the chunk; `shift` = byte offset of this tile's
    // taps against that reference (16 bins per tile = the same even/odd table, 16 entries on).
    // item = (feature column, q): rows {2q, 2q+1, 8+2q, 9+2q}.  With this interleave the four
    // q-lanes of a column read one contiguous span of the 16-tap slice per LDS read (taps
    // 2q,2q+1 first, taps 8+2q,9+2q second) and the host picks RP so that the basis rows
    // b = 0..3 of a neuron sit one such span apart: a 16-lane LDS group of one neuron covers
    // 16 distinct slots.  FT = double: taps, FMAs and F in f64 (ds_read_b128);
    // FT = float (PGL_OPT_FEATURE_F32): f32 taps / FMAs / F (ds_read_b64, half the LDS bytes).
    // The loop is LDS-bandwidth bound.
    typedef typename pgl_vec2<FT>::type V2;
    constexpr int ESZ = sizeof(FT);
    const int nitems = Kimp * 4;
    const int oddoff = B * RP * ESZ;
    for (int item = item_lo + tid; item < nitems; item += nthr) {
        const int q = item & 3;
        const int colx = item >> 2;
        const int np = (BB > 0) ? colx / BB : colx / B;
        const int b = colx - np * ((BB > 0) ? BB : B);
        const int cnt = s_cnt[np];
        const unsigned char* tb = phiBytes + (b * RP + q * 2) * ESZ;
        FT a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (cnt <= CAP) {
            const int2* sp = s_dec + np * CAP;
            const int start = s_lo[np];
            const unsigned char* tbs = tb + shift;
            for (int j = 0; j < cnt; j += 2) {
                const int2 e0 = sp[(start + j) & (CAP - 1)];
                int2 e1 = sp[(start + j + 1) & (CAP - 1)];
                if (j + 1 >= cnt) e1 = make_int2(-shift, 0);  // zero-weight dummy, valid offset
                const FT c0 = (FT)__int_as_float(e0.y);
                const FT c1 = (FT)__int_as_float(e1.y);
                const V2* p0 = reinterpret_cast<const V2*>(tbs + e0.x);
                const V2* p1 = reinterpret_cast<const V2*>(tbs + e1.x);
                const V2 u0 = p0[0], u1 = p0[4], w0 = p1[0], w1 = p1[4];
                a0 = fma(c0, u0.x, a0);
                a1 = fma(c0, u0.y, a1);
                a2 = fma(c0, u1.x, a2);
                a3 = fma(c0, u1.y, a3);
                a0 = fma(c1, w0.x, a0);
                a1 = fma(c1, w0.y, a1);
                a2 = fma(c1, w1.x, a2);
                a3 = fma(c1, w1.y, a3);
            }
        } else {                                              // window overflowed the staging
            const int2* sp = spk + s_lo[np];
            for (int j = 0; j < cnt; ++j) {
                const int2 e0 = pgl_decode_event<ESZ>(sp[j], t0, oddoff);
                const FT c0 = (FT)__int_as_float(e0.y);
                const V2* p0 = reinterpret_cast<const V2*>(tb + e0.x);
                const V2 u0 = p0[0], u1 = p0[4];
                a0 = fma(c0, u0.x, a0);
                a1 = fma(c0, u0.y, a1);
                a2 = fma(c0, u1.x, a2);
                a3 = fma(c0, u1.y, a3);
            }
        }
        FT* fr = Fs + (2 * q) * rsf + colx;
        fr[0] = a0;
        fr[rsf] = a1;
        fr[8 * rsf] = a2;
        fr[9 * rsf] = a3;
    }
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 2: 8 waves per workgroup (two per SIMD, <= 256
// registers each, VGPR-form MFMA).  A workgroup owns PTW post-synaptic tiles; the
// feature dimension K is split over KSPLIT = 8/PTW waves per tile, so a wave keeps
// only KTW = KT/KSPLIT accumulator tiles of G (<= 160 registers) and streams only its
// own slice of Wmat.  Per time tile:
//   gen F (all 512 threads) | forward partial X over the wave's K slice | partial X
//   -> LDS | the tile's 256 elements are split over its KSPLIT waves: sum of partials,
//   epilogue, r -> LDS | every wave re-reads r (MFMA B layout) | backward on its K slice.
// The two waves of a SIMD hide each other's LDS / L2 latencies; the event windows of
// the next tile are prefetched into registers during the MFMA phases.
// (f64 VALU work cannot hide under f64 MFMA on gfx950: both issue to the same DP
// pipeline -- tools/ubench -- so the win is latency hiding, not FP overlap.)
// ---------------------------------------------------------------------------
template <int KTW, int PTW, int NW, int CAP, typename FT>
__global__ __launch_bounds__(NW * 64, 2) void k_fused2(const FusedParams p)
{
    constexpr int TT = 16;
    constexpr int ESZ = sizeof(FT);              // element size of the F tile and of the basis tables
    constexpr int KSPLIT = NW / PTW;
    constexpr int KSW = KTW * 4;                 // forward k-steps of this wave
    constexpr int KS_ALL = KSW * KSPLIT;         // k-steps of the whole padded K
    constexpr int KT_ALL = KTW * KSPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int nthr = NW * 64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptl = wave % PTW;                  // post tile inside the workgroup
    const int ksl = wave / PTW;                  // K slice of this wave
    const int nPB = (p.nPT + PTW - 1) / PTW;
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
    const int pt = pb * PTW + ptl;
    const bool active = pt < p.nPT;

    const int N = p.N, B = p.B, R = p.R, rsf = p.rsf, RP = p.RP;
    // LDS carve (offsets multiples of 16)
    FT* Fs = reinterpret_cast<FT*>(smem);
    size_t off = ((size_t)TT * rsf * sizeof(FT) + 15) & ~(size_t)15;
    FT* phiE = reinterpret_cast<FT*>(smem + off);
    FT* phiO = phiE + (size_t)B * RP;
    off += (((size_t)2 * B * RP * ESZ) + 15) & ~(size_t)15;
    int2* s_spk = reinterpret_cast<int2*>(smem + off);
    off += (size_t)N * CAP * 8;
    int* s_lo = reinterpret_cast<int*>(smem + off);           // [2][N]
    off += (((size_t)2 * N * 4) + 15) & ~(size_t)15;
    int* s_cnt = reinterpret_cast<int*>(smem + off);          // [2][N]
    off += (((size_t)2 * N * 4) + 15) & ~(size_t)15;
    int* s_valid = reinterpret_cast<int*>(smem + off);        // [N] ring holds the neuron's current window
    off += (((size_t)N * 4) + 15) & ~(size_t)15;
    double* Xp = reinterpret_cast<double*>(smem + off);       // [NW][4][64] partial X
    off += (size_t)NW * 256 * 8;
    double* Rb = reinterpret_cast<double*>(smem + off);       // [PTW][4][64] residuals r
    off += (size_t)PTW * 256 * 8;
    double* Cs = reinterpret_cast<double*>(smem + off);       // [32] math constants (see epilogue)
    if (tid < 32) Cs[tid] = PGL_C[tid];

    for (int i = tid; i < B * RP; i += nthr) {
        const int b = i / RP, k = i - b * RP;
        phiE[i] = (FT)((k >= 16 && k < 16 + R) ? p.phi[b * R + k - 16] : 0.0);
        phiO[i] = (FT)((k + 1 >= 16 && k + 1 < 16 + R) ? p.phi[b * R + k + 1 - 16] : 0.0);
    }
    for (int i = tid; i < TT * rsf; i += nthr) Fs[i] = (FT)0;

    d4_t G[KTW];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);
    const double bias_l = valid_n ? p.bias[nloc] : 0.0;
    const double* __restrict__ wrow =
        p.Wfrag + ((size_t)(active ? pt : 0) * KS_ALL + (size_t)ksl * KSW) * 64;
    const int kcol0 = ksl * KTW * 16;            // first feature column of this wave's slice
    // epilogue ownership: the 256 elements (4 regs x 64 lanes) of a post tile are split over its
    // KSPLIT waves: KSPLIT=2 -> regs {2ks,2ks+1}; 4 -> reg ks; 8 -> reg ks/2, lane half ks&1
    constexpr int EPW = (KSPLIT >= 4) ? 1 : 4 / KSPLIT;
    int er[EPW];
#pragma unroll
    for (int e = 0; e < EPW; ++e) er[e] = (KSPLIT == 8) ? (ksl >> 1) : (KSPLIT == 4) ? ksl : ksl * EPW + e;
    const bool emine = (KSPLIT == 8) ? ((lane >> 5) == (ksl & 1)) : true;

    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;

    // prologue: windows of the first two tiles, events of the first tile
    if (tid < N) {
        for (int q = 0; q < 2; ++q) {
            const int tl = tile_beg + q;
            int lo = 0, cnt = 0;
            if (tl < tile_end) {
                lo = p.wlo[(size_t)tl * p.Nall + p.np0 + tid];
                cnt = p.whi[(size_t)tl * p.Nall + p.np0 + tid] - lo;
            }
            s_lo[(tl & 1) * N + tid] = lo;
            s_cnt[(tl & 1) * N + tid] = cnt;
        }
    }
    __syncthreads();
    const int t0_ref = tile_beg * TT;             // staged events are decoded relative to this tile
    {
        const int pb0 = (tile_beg & 1) * N;
        for (int id = tid; id < N * CAP; id += nthr) {
            const int np = id / CAP, sl = id % CAP;
            const int cnt = s_cnt[pb0 + np];
            if (cnt <= CAP && sl < cnt) {
                const int idx = s_lo[pb0 + np] + sl;
                s_spk[np * CAP + (idx & (CAP - 1))] = pgl_decode_event<ESZ>(p.spk[idx], t0_ref, B * RP * ESZ);
            }
        }
        if (tid < N) s_valid[tid] = (s_cnt[pb0 + tid] <= CAP) ? 1 : 0;
    }
    __syncthreads();

    constexpr int NPF = 3;                        // new events per neuron and tile taken on the fast path
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        const int t0 = tile * TT;
        const int cur = (tile & 1) * N;
        const int nxt = ((tile + 1) & 1) * N;
        // the staging code below indexes by thread id; with G taking 160 of the 256 registers at KTW = 20 the compiler's
        // hoisting of its thread-id arithmetic out of the tile loop (quotients by Dstim, ring addresses) ended in
        // scratch: behind this opaque copy it is recomputed per tile instead (a few integer instructions)
        int tid = threadIdx.x;
        if (KTW >= 20) asm volatile("" : "+v"(tid));
        // ---- prefetch (registers): the events that ENTER neuron tid's window with tile+1 (the
        // window slides by 16 bins: ~0.3 new events per neuron), windows of tile+2 ----
        int2 pf[NPF];
        int pf_new = -1;                           // -1: nothing to do; > NPF: restage at commit time
#pragma unroll
        for (int q = 0; q < NPF; ++q) pf[q] = make_int2(0, 0);
        if (tid < N && tile + 1 < tile_end && !PGL_DBG(2)) {
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int cnt_n = s_cnt[nxt + tid];
            pf_new = s_lo[nxt + tid] + cnt_n - hi;
            if (!s_valid[tid]) pf_new = NPF + 1;
            if (cnt_n <= CAP && pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new) pf[q] = p.spk[hi + q];
            }
        }
        int w2lo = 0, w2cnt = 0;
        if (tid < N && tile + 2 < tile_end && !PGL_DBG(32)) {
            w2lo = p.wlo[(size_t)(tile + 2) * p.Nall + p.np0 + tid];
            w2cnt = p.whi[(size_t)(tile + 2) * p.Nall + p.np0 + tid] - w2lo;
        }
        // dense stimulus feature columns of this tile
        if (p.Dstim > 0) {
            for (int id = tid; id < TT * p.Dstim; id += nthr) {
                const int t = id / p.Dstim;
                const int j = id % p.Dstim;
                const long long tg = (long long)t0 + t;
                Fs[t * rsf + p.Kimp + j] = (FT)((tg < p.nT) ? p.fstim[tg * p.DsAll + p.ds0 + j] : 0.0);
            }
        }
        // ---- F tile from the staged events ----
        if (!PGL_DBG(1)) {
            const unsigned char* phiBytes = reinterpret_cast<const unsigned char*>(phiE);
            if (B == 5)
                gen_items<5, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                    p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
            else if (B == 3)
                gen_items<3, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                    p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
            else
                gen_items<0, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                    p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
        }
        __syncthreads();

        // post-synaptic counts of the elements this wave owns in the epilogue: issued before the
        // forward pass so that the global-load latency hides under the MFMAs
        double sc[EPW];
#pragma unroll
        for (int e = 0; e < EPW; ++e) {
            const long long tg = (long long)t0 + grp + 4 * er[e];
            const long long tc = (tg < p.nT) ? tg : (p.nT - 1);
            sc[e] = (double)p.S[tc * p.Nall + nglob];
        }
        // ---- forward over this wave's K slice ----
        d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
        d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
        if (active && !PGL_DBG(8) && p.mode != 2) {
            const FT* fa = Fs + col * rsf + kcol0 + grp;
            const double* wr_s = wrow;
            asm volatile("" : "+s"(wr_s));
            // Wmat fragments: one 16-byte load per lane covers two k-steps (layout
            // [pt][ks/2][lane][2]); ring of PW2 loads = 2*PW2 MFMA steps ahead
            constexpr int PW2 = (KSW / 2 < PGL_PW / 2) ? KSW / 2 : PGL_PW / 2;
            constexpr int PA = (KSW < 4) ? KSW : 4;
            // explicit global address space: behind the opaque asm the compiler no longer knows
            // the provenance and would emit flat loads (vmcnt AND lgkmcnt, 64-bit VALU addresses)
            const pgl_glb_cd2p wr2 = (pgl_glb_cd2p)wr_s;
            pgl_d2 wr[PW2];
            double ar[PA];
            int lanew = lane;
            if (KTW >= 20) asm volatile("" : "+v"(lanew));        // (its 64-bit byte offset is not kept across the tile loop)
#pragma unroll
            for (int s = 0; s < PW2; ++s) wr[s] = wr2[s * 64 + lanew];
#pragma unroll
            for (int s = 0; s < PA; ++s) ar[s] = (double)fa[4 * s];
#pragma unroll
            for (int s = 0; s < KSW; ++s) {
                const double a = ar[s % PA];
                const double b = (s & 1) ? wr[(s / 2) % PW2].y : wr[(s / 2) % PW2].x;
                if (s + PA < KSW) ar[s % PA] = (double)fa[4 * (s + PA)];
                if ((s & 1) && (s / 2 + PW2 < KSW / 2)) wr[(s / 2) % PW2] = wr2[(s / 2 + PW2) * 64 + lanew];
                if (s & 1)
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            double* xw = Xp + (size_t)wave * 256 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[r * 64] = acc0[r] + acc1[r];
        }
        // ---- commit the prefetched staging for the next tile: the event slots and the
        // window buffer of `tile` were last read by gen(tile), i.e. before the barrier above;
        // committing here keeps the prefetch registers dead during epilogue and backward ----
        if (pf_new >= 0) {                        // tid < N and there is a next tile
            int2* ring = s_spk + tid * CAP;
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int lo_n = s_lo[nxt + tid], cnt_n = s_cnt[nxt + tid];
            if (cnt_n > CAP) {
                s_valid[tid] = 0;                 // gen takes the overflow path for this neuron
            } else if (pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new)
                        ring[(hi + q) & (CAP - 1)] = pgl_decode_event<ESZ>(pf[q], t0_ref, B * RP * ESZ);
            } else {                              // burst or ring lost during an overflow: restage
                for (int idx = lo_n; idx < lo_n + cnt_n; ++idx)
                    ring[idx & (CAP - 1)] = pgl_decode_event<ESZ>(p.spk[idx], t0_ref, B * RP * ESZ);
                s_valid[tid] = 1;
            }
        }
        if (tid < N) {                            // windows of tile+2 go to the buffer of `tile`
            s_lo[cur + tid] = w2lo;
            s_cnt[cur + tid] = w2cnt;
        }
        __syncthreads();

        // ---- epilogue: the tile's 256 elements are split over its KSPLIT waves ----
        if (active && p.mode == 1) {
            // forward-only launch of the sliced path: add this slice's partial currents to Xbuf
#pragma unroll
            for (int e = 0; e < EPW; ++e) {
                const int r = er[e];
                double x = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < KSPLIT; ++k2)
                    x += Xp[(size_t)(ptl + PTW * k2) * 256 + r * 64 + lane];
                const long long tg = (long long)t0 + grp + 4 * r;
                if (emine && tg < p.nT) p.Xbuf[tg * p.xstride + pt * 16 + col] += x;
            }
        } else if (active && p.mode == 0) {
            double xe[EPW], rese[EPW], terme[EPW];
            bool vte[EPW];
#pragma unroll
            for (int e = 0; e < EPW; ++e) {
                const int r = er[e];
                double x = bias_l;
                if (KTW >= 20) {                          // no register for it across the tile loop: read again (L1)
                    int nl = nloc;
                    asm volatile("" : "+v"(nl));
                    x = valid_n ? p.bias[nl] : 0.0;
                }
#pragma unroll
                for (int k2 = 0; k2 < KSPLIT; ++k2)
                    x += Xp[(size_t)(ptl + PTW * k2) * 256 + r * 64 + lane];
                const long long tg = (long long)t0 + grp + 4 * r;
                vte[e] = valid_n && (tg < p.t_hi) && emine;
                xe[e] = x;
            }
            if PGL_DBG(4) {
#pragma unroll
                for (int e = 0; e < EPW; ++e) {
                    terme[e] = xe[e] * sc[e];
                    rese[e] = xe[e] - sc[e];
                }
            } else {
                // constants come from LDS through an opaque pointer: as literals or hoisted
                // scalar loads they pin ~50 registers for the whole kernel and spill.  The
                // pointer keeps its LDS address space (ds_read, lgkmcnt only); a generic
                // pointer would turn every constant into a flat load
                pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                asm volatile("" : "+v"(Cl));
                pgl_rate_terms_n<EPW>(xe, sc, p.nlin | p.epi64, p.dt, terme, rese, Cl);
            }
#pragma unroll
            for (int e = 0; e < EPW; ++e) {
                const double res = vte[e] ? rese[e] : 0.0;
                ll_acc += vte[e] ? terme[e] : 0.0;
                gb_acc += res;
                if (emine) Rb[(size_t)ptl * 256 + er[e] * 64 + lane] = res;
            }
        }
        __syncthreads();

        // ---- backward on this wave's K slice ----
        if (active && p.want_grad && !PGL_DBG(16) && p.mode != 1) {
            double rr[4];
            if (p.mode == 2) {              // residuals of the sliced path come from Xbuf
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long tg = (long long)t0 + grp + 4 * r;
                    rr[r] = (tg < p.nT) ? p.Xbuf[tg * p.xstride + pt * 16 + col] : 0.0;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = Rb[(size_t)ptl * 256 + r * 64 + lane];
            }
            const FT* fb = Fs + grp * rsf + kcol0 + col;
            constexpr int NS = 4 * KTW;
            constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = (double)fb[(4 * (s / KTW)) * rsf + 16 * (s % KTW)];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS)
                    ar[s % PD] = (double)fb[(4 * ((s + PD) / KTW)) * rsf + 16 * ((s + PD) % KTW)];
                G[s % KTW] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KTW], G[s % KTW], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

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
}

// ---------------------------------------------------------------------------
// Fused ll + grad kernel, version 3 ("two-pass"): one workgroup = 8 waves = 8 post-synaptic
// tiles (128 neurons), one wave per tile, NO K split in the forward pass.
//
// The gradient accumulator G of a 128-neuron post block (128 x 640 f64 = 655 KB) does not fit one
// CU's registers; version 2 therefore runs two workgroups per time tile (4 post tiles each) and
// both generate the same feature tile.  Here the workgroup walks its time chunk twice instead:
//   pass 1, per tile: generate the full F tile once | forward X = F.Wmat over ALL of K (the wave
//           owns the whole 16x16 block of x: no partial exchange through LDS) | rate epilogue on
//           the accumulator registers (4 elements per lane, carried together) | residuals r stay
//           in registers in the B-operand layout and go to HBM once (2 KB per wave and tile) |
//           backward G += F^T r for the FIRST half of the feature columns.  2 barriers per tile.
//   pass 2, per tile: regenerate only the SECOND half of the columns into a double-buffered
//           half-width F tile | backward for that half with r read back (gen(tile+1) and the
//           MFMAs of tile are independent: one barrier pair per tile, waves drift).
// Feature generation per 128 post neurons and tile: 1.5 tiles' worth instead of 2.
// ---------------------------------------------------------------------------
template <int CAP>
__device__ __forceinline__ void pgl_stage_prologue(const FusedParams& p, int2* s_spk, int* s_lo, int* s_cnt,
                                                   int* s_valid, const int N, const int tile_beg,
                                                   const int tile_end, const int t0_ref, const int oddoff,
                                                   const int tid, const int nthr)
{
    if (tid < N) {
        for (int q = 0; q < 2; ++q) {
            const int tl = tile_beg + q;
            int lo = 0, cnt = 0;
            if (tl < tile_end) {
                lo = p.wlo[(size_t)tl * p.Nall + p.np0 + tid];
                cnt = p.whi[(size_t)tl * p.Nall + p.np0 + tid] - lo;
            }
            s_lo[(tl & 1) * N + tid] = lo;
            s_cnt[(tl & 1) * N + tid] = cnt;
        }
    }
    __syncthreads();
    const int pb0 = (tile_beg & 1) * N;
    for (int id = tid; id < N * CAP; id += nthr) {
        const int np = id / CAP, sl = id % CAP;
        const int cnt = s_cnt[pb0 + np];
        if (cnt <= CAP && sl < cnt) {
            const int idx = s_lo[pb0 + np] + sl;
            s_spk[np * CAP + (idx & (CAP - 1))] = pgl_decode_event<8>(p.spk[idx], t0_ref, oddoff);
        }
    }
    if (tid < N) s_valid[tid] = (s_cnt[pb0 + tid] <= CAP) ? 1 : 0;
    __syncthreads();
}

template <int KTH, int CAP, int PASS>
__global__ __launch_bounds__(512, 2) void k_fused3(const FusedParams p)
{
    // the two passes are two launches of this kernel (PASS = 1, 2) over the same grid: compiled
    // together, the register allocator spilled a third of G around every phase
    typedef double FT;
    constexpr int TT = 16, NW = 8, ESZ = 8, NPF = 3;
    constexpr int KT_ALL = 2 * KTH;
    constexpr int KS_ALL = 4 * KT_ALL;
    constexpr int C0 = KTH * 16;                 // first feature column of the second half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int nthr = NW * 64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nPB = (p.nPT + NW - 1) / NW;
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
    const int pt = pb * NW + wave;
    const bool active = pt < p.nPT;

    const int N = p.N, B = p.B, R = p.R, rsf = p.rsf, RP = p.RP;
    constexpr int rsfh = C0 + ((C0 % 32 == 0) ? 16 : 32);   // row stride of the half-width tiles of pass 2
                                                            // (rsfh % 32 == 16: rows 128 B apart mod 256 B)
    // LDS carve: [F tile (pass 1) | two half tiles (pass 2)] basis tables, event ring, windows, constants
    FT* Fs = reinterpret_cast<FT*>(smem);
    size_t off = ((size_t)TT * rsf * sizeof(FT) + 15) & ~(size_t)15;
    {
        const size_t two = ((size_t)2 * TT * rsfh * sizeof(FT) + 15) & ~(size_t)15;
        if (two > off) off = two;
    }
    FT* phiE = reinterpret_cast<FT*>(smem + off);
    FT* phiO = phiE + (size_t)B * RP;
    off += (((size_t)2 * B * RP * ESZ) + 15) & ~(size_t)15;
    int2* s_spk = reinterpret_cast<int2*>(smem + off);
    off += (size_t)N * CAP * 8;
    int* s_lo = reinterpret_cast<int*>(smem + off);           // [2][N]
    off += (((size_t)2 * N * 4) + 15) & ~(size_t)15;
    int* s_cnt = reinterpret_cast<int*>(smem + off);          // [2][N]
    off += (((size_t)2 * N * 4) + 15) & ~(size_t)15;
    int* s_valid = reinterpret_cast<int*>(smem + off);        // [N]
    off += (((size_t)N * 4) + 15) & ~(size_t)15;
    double* Cs = reinterpret_cast<double*>(smem + off);       // [32] math constants
    if (tid < 32) Cs[tid] = PGL_C[tid];

    for (int i = tid; i < B * RP; i += nthr) {
        const int b = i / RP, k = i - b * RP;
        phiE[i] = (FT)((k >= 16 && k < 16 + R) ? p.phi[b * R + k - 16] : 0.0);
        phiO[i] = (FT)((k + 1 >= 16 && k + 1 < 16 + R) ? p.phi[b * R + k + 1 - 16] : 0.0);
    }
    for (int i = tid; i < TT * rsf; i += nthr) Fs[i] = (FT)0;

    d4_t G[KTH];
#pragma unroll
    for (int kt = 0; kt < KTH; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.pidx ? p.pidx[valid_n ? nloc : 0] : p.n_lo + (valid_n ? nloc : 0);
    const double bias_l = valid_n ? p.bias[nloc] : 0.0;
    const double* __restrict__ wrow = p.Wfrag + (size_t)(active ? pt : 0) * KS_ALL * 64;
    const int oddoff = B * RP * ESZ;
    const unsigned char* phiBytes = reinterpret_cast<const unsigned char*>(phiE);

    const int tile_beg = p.tile0 + chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.tile0 + p.nTiles) tile_end = p.tile0 + p.nTiles;
    const int t0_ref = tile_beg * TT;
    // residual slab of this wave: [tile - tile0][pt][4][64]
    double* const rslab = p.Xbuf + ((size_t)(active ? pt : 0)) * 256 + lane;
    const size_t rstride = (size_t)p.nPT * 256;

    if constexpr (PASS == 1) {
    // =============================== pass 1 ===============================
    pgl_stage_prologue<CAP>(p, s_spk, s_lo, s_cnt, s_valid, N, tile_beg, tile_end, t0_ref, oddoff, tid, nthr);
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        const int t0 = tile * TT;
        const int cur = (tile & 1) * N;
        const int nxt = ((tile + 1) & 1) * N;
        int2 pf[NPF];
        int pf_new = -1;
#pragma unroll
        for (int q = 0; q < NPF; ++q) pf[q] = make_int2(0, 0);
        if (tid < N && tile + 1 < tile_end) {
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int cnt_n = s_cnt[nxt + tid];
            pf_new = s_lo[nxt + tid] + cnt_n - hi;
            if (!s_valid[tid]) pf_new = NPF + 1;
            if (cnt_n <= CAP && pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new) pf[q] = p.spk[hi + q];
            }
        }
        int w2lo = 0, w2cnt = 0;
        if (tid < N && tile + 2 < tile_end) {
            w2lo = p.wlo[(size_t)(tile + 2) * p.Nall + p.np0 + tid];
            w2cnt = p.whi[(size_t)(tile + 2) * p.Nall + p.np0 + tid] - w2lo;
        }
        if (p.Dstim > 0) {
            for (int id = tid; id < TT * p.Dstim; id += nthr) {
                const int t = id / p.Dstim;
                const int j = id % p.Dstim;
                const long long tg = (long long)t0 + t;
                Fs[t * rsf + p.Kimp + j] = (tg < p.nT) ? p.fstim[tg * p.DsAll + p.ds0 + j] : 0.0;
            }
        }
        if (!PGL_DBG(1)) {
            if (B == 5)
                gen_items<5, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
            else if (B == 3)
                gen_items<3, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
            else
                gen_items<0, CAP, FT>(Fs, rsf, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ);
        }
        __syncthreads();

        // post-synaptic counts of this lane's four elements (rows grp + 4r of neuron nglob)
        unsigned scb[4];                           // raw counts (converted in the epilogue)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long tg = (long long)t0 + grp + 4 * r;
            const long long tc = (tg < p.nT) ? tg : (p.nT - 1);
            scb[r] = p.S[tc * p.Nall + nglob];
        }
        // ---- forward over all of K ----
        d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
        d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
        if (active && !PGL_DBG(8)) {
            const FT* fa = Fs + col * rsf + grp;
            const double* wr_s = wrow;
            asm volatile("" : "+s"(wr_s));
            constexpr int PW2 = (KS_ALL / 2 < PGL_PW / 2) ? KS_ALL / 2 : PGL_PW / 2;
            constexpr int PA = 4;
            const pgl_glb_cd2p wr2 = (pgl_glb_cd2p)wr_s;
            pgl_d2 wr[PW2];
            double ar[PA];
#pragma unroll
            for (int s = 0; s < PW2; ++s) wr[s] = wr2[s * 64 + lane];
#pragma unroll
            for (int s = 0; s < PA; ++s) ar[s] = fa[4 * s];
#pragma unroll
            for (int s = 0; s < KS_ALL; ++s) {
                const double a = ar[s % PA];
                const double b = (s & 1) ? wr[(s / 2) % PW2].y : wr[(s / 2) % PW2].x;
                if (s + PA < KS_ALL) ar[s % PA] = fa[4 * (s + PA)];
                if ((s & 1) && (s / 2 + PW2 < KS_ALL / 2)) wr[(s / 2) % PW2] = wr2[(s / 2 + PW2) * 64 + lane];
                if (s & 1)
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- commit the staging of the next tile (the ring and the window buffer of `tile` were
        // last read by gen(tile), before the barrier above) ----
        if (pf_new >= 0) {
            int2* ring = s_spk + tid * CAP;
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int lo_n = s_lo[nxt + tid], cnt_n = s_cnt[nxt + tid];
            if (cnt_n > CAP) {
                s_valid[tid] = 0;
            } else if (pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new)
                        ring[(hi + q) & (CAP - 1)] = pgl_decode_event<ESZ>(pf[q], t0_ref, oddoff);
            } else {
                for (int idx = lo_n; idx < lo_n + cnt_n; ++idx)
                    ring[idx & (CAP - 1)] = pgl_decode_event<ESZ>(p.spk[idx], t0_ref, oddoff);
                s_valid[tid] = 1;
            }
        }
        if (tid < N) {
            s_lo[cur + tid] = w2lo;
            s_cnt[cur + tid] = w2cnt;
        }
        // ---- epilogue on the accumulator registers: 4 elements per lane, two at a time (the
        // temporaries of four interleaved chains would push G out of the register file) ----
        double rr[4];
        if (active) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                double xe[2], se[2], terme[2], rese[2];
                bool vte[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * h2 + e;
                    xe[e] = bias_l + (acc0[r] + acc1[r]);
                    se[e] = (double)scb[r];
                    const long long tg = (long long)t0 + grp + 4 * r;
                    vte[e] = valid_n && (tg < p.t_hi);
                }
                if PGL_DBG(4) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        terme[e] = xe[e] * se[e];
                        rese[e] = xe[e] - se[e];
                    }
                } else {
                    pgl_lds_cdp Cl = (pgl_lds_cdp)Cs;
                    asm volatile("" : "+v"(Cl));
                    pgl_rate_terms_n<2>(xe, se, p.nlin | p.epi64, p.dt, terme, rese, Cl);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double res = vte[e] ? rese[e] : 0.0;
                    rr[2 * h2 + e] = res;
                    ll_acc += vte[e] ? terme[e] : 0.0;
                    gb_acc += res;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) rr[r] = 0.0;
        }
        // ---- residuals to HBM for pass 2, backward for the first half of the columns ----
        if (active && p.want_grad && !PGL_DBG(16)) {
            double* rs = rslab + (size_t)(tile - p.tile0) * rstride;
            if (!PGL_DBG(512)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[r * 64] = rr[r];
            }
            const FT* fb = Fs + grp * rsf + col;
            constexpr int NS = 4 * KTH;
            constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = fb[(4 * (s / KTH)) * rsf + 16 * (s % KTH)];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS) ar[s % PD] = fb[(4 * ((s + PD) / KTH)) * rsf + 16 * ((s + PD) % KTH)];
                G[s % KTH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KTH], G[s % KTH], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    if (active) {
        pgl_store_ll(p, chunk, pt, 0, 1, lane, ll_acc, gb_acc);
    }
    if (active && p.want_grad) {
        double* gp = pgl_gpart(p.Gpart, pt, KT_ALL, 0, p.nChunks, chunk, lane);
            const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
        for (int kt = 0; kt < KTH; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
        }
    }
    } else {
    // =============================== pass 2 ===============================
    // columns [C0, Ktot): impulse columns of the presynaptic neurons >= C0 / B and the stimulus
    // columns; two half-width tiles Fh[buf][16][rsfh], column c at offset c - C0.
    FT* Fh0 = Fs;
    FT* Fh1 = Fs + (size_t)TT * rsfh;
    for (int i = tid; i < 2 * TT * rsfh; i += nthr) Fs[i] = (FT)0;
    const int np_first = C0 / B;                  // first neuron with a column in this half
    pgl_stage_prologue<CAP>(p, s_spk, s_lo, s_cnt, s_valid, N, tile_beg, tile_end, t0_ref, oddoff, tid, nthr);

    auto gen_half = [&](FT* Fdst, const int tile, const int cur) {
        const int t0 = tile * TT;
        if (p.Dstim > 0) {
            for (int id = tid; id < TT * p.Dstim; id += nthr) {
                const int t = id / p.Dstim;
                const int j = id % p.Dstim;
                const long long tg = (long long)t0 + t;
                Fdst[t * rsfh + p.Kimp - C0 + j] = (tg < p.nT) ? p.fstim[tg * p.DsAll + p.ds0 + j] : 0.0;
            }
        }
        if (p.Kimp > C0 && !PGL_DBG(1)) {
            FT* Fv = Fdst - C0;                    // column c of the full layout lands at c - C0
            if (B == 5)
                gen_items<5, CAP, FT>(Fv, rsfh, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ, C0 * 4);
            else if (B == 3)
                gen_items<3, CAP, FT>(Fv, rsfh, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ, C0 * 4);
            else
                gen_items<0, CAP, FT>(Fv, rsfh, phiBytes, RP, s_spk, s_lo + cur, s_cnt + cur, p.spk, t0, B,
                                      p.Kimp, tid, nthr, (t0 - t0_ref) * ESZ, C0 * 4);
        }
    };
    auto bwd_half = [&](const FT* Fsrc, const double (&rq)[4]) {
        const FT* fb = Fsrc + grp * rsfh + col;
        constexpr int NS = 4 * KTH;
        constexpr int PD = (NS < PGL_PD) ? NS : PGL_PD;
        double ar[PD];
#pragma unroll
        for (int s = 0; s < PD; ++s) ar[s] = fb[(4 * (s / KTH)) * rsfh + 16 * (s % KTH)];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double a = ar[s % PD];
            if (s + PD < NS) ar[s % PD] = fb[(4 * ((s + PD) / KTH)) * rsfh + 16 * ((s + PD) % KTH)];
            G[s % KTH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rq[s / KTH], G[s % KTH], 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    };

    gen_half((tile_beg & 1) ? Fh1 : Fh0, tile_beg, (tile_beg & 1) * N);
    double rv[4] = {0.0, 0.0, 0.0, 0.0};
    // staging registers for the windows of tile_beg + 1 (committed at the top of the first iteration)
    int2 pf[NPF];
    int pf_new = -1;
    int w2lo = 0, w2cnt = 0;
    auto prefetch_stage = [&](const int g) {       // events entering window(g+1), windows of g+2
        const int cur = (g & 1) * N, nxt = ((g + 1) & 1) * N;
        pf_new = -1;
#pragma unroll
        for (int q = 0; q < NPF; ++q) pf[q] = make_int2(0, 0);
        if (tid >= np_first && tid < N && g + 1 < tile_end) {
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int cnt_n = s_cnt[nxt + tid];
            pf_new = s_lo[nxt + tid] + cnt_n - hi;
            if (!s_valid[tid]) pf_new = NPF + 1;
            if (cnt_n <= CAP && pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new) pf[q] = p.spk[hi + q];
            }
        }
        w2lo = 0;
        w2cnt = 0;
        if (tid < N && g + 2 < tile_end) {
            w2lo = p.wlo[(size_t)(g + 2) * p.Nall + p.np0 + tid];
            w2cnt = p.whi[(size_t)(g + 2) * p.Nall + p.np0 + tid] - w2lo;
        }
    };
    auto commit_stage = [&](const int g) {         // ring <- window(g+1), window buffer of g <- g+2
        const int cur = (g & 1) * N, nxt = ((g + 1) & 1) * N;
        if (pf_new >= 0) {
            int2* ring = s_spk + tid * CAP;
            const int hi = s_lo[cur + tid] + s_cnt[cur + tid];
            const int lo_n = s_lo[nxt + tid], cnt_n = s_cnt[nxt + tid];
            if (cnt_n > CAP) {
                s_valid[tid] = 0;
            } else if (pf_new <= NPF) {
#pragma unroll
                for (int q = 0; q < NPF; ++q)
                    if (q < pf_new)
                        ring[(hi + q) & (CAP - 1)] = pgl_decode_event<ESZ>(pf[q], t0_ref, oddoff);
            } else {
                for (int idx = lo_n; idx < lo_n + cnt_n; ++idx)
                    ring[idx & (CAP - 1)] = pgl_decode_event<ESZ>(p.spk[idx], t0_ref, oddoff);
                s_valid[tid] = 1;
            }
        }
        if (tid < N) {
            s_lo[cur + tid] = w2lo;
            s_cnt[cur + tid] = w2cnt;
        }
    };
    prefetch_stage(tile_beg);
    __syncthreads();                               // first half tile complete

    for (int tile = tile_beg; tile < tile_end; ++tile) {
        // gen(tile) has finished everywhere (barrier): the ring may move on to window(tile+1)
        commit_stage(tile);
        __syncthreads();
        prefetch_stage(tile + 1);
        const FT* Fcur = (tile & 1) ? Fh1 : Fh0;
        FT* Fnext = (tile & 1) ? Fh0 : Fh1;
        const bool more = tile + 1 < tile_end;
        // r(tile) is loaded before the generation of the next half tile and consumed after it; the
        // two half tiles make gen(tile+1) and the MFMAs of `tile` independent, so no barrier
        // separates them and waves drift apart (a wave that finishes generating early multiplies
        // while its SIMD neighbour still generates).  Forcing that overlap -- waves 0-3 generate
        // first, 4-7 multiply first -- measured 5 % slower: one MFMA wave per SIMD does not fill
        // the pipe.
        if (active && !PGL_DBG(512)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = rslab[(size_t)(tile - p.tile0) * rstride + r * 64];
        }
        if (more) gen_half(Fnext, tile + 1, ((tile + 1) & 1) * N);
        if (active && !PGL_DBG(16)) bwd_half(Fcur, rv);
        __syncthreads();
    }

    if (active) {
        double* gp = pgl_gpart(p.Gpart, pt, KT_ALL, KTH, p.nChunks, chunk, lane);
            const size_t gcs = (size_t)p.nChunks * 64;
#pragma unroll
        for (int kt = 0; kt < KTH; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * gcs] = G[kt][r];
        }
    }
    }
}
