// shared definitions: launch parameters, f64 elementary functions, the rate epilogue (pgl_rate_fx)
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef double d4_t __attribute__((ext_vector_type(4)));

#define PGL_CAP 16          // staged spike events per presynaptic neuron and tile
#ifndef PGL_PD
#define PGL_PD 4             // F^T fragment prefetch depth (MFMA steps) of the backward passes
#endif
#ifndef PGL_PW
#define PGL_PW 8             // Wmat fragment prefetch depth (MFMA steps) of the forward passes
#endif
#define PGL_MAXB 8
// timing-ablation switches of the tile loops (dev option 99, tools/quick_bench.py, tools/gibbs_ablate.py): compiled in
// only with -DPGL_ABLATE (tools/build_variant.sh); the shipped library has none of these branches
#ifdef PGL_ABLATE
#define PGL_DBG(bit) ((p.dbg & (bit)) != 0)
#else
#define PGL_DBG(bit) (false)
#endif
#ifndef PGL_PRIO
#define PGL_PRIO 1           // k_fused5: waves 4-7 lead the first half of every MFMA loop (s_setprio)
#ifndef PGL_EPI_PRIO
#define PGL_EPI_PRIO 1       // k_fused6 / k_fused7: issue priority of a wave inside its rate epilogue (s_setprio): the epilogue is
                             // a chain of dependent VALU instructions between two workgroup barriers, the MFMA loops of the other
                             // workgroups on the SIMD are throughput work that fills whatever it leaves -- measured C2 0.1447 ->
                             // 0.1412 ms (priority 1, 2 and 3 alike), C5 and C1 within the noise
#endif
#endif
#ifndef PGL_EBAR
#define PGL_EBAR 1           // k_fused5: barrier between the epilogue and the backward loop
#endif
#ifndef PGL_EPI_F32
#define PGL_EPI_F32 1        // pgl_rate4: single-precision exp for the log1p / sigmoid corrections when the whole wave has x > 12
#endif
#ifndef PGL_DS1
#define PGL_DS1 4            // k_fused5: cap on the MFMAs between two DMA rounds of a backward pass (0 = spread evenly over it).
                             // 4: the requests leave in the first half of the pass and have the second half to land (A/B over
                             // four interleaved runs each: 3.23-3.26 ms against 3.27-3.30 ms spread evenly, 2: 3.20-3.35)
#endif
#ifndef PGL_ENE
#define PGL_ENE 4            // elements a lane carries through the rate epilogue together (k_fused5):
                             // 4 = pgl_rate4 (fixed instruction order), 2 = pgl_rate_terms_n<2> only
#endif

#ifndef PGL_DMA_IL
#define PGL_DMA_IL 1         // k_fused6 / k_fused7: the LDS-DMA pieces of the next tile go out BETWEEN the MFMAs of the forward
                             // (k_fused7 with streamed Wmat: backward) loop instead of as a burst behind the "landed" barrier:
                             // a wave waits ~500 cycles per piece for the CU's address path (phase profile, 3 workgroups per
                             // CU: 3 000 of 11 600 cycles per tile and wave at C2) -- behind an MFMA that wait is free
#endif
#ifndef PGL_WREG_MAX
#define PGL_WREG_MAX 28      // k_fused7: up to this many k-steps the wave's Wmat fragments stay in registers for the whole chunk
                             // (40 = ten k-tiles: 80 registers of fragments beside 80 of G and the epilogue spilled 4-74 VGPRs)
#endif
#ifndef PGL_LDS1
#define PGL_LDS1 1           // A-fragment reads of the resident-tile kernels as single ds_read_b64 (volatile: the compiler's
                             // load/store optimizer otherwise pairs them into ds_read2_b64, which is served in 16-lane groups
                             // with a 32-bank modulus at half the rate -- and the forward pattern col*RS + grp conflicts 2-way
                             // there, SQ_LDS_BANK_CONFLICT = 4 cycles per read; ds_read_b64: 32-lane groups, 64 banks, no conflict)
#endif
__device__ __forceinline__ double pgl_lds_f64(const double* p)
{
#if PGL_LDS1
    return *(const volatile __attribute__((address_space(3))) double*)p;
#else
    return *p;
#endif
}

// Per-chunk partials of G: [post tile][k-tile][r][chunk][64 lanes] -- all chunk partials of one 64-element
// fragment are contiguous (chunk stride 512 B), so the reduction over chunks (k_finalize) is a streaming read;
// the fused kernels write one 512-byte piece per fragment at the end of their chunk.
__device__ __forceinline__ double* pgl_gpart(double* G, const int pt, const int KT, const int kt0,
                                             const int nChunks, const int chunk, const int lane)
{
    return G + (((size_t)pt * KT + kt0) * 4) * ((size_t)nChunks * 64) + (size_t)chunk * 64 + lane;
}

struct FusedParams {
    // problem
    long long nT;
    int N, B, R, nlin;
    int Dstim, Kimp, Ktot;
    double dt;
    // data
    const int2* __restrict__ spk;        // events (t, count), grouped by neuron, time-sorted
    const int* __restrict__ wlo;         // [nT16][N] first event with s >= 16*tile - R
    const int* __restrict__ whi;         // [nT16][N] first event with s >= 16*tile + 15
    const uint8_t* __restrict__ S;       // (nT,N) counts
    const double* __restrict__ fstim;    // (nT,Dstim) or null
    const double* __restrict__ phi;      // [B][R] transposed basis
    // per-call
    const double* __restrict__ Wfrag;    // [nPT][KS][64]
    const double* __restrict__ bias;     // [nPT*16]
    int n_lo, npost, nPT;
    int nT16, tilesPerChunk, nChunks, nTiles;
    int rsf;                             // F row stride in elements
    int RP;                              // padded basis-table length (>= R+32, RP % 32 == 8)
    double* __restrict__ Gpart;          // [nChunks][nPT][KT][4][64]
    double* __restrict__ llpart;         // [nPT * 16 neurons][nChunks][KSPLIT] (pgl_store_ll)
    double* __restrict__ gbpart;         // same layout
    // feature-column slice (general path for N > 128 or more than 640 columns): this launch
    // covers presynaptic neurons [np0, np0+N) and stimulus columns [ds0, ds0+Dstim); N / Dstim /
    // Kimp above are then the slice's, Nall / DsAll the strides of S, the window tables and fstim
    int Nall, np0, DsAll, ds0;
    int mode;                            // 0 fused; 1 forward only: X += F.W; 2 backward only: r from Rbuf
    double* __restrict__ Xbuf;           // (nT, xstride) currents / residuals of the 3-phase path
    int xstride;
    int tile0;                           // first 16-row tile of the evaluated time range
    long long t_hi;                      // rows >= t_hi are excluded from ll / gradient
    int want_grad;
    int dbg;                             // timing ablation bits (results invalid when != 0)
    const unsigned char* __restrict__ Fimg;   // resident feature tiles (k_fused5), else null
    int img_tile0;                       // first tile the resident images cover
    const int* __restrict__ pidx;        // post-synaptic neuron of local column j (null: n_lo + j) -- an
                                         // arbitrary subset of neurons per launch (pgl_ll_grad_list_dev)
    // kernels that keep their Wmat fragments in registers (k_fused6, k_fused7 up to 40 k-steps) gather them
    // straight from the caller's theta (npost, P) and Weff (Nall, Nall) -- no k_prep_w launch, no Wfrag round
    // trip; null = read Wfrag / bias
    const double* __restrict__ theta;
    const double* __restrict__ Weff;
    int P;
    int epi64;                           // 2: all-f64 rate epilogue (PGL_OPT_EPI_F64), 0: default (see pgl_rate4)
    // separable stimulus at the frame rate INSIDE the forward contraction (k_fused7<.., XIO = 2>): the stimulus current
    // of a tile is five more k-steps, A = coefficient rows of the tile's 16 bins (sepA, by tile phase), B = w_t (x) z
    const double* __restrict__ sepA;     // [sepNH + q / gcd(q, 16)][5][64] A fragments (build_frame_table)
    const double* __restrict__ sepZ;     // [Tstim][sepLdy] frame-rate projections z_n of the listed rows (YfT)
    const double* __restrict__ sepTheta; // (npost, P): w_t = columns 1 .. 3
    long long sepT;                      // stimulus frames
    int sepLdy, sepQ, sepM, sepNH, sepG; // NH head tiles with their own A fragments, gcd(q, 16) = 1 << sepG
    int sepBt;                           // temporal bases in use (<= 3)
    // ... and its backward in the same kernel (want_grad): D[(j', bt)][n] += sum_i A[i][(j', bt)] r[i][n] as eight more
    // MFMAs per tile (A^T fragments by tile phase: sepAT), accumulated in registers over the tiles that share a frame
    // base and written out when the base changes -- sepD[base - sepB0][slot][post tile][5][64] (registers 0..3 = columns
    // 0..15 in the accumulator layout, 4 = columns 16, 17 in lane groups 0, 1), slot = chunk - (chunk of the base's first
    // tile).  No residual slab, no k_sepf_bwd pass over it; k_sepf_finish_d folds the pieces.  Null: slab form.
    const double* __restrict__ sepAT;    // [phases as sepA][8][64]
    double* __restrict__ sepD;
    long long sepB0;                     // frame base of the evaluated range's first tile
    int sepSL;                           // slots per base
    // post-block-major grid (k_fused5 on the column slices of a wide population whose last post block is light): workgroup b
    // takes post block b / nChunks, so that every CU first runs a full block, then a light one (1 = on)
    int pb_major;
};

// A wave's ll / d ll / d bias partials (accumulator layout: lane = 16 r + column) leave the kernel summed over the four lane
// groups and NEURON-major: [neuron][chunk][K-slice wave] -- the reduction over chunks (pgl_reduce_ll, one block per neuron)
// then reads one contiguous run.  (In the accumulator layout a neuron owns 8 bytes of every 128-byte line: each load
// instruction of the reduction touched 64 lines and the block sat on its CU's L1 for 19 us at C1 / C2 -- two thirds of
// k_finalize.)
__device__ __forceinline__ void pgl_store_ll(const FusedParams& p, const int chunk, const int pt, const int sub,
                                             const int nsub, const int lane, double ll, double gb)
{
    ll += __shfl_xor(ll, 16, 64);
    gb += __shfl_xor(gb, 16, 64);
    ll += __shfl_xor(ll, 32, 64);
    gb += __shfl_xor(gb, 32, 64);
    if (lane < 16) {
        const size_t o = (((size_t)pt * 16 + lane) * p.nChunks + chunk) * nsub + sub;
        p.llpart[o] = ll;
        p.gbpart[o] = gb;
    }
}

// geometry shared by k_fused7 (writer) and k_sepf_finish_d (reader): the first tile whose frame base is b, for the
// evaluated tiles [tile0, ...), and the chunk that holds it
__device__ __forceinline__ long long pgl_sepd_first_tile(const long long b, const int M, const int q, const long long tile0)
{
    const long long t = (b <= 0) ? 0 : ((b + M) * (long long)q + 15) / 16;
    return t > tile0 ? t : tile0;
}

// The Wmat B fragments (k-steps ks0 .. ks0 + NS - 1, lane group grp) of local post neuron nloc, as k_prep_w would
// write them:  Wmat[k][n] = theta_n[1 + Dstim + k] * Weff[k / B][n]  (impulse columns),  theta_n[1 + k - Kimp]
// (stimulus columns), k = 4 ks + grp.  Branch-free: every load goes to a clamped, always valid address and all
// 2 NS loads are in flight together (one L2 / HBM latency per workgroup instead of one per k-step).
template <int NS>
__device__ __forceinline__ void pgl_wfrag_direct(const FusedParams& p, const int ks0, const int grp, const int nloc,
                                                 const int nglob, const bool valid_n, double (&w)[NS])
{
    const double* row = p.theta + (size_t)(valid_n ? nloc : 0) * p.P;
    const double* wcol = p.Weff + (valid_n ? nglob : 0);
    const int imp0 = 1 + p.DsAll + p.np0 * p.B, st0 = 1 + p.ds0 - p.Kimp;
    const float rB = 1.0f / (float)p.B;      // k / B for k < 2^20, B <= 8: (k + 0.5) / B is >= 1/16 away from an integer
    double tv[NS], wv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * (ks0 + s) + grp;
        const int kc = (k < p.Ktot) ? k : 0;
        const bool imp = kc < p.Kimp;
        const int npre = p.np0 + (int)(((float)(imp ? kc : 0) + 0.5f) * rB);
        tv[s] = row[(imp ? imp0 : st0) + kc];
        wv[s] = wcol[(size_t)npre * p.Nall];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * (ks0 + s) + grp;
        const double v = (k < p.Kimp) ? tv[s] * wv[s] : tv[s];
        w[s] = (valid_n && k < p.Ktot) ? v : 0.0;
    }
}

// ---------------------------------------------------------------------------
// f64 elementary functions of the epilogue.  Hand-rolled (instead of the ocml calls)
// because on gfx950 every f64 VALU instruction competes with the f64 MFMA for the same
// DP pipeline (tools/ubench): the rate epilogue is pure overhead on the MFMA roofline, so
// its instruction count matters.  Accuracy ~1 ulp (prototype: tools/ubench/proto_math.py).
// ---------------------------------------------------------------------------
// polynomial / reduction constants, read with scalar loads (SGPRs): as f64 literals they would
// be hoisted into ~50 loop-invariant VGPRs and spilled around the MFMA accumulators
__constant__ double PGL_C[32] = {
    1.4426950408889634,            //  0 log2(e)
    6.93147180369123816490e-01,    //  1 ln2 hi
    1.90821492927058770002e-10,    //  2 ln2 lo
    1.6059043836821613e-10,        //  3 1/13!
    2.08767569878681e-09,          //  4 1/12!
    2.505210838544172e-08,         //  5 1/11!
    2.755731922398589e-07,         //  6 1/10!
    2.7557319223985893e-06,        //  7 1/9!
    2.48015873015873e-05,          //  8 1/8!
    0.0001984126984126984,         //  9 1/7!
    0.001388888888888889,          // 10 1/6!
    0.008333333333333333,          // 11 1/5!
    0.041666666666666664,          // 12 1/4!
    0.16666666666666666,           // 13 1/3!
    1.479819860511658591e-01,      // 14 Lg7
    1.531383769920937332e-01,      // 15 Lg6
    1.818357216161805012e-01,      // 16 Lg5
    2.222219843214978396e-01,      // 17 Lg4
    2.857142874366239149e-01,      // 18 Lg3
    3.999999999940941908e-01,      // 19 Lg2
    6.666666666666735130e-01,      // 20 Lg1
    0.70710678118654752440,        // 21 sqrt(1/2)
    1.0 / 3.0,                     // 22
    9.6e-5,                        // 23 series threshold on exp(-|x|)
    0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ double pgl_rcp(const double b)
{
    double r = __builtin_amdgcn_rcp(b);           // v_rcp_f64 seed + two Newton steps
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}

// exp(y): k = rint(y/ln2), r = y - k ln2 (hi/lo split), degree-13 Taylor in |r| <= 0.347, ldexp
template <typename CP>
__device__ __forceinline__ double pgl_exp(const double y, const CP C)
{
    const double k = rint(y * C[0]);
    double r = fma(-k, C[1], y);
    r = fma(-k, C[2], r);
    double p = C[3];
#pragma unroll
    for (int i = 4; i <= 13; ++i) p = fma(p, r, C[i]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double kc = fmin(fmax(k, -2200.0), 2200.0);
    return ldexp(p, (int)kc);
}

// log(v) for v >= 0 (fdlibm e_log.c scheme: v = 2^e m, m in [sqrt(1/2), sqrt 2),
// s = f/(2+f), 7-term polynomial in s^2); log(0) = -inf, inf/NaN pass through
template <typename CP>
__device__ __forceinline__ double pgl_log(const double v, const CP C)
{
    double m = __builtin_amdgcn_frexp_mant(v);    // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(v);
    const bool lt = m < C[21];
    m = lt ? m + m : m;
    e = lt ? e - 1 : e;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    const double rc = pgl_rcp(den);
    double s = f * rc;
    s = fma(fma(-den, s, f), rc, s);
    const double z = s * s;
    double R = C[14];
#pragma unroll
    for (int i = 15; i <= 20; ++i) R = fma(R, z, C[i]);
    R = R * z;
    const double hfsq = 0.5 * f * f;
    const double de = (double)e;
    double res = de * C[1] - ((hfsq - (s * (hfsq + R) + de * C[2])) - f);
    res = (v == 0.0) ? -__builtin_huge_val() : res;
    res = (v < __builtin_huge_val()) ? res : v;
    return res;
}

// One element of the rate epilogue (glm.py:43-52 and its derivative w.r.t. x):
//   explinear: lam = log(1+exp(x)) in the overflow-safe form max(x,0) + log1p(exp(-|x|)),
//              term = -dt*lam + s*log(lam),  r = (-dt + s/lam) * sigmoid(x)
//   exp:       lam = exp(x), term = -dt*lam + s*x, r = -dt*lam + s
// log(lam) and 1/lam are only evaluated in waves where some lane has a spike (s > 0);
// when every lane of the wave has exp(-|x|) < 9.6e-5 (|x| > 9.25, the operating regime of
// standard_glm's bias ~ 20) log1p and 1/(1+e) come from their alternating series (error < e^6).
template <int NE, typename CP>
__device__ __forceinline__ void pgl_rate_terms_n(const double (&x)[NE], const double (&s)[NE],
                                                 const int nlin_, const double dt, double (&term)[NE],
                                                 double (&res)[NE], const CP C)
{
    // nlin_: PGL_NLIN_* in bit 0; bit 1 set = all-f64 epilogue (PGL_OPT_EPI_F64: no single-precision correction)
    const int nlin = nlin_ & 1;
    const bool allf64 = (nlin_ & 2) != 0;
    // NE independent elements per lane are carried through every stage together: the Horner chains
    // are latency bound (dependent f64 FMAs), two of them interleave in the same issue slots
    if (nlin == 1) {
        double e[NE], l1p[NE], inv[NE], lam[NE], sig[NE];
        bool small = true, spike = false, hi = (PGL_EPI_F32 != 0) && !allf64;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            hi = hi && (x[i] > 12.0);
            spike = spike || (s[i] > 0.0);
        }
        const bool fast = __all(hi);               // see pgl_rate4: single-precision exp for the corrections
        if (fast) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i] = (double)__builtin_amdgcn_exp2f((float)x[i] * -1.44269504088896340736f);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                e[i] = pgl_exp(-fabs(x[i]), C);
                small = small && (e[i] < C[23]);
            }
        }
        if (fast) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double ei = e[i];
                l1p[i] = ei * fma(ei, -0.5, 1.0);
                inv[i] = fma(-ei, fma(-ei, fma(-ei, 1.0, 1.0), 1.0), 1.0);
            }
        } else if (__all(small)) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double ei = e[i];
                l1p[i] = ei * fma(-ei, fma(-ei, fma(-ei, fma(-ei, 0.2, 0.25), C[22]), 0.5), 1.0);
                inv[i] = fma(-ei, fma(-ei, fma(-ei, fma(-ei, fma(-ei, 1.0, 1.0), 1.0), 1.0), 1.0), 1.0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double u = 1.0 + e[i];
                inv[i] = pgl_rcp(u);
                l1p[i] = pgl_log(u, C) + (e[i] - (u - 1.0)) * inv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            lam[i] = fmax(x[i], 0.0) + l1p[i];
            sig[i] = (x[i] >= 0.0) ? inv[i] : e[i] * inv[i];
            term[i] = -dt * lam[i];
            res[i] = -dt * sig[i];
        }
        if (spike) {                       // some lane of the wave holds a spike in one of its elements
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double tl = fma(pgl_log(lam[i], C), s[i], term[i]);
                const double rl = (-dt + s[i] * pgl_rcp(lam[i])) * sig[i];
                term[i] = (s[i] > 0.0) ? tl : term[i];
                res[i] = (s[i] > 0.0) ? rl : res[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            // reference semantics at lam == 0 (x < -745): log(0)*S = -inf*0 = NaN (glm.py:52), and
            // its derivative S*lam'/lam = 0/0 = NaN (callers zero NaN gradients, coord_descent.py:179)
            term[i] = (lam[i] == 0.0) ? __builtin_nan("") : term[i];
            res[i] = (lam[i] == 0.0 || x[i] != x[i]) ? __builtin_nan("") : res[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const double lam = pgl_exp(x[i], C);
            term[i] = fma(x[i], s[i], -dt * lam);
            res[i] = fma(-dt, lam, s[i]);
        }
    }
}

// ---------------------------------------------------------------------------
// Rate epilogue of one 16x16 tile on the accumulator layout: four elements per lane carried
// through every stage together, in a fixed instruction order (a scheduling barrier after every
// row of four).  The f64 VALU shares its pipeline with the f64 MFMA and a dependent v_fma_f64
// costs ~11 cycles against 4 of issue: the compiler's own order (register pressure first) runs
// the Horner chains one after the other, i.e. latency bound.  Constants arrive by scalar loads
// (SGPR operands: no VGPRs, no LDS reads) through a pointer the caller has made opaque inside the
// tile loop, so the loads are not hoisted out of it.
//   explinear, every lane in the series regime (exp(-|x|) < 9.6e-5, i.e. |x| > 9.25): handled here;
//   exp nonlinearity: handled here; anything else (or lam == 0 / NaN): returns false and the caller
//   takes the general two-at-a-time path (pgl_rate_terms_n).
// The spike terms  S log(lam)  and  S/lam  concern ~2 % of the elements: they are compacted
// through a per-wave LDS scratch (one element per lane) and evaluated once per tile instead of
// once per accumulator register.
//   x[4] currents, sc[4] spike counts; res[4] = d ll / d x; returns the tile's ll contribution
//   of this lane in `term`.
// ---------------------------------------------------------------------------
// Phase timeline instrumentation (dev builds only: hipcc -DPGL_PROF, tools/phase_profile.py):
// per wave, cycles between the phase boundaries of a tile summed over the chunk.
#ifdef PGL_PROF
__device__ long long g_pgl_prof[2][4096][8][12];          // [pass-1][workgroup][wave][phase]
__device__ long long g_pgl_prof_ts[4096][5];              // per workgroup: entry, loop start, loop end, exit (100 MHz ticks), XCC / SE / CU id
#define PGL_PROF_ENTRY const long long prof_entry = __builtin_amdgcn_s_memrealtime();
#define PGL_PROF_EXIT                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                               \
            g_pgl_prof_ts[blockIdx.x][0] = prof_entry;                                             \
            g_pgl_prof_ts[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();                       \
            g_pgl_prof_ts[blockIdx.x][4] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16) | \
                                           (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11));   \
        }                                                                                          \
    } while (0)
#define PGL_PROF_DECL long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long prof_t = __builtin_amdgcn_s_memtime(); \
    const long long prof_rt0 = __builtin_amdgcn_s_memrealtime(), prof_t0 = prof_t;
#define PGL_PROF_MARK(i)                                           \
    do {                                                           \
        const long long t_ = __builtin_amdgcn_s_memtime();         \
        prof_acc[i] += t_ - prof_t;                                \
        prof_t = t_;                                               \
    } while (0)
#define PGL_PROF_STORE(pass)                                                                  \
    do {                                                                                      \
        prof_acc[10] = __builtin_amdgcn_s_memtime() - prof_t0;          /* shader cycles of the loop */ \
        prof_acc[11] = __builtin_amdgcn_s_memrealtime() - prof_rt0;     /* 100 MHz ticks of the loop */  \
        if (lane == 0 && blockIdx.x < 4096)                                                   \
            for (int i_ = 0; i_ < 12; ++i_) g_pgl_prof[pass - 1][blockIdx.x][wave][i_] = prof_acc[i_]; \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                          \
            g_pgl_prof_ts[blockIdx.x][1] = prof_rt0;                                          \
            g_pgl_prof_ts[blockIdx.x][2] = prof_rt0 + prof_acc[11];                           \
        }                                                                                     \
    } while (0)
#else
#define PGL_PROF_ENTRY
#define PGL_PROF_EXIT
#define PGL_PROF_DECL
#define PGL_PROF_MARK(i)
#define PGL_PROF_STORE(pass)
#endif
#ifdef PGL_PROF
#define PGL_PROF_ARGS , long long (&prof_acc)[12], long long& prof_t
#define PGL_PROF_PASS , prof_acc, prof_t
#define PGL_PROF_DUMMY , pgl_prof_dummy_acc, pgl_prof_dummy_t
#else
#define PGL_PROF_ARGS
#define PGL_PROF_PASS
#define PGL_PROF_DUMMY
#endif
typedef double pgl_d2 __attribute__((ext_vector_type(2)));
typedef double pgl_d16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) double* pgl_k_cdp;
#define PGL_ROW __builtin_amdgcn_sched_barrier(0)

template <int NE, int CAP>
__device__ __forceinline__ bool pgl_rate_fx(const double (&x)[NE], const unsigned (&sc)[NE], const int nlin_,
                                            const double dt, const pgl_k_cdp C, double* scratch,
                                            const int lane, double& term, double (&res)[NE] PGL_PROF_ARGS)
{
    const int nlin = nlin_ & 1;                       // bit 1 of nlin_: all-f64 epilogue (PGL_OPT_EPI_F64)
    const bool allf64 = (nlin_ & 2) != 0;
    double k[NE], r[NE], q[NE], e[NE];
    // every constant of the exp / series stages is requested up front: one scalar-memory wait
    double c[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) c[j] = C[j];
    const double c13 = C[22], cthr = C[23];
    double lam[NE], sig[NE];
    bool fastdone = false;
#if PGL_EPI_F32
    if (nlin == 1 && !allf64) {
        // every element of the wave at x > 12 (the operating regime of standard_glm, bias ~ 20): exp(-x) < 6.2e-6
        // only enters lam = x + log1p(e) and sigmoid = 1/(1+e) as a correction that single precision resolves --
        // e = v_exp_f32(-x log2 e) (relative error ~1e-6: the f32 rounding of x in the exponent), so lam and the
        // sigmoid are within 6e-12 absolute = 5e-13 relative of the f64 result at x = 12 and 1e-16 at x = 20.
        // 9 instructions per element instead of 36: every VALU instruction here is an MFMA issue slot lost.
        bool hi = true;
#pragma unroll
        for (int i = 0; i < NE; ++i) hi = hi && (x[i] > 12.0);
        if (__all(hi)) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double ef = (double)__builtin_amdgcn_exp2f((float)x[i] * -1.44269504088896340736f);
                lam[i] = fma(ef, fma(ef, -0.5, 1.0), x[i]);                  // x + e (1 - e/2), e^3/3 < 8e-17
                sig[i] = fma(-ef, fma(-ef, fma(-ef, 1.0, 1.0), 1.0), 1.0);   // 1 - e + e^2 - e^3
            }
            fastdone = true;
        }
    }
#endif
    if (!fastdone) {
    // ---- e = exp(y), y = -|x| (explinear) or x (exp) ----
#pragma unroll
    for (int i = 0; i < NE; ++i) k[i] = rint((nlin == 1 ? -fabs(x[i]) : x[i]) * c[0]);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) r[i] = fma(-k[i], c[1], (nlin == 1 ? -fabs(x[i]) : x[i]));
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) r[i] = fma(-k[i], c[2], r[i]);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = fma(c[3], r[i], c[4]);
    PGL_ROW;
#pragma unroll
    for (int j = 5; j <= 13; ++j) {
#pragma unroll
        for (int i = 0; i < NE; ++i) q[i] = fma(q[i], r[i], c[j]);
        PGL_ROW;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = fma(q[i], r[i], 0.5);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = fma(q[i], r[i], 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = fma(q[i], r[i], 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) k[i] = fmin(fmax(k[i], -2200.0), 2200.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = ldexp(q[i], (int)k[i]);
    PGL_ROW;
    PGL_PROF_MARK(7);
    if (nlin != 1) {
        // exp nonlinearity: lam = e, term = x s - dt lam, r = s - dt lam
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const double sd = (double)sc[i];
            res[i] = fma(-dt, e[i], sd);
            t += fma(x[i], sd, -dt * e[i]);
        }
        term = t;
        return true;
    }
    bool small = true;
#pragma unroll
    for (int i = 0; i < NE; ++i) small = small && (e[i] < cthr);
    double l1p[NE], inv[NE];
    if (!__all(small)) {
        // ---- general regime (some |x| < 9.25: low firing rates): log1p(e) and 1/(1+e) in full, from the e at hand;
        // the spike terms stay compacted below -- the fully general pgl_rate_terms_n (a second exp, log and
        // reciprocal of lam for every element of a wave that holds a spike) is left for lam == 0 / NaN only ----
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const double u = 1.0 + e[i];
            inv[i] = pgl_rcp(u);
            l1p[i] = pgl_log(u, C) + (e[i] - (u - 1.0)) * inv[i];
        }
        PGL_ROW;
    } else {
    // ---- series regime: log1p(e) and 1/(1+e) from their alternating series (error < e^6) ----
#pragma unroll
    for (int i = 0; i < NE; ++i) l1p[i] = fma(-e[i], 0.2, 0.25);
#pragma unroll
    for (int i = 0; i < NE; ++i) inv[i] = fma(-e[i], 1.0, 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) l1p[i] = fma(-e[i], l1p[i], c13);
#pragma unroll
    for (int i = 0; i < NE; ++i) inv[i] = fma(-e[i], inv[i], 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) l1p[i] = fma(-e[i], l1p[i], 0.5);
#pragma unroll
    for (int i = 0; i < NE; ++i) inv[i] = fma(-e[i], inv[i], 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) l1p[i] = fma(-e[i], l1p[i], 1.0);
#pragma unroll
    for (int i = 0; i < NE; ++i) inv[i] = fma(-e[i], inv[i], 1.0);
    PGL_ROW;
#pragma unroll
    for (int i = 0; i < NE; ++i) l1p[i] = e[i] * l1p[i];
#pragma unroll
    for (int i = 0; i < NE; ++i) inv[i] = fma(-e[i], inv[i], 1.0);
    PGL_ROW;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) lam[i] = fmax(x[i], 0.0) + l1p[i];
#pragma unroll
    for (int i = 0; i < NE; ++i) sig[i] = (x[i] >= 0.0) ? inv[i] : e[i] * inv[i];
    PGL_ROW;
    }
    // reference semantics at lam == 0 / NaN are the general path's business
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NE; ++i) ok = ok && (lam[i] > 0.0);
    if (!__all(ok)) return false;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NE; ++i) t = fma(-dt, lam[i], t);
    PGL_PROF_MARK(8);
    // ---- spike terms, compacted: slot = rank of the (register, lane) pair among the tile's spikes ----
    // (scratch: CAP rates + CAP (log, 1/x) pairs = 3 CAP doubles per wave; CAP = 64 or, where LDS is short, 32)
    unsigned long long m[NE];
    int base[NE], total = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        m[i] = __ballot(sc[i] != 0u);
        base[i] = total;
        total += __popcll(m[i]);
    }
    if (total == 0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) res[i] = -dt * sig[i];
    } else if (total < CAP) {
        int slot[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            slot[i] = base[i] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[i] >> 32),
                                         __builtin_amdgcn_mbcnt_lo((unsigned)m[i], 0u));
            if (sc[i] != 0u) scratch[slot[i]] = lam[i];
            slot[i] = (sc[i] != 0u) ? slot[i] : CAP - 1;   // lanes without a spike read the unused last slot, weight 0
        }
        __builtin_amdgcn_wave_barrier();               // one wave: its LDS operations execute in order
        const double lc = (lane < total) ? scratch[lane] : 1.0;
        pgl_d2 LI;
        LI.x = pgl_log(lc, C);
        LI.y = pgl_rcp(lc);
        pgl_d2* const so = reinterpret_cast<pgl_d2*>(scratch + CAP);
        if (CAP == 64 || lane < CAP) so[lane] = LI;
        __builtin_amdgcn_wave_barrier();
        pgl_d2 g[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) g[i] = so[slot[i]];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            // elements without a spike read the last slot with weight 0: that slot is never a real spike here
            // (total < CAP) and holds (log 1, 1/1) -- slot 0 would be ANOTHER neuron's spike, and 0 * (1/lam)
            // of a denormal or infinite rate is NaN
            const double sd = (double)sc[i];
            t = fma(g[i].x, sd, t);
            res[i] = fma(sd, g[i].y, -dt) * sig[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            res[i] = -dt * sig[i];
            if (m[i] != 0ull) {
                const double sd = (double)sc[i];
                const double L = pgl_log(lam[i], C);
                const double I = pgl_rcp(lam[i]);
                t = (sc[i] != 0u) ? fma(L, sd, t) : t;
                res[i] = (sc[i] != 0u) ? fma(sd, I, -dt) * sig[i] : res[i];
            }
        }
    }
    PGL_PROF_MARK(9);
    term = t;
    return true;
}

__device__ __forceinline__ bool pgl_rate4(const double (&x)[4], const unsigned (&sc)[4], const int nlin,
                                          const double dt, const pgl_k_cdp C, double* scratch,
                                          const int lane, double& term, double (&res)[4] PGL_PROF_ARGS)
{
    return pgl_rate_fx<4, 64>(x, sc, nlin, dt, C, scratch, lane, term, res PGL_PROF_PASS);
}

template <typename CP>
__device__ __forceinline__ void pgl_rate_terms(const double x, const double s, const int nlin,
                                               const double dt, double& term, double& res,
                                               const CP C)
{
    const double xa[1] = {x}, sa[1] = {s};
    double ta[1], ra[1];
    pgl_rate_terms_n<1>(xa, sa, nlin, dt, ta, ra, C);
    term = ta[0];
    res = ra[0];
}

__device__ __forceinline__ double pgl_softplus_parts(double x, double& sig, double& loglam)
{
    // stable log(1+exp(x)) (nlin.py:43); ocml form, used by the non-hot helper kernels
    const double e = exp(-fabs(x));
    const double lam = fmax(x, 0.0) + log1p(e);
    const double inv = 1.0 / (1.0 + e);
    sig = (x >= 0.0) ? inv : e * inv;
    loglam = log(lam);
    return lam;
}
